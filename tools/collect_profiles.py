"""Copies the judged summaries of a tools/gpu_profile.sh run into profiles/ :
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary
  profiles/<tag>_pmc_summary.json   per-kernel sums of the PMC passes + derived figures
  profiles/traffic.json             HBM-side bytes per k_intersect_kd launch (read by bench.py)
Usage: collect_profiles.py gpurun_out/prof_<tag> <tag>"""
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parse_prof  # noqa: E402

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(out, tag + "_kernel_stats.csv"))
summary = os.path.join(out, tag + "_pmc_summary.json")
parse_prof.main(src, summary)
d = json.load(open(summary))
key = [k for k in d["counters"] if "k_intersect_kd<false" in k][0]
c = d["counters"][key]
n = c["_dispatches"]["pmc_rdreq"]
read_bytes = 32 * c["TCC_EA0_RDREQ_32B_sum"] + 64 * c["TCC_EA0_RDREQ_64B_sum"] + 128 * c["TCC_EA0_RDREQ_128B_sum"]
write_bytes = c["WRITE_SIZE"] * 1024
derived = {
    "kernel": key,
    "dispatches": n,
    "read_bytes_per_launch": read_bytes / n,
    "write_bytes_per_launch": write_bytes / c["_dispatches"]["pmc_write"],
    "fetch_size_kb_per_launch": c["FETCH_SIZE"] / c["_dispatches"]["pmc_fetch"],
    "fetch_size_x2_bytes_per_launch": 2 * 1024 * c["FETCH_SIZE"] / c["_dispatches"]["pmc_fetch"],
    "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
    "valu_lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]),
    "wave_cycles_waiting": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
    "wave_cycles_issue_stalled": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
    "avg_launch_us_traced": d["kernels"][key]["avg_us"],
    # share of the SIMDs' issue slots taken by vector-ALU instructions: SQ_ACTIVE_INST_VALU counts
    # 4-cycle slots summed over all SIMDs, GRBM_GUI_ACTIVE busy cycles summed over the 8 XCDs
    "valu_busy": 4.0 * c["SQ_ACTIVE_INST_VALU"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0),
    "valu_insts_per_launch": c["SQ_INSTS_VALU"] / c["_dispatches"]["pmc_sq"],
    "salu_insts_per_launch": c["SQ_INSTS_SALU"] / c["_dispatches"]["pmc_sq2"],
}
# the bench line of the traced run (its JSON is in the pass's log) tells how many path-bounces a launch held
pb_per_launch = None
counter_config = None
for line in open(os.path.join(src, "trace.log"), errors="replace"):
    if line.startswith("{"):
        pb_per_launch = json.loads(line)["roofline"]["path_bounces_per_launch"]
        counter_config = json.loads(line)["roofline"]["counter_config"]
        shutil.copy(os.path.join(src, "trace.log"), os.path.join(out, tag + "_bench_under_trace.log"))
derived["path_bounces_per_launch"] = pb_per_launch
d["derived_k_intersect_kd"] = derived


def per_kernel(name):
    """HBM-side bytes over kernel time, VALU issue-slot occupancy and lane activity of one kernel (all its dispatches)."""
    keys = [k for k in d["counters"] if name in k]
    if not keys:
        return None
    c = d["counters"][keys[0]]
    t_us = d["kernels"][keys[0]]["total_us"]
    rd = 32 * c["TCC_EA0_RDREQ_32B_sum"] + 64 * c["TCC_EA0_RDREQ_64B_sum"] + 128 * c["TCC_EA0_RDREQ_128B_sum"]
    wr = c["WRITE_SIZE"] * 1024
    return {
        "kernel": keys[0], "dispatches": c["_dispatches"]["pmc_sq"], "total_ms_traced": t_us / 1e3,
        "hbm_read_gb": rd / 1e9, "hbm_write_gb": wr / 1e9, "hbm_tb_per_s": (rd + wr) / 1e12 / (t_us / 1e6),
        "hbm_frac_of_8tb": (rd + wr) / 1e12 / (t_us / 1e6) / 8.0,
        "l2_hit_rate": c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
        "valu_busy": 4.0 * c["SQ_ACTIVE_INST_VALU"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0),
        "valu_lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]),
        "wave_cycles_waiting": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
        "valu_wave_insts": c["SQ_INSTS_VALU"], "salu_wave_insts": c["SQ_INSTS_SALU"],
    }


d["derived_per_kernel"] = {n: per_kernel(n) for n in ("k_intersect_kd<false", "k_shade", "k_camera", "k_finish")}
json.dump(d, open(summary, "w"), indent=1, sort_keys=True)
json.dump({
    "k_intersect_hbm_bytes_per_launch": derived["read_bytes_per_launch"] + derived["write_bytes_per_launch"],
    "path_bounces_per_launch": pb_per_launch,
    # bench.py reports these counters only for a run of this very configuration
    "config": counter_config,
    "how": "rocprofv3 --pmc, separate passes: sum(TCC_EA0_RDREQ_{32,64,128}B_sum x request size) + WRITE_SIZE*1024, "
           "divided by the kernel's dispatches; bench.py --spp-per-step 1 --steps 512 --warmup 0 (8 launches of 512 passes)",
    "source": tag + "_pmc_summary.json",
}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(derived, indent=1))
