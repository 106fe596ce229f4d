"""Copies the judged summaries of a tools/gpu_profile.sh run into profiles/ :
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary
  profiles/<tag>_pmc_summary.json   per-kernel sums of the PMC passes + derived figures
  profiles/traffic_<workload>.json  per kernel: HBM-side bytes per launch, VALU occupancy, lane activity, L2 hit rate
                                    and the configuration they were collected on (read by bench.py)
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
for f in parse_prof.newest(os.path.join(src, "trace", "**", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(out, tag + "_kernel_stats.csv"))
summary = os.path.join(out, tag + "_pmc_summary.json")
parse_prof.main(src, summary)
d = json.load(open(summary))
kd_keys = [k for k in d["counters"] if "k_intersect_kd<false" in k]
if not kd_keys:          # a scene without triangles launches no intersect kernel: the derived block describes k_shade
    kd_keys = [k for k in d["counters"] if "k_shade" in k]
key = kd_keys[0]
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
# what bench.py reads: per kernel (its own names) the HBM-side bytes of one launch and the path-bounces that launch held.
# k_intersect_kd and k_shade run once per depth on the same queues: their launches hold the same path-bounces.
kernels = {}
for bench_name, prof_name in (("k_intersect_kd", "k_intersect_kd<false"), ("k_shade", "k_shade")):
    pk = d["derived_per_kernel"].get(prof_name)
    if not pk:
        continue
    cc = d["counters"][pk["kernel"]]
    n_rd, n_wr = cc["_dispatches"]["pmc_rdreq"], cc["_dispatches"]["pmc_write"]
    # The run is one batch of passes, i.e. one launch per depth, preceded by the bench's single instrumented pass.  That
    # pass uses another instantiation of the intersect kernel but the SAME k_shade: `depth` of k_shade's dispatches are
    # its one-pass launches (1 / 512 of a batch's work, counted into the sums) -- "per launch" divides by the batches' launches.
    depth = (counter_config or {}).get("depth") or 0
    if bench_name == "k_shade" and n_rd > depth:
        n_rd = n_wr = n_rd - depth
    kernels[bench_name] = {
        "hbm_bytes_per_launch": pk["hbm_read_gb"] * 1e9 / n_rd + pk["hbm_write_gb"] * 1e9 / n_wr,
        "path_bounces_per_launch": pb_per_launch, "dispatches": n_rd,
        "valu_busy": pk["valu_busy"], "valu_lane_utilisation": pk["valu_lane_utilisation"],
        "l2_hit_rate": pk["l2_hit_rate"], "wave_cycles_waiting": pk["wave_cycles_waiting"],
        "hbm_tb_per_s": pk["hbm_tb_per_s"], "avg_launch_us_traced": d["kernels"][pk["kernel"]]["total_us"] / n_rd,
    }
workload = (counter_config or {}).get("workload", "unknown")
json.dump({
    "kernels": kernels,
    # bench.py reports these counters only for a run of this very configuration
    "config": counter_config,
    "how": "rocprofv3 --pmc, separate passes: sum(TCC_EA0_RDREQ_{32,64,128}B_sum x request size) + WRITE_SIZE*1024, "
           "divided by the kernel's dispatches; valu_busy = 4 x SQ_ACTIVE_INST_VALU / SIMD cycles, lane utilisation = "
           "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); bench.py --spp-per-step 1 --warmup 0, one batch of passes",
    "source": tag + "_pmc_summary.json",
}, open(os.path.join(out, "traffic_%s.json" % workload), "w"), indent=1)
print(json.dumps(derived, indent=1))
