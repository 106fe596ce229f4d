"""Summarises a tools/gpu_profile.sh output directory into one JSON file:
per-kernel call counts / average durations from the kernel trace and per-kernel
sums of every PMC counter.  Usage: parse_prof.py gpurun_out/prof_<tag> out.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    for k in ("k_intersect<true, false>", "k_intersect<true, true>", "k_intersect<false, false>",
              "k_intersect<false, true>", "k_shade", "k_camera", "k_finish", "k_checksum", "k_debug_intersect"):
        if k in name:
            return k
    return name.split("(")[0][:60]


def newest(pattern):
    """The most recent file matching `pattern` (results of repeated runs merged into one directory keep the older
    runs' files, named by process id, beside the new ones)."""
    files = glob.glob(pattern, recursive=True)
    return [max(files, key=os.path.getmtime)] if files else []


def main(d, out):
    res = {"kernels": {}, "counters": {}}
    for f in newest(os.path.join(d, "trace", "**", "*kernel_trace.csv")):
        agg = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for k, (n, us) in agg.items():
            res["kernels"][k] = {"calls": n, "total_us": round(us, 1), "avg_us": round(us / n, 2)}
    for sub in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        if not os.path.isdir(sub):
            continue
        for f in newest(os.path.join(sub, "**", "*counter_collection.csv")):
            agg = defaultdict(lambda: defaultdict(float))
            calls = defaultdict(set)
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                calls[k].add(r["Dispatch_Id"])
            for k, cs in agg.items():
                e = res["counters"].setdefault(k, {})
                for c, v in cs.items():
                    e[c] = v
                e.setdefault("_dispatches", {})[os.path.basename(sub)] = len(calls[k])
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
