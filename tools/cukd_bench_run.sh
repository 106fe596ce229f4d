#!/bin/bash
# tools/cukd_bench_run.sh [MESH] -- on the GPU box: tools/cukd_bench.py under rocprofv3 --kernel-trace, then the average duration
# of its traversal launches as rays/s and frames/s, the two figures the reference's benchmark prints (ray_traversal.cu:179-181).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
M=${1:-dragon_res3}
OUT=$ROOT/gpurun_out/cukd_bench_$M
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 $ROOT/tools/cukd_bench.py $M 100 > "$OUT/log.txt" 2>&1 || exit 1
grep -v "^[EW]2" "$OUT/log.txt" | grep "tree\|rays"
python3 - "$OUT" <<'PY'
import csv, glob, sys
d = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_intersect_kd<false" in r["Kernel_Name"]:
            d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e9)
d = d[1:]   # (the untimed first call)
n = 1024 * 864
avg = sum(d) / len(d)
print("k_intersect_kd: %d launches, %.3f ms average -> %.0f Mrays/s, %.0f frames/s of 1024 x 864" % (len(d), avg * 1e3, n / avg / 1e6, 1 / avg))
PY
