#!/bin/bash
# tools/trace_small.sh PASSES -- on the GPU box: rocprofv3 --kernel-trace --stats around tools/depth_profile.py PASSES
# (per-kernel times of small launches: one pass, a few passes).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=${1:-1}
OUT=$ROOT/gpurun_out/trace_small_$P
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ROOT/tools/depth_profile.py $P > "$OUT/log.txt" 2>&1 || exit 1
grep "batch of" "$OUT/log.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
rows.sort()
# the last batch: from the last k_camera launch on
last = max(i for i, r in enumerate(rows) if r[2].startswith("k_camera(") or r[2] == "k_camera")
seq = rows[last:]
t0 = seq[0][0]
for a, b, n in seq:
    print("%8.3f ms  +%7.3f ms  %s" % ((a - t0) / 1e6, (b - a) / 1e6, n))
print("span %.3f ms, kernels %.3f ms" % ((seq[-1][1] - t0) / 1e6, sum(b - a for a, b, _ in seq) / 1e6))
PY
