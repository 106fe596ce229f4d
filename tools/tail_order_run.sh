#!/bin/bash
# tools/tail_order_run.sh DEPTH [MAX_RAYS] [PASSES] -- on the GPU box: tools/tail_order_experiment.py under rocprofv3 --kernel-trace,
# then the durations of its k_intersect_kd launches in launch order.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=${1:-3}; M=${2:-16000000}; P=${3:-64}
OUT=$ROOT/gpurun_out/tail_order_d$D
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 $ROOT/tools/tail_order_experiment.py $D $M $P > "$OUT/log.txt" 2>&1 || exit 1
grep -v "^[EW]2" "$OUT/log.txt" | grep "depth\|hits"
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_intersect_kd<false" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
rows.sort()
print("k_intersect_kd launches in order (ms):", " ".join("%.3f" % d for _, d in rows))
PY
