#!/bin/bash
# tools/sweep.sh VAR v1 v2 ... -- runs bench.py once per value of the environment variable VAR
# (extra bench arguments through BENCH_ARGS) and prints one compact line per run.
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 200 python bench.py --no-cpu-baseline --spp-per-step 1 --steps 256 $BENCH_ARGS 2>/dev/null | python tools/brief.py "$var=$v" || exit 1
done
