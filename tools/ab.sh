#!/bin/bash
# tools/ab.sh NAME... -- bench.py once per library variant gpupathtracer_amd/lib/libgpt_hip_NAME.so
# ("base" = the default library), one compact line each.
for v in "$@"; do
  if [ "$v" = base ]; then unset GPT_LIB_PATH; else export GPT_LIB_PATH=$PWD/gpupathtracer_amd/lib/libgpt_hip_$v.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --spp-per-step 1 --steps 256 $BENCH_ARGS 2>/dev/null | python tools/brief.py "$v" || exit 1
done
