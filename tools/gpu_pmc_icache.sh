#!/bin/bash
# Instruction-cache and scalar-cache counters of a short bench run (one rocprofv3 --pmc pass).
# Usage: tools/gpu_pmc_icache.sh [bench args...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_icache
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/p" -- python3 $ROOT/bench.py --spp-per-step 1 --steps 256 --warmup 0 --no-cpu-baseline "$@" > "$OUT/p.log" 2>&1
echo "rc=$?"
