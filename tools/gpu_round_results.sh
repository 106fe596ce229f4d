#!/bin/bash
# tools/gpu_round_results.sh <tag> -- on the GPU box: the bench lines and timings a round's profiles/ directory keeps
# (default bench line as the driver runs it, the other workloads, launchKernel call time, path-flip measurement).
set -u
TAG=${1:-rXX}
OUT=gpurun_out/results_$TAG
mkdir -p $OUT
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || exit 1
: > $OUT/workloads.jsonl
timeout -k 10 300 python bench.py --workload spheres --steps 4 --warmup 1 --no-cpu-baseline >> $OUT/workloads.jsonl 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload diamond_coil --steps 4 --warmup 1 --no-cpu-baseline >> $OUT/workloads.jsonl 2>/dev/null || exit 1
timeout -k 10 400 python bench.py --workload soup --width 3840 --height 2160 --steps 1 --warmup 0 --spp-per-step 256 --no-cpu-baseline >> $OUT/workloads.jsonl 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --spp-per-step 128 --batch 128 --no-cpu-baseline >> $OUT/workloads.jsonl 2>/dev/null || exit 1
timeout -k 10 300 python tools/launch_kernel_timing.py > $OUT/launch_kernel_timing.txt 2>&1 || exit 1
timeout -k 10 600 python -u tools/measure_flips.py > $OUT/flips.log 2>&1 || exit 1
timeout -k 10 200 python tools/depth_profile.py 64 > $OUT/depth_profile.txt 2>&1 || exit 1
echo done
