#!/bin/bash
# tools/sweep_tree.sh -- bench workload over the kd-tree builder's SAH constants (GPT_KD_COST x GPT_KD_EMPTY_RATIO)
for c in ${COSTS:-1.5 2 3 4}; do for e in ${RATIOS:-0.12}; do
  GPT_KD_COST=$c GPT_KD_EMPTY_RATIO=$e timeout -k 10 200 python bench.py --no-cpu-baseline --spp-per-step 1 --steps 256 2>/dev/null | python tools/brief.py "cost=$c empty=$e" || exit 1
done; done
