#!/bin/bash
# tools/sweep_walk.sh -- bench workload over k_intersect_kd's two scheduling thresholds (GPT_LEAF_MIN x GPT_REFILL_BELOW)
for lm in ${LEAF_MINS:-20 28 36 44}; do for rb in ${REFILLS:-44 52 60}; do
  GPT_LEAF_MIN=$lm GPT_REFILL_BELOW=$rb timeout -k 10 200 python bench.py --no-cpu-baseline --spp-per-step 1 --steps 256 2>/dev/null | python tools/brief.py "leaf_min=$lm refill=$rb" || exit 1
done; done
