# tools/sweep_soup.sh -- kd builder parameters (SAH stage threshold, traversal cost) on the soup and cow workloads
for small in ${SMALLS:-24 96 384}; do for cost in ${COSTS:-1.5 3 6}; do
  GPT_KD_COST=$cost timeout -k 10 300 python bench.py --workload soup --width 3840 --height 2160 --steps 1 --warmup 0 --spp-per-step 64 --batch 32 --kd-small $small --no-cpu-baseline 2>/dev/null | python tools/brief.py soup_cost${cost}_small${small} || exit 1
done; done
for small in 24 48 96; do
  timeout -k 10 300 python bench.py --steps 1 --warmup 0 --spp-per-step 512 --kd-small $small --no-cpu-baseline 2>/dev/null | python tools/brief.py cow_small${small} || exit 1
done
