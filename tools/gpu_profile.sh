#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + PMC passes of a short bench run.
# Each rocprofv3 invocation is its own process with its own time limit; counters
# are collected in separate passes (SQ / TCC / FETCH / WRITE), never together with
# API tracing.  Usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --spp-per-step 1 --steps ${STEPS:-512} --warmup 0 --no-cpu-baseline $*"

# ONLY="trace pmc_sq ..." restricts the run to the named passes
run() {  # name, rocprof args...
  local name=$1; shift
  if [ -n "${ONLY:-}" ] && ! [[ " $ONLY " == *" $name "* ]]; then return 0; fi
  timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc"
  return $rc
}
run trace --kernel-trace --stats || exit 1
run pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY || exit 1
run pmc_sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT || exit 1
run pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE || exit 1
run pmc_fetch --pmc FETCH_SIZE || exit 1
run pmc_write --pmc WRITE_SIZE TCC_EA0_RDREQ_DRAM_sum || exit 1
run pmc_rdreq --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum || echo "rdreq pass failed (optional)"
run pmc_tcp --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum || echo "tcp pass failed (optional)"
echo done
