#!/bin/bash
# Profiling recipe for the bulk count kernel (run on the GPU box through gpurun):
#   bash tools/prof.sh <tag> [bench args]        (HBM_ONLY=1: kernel trace + the passes the HBM figure needs + the SQ pass)
# One rocprofv3 run per counter group (no tracing flags next to --pmc); results under
# gpurun_out/prof_<tag>/; tools/traffic.py condenses them into profiles/.
set -o pipefail
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
echo "$*" > $OUT/bench_args.txt
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --legs headline --steps 5 --warmup 1 $*"
run() { # name, counters...
  local name=$1; shift
  echo "[prof] $name: $*"
  timeout -k 10 400 rocprofv3 "$@" --output-format csv -d $OUT/$name -o out -- $BENCH > $OUT/$name.log 2>&1 || { echo "[prof] $name FAILED"; tail -3 $OUT/$name.log; return 1; }
}
run trace --kernel-trace --stats || exit 1
run pmc_fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE SQ_WAVES
run pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run pmc_rdreq --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum
run pmc_sq --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
[ -n "$HBM_ONLY" ] && { echo "[prof] done (HBM passes only)"; exit 0; }
run pmc_l1 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
run pmc_atomic --pmc TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_REQ_sum SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT
echo "[prof] done"
