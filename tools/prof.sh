#!/bin/bash
# Profiling recipe for the bulk count kernel (run on the GPU box through gpurun).
# Usage: bash tools/prof.sh <tag> [extra bench args]
set -o pipefail
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 5 --warmup 1 --cpu-sample 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1 || exit 2
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1 || exit 3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1 || exit 4
rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc_atomic -o pmc -- $BENCH > $OUT/pmc_atomic.log 2>&1 || echo "atomic pass failed (counter names?)"
find $OUT -name "*.csv" | head -40
