#!/bin/bash
# Kernel trace and HBM-traffic passes of the single-cell pipeline on host batches, BASELINE config 4
# (run on the GPU box through gpurun):  bash tools/prof_sc.sh <tag> [bench_sc args]
# One rocprofv3 run per counter group; tools/traffic_sc.py condenses gpurun_out/prof_<tag>/ into profiles/.
set -o pipefail
TAG=${1:-r03_sc}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
echo "--check 0 $*" > $OUT/bench_args.txt
cd /tmp && export TMPDIR=/tmp
run() { # name, flags...
  local name=$1; shift
  echo "[prof] $name: $*"
  timeout -s KILL 300 rocprofv3 "$@" --output-format csv -d $OUT/$name -o out -- python3 $REPO/tools/bench_sc.py --check 0 $BENCH_ARGS > $OUT/$name.log 2>&1 || { echo "[prof] $name FAILED"; tail -3 $OUT/$name.log; return 1; }
}
BENCH_ARGS="$*"
run trace --kernel-trace --stats || exit 1
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_rdreq --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum
echo "[prof] done"
