#!/bin/bash
# rocprofv3 kernel trace of the single-cell benchmark: bash tools/prof_sc.sh <tag> [bench_sc args]
TAG=${1:-r01_sc}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o out -- python3 $REPO/tools/bench_sc.py --check 0 "$@" > $OUT/trace.log 2>&1 || { echo FAILED; tail -5 $OUT/trace.log; exit 1; }
tail -1 $OUT/trace.log | cut -c1-600
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $REPO/gpurun_out/${TAG}_kernel_stats.csv
head -40 $REPO/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
