#!/bin/bash
# Kernel timeline of the on-GPU decoder: rocprofv3 --kernel-trace over tools/bench_e2e.py, then, for the last pass over
# the file, when each kernel family ran and how much of the wall the GPU was busy.
#   bash tools/timeline_e2e.sh <tag> [bench_e2e args]   ->  gpurun_out/timeline_<tag>/{kernel_trace.csv,summary.txt}
TAG=${1:-x}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/timeline_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -o out -- python3 $REPO/tools/bench_e2e.py "$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
tail -1 $OUT/run.log | cut -c1-1500
cp $(find $OUT/raw -name '*kernel_trace.csv' | head -1) $OUT/kernel_trace.csv
rm -rf $OUT/raw
python3 $REPO/tools/timeline_e2e.py $OUT/kernel_trace.csv | tee $OUT/summary.txt
