#!/bin/bash
# Kernel trace only of the single-cell job (BASELINE config 4 from host batches), for quick A/B of kernel changes:
#   bash tools/trace_sc.sh <tag>   ->  gpurun_out/trace_<tag>/kernel_stats.csv + a top-25 table on stdout
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o out -- python3 $REPO/tools/bench_sc.py --check 0 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
tail -1 $OUT/run.log
cp $(find $OUT/raw -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/raw
python3 - $OUT/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows if 'rocclr' not in r['Name']) / 1e6
print('kernel sum (without copy / fill kernels): %.2f ms' % tot)
for r in rows[:28]:
    print('%9.3f ms %5s  %s' % (float(r['TotalDurationNs']) / 1e6, r['Calls'], r['Name'][:120]))
PY
