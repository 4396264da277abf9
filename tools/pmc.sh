#!/bin/bash
# tools/pmc.sh <tag> "<bench args>" "<counters pass 1>" "<counters pass 2>" ...
# One rocprofv3 --pmc pass per counter group (no tracing flags alongside); prints per-kernel means.
# PMC_ENV="NAME=value ..." exports variables for the bench process (rocprofv3 starts python3 directly).
TAG=$1; BARGS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in $PMC_ENV; do export $kv; done
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 $REPO/bench.py --legs headline --steps 3 --warmup 1 $BARGS > $OUT/p$i.log 2>&1 || { echo "pass $i ($grp) failed"; tail -3 $OUT/p$i.log; continue; }
  python3 - $OUT/p$i/pmc_counter_collection.csv <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'count_bulk' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print('%-40s %16.1f  (n=%d)' % (k, sum(v) / len(v), len(v)))
PY
done
