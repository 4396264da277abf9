#!/bin/bash
# tools/pmc_env.sh <tag> "<ENV=1 ...>" "<bench args>" "<counters pass 1>" "<counters pass 2>" ...
# As tools/pmc.sh, with environment variables for the bench process (exported before rocprofv3, which
# starts python3 directly).
TAG=$1; ENVS=$2; BARGS=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in $ENVS; do export $kv; done
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-host-stream $BARGS > $OUT/p$i.log 2>&1 || { echo "pass $i ($grp) failed"; tail -3 $OUT/p$i.log; continue; }
  python3 - $OUT/p$i/pmc_counter_collection.csv <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    short = 'count' if 'count_bulk' in k else ('scatter' if 'part_pass_kernel<true>' in k else ('hist' if 'part_pass' in k else ('scan' if 'part_scan' in k else None)))
    if short:
        agg[(short, r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print('%-8s %-36s %16.1f  (n=%d)' % (k[0], k[1], sum(v) / len(v), len(v)))
PY
done
