#!/bin/bash
# tools/pmc_env.sh "<ENV=val ...>" "<bench args>" "<counters>" -- one PMC pass under an environment
E=$1; BARGS=$2; CTRS=$3
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_env_$$
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export $E
rocprofv3 --pmc $CTRS --output-format csv -d $OUT -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --cpu-sample 0 $BARGS > $OUT/log.txt 2>&1 || { echo failed; tail -3 $OUT/log.txt; exit 1; }
python3 - $OUT/pmc_counter_collection.csv <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'count_bulk' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print('%-40s %16.1f  (n=%d)' % (k, sum(v) / len(v), len(v)))
PY
