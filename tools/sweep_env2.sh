#!/bin/bash
# tools/sweep_env2.sh "ENV1=a ENV2=b" [bench args] -- one bench run under the given environment
cd ${GRAFT_REPO_ROOT:-/root/repo}
E=$1; shift
echo "== $E $*"
env $E timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-host-stream "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], 'Mreads/s', d['ms_per_step'], 'ms', d['roofline']['avg_launch_ms'], d['assigned_frac'])"
