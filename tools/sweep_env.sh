#!/bin/bash
# sweep an environment knob: tools/sweep_env.sh VAR "v1 v2 ..." [bench args]
cd ${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  echo "== $VAR=$v $*"
  env $VAR=$v timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-host-stream "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], 'Mreads/s', d['ms_per_step'], 'ms', d['roofline']['avg_launch_ms'], d['assigned_frac'])"
done
