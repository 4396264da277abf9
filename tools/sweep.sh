#!/bin/bash
# quick parameter sweeps of bench.py on the GPU box; prints one line per variant
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  echo "== $v"
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-host-stream $v 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], 'Mreads/s', d['ms_per_step'], 'ms', d['roofline']['avg_launch_ms'], d['config']['index'], d['assigned_frac'])"
done
