#!/bin/bash
# Quick variants of bench.py on the GPU box, one line per variant.  Each argument is one variant: leading
# NAME=value words are exported for that run, the rest are bench.py arguments, e.g.
#   tools/sweep.sh "" "TEC_UNROLL=1" "TEC_SPILL=0 --genome custom3m --layout se --records 100000000" "--order sorted"
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  envs=(); args=()
  for w in $v; do
    if [[ ${#args[@]} -eq 0 && $w == [A-Za-z_]*=* ]]; then envs+=("$w"); else args+=("$w"); fi
  done
  echo "== $v"
  env "${envs[@]}" timeout -k 10 300 python bench.py --legs headline --steps 10 --warmup 2 "${args[@]}" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], 'Mreads/s', d['ms_per_step'], 'ms', d['roofline'].get('avg_launch_ms'), d['config'].get('index'), d.get('assigned_frac'))"
done
