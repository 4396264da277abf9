#!/bin/bash
# Config 5 (3 M unique-name intervals, 100 M SE units) at several fine-bin widths (development aid, run on the GPU box):
#   bash tools/ab_cfg5.sh "0 9 10 11 12"
cd $GRAFT_REPO_ROOT
for s in ${1:-0 10 11 12}; do
  timeout -k 10 300 python3 bench.py --legs headline --steps 10 --warmup 2 --genome custom3m --layout se --records 100000000 --bin-shift $s --cpu-sample 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bin-shift $s', 'ms_per_step', d['ms_per_step'], 'roofline', {k: d['roofline'].get(k) for k in ('avg_launch_ms','achieved')}, {k: d.get(k) for k in ('spill_fold_ms_per_launch','gpu_ms_per_step')})"
done
