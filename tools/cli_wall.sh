#!/bin/bash
# Wall time of a whole `bin/te_count` process on a synthetic PE BAM (GPU box): bash tools/cli_wall.sh [records]
cd $GRAFT_REPO_ROOT
N=${1:-40000000}
python3 - <<PY
import sys; sys.path.insert(0, '.')
from te_counter_amd import bam, synth
ann = synth.make_annotation('hg38')
u = synth.make_units(ann, $N // 2, seed=4242, paired=True)
refs = ann.chrom_names + ['chrUn_1']
tid, pos, fl, mq = bam.units_to_records(u, len(refs) - 1, True)
bam.write_bam_arrays('/dev/shm/w.bam', refs, tid, pos, fl, mq, pair_names=True, payload='illumina')
ann.flat.save('/dev/shm/w.tecidx')
PY
for rep in 1 2 3; do
  t0=$(date +%s.%N)
  TEC_INDEX_TIMING=1 python3 bin/te_count -i /dev/shm/w.bam -o /dev/shm/w.tsv -g /dev/shm/w.tecidx -m custom 2>&1 | grep -v "^INFO" | tail -12
  t1=$(date +%s.%N)
  python3 -c "print('te_count process: %.2f s wall' % ($t1 - $t0))"
done
for rep in 1 2; do t0=$(date +%s.%N); python3 tools/cli_phases.py; t1=$(date +%s.%N); python3 -c "print('phase script process: %.2f s wall' % ($t1 - $t0))"; done
rm -f /dev/shm/w.bam /dev/shm/w.tecidx /dev/shm/w.tsv
