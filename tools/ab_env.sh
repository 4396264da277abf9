#!/bin/bash
# A/B of decoder settings on one large synthetic BAM (development aid, run on the GPU box):
#   bash tools/ab_env.sh <records> <payload> "VAR=a VAR2=b" "VAR=c" ...     (an empty string "" = the defaults)
# One process per setting (the variables are read when a file is opened), three passes each.
cd $GRAFT_REPO_ROOT
N=$1; P=$2; shift 2
python3 - <<PY
import sys; sys.path.insert(0, '.')
from te_counter_amd import bam, synth
ann = synth.make_annotation('hg38', scale=0.05)
u = synth.make_units(ann, $N // 2, seed=4242, paired=True)
refs = ann.chrom_names + ['chrUn_1']
tid, pos, fl, mq = bam.units_to_records(u, len(refs) - 1, True)
bam.write_bam_arrays('/dev/shm/ab.bam', refs, tid, pos, fl, mq, pair_names=True, payload='$P')
ann.flat.save('/dev/shm/ab.tecidx')
PY
cat > /tmp/ab_one.py <<'PY'
import sys, time
sys.path.insert(0, '.')
from te_counter_amd.measure import measureTE
m = measureTE('x', 20); m.bind_genome('/dev/shm/ab.tecidx'); m.load_genome()
for rep in range(3):
    t = time.time(); r = m.parse_bampe('/dev/shm/ab.bam'); dt = time.time() - t
    print('[%s]' % sys.argv[1], 'decode_and_count %.3f s' % dt, m.decode_report['blocks'], m.decode_report['seconds'], flush=True)
PY
for cfg in "$@"; do env $cfg timeout -k 10 120 python3 /tmp/ab_one.py "$cfg" || break; done
rm -f /dev/shm/ab.bam /dev/shm/ab.tecidx
