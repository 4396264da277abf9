#!/bin/bash
# Full-occupancy comparison of two inflate builds on a BAM of SMALL BGZF blocks (so that 131 072 of them stay below
# the 4 GiB a chunk's inflated stream may hold): bash tools/ab_small_blocks.sh libA libB
cd $GRAFT_REPO_ROOT
python3 - <<PY
import sys; sys.path.insert(0, '.')
from te_counter_amd import bam, synth
ann = synth.make_annotation('hg38', scale=0.05)
u = synth.make_units(ann, 7_000_000, seed=4242, paired=True)
refs = ann.chrom_names + ['chrUn_1']
tid, pos, fl, mq = bam.units_to_records(u, len(refs) - 1, True)
bam.write_bam_arrays('/dev/shm/ab.bam', refs, tid, pos, fl, mq, pair_names=True, payload='illumina', block_bytes=30000)
ann.flat.save('/dev/shm/ab.tecidx')
PY
cat > /tmp/ab_one.py <<'PY'
import sys, time
sys.path.insert(0, '.')
from te_counter_amd.measure import measureTE
m = measureTE('x', 20); m.bind_genome('/dev/shm/ab.tecidx'); m.load_genome()
for rep in range(3):
    t = time.time(); r = m.parse_bampe('/dev/shm/ab.bam'); dt = time.time() - t
    print(sys.argv[1], 'decode_and_count %.3f s' % dt, m.decode_report['blocks'], m.decode_report['seconds'], flush=True)
PY
for lib in "$@"; do cp $lib te_counter_amd/libtecount.so; TEC_GBAM_CHUNK_BLOCKS=131072 python3 /tmp/ab_one.py $lib; done
rm -f /dev/shm/ab.bam /dev/shm/ab.tecidx
