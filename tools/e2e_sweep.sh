cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys, os, json, time
sys.path.insert(0, '.')
import numpy as np
from te_counter_amd import bam, synth, Engine
from te_counter_amd.gbam import GpuBam
ann = synth.make_annotation('hg38')
u = synth.make_units(ann, 12_000_000, seed=4242, paired=True)
refs = ann.chrom_names + ['chrUn_1']
tid, pos, fl, mq = bam.units_to_records(u, len(refs) - 1, True)
p = 'illumina'
bam.write_bam_arrays('/tmp/e_%s.bam' % p, refs, tid, pos, fl, mq, pair_names=True, payload=p)
eng = Engine(0); eng.upload_index(ann.flat)
for rep in range(3):
    t0 = time.time(); g = GpuBam(eng, '/tmp/e_%s.bam' % p); t1 = time.time()
    tm = ann.flat.tid_map(g.reference_names); eng.bulk_begin(20); t2 = time.time()
    r = g.run_bulk(True, 20, tm); t3 = time.time()
    c, st = eng.bulk_finish(); t4 = time.time()
    g.close(); t5 = time.time()
    print(json.dumps({'open': round(t1-t0,3), 'begin': round(t2-t1,3), 'run': round(t3-t2,3), 'finish': round(t4-t3,3), 'close': round(t5-t4,3), 'seconds': r['seconds']}))
PY
