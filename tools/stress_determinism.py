#!/usr/bin/env python3
"""Repeats the same device-resident count many times and checks that every run gives the same
vector (and k passes = k x one pass): a detector for races / counter overflow in the kernels."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from te_counter_amd import Engine, synth
from oracle import c_oracle

order = sys.argv[1] if len(sys.argv) > 1 else 'sorted'
n_units = int(sys.argv[2]) if len(sys.argv) > 2 else 62_500_000
ann = synth.make_annotation('hg38')
units = synth.make_units(ann, n_units, seed=2003, paired=True, order=order)
dev = [torch.from_numpy(u.view(np.int16) if u.dtype == np.uint16 else u).cuda() for u in units]
ptrs = [d.data_ptr() for d in dev]
torch.cuda.synchronize()
REPS = int(os.environ.get('STRESS_REPS', 12))
for env in ({}, {'TEC_UNROLL': '1'}, {'TEC_KERNEL': '1'}, {'TEC_LDS_KB': '40'}, {'TEC_UNROLL': '4'}):
    for k in ('TEC_UNROLL', 'TEC_KERNEL', 'TEC_LDS_KB'):
        os.environ.pop(k, None)
    os.environ.update(env)
    eng = Engine(0)
    eng.upload_index(ann.flat)
    ref = None
    bad = 0
    for rep in range(REPS):
        eng.bulk_begin(20)
        eng.bulk_submit_device(n_units, *ptrs)
        c, st = eng.bulk_finish()
        if ref is None:
            ref = c.copy()
        elif not np.array_equal(c, ref):
            bad += 1
            d = np.nonzero(c != ref)[0]
            if bad <= 6:
                print('  rep', rep, 'differs at', len(d), 'names; e.g.', d[:5], (c[d[:5]].astype(np.int64) - ref[d[:5]].astype(np.int64)))
    print(order, env, 'mismatching repeats:', bad, 'sum', int(ref.sum()))
    for steps in ():
        badk = 0
        for rep in range(4):
            eng.bulk_begin(20)
            for _ in range(steps):
                eng.bulk_submit_device(n_units, *ptrs)
            c, st = eng.bulk_finish()
            if not np.array_equal(c, ref * steps):
                badk += 1
                d = np.nonzero(c != ref * steps)[0]
                print('  steps', steps, 'rep', rep, 'differs at', len(d), 'names; e.g.', d[:4], (c[d[:4]].astype(np.int64) - (ref[d[:4]] * steps).astype(np.int64)))
        print('   steps', steps, 'bad', badk)
    eng.close()
# against the oracle on a prefix (sorted prefix = one genome region)
ci = c_oracle.COracleIndex.from_flat(ann.flat)
m = min(n_units, 3_000_000)
want, _ = ci.bulk_count(*[u[:m] for u in units], 20)
eng = Engine(0)
eng.upload_index(ann.flat)
eng.bulk_begin(20)
eng.bulk_submit(*[u[:m] for u in units])
got, _ = eng.bulk_finish()
print('oracle prefix identical:', bool(np.array_equal(got, want)))
