"""Run-to-run determinism of the bulk kernel on coordinate-sorted input (every lane of a wave
hits the same hot name: the pattern that exposed a lost-update race between in-flight LDS adds
and the histogram flush).  Every repeat must give the same vector, equal to the oracle's."""

import numpy as np
import pytest

from oracle import c_oracle
from te_counter_amd import Engine, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('order', ['sorted', 'shuffled'])
def test_repeated_launches_are_identical(order):
    import torch
    ann = synth.make_annotation('hg38', scale=0.25)
    n = 16_000_000
    units = synth.make_units(ann, n, seed=2003, paired=True, order=order)
    want, _ = c_oracle.COracleIndex.from_flat(ann.flat).bulk_count(*[u[:4_000_000] for u in units], 20)
    dev = [torch.from_numpy(u.view(np.int16) if u.dtype == np.uint16 else u).cuda() for u in units]
    ptrs = [d.data_ptr() for d in dev]
    torch.cuda.synchronize()
    eng = Engine(0)
    eng.upload_index(ann.flat)
    eng.bulk_begin(20)
    eng.bulk_submit(*[u[:4_000_000] for u in units])
    got, _ = eng.bulk_finish()
    assert np.array_equal(got, want)
    ref = None
    for rep in range(200):
        eng.bulk_begin(20)
        eng.bulk_submit_device(n, *ptrs)
        c, _ = eng.bulk_finish()
        if ref is None:
            ref = c
        assert np.array_equal(c, ref), 'launch %d differs' % rep
    eng.bulk_begin(20)
    for _ in range(7):
        eng.bulk_submit_device(n, *ptrs)
    c, _ = eng.bulk_finish()
    assert np.array_equal(c, 7 * ref)
    eng.close()


def test_sc_repeated_runs_are_identical():
    """The single-cell pipeline (atomics, rocPRIM sorts, epoch search) must be run-to-run stable."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import bench_sc
    ann = synth.make_annotation('hg38', scale=0.05)
    eng = Engine(0)
    eng.upload_index(ann.flat)
    arrays, wl, tables = synth.make_sc_arrays(ann, 3_000_000, n_cells=1500, n_whitelist=40000, seed=5, order='sorted')
    ref = None
    for rep in range(25):
        out, st, _, (ids, counts) = bench_sc.run(eng, arrays, wl, tables, 1000, bool(rep % 2), 250_000, batch=1 << 20)
        key = (rep % 2)
        cur = (st, ids.tobytes(), counts.tobytes()) + tuple(out[k].tobytes() for k in ('cells', 'cell_hits', 'coo_cell', 'coo_name', 'coo_count'))
        if ref is None:
            ref = {}
        if key not in ref:
            ref[key] = cur
        assert cur == ref[key], 'run %d differs' % rep
    eng.close()
