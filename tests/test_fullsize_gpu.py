"""BASELINE.json-sized runs on the GPU, checked through size-independent properties:
sample parity against the C oracle, linearity, invariance under sharding (1/2/4/8 ways, the
multi-GPU decomposition) and under permutation of the input order, and conservation of units."""

import numpy as np
import pytest

from oracle import c_oracle
from te_counter_amd import Engine, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hg38():
    ann = synth.make_annotation('hg38')
    eng = Engine(0)
    eng.upload_index(ann.flat)
    yield ann, eng
    eng.close()


def _count(eng, units, parts=1, q=20):
    eng.bulk_begin(q)
    n = len(units[0])
    step = (n + parts - 1) // parts
    for a in range(0, n, step):
        eng.bulk_submit(*[u[a:a + step] for u in units])
    return eng.bulk_finish()


@pytest.mark.parametrize('paired,n_units', [(True, 62_500_000), (False, 100_000_000)])
def test_full_size_properties(hg38, paired, n_units):
    """Headline shape (125 M PE records) and BASELINE config 2 (100 M SE records, -q 20)."""
    ann, eng = hg38
    units = synth.make_units(ann, n_units, seed=2002 if not paired else 2003, paired=paired)
    full, st = _count(eng, units, parts=4)
    assert st['units'] == n_units
    assert st['qcfail'] + st['lowq'] + st['badchrom'] <= n_units and st['assigned'] <= n_units
    # conservation: every unit is filtered, or examined
    assert int(full.sum()) >= st['assigned']

    # (1) sample parity against the oracle: a strided 1.5 M-unit sample
    idx = np.arange(0, n_units, n_units // 1_500_000)
    sample = [u[idx] for u in units]
    want, wst = c_oracle.COracleIndex.from_flat(ann.flat).bulk_count(*sample, 20)
    got, gst = _count(eng, sample)
    assert np.array_equal(got, want) and all(gst[k] == v for k, v in wst.items())

    # (2) sharding: the sum over 8 disjoint shards (the 8-GPU decomposition) is the whole
    acc = np.zeros_like(full)
    acc_assigned = 0
    for r in range(8):
        c, s = _count(eng, [u[r::8] for u in units])
        acc += c
        acc_assigned += s['assigned']
    assert np.array_equal(acc, full) and acc_assigned == st['assigned']

    # (3) permutation invariance: reversed input order
    rev, rst = _count(eng, [u[::-1].copy() for u in units], parts=3)
    assert np.array_equal(rev, full) and rst == st

    # (4) linearity: the batch counted twice
    eng.bulk_begin(20)
    for _ in range(2):
        eng.bulk_submit(*units)
    twice, tst = eng.bulk_finish()
    assert np.array_equal(twice, 2 * full) and tst['assigned'] == 2 * st['assigned']


def test_custom_3M_interval_annotation():
    """BASELINE config 5 shape: 3 M unique-name intervals (name space far beyond the LDS
    histogram -> hot/cold split with global atomics), 40 M SE units; sample parity + sharding."""
    ann = synth.make_custom_annotation()
    eng = Engine(0)
    eng.upload_index(ann.flat)
    units = synth.make_units(ann, 40_000_000, seed=2005, paired=False)
    full, st = _count(eng, units, parts=2)
    idx = np.arange(0, 40_000_000, 40)
    sample = [u[idx] for u in units]
    want, wst = c_oracle.COracleIndex.from_flat(ann.flat).bulk_count(*sample, 20)
    got, gst = _count(eng, sample)
    assert np.array_equal(got, want) and all(gst[k] == v for k, v in wst.items())
    acc = np.zeros_like(full)
    for r in range(2):
        acc += _count(eng, [u[r::2] for u in units])[0]
    assert np.array_equal(acc, full)
    eng.close()


def test_sc_batching_invariance_and_oracle_sample():
    """Single-cell path at 6 M records (config 4 generator, several spill epochs): the result must
    not depend on how the file is cut into batches; a 120 k-record run of the same generator is
    checked bit-for-bit against the Python oracle."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from oracle import te_oracle as orc
    import bench_sc
    ann = synth.make_annotation('hg38', scale=0.05)
    eng = Engine(0)
    eng.upload_index(ann.flat)
    arrays, wl, tables = synth.make_sc_arrays(ann, 6_000_000, n_cells=2000, n_whitelist=50000, seed=77)
    outs = []
    for batch in (1 << 20, 1 << 22, 1 << 23, 777_777):
        out, st, _, (ids, counts) = bench_sc.run(eng, arrays, wl, tables, 1500, True, 400_000, batch=batch)
        outs.append((out, st, ids, counts))
    assert outs[0][1]['bundles'] >= 4
    for o in outs[1:]:
        assert o[1] == outs[0][1]
        assert np.array_equal(o[2], outs[0][2]) and np.array_equal(o[3], outs[0][3])
        for k in ('cells', 'cell_hits', 'coo_cell', 'coo_name', 'coo_count'):
            assert np.array_equal(o[0][k], outs[0][0][k]), k
    # oracle sample
    small = synth.make_annotation('hg38', scale=0.01)
    eng.upload_index(small.flat)
    arrays, wl, tables = synth.make_sc_arrays(small, 120_000, n_cells=60, n_whitelist=3000, seed=78)
    out, st, _, _ = bench_sc.run(eng, arrays, wl, tables, 20, False, 9000, batch=50_000)
    wl_str = [bytes(r).rstrip(b'\0').decode() for r in wl]
    f = small.flat
    feats = [{'chr': f.chrom_keys[c], 'left': int(l), 'right': int(r), 'strand': '+-'[(m >> 29) & 1], 'name': 'x',
              'type': 'TE' if ((m >> 26) & 7) == 2 else 'protein_coding', 'ensg': f.names[m & ((1 << 26) - 1)]}
             for c, l, r, m in zip(f.chrom_id.tolist(), f.left.tolist(), f.right.tolist(), f.meta.tolist())]
    want, wbc, _ = orc.sc_count(feats, synth.sc_arrays_to_records(arrays, tables[0]), wl_str, 20, strand=False, bundle_threshold=9000)
    got = {}
    for c, nm, v in zip(out['coo_cell'].tolist(), out['coo_name'].tolist(), out['coo_count'].tolist()):
        got.setdefault(f.names[nm], {})[wl_str[c]] = v
    assert got == {k: v for k, v in want.items() if v}
    assert [(wl_str[c], h) for c, h in zip(out['cells'].tolist(), out['cell_hits'].tolist())] == list(wbc.items())
    eng.close()
