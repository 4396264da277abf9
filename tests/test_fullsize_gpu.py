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
