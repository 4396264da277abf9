"""The packed 16-bit LDS counters of the bulk kernel are never flushed inside a launch: the add
that carries out of a half books it on the global counter (csrc/bulk.hip, wrap_account).  These
cases make single workgroups give far more than 65 535 hits to a few names that share LDS words
(the hottest names are internal ids 0 and 1: one word), so that low-half carries, high-half
wraps and both at once all occur, in every kernel variant, and check the totals exactly."""

import numpy as np
import pytest

from oracle import c_oracle
from te_counter_amd import Engine, _lib
from te_counter_amd.index import FlatIndex

pytestmark = pytest.mark.gpu

# unit kinds (p1, p2): A only | A+B | A+B+C | A+B+C+D | E (the last name) | nothing
KINDS = np.array([(1200, 1300), (1600, 1700), (2100, 2200), (2600, 2700), (50100, 50200), (20000, 20100)], np.int32)
N = 100_000_000  # <= 512 workgroups -> >= 195 k units each: A, B, C wrap twice per workgroup, D once


def _annotation(odd_names):
    # four overlapping features on chr1 (two genes, two TEs) + far away genes; internal ids are
    # ordered by covered bp, so G_A/G_B and T_C/T_D pair up in LDS words
    names = ['G_A', 'G_B', 'T_C', 'T_D', 'Z_E'] + ([] if odd_names else ['Z_F'])
    feats = [(0, 1000, 9000, 0, _lib.CLASS_GENE), (0, 1500, 9400, 1, _lib.CLASS_GENE), (0, 2000, 5000, 2, _lib.CLASS_TE),
             (0, 2500, 4500, 3, _lib.CLASS_TE), (0, 50000, 50400, 4, _lib.CLASS_GENE)]
    if not odd_names:
        feats.append((0, 60000, 60100, 5, _lib.CLASS_GENE))
    chrom = np.array([f[0] for f in feats], np.int32)
    left = np.array([f[1] for f in feats], np.int32)
    right = np.array([f[2] for f in feats], np.int32)
    meta = np.array([f[3] | (f[4] << _lib.META_CLASS_SHIFT) for f in feats], np.uint32)
    return FlatIndex(names, ['1'], ['+', '-'], chrom, left, right, meta)


@pytest.fixture(scope='module')
def units():
    rng = np.random.default_rng(11)
    k = rng.choice(len(KINDS), N, p=[0.05, 0.10, 0.25, 0.50, 0.05, 0.05]).astype(np.int8)
    u = (np.zeros(N, np.int32), KINDS[k, 0].copy(), KINDS[k, 1].copy(), np.zeros(N, np.uint16), np.full(N, 255, np.uint8))
    return u, np.bincount(k, minlength=len(KINDS))


@pytest.mark.parametrize('odd_names', [False, True])
@pytest.mark.parametrize('env', [{}, {'TEC_UNROLL': '1'}, {'TEC_KERNEL': '1'}, {'TEC_AGG': '1'}, {'TEC_FORCE_MIXED': '1'}])
def test_counters_carry_exactly(units, odd_names, env, monkeypatch):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    units, mult = units
    flat = _annotation(odd_names)
    orc = c_oracle.COracleIndex.from_flat(flat)
    one = lambda i: orc.bulk_count(np.zeros(1, np.int32), KINDS[i:i + 1, 0].copy(), KINDS[i:i + 1, 1].copy(),
                                   np.zeros(1, np.uint16), np.full(1, 255, np.uint8), 20)[0]
    want = sum(int(m) * one(i).astype(np.int64) for i, m in enumerate(mult))
    assert want[0] > 256 * 3 * 65536  # the case is what it claims to be
    sample = [u[:200_000] for u in units]
    want_s, _ = orc.bulk_count(*sample, 20)
    eng = Engine(0)
    eng.upload_index(flat)
    eng.bulk_begin(20)
    eng.bulk_submit(*sample)
    got_s, _ = eng.bulk_finish()
    assert np.array_equal(got_s, want_s)
    for parts in (1, 3):
        eng.bulk_begin(20)
        step = (N + parts - 1) // parts
        for a in range(0, N, step):
            eng.bulk_submit(*[u[a:a + step] for u in units])
        got, st = eng.bulk_finish()
        assert np.array_equal(got.astype(np.int64), want), (env, parts, got, want)
        assert st['units'] == N
    eng.close()
