"""Edge cases on the GPU against the oracles: extreme coordinates, degenerate features, ragged /
empty inputs, odd barcodes and UMIs."""

import numpy as np
import pytest

from oracle import c_oracle
from oracle import te_oracle as orc
from te_counter_amd import Engine, FlatIndex, _lib, measureTE
from te_counter_amd.reads import ListSource

pytestmark = pytest.mark.gpu

INT_MAX = 2 ** 31 - 1


def test_extreme_unit_coordinates():
    """Negative, zero and INT_MAX positions, chrom ids outside the table, features with
    right < left / right == left / at 0 / huge, positions exactly on 10 kb boundaries."""
    rng = np.random.default_rng(5)
    left = np.array([0, 0, 1, 9999, 10000, 19999, 20000, 30, 500, 2 ** 31 - 5000, 9990, 10000, 29999, 0], np.int64)
    right = np.array([0, 5, 1, 10001, 10000, 20000, 29999, 20, 499, 2 ** 31 - 3, 10010, 19999, 30001, 2 ** 31 - 3], np.int64)
    n_feat = len(left)
    meta = (np.arange(n_feat) % 7 | (rng.integers(1, 4, n_feat) << 26)).astype(np.uint32)
    flat = FlatIndex(['n%d' % i for i in range(7)], ['1', '2', 'EMPTY'], ['+', '-'], np.array([0] * 10 + [1] * 4), left, right, meta)
    specials = np.array([-2 ** 31, -2, -1, 0, 1, 2, 4, 5, 6, 19, 20, 21, 29, 30, 31, 498, 499, 500, 501, 9989, 9990, 9991, 9998,
                         9999, 10000, 10001, 10002, 10009, 10010, 10011, 19998, 19999, 20000, 20001, 29998, 29999, 30000,
                         30001, 30002, INT_MAX - 5001, INT_MAX - 5000, INT_MAX - 4, INT_MAX - 3, INT_MAX - 2, INT_MAX - 1,
                         INT_MAX], np.int64)
    p1, p2 = np.meshgrid(specials, specials)
    p1, p2 = p1.ravel(), p2.ravel()
    n = len(p1)
    chrom = rng.choice([-5, -1, 0, 0, 0, 1, 1, 2, 3, 100], n).astype(np.int32)
    units = (chrom, p1.astype(np.int32), p2.astype(np.int32), np.zeros(n, np.uint16), np.full(n, 255, np.uint8))
    want, wst = c_oracle.COracleIndex.from_flat(flat).bulk_count(*units, 20)
    eng = Engine(0)
    for shift in (0, 7, 12, 15, 16, 20):
        eng.upload_index(flat, shift)
        eng.bulk_begin(20)
        eng.bulk_submit(*units)
        got, gst = eng.bulk_finish()
        assert np.array_equal(got, want), shift
        assert all(gst[k] == v for k, v in wst.items()), shift
    eng.close()


def test_index_without_hittable_features_and_single_feature():
    eng = Engine(0)
    flat = FlatIndex(['a', 'b'], ['1'], ['+', '-'], [0, 0], [50, 10], [10, 10], np.array([0 | (1 << 26), 1 | (1 << 26)], np.uint32))
    eng.upload_index(flat)
    units = (np.zeros(100, np.int32), np.arange(100, dtype=np.int32), np.arange(100, dtype=np.int32) + 5,
             np.zeros(100, np.uint16), np.full(100, 255, np.uint8))
    want, _ = c_oracle.COracleIndex.from_flat(flat).bulk_count(*units, 20)
    eng.bulk_begin(20)
    eng.bulk_submit(*units)
    got, st = eng.bulk_finish()
    assert np.array_equal(got, want) and st['units'] == 100
    eng.close()


def _sc(feats, recs, wl, tmp_path, maxcells=10, strand=False, thr=10 ** 7):
    wlf = tmp_path / 'wl.txt'
    wlf.write_text(''.join(b + '\n' for b in wl))
    m = measureTE('t', 20)
    m.BUNDLE_KEYS = thr
    m.bind_genome(FlatIndex.from_features(feats))
    res = m.sc_parse_bamse(ListSource(recs, batch_size=13), whitelistfilename=str(wlf), strand=strand, label='x', maxcells=maxcells)
    want, wbc, wtotal = orc.sc_count(feats, recs, wl, maxcells, strand=strand, bundle_threshold=thr)
    assert res == want and list(m.barcodes.items()) == list(wbc.items()) and m.total_reads == wtotal
    out = str(tmp_path / 'o.tsv')
    m.sc_save_result(res, out, maxcells=maxcells)
    assert (open(out).read(), open(out.replace('.tsv', '.barcode_freq.tsv')).read()) == orc.format_sc_tsv(want, wbc, maxcells)
    return m


def test_sc_odd_barcodes_and_umis(tmp_path):
    """10x-style '-1' suffixes, CR/UR fallbacks, N in UMIs, UMIs of different lengths (string
    order!), an over-long barcode, whitelist with duplicates and padding whitespace."""
    feats = [{'chr': 'chr1', 'left': 100, 'right': 5000, 'strand': '+', 'name': 'g', 'type': 'protein_coding', 'ensg': 'G1'},
             {'chr': 'chr1', 'left': 200, 'right': 900, 'strand': '-', 'name': 't', 'type': 'TE', 'ensg': 'LINE:L1:x'},
             {'chr': 'chr1', 'left': 300, 'right': 800, 'strand': '+', 'name': 't', 'type': 'TE', 'ensg': 'LINE:L1:x'},
             {'chr': 'chr2', 'left': 100, 'right': 900, 'strand': '-', 'name': 's', 'type': 'snRNA', 'ensg': 'S1'}]
    wl = ['ACGTACGTACGTACGT-1', 'TTTTACGTACGTACGT-1', ' AAAAACGTACGTACGT-1 ', 'ACGTACGTACGTACGT-1', 'CCCC', 'GGGGGGGGGGGGGGGGGGGGGGGG']
    umis = ['A', 'AA', 'AAA', 'AAC', 'AC', 'ANNT', 'C', 'CA', 'TTTTTTTTTTTT', 'TTTTTTTTTTTTA', 'N', 'NA']
    recs = []
    for i, u in enumerate(umis * 3):
        cb = ['ACGTACGTACGTACGT-1', 'TTTTACGTACGTACGT-1', 'AAAAACGTACGTACGT-1', 'CCCC', 'GGGGGGGGGGGGGGGGGGGGGGGG',
              'GGGGGGGGGGGGGGGGGGGGGGGGG', 'NOTONLIST'][i % 7]
        tags = {'CB': cb, 'UB': u} if i % 5 else {'CR': cb, 'UR': u}
        recs.append({'flag': 0x10 if i % 3 == 0 else 0, 'mapq': 255, 'rname': 'chr1' if i % 4 else 'chr2',
                     'pos': 150 + 37 * (i % 11), 'end': 250 + 37 * (i % 11), 'tags': tags})
    for strand in (False, True):
        _sc(feats, recs, wl, tmp_path, strand=strand, thr=7)


def test_sc_degenerate_feature_across_a_bucket_boundary(tmp_path):
    """A feature with right == left - 1 passes the single-cell 2-bp window test for a read starting
    at left - 1, but when it straddles a 10 kb boundary (left = 10000, right = 9999) the reference
    files it under no bucket at all (genelist.py:371-373) and never looks at it; the same feature
    inside one bucket (left = 10005, right = 10004) is counted."""
    feats = [{'chr': 'chr1', 'left': 10000, 'right': 9999, 'strand': '+', 'name': 'a', 'type': 'protein_coding', 'ensg': 'STRADDLE'},
             {'chr': 'chr1', 'left': 10005, 'right': 10004, 'strand': '+', 'name': 'b', 'type': 'protein_coding', 'ensg': 'INSIDE'},
             {'chr': 'chr1', 'left': 20000, 'right': 19999, 'strand': '-', 'name': 'c', 'type': 'TE', 'ensg': 'LINE:L1:s'},
             {'chr': 'chr1', 'left': 500, 'right': 900, 'strand': '+', 'name': 'd', 'type': 'protein_coding', 'ensg': 'PLAIN'}]
    wl = ['AAAA', 'CCCC']
    recs = []
    for k, (pos, end) in enumerate([(9999, 10050), (9950, 10000), (10004, 10100), (9990, 10005), (19999, 20050),
                                    (19900, 20000), (600, 700), (9998, 9999), (10000, 10001)] * 3):
        recs.append({'flag': 0, 'mapq': 255, 'rname': 'chr1', 'pos': pos, 'end': end,
                     'tags': {'CB': wl[k % 2], 'UB': 'ACGT'[k % 4] + 'ACGT'[(k // 4) % 4] + 'ACGT'[(k // 16) % 4]}})
    m = _sc(feats, recs, wl, tmp_path)
    assert 'INSIDE' in {k for k, v in orc.sc_count(feats, recs, wl, 10)[0].items() if v}
    assert m is not None


def test_sc_everything_filtered_raises_like_the_reference(tmp_path):
    feats = [{'chr': 'chr1', 'left': 100, 'right': 500, 'strand': '+', 'name': 'g', 'type': 'protein_coding', 'ensg': 'G1'}]
    recs = [{'flag': 0x4, 'mapq': 255, 'rname': 'chr1', 'pos': 150, 'end': 250, 'tags': {'CB': 'AAAA', 'UB': 'CC'}},
            {'flag': 0, 'mapq': 3, 'rname': 'chr1', 'pos': 150, 'end': 250, 'tags': {'CB': 'AAAA', 'UB': 'CC'}},
            {'flag': 0, 'mapq': 255, 'rname': 'chr1_random', 'pos': 150, 'end': 250, 'tags': {'CB': 'AAAA', 'UB': 'CC'}}]
    wlf = tmp_path / 'wl.txt'
    wlf.write_text('AAAA\n')
    for rs in (recs, []):
        m = measureTE('t', 20)
        m.bind_genome(FlatIndex.from_features(feats))
        with pytest.raises(ZeroDivisionError):
            m.sc_parse_bamse(ListSource(rs), whitelistfilename=str(wlf), label='x', maxcells=5)
        with pytest.raises(ZeroDivisionError):
            orc.sc_count(feats, rs, ['AAAA'], 5)


def test_many_chromosomes():
    """More chromosomes than the kernel keeps in LDS (custom GTFs with thousands of contigs)."""
    rng = np.random.default_rng(9)
    n_chrom, n_feat, n_units = 700, 20000, 300000
    chrom = rng.integers(0, n_chrom, n_feat)
    left = rng.integers(0, 50000, n_feat)
    right = left + rng.integers(1, 3000, n_feat)
    meta = (rng.integers(0, 500, n_feat) | (rng.integers(1, 4, n_feat) << 26)).astype(np.uint32)
    flat = FlatIndex(['n%03d' % i for i in range(500)], ['c%d' % i for i in range(n_chrom)], ['+', '-'], chrom, left, right, meta)
    uc = rng.integers(-1, n_chrom + 2, n_units).astype(np.int32)
    p1 = rng.integers(0, 52000, n_units).astype(np.int32)
    units = (uc, p1, (p1 + rng.integers(1, 400, n_units)).astype(np.int32), np.zeros(n_units, np.uint16), np.full(n_units, 255, np.uint8))
    want, wst = c_oracle.COracleIndex.from_flat(flat).bulk_count(*units, 20)
    eng = Engine(0)
    eng.upload_index(flat)
    eng.bulk_begin(20)
    eng.bulk_submit(*units)
    got, gst = eng.bulk_finish()
    assert np.array_equal(got, want) and all(gst[k] == v for k, v in wst.items())
    eng.close()


@pytest.mark.parametrize('mixed', [False, True])
@pytest.mark.parametrize('paired', [True, False])
def test_dense_pileups_long_lists_and_many_names(mixed, paired):
    """Fine bins with dozens of entries (continuation pieces AND the walk beyond them), units that
    hit more than six distinct names (exact walk), edges on and next to 10 kb boundaries, in every
    record-kernel variant; against the C oracle."""
    rng = np.random.default_rng(29 + mixed + 2 * paired)
    n_names = 90
    feats = []
    # pile-ups: around a few centres, 40-120 features of many names overlapping within ~1.5 kb
    centres = [9_000, 10_000, 19_990, 20_000, 30_010, 55_555, 99_999, 100_000, 250_000]
    for c in centres:
        for _ in range(int(rng.integers(40, 120))):
            left = int(c + rng.integers(-900, 900))
            right = left + int(rng.integers(1, 700))
            feats.append((0, max(0, left), right, int(rng.integers(0, n_names))))
    # a sparse background on two chromosomes
    for _ in range(3000):
        left = int(rng.integers(0, 400_000))
        feats.append((int(rng.integers(0, 2)), left, left + int(rng.integers(1, 3000)), int(rng.integers(0, n_names))))
    classes = [_lib.CLASS_GENE, _lib.CLASS_TE, _lib.CLASS_SNRNA] + ([_lib.CLASS_OTHER] if mixed else [])
    name_class = rng.integers(0, len(classes), n_names)
    chrom = np.array([f[0] for f in feats], np.int32)
    left = np.array([f[1] for f in feats], np.int32)
    right = np.array([f[2] for f in feats], np.int32)
    meta = np.array([f[3] | (classes[name_class[f[3]]] << _lib.META_CLASS_SHIFT) for f in feats], np.uint32)
    flat = FlatIndex(['N%03d' % i for i in range(n_names)], ['1', '2'], ['+', '-'], chrom, left, right, meta)
    n = 400_000
    c = rng.choice(centres, n)
    p1 = (c + rng.integers(-1200, 1200, n)).astype(np.int32)
    p1[::7] = (rng.integers(0, 40, len(p1[::7])) * 10_000 + rng.integers(-2, 3, len(p1[::7]))).astype(np.int32)
    p2 = (p1 + (rng.integers(-400, 400, n) if paired else rng.integers(1, 2500, n))).astype(np.int32)
    p2[::11] = (rng.integers(1, 40, len(p2[::11])) * 10_000 - 1 + rng.integers(-2, 3, len(p2[::11]))).astype(np.int32)
    units = (rng.integers(0, 2, n).astype(np.int32), p1, p2, np.zeros(n, np.uint16), np.full(n, 255, np.uint8))
    want, wst = c_oracle.COracleIndex.from_flat(flat).bulk_count(*units, 20)
    assert want.max() > 0
    for shift in (0, 7, 9, 12):
        eng = Engine(0)
        eng.upload_index(flat, shift)
        eng.bulk_begin(20)
        eng.bulk_submit(*units)
        got, gst = eng.bulk_finish()
        eng.close()
        assert np.array_equal(got, want), (shift, int(np.abs(got.astype(np.int64) - want.astype(np.int64)).sum()))
        assert all(gst[k] == v for k, v in wst.items())
