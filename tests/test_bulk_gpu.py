"""GPU parity for the bulk SE/PE path: libtecount (through the C ABI) against the golden vectors
captured from the reference, and against the C oracle on larger seeded inputs."""

import numpy as np
import pytest

import datagen as dg
from conftest import golden_names, load_golden
from helpers import flat_from_golden, units_from_records
from oracle import c_oracle
from te_counter_amd import Engine, FlatIndex, measureTE
from te_counter_amd.reads import ListSource

pytestmark = pytest.mark.gpu

BULK = [n for n in golden_names('bulk_') if '_err_' not in n]


@pytest.fixture(scope='module')
def engine():
    e = Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize('name', BULK)
def test_golden_through_measureTE(name, tmp_path):
    g = load_golden(name)
    m = measureTE('test', g['params']['q'])
    m.bind_genome(flat_from_golden(g))
    m.load_genome()
    src = ListSource(g['records'], batch_size=5 if 'quirks' in name else 777)
    res = (m.parse_bampe if g['params']['paired'] else m.parse_bamse)(src)
    assert m.total_reads == g['expected']['total_reads']
    assert res == g['expected']['result']
    out = str(tmp_path / 'o.tsv')
    m.save_result_bulk(res, out)
    assert open(out).read() == g['expected']['tsv']


@pytest.mark.parametrize('name', ['bulk_se_quirks', 'bulk_pe_quirks_odd', 'bulk_pe_rand11'])
def test_golden_from_a_pysam_style_file(name):
    """measureTE takes what the reference itself holds after `pysam.AlignmentFile(...)` (te_count.py:65,190)."""
    from helpers import FakeAlignmentFile
    g = load_golden(name)
    m = measureTE('test', g['params']['q'])
    m.bind_genome(flat_from_golden(g))
    m.load_genome()
    res = (m.parse_bampe if g['params']['paired'] else m.parse_bamse)(FakeAlignmentFile(g['records']))
    assert m.total_reads == g['expected']['total_reads']
    assert res == g['expected']['result']


@pytest.mark.parametrize('name,exc', [('bulk_err_enhancer', NameError), ('bulk_err_strand', NotImplementedError),
                                      ('bulk_err_qlist', TypeError), ('bulk_err_pe_mismatch', AttributeError)])
def test_failure_modes(name, exc):
    g = load_golden(name)
    assert g['expected']['exception'] == exc.__name__
    p = g['params']
    m = measureTE('test', p['q'])
    m.bind_genome(flat_from_golden(g))
    m.load_genome()
    with pytest.raises(exc):
        (m.parse_bampe if p['paired'] else m.parse_bamse)(ListSource(g['records']), strand=p.get('strand', False))


@pytest.mark.parametrize('shift', [0, 7, 9, 12, 16])
@pytest.mark.parametrize('seed,paired', [(31, False), (32, True)])
def test_random_vs_c_oracle_all_bin_shifts(engine, seed, paired, shift):
    """The result must not depend on the fine-bin width."""
    feats = dg.make_features(seed, 3000, span=400000)
    flat = FlatIndex.from_features(feats)
    units, _ = units_from_records(flat, dg.make_records(seed + 1, 40000, feats, span=400000, paired=paired), paired)
    want, wst = c_oracle.COracleIndex.from_flat(flat).bulk_count(*units, 20)
    engine.upload_index(flat, shift)
    engine.bulk_begin(20)
    for a in range(0, len(units[0]), 9973):  # ragged batches
        engine.bulk_submit(*[u[a:a + 9973] for u in units])
    got, gst = engine.bulk_finish()
    assert np.array_equal(got, want)
    assert {k: gst[k] for k in wst} == wst and gst['units'] == len(units[0])


def _dense_case(seed, n_feat, n_units, span, countable_only):
    rng = np.random.default_rng(seed)
    n_names = max(8, n_feat // 4)
    left = rng.integers(0, span, n_feat)
    ln = np.where(rng.random(n_feat) < 0.7, rng.integers(1, 400, n_feat), rng.integers(1, 30000, n_feat))
    k = rng.integers(0, span // 10000 + 1, n_feat) * 10000
    left = np.where(rng.random(n_feat) < 0.3, np.maximum(0, k + rng.integers(-2, 3, n_feat)), left)
    right = left + ln
    cls = rng.integers(1, 4, n_feat) if countable_only else rng.integers(0, 4, n_feat)
    meta = (rng.integers(0, n_names, n_feat) | (cls << 26) | (rng.integers(0, 2, n_feat) << 29)).astype(np.uint32)
    chrom = rng.integers(0, 3, n_feat)
    flat = FlatIndex(['n%07d' % i for i in range(n_names)], ['1', '2', 'X'], ['+', '-'], chrom, left, right, meta)
    anchor = rng.integers(0, n_feat, n_units)
    p1 = np.where(rng.random(n_units) < 0.6, left[anchor] + rng.integers(-2, 3, n_units), rng.integers(0, span, n_units))
    p1 = np.maximum(p1, 0)
    p2 = p1 + rng.choice([1, 2, 50, 100, 300, 9999, 10000, 10001, 25000], n_units)
    uc = np.where(rng.random(n_units) < 0.6, chrom[anchor], rng.integers(-1, 3, n_units))
    flag = np.where(rng.random(n_units) < 0.05, rng.choice([0x4, 0x200, 0x400, 0x100, 0x800], n_units), 0) | \
        np.where(rng.random(n_units) < 0.5, 0x10, 0)
    mapq = rng.choice([0, 3, 19, 20, 21, 255, 255, 255], n_units)
    return flat, (uc.astype(np.int32), p1.astype(np.int32), p2.astype(np.int32), flag.astype(np.uint16), mapq.astype(np.uint8))


@pytest.mark.parametrize('countable_only', [True, False])
def test_dense_random_2M_units(engine, countable_only):
    """Many overlapping features per position (long hit lists, both kernel variants)."""
    flat, units = _dense_case(77, 200000, 2000000, 3000000, countable_only)
    want, wst = c_oracle.COracleIndex.from_flat(flat).bulk_count(*units, 20)
    engine.upload_index(flat)
    engine.bulk_begin(20)
    engine.bulk_submit(*units)
    got, gst = engine.bulk_finish()
    assert int(want.sum()) > 1000000
    assert np.array_equal(got, want)
    assert {k: gst[k] for k in wst} == wst


def test_device_resident_submit_and_linearity(engine):
    """submit_device on HBM-resident batches; counting the same batch k times gives k x counts."""
    import torch
    flat, units = _dense_case(78, 50000, 500000, 2000000, True)
    want, _ = c_oracle.COracleIndex.from_flat(flat).bulk_count(*units, 20)
    engine.upload_index(flat)
    dev = [torch.from_numpy(u.view(np.int16) if u.dtype == np.uint16 else u).cuda() for u in units]
    torch.cuda.synchronize()
    engine.bulk_begin(20)
    for _ in range(3):
        engine.bulk_submit_device(len(units[0]), *[d.data_ptr() for d in dev])
    got, st = engine.bulk_finish()
    assert np.array_equal(got, want * 3) and st['units'] == 3 * len(units[0])
    ms, launches = engine.bulk_kernel_time()
    assert launches == 3 and ms > 0


def test_empty_and_tiny(engine):
    flat, units = _dense_case(79, 100, 1000, 100000, True)
    engine.upload_index(flat)
    engine.bulk_begin(20)
    engine.bulk_submit(*[u[:0] for u in units])
    got, st = engine.bulk_finish()
    assert got.sum() == 0 and st['units'] == 0
    engine.bulk_begin(20)
    engine.bulk_submit(*[u[:1] for u in units])
    got, st = engine.bulk_finish()
    want, _ = c_oracle.COracleIndex.from_flat(flat).bulk_count(*[u[:1] for u in units], 20)
    assert np.array_equal(got, want)


def _one_strand(flat, strand):
    sid = (flat.meta >> 29) & 7
    keep = sid == flat.strands.index(strand)
    return FlatIndex(flat.names, flat.chrom_keys, flat.strands, flat.chrom_id[keep], flat.left[keep], flat.right[keep], flat.meta[keep])


def _stranded_oracle(flat, units, q):
    """The extension's definition, restated with the pinned oracle: forward units against the '+' features only,
    reverse units against the '-' features only (te_count.py:654, 665 is the rule; bulk `--strand` itself has no
    reference output, te_count.py:58-59)."""
    rev = (units[3] & 0x10) != 0
    total, stats = None, None
    for strand, pick in (('+', ~rev), ('-', rev)):
        c, st = c_oracle.COracleIndex.from_flat(_one_strand(flat, strand)).bulk_count(*[u[pick] for u in units], q)
        total = c if total is None else total + c
        stats = st if stats is None else {k: stats[k] + st[k] for k in st}
    return total, stats


@pytest.mark.parametrize('countable_only', [True, False])
@pytest.mark.parametrize('csr_kernel', [False, True])
def test_stranded_extension_counts_per_strand(countable_only, csr_kernel, monkeypatch):
    flat, units = _dense_case(91, 60000, 400000, 2000000, countable_only)
    want, wst = _stranded_oracle(flat, units, 20)
    plain, _ = c_oracle.COracleIndex.from_flat(flat).bulk_count(*units, 20)
    assert 0 < int(want.sum()) < int(plain.sum())
    if csr_kernel:
        monkeypatch.setenv('TEC_KERNEL', '1')
    eng = Engine(0)
    try:
        split = flat.strand_split()
        assert len(split.chrom_keys) == 2 * len(flat.chrom_keys) and split.names == flat.names
        eng.upload_index(split)
        eng.bulk_begin(20)
        eng.bulk_set_stranded(True)
        for a in range(0, len(units[0]), 99991):
            eng.bulk_submit(*[u[a:a + 99991] for u in units])
        got, gst = eng.bulk_finish()
        assert np.array_equal(got, want)
        assert {k: gst[k] for k in wst} == wst
        # the switch does not outlive its run, and an index with an odd number of chromosomes is refused
        eng.upload_index(flat)
        eng.bulk_begin(20)
        with pytest.raises(Exception, match='two per chromosome'):
            eng.bulk_set_stranded(True)
        eng.bulk_submit(*units)
        got, _ = eng.bulk_finish()
        assert np.array_equal(got, plain)
    finally:
        eng.close()


@pytest.mark.parametrize('paired', [False, True])
def test_stranded_extension_through_measureTE(paired, tmp_path):
    """strand=True stays the reference's NotImplementedError unless the extension is switched on; with it, records ->
    units (PE: read 1's strand) -> per-strand counts, the same from a BAM file (GPU decoder) and from records in memory;
    an unstranded call afterwards sees the plain index again."""
    from te_counter_amd import bam
    feats = dg.make_features(41, 4000, span=600000)
    recs = dg.make_records(42, 30001, feats, span=600000, paired=paired)
    flat = FlatIndex.from_features(feats)
    units, _ = units_from_records(flat, recs, paired)
    want, _ = _stranded_oracle(flat, units, 20)
    plain, _ = c_oracle.COracleIndex.from_flat(flat).bulk_count(*units, 20)
    assert 0 < int(want.sum()) < int(plain.sum())
    m = measureTE('test', 20)
    m.bind_genome(flat)
    m.load_genome()
    call = m.parse_bampe if paired else m.parse_bamse
    with pytest.raises(NotImplementedError):
        call(ListSource(recs), strand=True)
    m.bulk_strand_extension = True
    path = str(tmp_path / 'x.bam')
    bam.write_bam(path, ListSource(recs).reference_names, recs)
    for src in (ListSource(recs, batch_size=4001), path):
        res = call(src, strand=True)
        assert [res[n] for n in flat.names] == want.tolist()
    res = call(path)
    assert [res[n] for n in flat.names] == plain.tolist()
