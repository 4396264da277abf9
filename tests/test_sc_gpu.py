"""GPU parity for the single-cell path: measureTE.sc_parse_bamse / sc_save_result on the MI355X
against the golden vectors captured from the reference, and against the Python oracle on
seeded inputs (several bundle thresholds, binding cell selection, ragged batches)."""

import pytest

import datagen as dg
from conftest import golden_names, load_golden
from helpers import flat_from_golden
from oracle import te_oracle as orc
from te_counter_amd import FlatIndex, measureTE
from te_counter_amd.reads import ListSource

pytestmark = pytest.mark.gpu

SC = [n for n in golden_names('sc_') if '_err_' not in n]


def run_sc(flat, records, whitelist, maxcells, strand, q, bundle_keys, tmp_path, batch_size=997, umis=True, source=None):
    wl = tmp_path / 'wl.txt'
    wl.write_text(''.join(b + '\n' for b in whitelist))
    m = measureTE('test', q)
    m.BUNDLE_KEYS = bundle_keys
    m.bind_genome(flat)
    res = m.sc_parse_bamse(source or ListSource(records, batch_size=batch_size), UMIS=umis, whitelistfilename=str(wl),
                           strand=strand, label='lab', maxcells=maxcells)
    out = str(tmp_path / 'o.tsv')
    m.sc_save_result(res, out, maxcells=maxcells)
    return m, res, open(out).read(), open(out.replace('.tsv', '.barcode_freq.tsv')).read()


@pytest.mark.parametrize('name', SC)
def test_golden(name, tmp_path):
    g = load_golden(name)
    p = g['params']
    m, res, tsv, freq = run_sc(flat_from_golden(g), g['records'], g['whitelist'], p['maxcells'], p['strand'], p['q'],
                               p['bundle_threshold'], tmp_path)
    assert m.total_reads == g['expected']['total_reads']
    assert [list(kv) for kv in m.barcodes.items()] == g['expected']['barcodes']
    assert res == g['expected']['result']
    assert tsv == g['expected']['tsv']
    assert freq == g['expected']['barcode_freq']


@pytest.mark.parametrize('name', ['sc_rand21_strand', 'sc_topcells_binding_multibundle'])
def test_golden_from_a_pysam_style_file(name, tmp_path):
    """sc_parse_bamse over what the reference holds after `pysam.AlignmentFile(...)` (te_count.py:351)."""
    from helpers import FakeAlignmentFile
    g = load_golden(name)
    p = g['params']
    m, res, tsv, freq = run_sc(flat_from_golden(g), g['records'], g['whitelist'], p['maxcells'], p['strand'], p['q'],
                               p['bundle_threshold'], tmp_path, source=FakeAlignmentFile(g['records']))
    assert m.total_reads == g['expected']['total_reads']
    assert res == g['expected']['result']
    assert tsv == g['expected']['tsv']


@pytest.mark.parametrize('name,exc', [('sc_err_no_strand_key', KeyError), ('sc_err_noumi', ZeroDivisionError),
                                      ('sc_err_no_cb', AssertionError), ('sc_err_no_ub', AssertionError)])
def test_failure_modes(name, exc, tmp_path):
    g = load_golden(name)
    assert g['expected']['exception'] == exc.__name__
    p = g['params']
    with pytest.raises(exc):
        run_sc(flat_from_golden(g), g['records'], p['whitelist'], p['maxcells'], p['strand'], 20, 10 ** 7, tmp_path,
               umis=p.get('umis', True))


@pytest.mark.parametrize('seed', [41, 42])
@pytest.mark.parametrize('strand', [False, True])
@pytest.mark.parametrize('thr', [10 ** 7, 700, 97])
def test_random_vs_oracle(seed, strand, thr, tmp_path):
    types = ['protein_coding', 'lncRNA', 'TE', 'TE', 'snRNA', 'enhancer', 'miRNA']
    feats = dg.make_features(seed, 1500, span=300000, types=types)
    wl = dg.make_whitelist(seed, 400)
    recs = dg.make_sc_records(seed + 1, 20000, wl, feats, span=300000, n_cells=60)
    want_res, want_bc, want_total = orc.sc_count(feats, recs, wl, 25, strand=strand, bundle_threshold=thr)
    m, res, tsv, freq = run_sc(FlatIndex.from_features(feats), recs, wl, 25, strand, 20, thr, tmp_path, batch_size=4093)
    assert m.total_reads == want_total
    assert list(m.barcodes.items()) == list(want_bc.items())
    assert res == want_res
    assert (tsv, freq) == orc.format_sc_tsv(want_res, want_bc, 25)


WIDE = str.maketrans('ACGTN', 'acgtn')  # UMI bytes outside the NUL / ACGNT alphabet: five bits of a byte differ


@pytest.mark.parametrize('umi_len,wide,env,form', [(10, False, {}, 'one round'), (24, False, {}, 'two rounds'),
                                                   (10, True, {}, 'one round'), (14, True, {}, 'two rounds'),
                                                   (24, True, {}, 'columns'), (10, False, {'TEC_SC_SORT': 'columns'}, 'columns'),
                                                   (-16, False, {}, 'one round'), (-16, True, {}, 'two rounds')])
@pytest.mark.parametrize('thr', [10 ** 7, 450])
def test_every_form_of_the_part1_sort(umi_len, wide, env, form, thr, tmp_path, monkeypatch):
    """Part 1 orders the survivors by (barcode, UMI) through a key string that leaves out what is equal in all of them
    (symbol ranks for UMIs over NUL / ACGNT, the differing bits otherwise): up to 64 bits in one radix round, up to 96 in
    two, beyond that (or on request) column by column.  Every form must give the reference's result (te_count.py:371-476:
    the dict keyed by 'CB-UB' and its sorted bundles)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    types = ['protein_coding', 'lncRNA', 'TE', 'TE', 'snRNA', 'enhancer', 'miRNA']
    feats = dg.make_features(77, 1200, span=250000, types=types)
    wl = dg.make_whitelist(77, 300)
    recs = dg.make_sc_records(78, 12000, wl, feats, span=250000, n_cells=40, umi_len=abs(umi_len))
    if umi_len < 0:  # ragged: UMIs of 14..16 letters (a shorter one is a prefix of nothing else: NUL pads take part in the order)
        for r in recs:
            r['tags'] = {k: (v[:len(v) - sum(map(ord, v)) % 3] if k in ('UB', 'UR') else v) for k, v in r['tags'].items()}
    if wide:
        for r in recs:
            r['tags'] = {k: (v.translate(WIDE) if k in ('UB', 'UR') else v) for k, v in r['tags'].items()}
    want_res, want_bc, want_total = orc.sc_count(feats, recs, wl, 20, strand=True, bundle_threshold=thr)
    m, res, tsv, freq = run_sc(FlatIndex.from_features(feats), recs, wl, 20, True, 20, thr, tmp_path, batch_size=2999)
    bits = m.stats['sort_key_bits']
    assert {'one round': 0 < bits <= 64, 'two rounds': 64 < bits <= 96, 'columns': bits == 0}[form], bits
    assert m.total_reads == want_total
    assert list(m.barcodes.items()) == list(want_bc.items())
    assert res == want_res
    assert (tsv, freq) == orc.format_sc_tsv(want_res, want_bc, 20)


def test_sort_key_bits_needs_the_sort(tmp_path):
    """tec_sc_sort_key_bits before tec_sc_part1_sort is a state error, not a stale number."""
    import numpy as np
    from te_counter_amd import Engine
    feats = dg.make_features(5, 50)
    eng = Engine(0)
    eng.upload_index(FlatIndex.from_features(feats))
    wl24 = np.zeros((2, 24), np.uint8)
    wl24[0, :4] = np.frombuffer(b'AAAC', np.uint8)
    wl24[1, :4] = np.frombuffer(b'AAAG', np.uint8)
    eng.sc_begin(False, 20, wl24, 10 ** 7, np.zeros(1, np.int32), np.zeros(1, np.uint8))
    from te_counter_amd._lib import TecError
    with pytest.raises(TecError):
        eng.sc_sort_key_bits()
    assert eng.sc_part1_sort() == 0 and eng.sc_sort_key_bits() == 0  # no survivors: nothing was sorted
    eng.sc_end()
    eng.close()


@pytest.mark.parametrize('world', [2, 3])
@pytest.mark.parametrize('name', ['sc_rand21_strand_multibundle', 'sc_rand22_nostrand_multibundle', 'sc_topcells_binding_multibundle',
                                  'sc_rand22_strand', 'sc_degenerate_boundary'])
def test_cells_dealt_over_several_gpu_contexts(name, world, tmp_path):
    """The multi-GPU single-cell decomposition with the REAL kernels: N engine contexts on this GPU, one per
    rank (threads), each decoding its own byte range of one BAM file on the device; survivors exchanged by
    barcode id, bundle boundaries by distributed bisection, global selection and presence tables -- merged
    result, barcode order, TSV and barcode_freq equal the reference's goldens."""
    import threading
    from helpers import ThreadWorld
    from te_counter_amd import bam
    g = load_golden(name)
    p = g['params']
    flat = flat_from_golden(g)
    wlf = tmp_path / 'wl.txt'
    wlf.write_text(''.join(b + '\n' for b in g['whitelist']))
    path = str(tmp_path / 'in.bam')
    bam.write_bam(path, ListSource(g['records']).reference_names, g['records'], block_bytes=2500)
    shared = ThreadWorld.Shared(world)
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            m = measureTE('t', p['q'], world=ThreadWorld(r, shared))
            m.BUNDLE_KEYS = p['bundle_threshold']
            m.bind_genome(flat)
            m.load_genome()
            host_bytes = []
            end = m._engine.sc_end
            m._engine.sc_end = lambda: (host_bytes.append(m._engine.sc_host_row_bytes()), end())[1]
            res = m.sc_parse_bamse(path, whitelistfilename=str(wlf), strand=p['strand'], label='lab', maxcells=p['maxcells'])
            assert host_bytes == [0]  # the survivors changed contexts in HBM: no column of them went through host memory
            if r == 0:
                o = str(tmp_path / 'o.tsv')
                m.sc_save_result(res, o, maxcells=p['maxcells'])
                out[0] = (m, res, open(o).read(), open(o.replace('.tsv', '.barcode_freq.tsv')).read())
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
            shared.barrier.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    m, res, tsv, freq = out[0]
    assert m.total_reads == g['expected']['total_reads']
    assert [list(kv) for kv in m.barcodes.items()] == g['expected']['barcodes']
    assert res == g['expected']['result']
    assert tsv == g['expected']['tsv'] and freq == g['expected']['barcode_freq']
    if 'multibundle' in name:
        assert m.stats['bundles'] > 1
