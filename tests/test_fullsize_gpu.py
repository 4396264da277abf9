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


@pytest.fixture(scope='module')
def mm10():
    ann = synth.make_annotation('mm10')
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


@pytest.mark.parametrize('genome,paired,n_units', [('hg38', True, 62_500_000), ('hg38', False, 100_000_000),
                                                   ('mm10', True, 62_500_000)])
def test_full_size_properties(request, genome, paired, n_units):
    """Headline shape (125 M PE records), BASELINE config 2 (100 M SE records, -q 20) and BASELINE
    config 3 (mm10 genes_tes, PE, one GPU's share of the 500 M records = 62.5 M pair units; counted
    without --strand because the reference refuses bulk --strand, te_count.py:58-59)."""
    ann, eng = request.getfixturevalue(genome)
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


def test_config1_mm10_pe_2M_records_full_oracle_and_cli(mm10, tmp_path):
    """BASELINE config 1 (test/data/bulk_rnaseq_PE.bam is a missing blob: 2 M synthetic PE records
    against the mm10-like genes_tes annotation).  Every count is compared with the C
    oracle, and the same records go BAM file -> bin/te_count's code path -> TSV, whose bytes must be
    what the reference's writer (te_count.py:279-296, restated in the oracle) gives for those counts."""
    from oracle import te_oracle as orc
    from te_counter_amd import bam, cli
    ann, eng = mm10
    n_units = 1_000_000
    units = synth.make_units(ann, n_units, seed=2001, paired=True)
    ci = c_oracle.COracleIndex.from_flat(ann.flat)
    want, wst = ci.bulk_count(*units, 20)
    got, gst = _count(eng, units, parts=3)
    assert np.array_equal(got, want) and all(gst[k] == v for k, v in wst.items())
    assert int(want.sum()) > 300_000
    # the same pairs as a BAM file through the command line
    refs = ann.chrom_names + ['chrUn_GL456239']
    tid, pos, flag, mapq = bam.units_to_records(units, len(refs) - 1, paired=True)
    path, idx, out = str(tmp_path / 'cfg1.bam'), str(tmp_path / 'mm10.tecidx'), str(tmp_path / 'cfg1.tsv')
    bam.write_bam_arrays(path, refs, tid, pos, flag, mapq, pair_names=True)
    ann.flat.save(idx)
    cli.main(['-i', path, '-o', out, '-g', idx, '-m', 'custom'])
    result = {n: int(c) for n, c in zip(ann.flat.names, want)}
    assert open(out).read() == orc.format_bulk_tsv(result, n_units + 1)


def test_config4_full_size_single_cell():
    """BASELINE config 4 at full size: 10 k cells x 5 k reads = 50 M SE records, 737 280-barcode
    whitelist, 12-mer UMIs, full hg38-like annotation, bundle threshold 1e7 (several spill epochs).
    Checked through properties: the result does not depend on how the file is cut into batches, every
    record is accounted for, per-cell totals add up; a 150 k-record run of the same generator on
    the same annotation is compared bit for bit with the Python oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import bench_sc
    ann = synth.make_annotation('hg38')
    eng = Engine(0)
    eng.upload_index(ann.flat)
    arrays, wl, tables = synth.make_sc_arrays(ann, 50_000_000, n_cells=10000, seed=2004)
    n = len(arrays['pos'])
    assert n > 49_000_000 and wl.shape[0] == 737280
    outs = []
    for batch in (1 << 22, 5_000_011):
        out, st, _, (ids, counts) = bench_sc.run(eng, arrays, wl, tables, 10000, False, 10_000_000, batch=batch)
        outs.append((out, st, ids, counts))
    out, st, ids, counts = outs[0]
    assert st['bundles'] >= 2 and st['records'] == n
    # conservation: every record is filtered by exactly one rule or survives
    assert st['qcfail'] + st['lowq'] + st['invalid_barcode'] + st['skipped_chrom'] + st['survivors'] == n
    # a (CB,UB) key has at most one bundle line per spill epoch; the reader keeps a subset of the lines
    assert st['keys'] <= st['survivors'] and st['kept_lines'] <= st['keys'] * st['bundles']
    # (a mutated off-whitelist barcode lands on another whitelist entry now and then: a few more than 10 000 cells)
    assert 10000 <= len(ids) <= 10100 and int(counts.sum()) >= st['keys']
    # per-cell totals: the COO triplets of a cell sum to at least its number of assigned UMIs
    assert len(out['cells']) <= len(ids) and int(out['coo_count'].sum()) >= st['assigned'] > 0
    assert np.all(np.diff(out['coo_cell'].astype(np.int64) * (1 << 32) + out['coo_name']) > 0)  # sorted, unique
    o2 = outs[1]
    assert o2[1] == st and np.array_equal(o2[2], ids) and np.array_equal(o2[3], counts)
    for k in ('cells', 'cell_hits', 'coo_cell', 'coo_name', 'coo_count'):
        assert np.array_equal(o2[0][k], out[k]), k
    del arrays, outs, o2
    # oracle sample: same generator, same (full) annotation, 150 k records, 40 cells, several epochs
    from oracle import te_oracle as orc
    arrays, wl, tables = synth.make_sc_arrays(ann, 150_000, n_cells=40, n_whitelist=5000, seed=2014)
    out, st, _, _ = bench_sc.run(eng, arrays, wl, tables, 25, True, 12_000, batch=40_000)
    assert st['bundles'] >= 3
    wl_str = [bytes(r).rstrip(b'\0').decode() for r in wl]
    recs = synth.sc_arrays_to_records(arrays, tables[0])
    # the oracle needs feature dicts: only features that overlap a record's window can be hit (5.7 M dicts otherwise)
    f = ann.flat
    feats = synth.features_overlapping(f, tables[1][arrays['cls']], arrays['pos'] - 8, arrays['end'] + 8)
    assert 1000 < len(feats) < 600_000
    want, wbc, _ = orc.sc_count(feats, recs, wl_str, 25, strand=True, bundle_threshold=12_000)
    got = {}
    for c, nm, v in zip(out['coo_cell'].tolist(), out['coo_name'].tolist(), out['coo_count'].tolist()):
        got.setdefault(f.names[nm], {})[wl_str[c]] = v
    assert got == {k: v for k, v in want.items() if v}
    assert [(wl_str[c], h) for c, h in zip(out['cells'].tolist(), out['cell_hits'].tolist())] == list(wbc.items())
    eng.close()


@pytest.mark.parametrize('umi_len,lower,form', [(12, False, 'one round'), (24, False, 'two rounds'), (16, True, 'two rounds')])
def test_sort_forms_agree_at_3M_records(umi_len, lower, form, monkeypatch):
    """Part 1's sort through the compact key string (symbol ranks / differing bits, one and two radix rounds) against the
    column-by-column form on 3 M records (te_count.py:371-476 needs the keys grouped and each key's records in file
    order; both forms must lead to the same cells, totals and triplets, spill epochs included)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import bench_sc
    ann = synth.make_annotation('hg38', scale=0.05)
    eng = Engine(0)
    eng.upload_index(ann.flat)
    arrays, wl, tables = synth.make_sc_arrays(ann, 3_000_000, n_cells=500, n_whitelist=100_000, seed=77, umi_len=umi_len)
    if lower:
        ub = arrays['ub24']
        ub[ub != 0] |= 0x20  # lower-case letters: outside the NUL / ACGNT alphabet
    bits = []
    end = eng.sc_end
    monkeypatch.setattr(eng, 'sc_end', lambda: (bits.append(eng.sc_sort_key_bits()), end())[1])
    out, st, _, _ = bench_sc.run(eng, arrays, wl, tables, 400, True, 400_000)
    assert {'one round': 0 < bits[-1] <= 64, 'two rounds': 64 < bits[-1] <= 96}[form], bits
    assert st['bundles'] >= 2
    monkeypatch.setenv('TEC_SC_SORT', 'columns')
    out2, st2, _, _ = bench_sc.run(eng, arrays, wl, tables, 400, True, 400_000)
    assert bits[-1] == 0 and st2 == st
    for k in ('cells', 'cell_hits', 'coo_cell', 'coo_name', 'coo_count'):
        assert np.array_equal(out2[k], out[k]), k
    eng.close()


def test_custom_3M_interval_annotation():
    """BASELINE config 5 shape: 3 M unique-name intervals (name space far beyond the LDS
    histogram: the remaining names go through the spill list, sorted and run-length encoded after every
    launch), 125 M SE units = one GPU's share of the 1 B reads; sample parity + sharding + the
    global-atomic path (TEC_SPILL=0) as a second opinion."""
    ann = synth.make_custom_annotation()
    eng = Engine(0)
    eng.upload_index(ann.flat)
    units = synth.make_units(ann, 125_000_000, seed=2005, paired=False)
    full, st = _count(eng, units, parts=2)
    idx = np.arange(0, 125_000_000, 125)
    sample = [u[idx] for u in units]
    want, wst = c_oracle.COracleIndex.from_flat(ann.flat).bulk_count(*sample, 20)
    got, gst = _count(eng, sample)
    assert np.array_equal(got, want) and all(gst[k] == v for k, v in wst.items())
    acc = np.zeros_like(full)
    for r in range(2):
        acc += _count(eng, [u[r::2] for u in units])[0]
    assert np.array_equal(acc, full)
    eng.close()
    import os
    os.environ['TEC_SPILL'] = '0'
    try:
        eng = Engine(0)
        eng.upload_index(ann.flat)
        atomics, _ = _count(eng, [u[:30_000_000] for u in units])
        eng.close()
    finally:
        del os.environ['TEC_SPILL']
    eng = Engine(0)
    eng.upload_index(ann.flat)
    assert np.array_equal(_count(eng, [u[:30_000_000] for u in units])[0], atomics)
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
