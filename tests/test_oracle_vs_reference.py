"""Live check of the oracle against the real reference on fresh random inputs.
Only runs where /root/reference is mounted (the build container); skipped elsewhere."""

import pytest

import datagen as dg
from oracle import ref_harness as rh
from oracle import te_oracle as orc

pytestmark = pytest.mark.skipif(not rh.reference_available(), reason='reference not mounted')


@pytest.mark.parametrize('seed', [1, 2, 3])
@pytest.mark.parametrize('paired', [False, True])
def test_bulk_live(seed, paired):
    feats = dg.make_features(seed, 400)
    recs = dg.make_records(seed + 50, 3000, feats, paired=paired)
    ref = rh.run_bulk(feats, recs, paired)
    res, total, _ = orc.bulk_count(feats, recs, paired)
    assert (res, total) == (ref['result'], ref['total_reads'])
    assert orc.format_bulk_tsv(res, total) == ref['tsv']


@pytest.mark.parametrize('seed', [4, 5])
@pytest.mark.parametrize('strand', [False, True])
@pytest.mark.parametrize('thr', [None, 120])
def test_sc_live(seed, strand, thr):
    feats = dg.make_features(seed, 250, types=['protein_coding', 'lncRNA', 'TE', 'TE', 'snRNA', 'enhancer', 'miRNA'])
    wl = dg.make_whitelist(seed, 100)
    recs = dg.make_sc_records(seed + 70, 2500, wl, feats, n_cells=10)
    ref = rh.run_sc(feats, recs, wl, 6, strand=strand, bundle_threshold=thr)
    res, bcs, total = orc.sc_count(feats, recs, wl, 6, strand=strand, bundle_threshold=thr or 10 ** 7)
    assert res == ref['result'] and total == ref['total_reads']
    assert list(bcs.items()) == ref['barcodes']
    assert orc.format_sc_tsv(res, bcs, 6) == (ref['tsv'], ref['barcode_freq'])


@pytest.mark.parametrize('seed', [5, 6])
def test_custom_gtf_live(seed, tmp_path):
    from te_counter_amd import genome
    p = tmp_path / 'x.gtf'
    p.write_text(dg.make_gtf_text(seed, 50))
    assert genome.custom_features(str(p)) == rh.run_make_custom(str(p))
