"""The CPU oracle (oracle/te_oracle.py) against golden vectors captured from the real
reference (tests/golden/make_golden.py).  Runs anywhere, no GPU, no /root/reference."""

import pytest

from conftest import golden_names, load_golden
from oracle import te_oracle as orc

BULK = [n for n in golden_names('bulk_') if '_err_' not in n]
SC = [n for n in golden_names('sc_') if '_err_' not in n]
ERR = golden_names('bulk_err_') + golden_names('sc_err_')


def test_fixture_inventory():
    assert len(BULK) >= 11 and len(SC) >= 10 and len(ERR) >= 8


@pytest.mark.parametrize('name', BULK)
def test_bulk(name):
    g = load_golden(name)
    res, total, _ = orc.bulk_count(g['features'], g['records'], g['params']['paired'], g['params']['q'])
    assert total == g['expected']['total_reads']
    assert res == g['expected']['result']
    assert orc.format_bulk_tsv(res, total) == g['expected']['tsv']


@pytest.mark.parametrize('name', SC)
def test_sc(name):
    g = load_golden(name)
    p = g['params']
    res, bcs, total = orc.sc_count(g['features'], g['records'], g['whitelist'], p['maxcells'], p['q'],
                                   strand=p['strand'], bundle_threshold=p['bundle_threshold'])
    assert total == g['expected']['total_reads']
    assert res == g['expected']['result']
    assert [list(kv) for kv in bcs.items()] == g['expected']['barcodes']
    tsv, freq = orc.format_sc_tsv(res, bcs, p['maxcells'])
    assert tsv == g['expected']['tsv']
    assert freq == g['expected']['barcode_freq']


@pytest.mark.parametrize('name', ERR)
def test_failure_modes(name):
    g = load_golden(name)
    p = g['params']
    want = g['expected']['exception']
    assert want is not None
    with pytest.raises(BaseException) as ei:
        if g['kind'].startswith('bulk'):
            orc.bulk_count(g['features'], g['records'], p['paired'], p['q'], strand=p.get('strand', False))
        else:
            orc.sc_count(g['features'], g['records'], p['whitelist'], p['maxcells'], strand=p['strand'],
                         umis=p.get('umis', True))
    assert type(ei.value).__name__ == want or want in [c.__name__ for c in type(ei.value).__mro__]


def test_quirks_are_present():
    """The bucket-candidate misses of SURVEY.md 3.4 note (iii) really are in the fixture."""
    g = load_golden('bulk_se_quirks')
    res = g['expected']['result']
    # reads 20000-31000, 20000-20050 and 19999-20001 all have an edge inside G1 20000-20100;
    # the first is not a bucket candidate (b1=1, b2=3, G1 sits in bucket 2 only) -> 2, not 3
    assert res['ENSG00001'] == 2
    # read 10000-19999 vs G2 10000-19999: b1=0, b2=2, G2 sits in bucket 1 only -> only 10000-10050 counts
    assert res['ENSG00002'] == 1
    assert g['expected']['total_reads'] == len(g['records']) + 1


@pytest.mark.parametrize('name', ['custom_e2e_se', 'custom_e2e_pe', 'glb_e2e_se'])
def test_index_end_to_end_fixtures(name):
    """GTF -> make_custom -> count and .glb -> count as the reference ran them: this build's GTF reader gives the
    reference's rows, its .glb importer the reference's features, and the oracle on them the reference's TSV."""
    import gzip
    import json
    import os
    from conftest import GOLDEN_DIR
    from te_counter_amd import genome
    from te_counter_amd.index import FlatIndex
    g = load_golden(name)
    if g['kind'] == 'custom_e2e':
        feats = genome.custom_features(os.path.join(GOLDEN_DIR, g['gtf_file']))
        assert feats == g['expected']['rows']
    else:
        feats = json.load(gzip.open(os.path.join(GOLDEN_DIR, 'ref_built_index.features.json.gz'), 'rt'))
        flat = FlatIndex.from_glb(os.path.join(GOLDEN_DIR, g['glb_file']))
        assert flat.names == sorted(set(f['ensg'] for f in feats))
    res, total, _ = orc.bulk_count(feats, g['records'], g['params']['paired'], g['params']['q'])
    assert total == g['expected']['total_reads'] and res == g['expected']['result']
    assert orc.format_bulk_tsv(res, total) == g['expected']['tsv']
