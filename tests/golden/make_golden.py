#!/usr/bin/env python3
"""Regenerates tests/golden/*.json.gz by running the REAL reference (mounted at
/root/reference) through oracle/ref_harness.py.  Only runs in the build container; the
fixtures it writes are pure data (inputs + the reference's outputs) and are what travels.

    python tests/golden/make_golden.py
"""

import gzip
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import datagen as dg  # noqa: E402
from oracle import ref_harness as rh  # noqa: E402


def dump(name, obj):
    path = os.path.join(HERE, name + '.json.gz')
    raw = json.dumps(obj, sort_keys=True, separators=(',', ':')).encode()
    with open(path, 'wb') as fh:  # mtime=0 -> byte-stable output
        with gzip.GzipFile(fileobj=fh, mode='wb', mtime=0) as gz:
            gz.write(raw)
    print('%-28s %8d bytes' % (name, os.path.getsize(path)))


def bulk_case(name, feats, recs, paired, q=20):
    ref = rh.run_bulk(feats, recs, paired, quality_threshold=q)
    dump(name, {'kind': 'bulk', 'features': feats, 'records': recs,
                'params': {'paired': paired, 'q': q},
                'expected': {'result': ref['result'], 'total_reads': ref['total_reads'], 'tsv': ref['tsv']}})


def sc_case(name, feats, recs, wl, maxcells, strand, thr=None, q=20):
    ref = rh.run_sc(feats, recs, wl, maxcells, quality_threshold=q, strand=strand, bundle_threshold=thr)
    dump(name, {'kind': 'sc', 'features': feats, 'records': recs, 'whitelist': wl,
                'params': {'maxcells': maxcells, 'strand': strand, 'q': q,
                           'bundle_threshold': thr if thr is not None else 10000000},
                'expected': {'result': ref['result'], 'barcodes': ref['barcodes'],
                             'total_reads': ref['total_reads'], 'tsv': ref['tsv'],
                             'barcode_freq': ref['barcode_freq']}})


def error_case(name, kind, feats, recs, params, fn):
    try:
        fn()
        exc = None
    except BaseException as e:  # the reference dies in various ways, record which
        exc = type(e).__name__
    dump(name, {'kind': kind + '_error', 'features': feats, 'records': recs, 'params': params,
                'expected': {'exception': exc}})


def quirk_features():
    g = lambda i, l, r, t='protein_coding', s='+', c='chr1': {  # noqa: E731
        'chr': c, 'left': l, 'right': r, 'strand': s, 'name': 'G%d' % i, 'type': t, 'ensg': 'ENSG%05d' % i}
    te = lambda n, l, r, s='+', c='chr1': {  # noqa: E731
        'chr': c, 'left': l, 'right': r, 'strand': s, 'name': n, 'type': 'TE', 'ensg': n}
    return [
        g(1, 20000, 20100),            # missed by a read 20000-31000 (bucket-candidate quirk)
        g(2, 10000, 19999),            # missed by a read 10000-19999
        g(3, 5000, 5300), te('SINE:Alu:AluY', 5100, 5400), te('SINE:Alu:AluY', 5150, 5350, '-'),
        g(4, 40000, 40001), g(5, 49999, 50000), g(6, 50000, 50001), g(7, 59998, 59999),
        g(8, 0, 5), g(9, 1, 2),
        g(10, 70000, 95000, 'lncRNA'), g(11, 70500, 70600, 'miRNA'), g(12, 71000, 71100, 'snRNA'),
        te('LINE:L1:L1HS', 100, 6200, '-', 'chrX'), g(13, 150, 160, 'lincRNA', '-', 'chrX'),
        g(14, 30, 20),                  # right < left: listed in a bucket, can never hit
        g(15, 300, 400, c='chrUn_GL000220'), g(16, 300, 400, c='chrM'), g(17, 300, 400, c='MT'),
        g(18, 120000, 120050, 'miRNA'),
    ]


def quirk_records():
    def r(rname, pos, end, flag=0, mapq=255, qn='q'):
        return {'flag': flag, 'mapq': mapq, 'qname': qn, 'rname': rname, 'pos': pos, 'end': end}
    recs = [
        r('chr1', 20000, 31000), r('chr1', 10000, 19999), r('chr1', 20000, 20050), r('chr1', 10000, 10050),
        r('chr1', 19999, 20001), r('chr1', 5200, 5250), r('chr1', 5120, 5160), r('chr1', 4000, 6000),
        r('chr1', 40000, 40001), r('chr1', 39999, 40001), r('chr1', 49999, 50001), r('chr1', 50000, 50001),
        r('chr1', 59998, 59999), r('chr1', 59997, 59999), r('chr1', 0, 1), r('chr1', 0, 5), r('chr1', 1, 2),
        r('chr1', 0, 0), r('chr1', 69999, 70001), r('chr1', 70550, 70560), r('chr1', 71050, 71060),
        r('chr1', 94999, 95001), r('chr1', 80000, 120020), r('chr1', 120010, 120020),
        r('chrX', 155, 158), r('chrX', 155, 158, 0x10), r('X', 155, 158), r('chrx', 155, 158),
        r('chr1', 25, 28), r('chrUn_GL000220', 320, 340), r('chrM', 320, 340), r('chrMT', 320, 340), r('MT', 320, 340),
        r('chr1', 5200, 5250, 0x4), r('chr1', 5200, 5250, 0x400), r('chr1', 5200, 5250, 0x200),
        r('chr1', 5200, 5250, 0x100), r('chr1', 5200, 5250, 0x800), r('chr1', 5200, 5250, 0, 19),
        r('chr1', 5200, 5250, 0, 20), r('chr1', 5200, 5250, 0, 0), r('chr9', 5200, 5250),
    ]
    return recs


def main():
    if not rh.reference_available():
        sys.exit('needs /root/reference')
    # ---------------- bulk
    qf, qr = quirk_features(), quirk_records()
    bulk_case('bulk_se_quirks', qf, qr, False)
    bulk_case('bulk_pe_quirks', qf, qr, True)
    bulk_case('bulk_pe_quirks_odd', qf, qr[:-1], True)
    bulk_case('bulk_se_empty', qf, [], False)
    bulk_case('bulk_pe_one_record', qf, qr[:1], True)
    for seed in (11, 12):
        feats = dg.make_features(seed, 500)
        bulk_case('bulk_se_rand%d' % seed, feats, dg.make_records(seed + 100, 4000, feats), False)
        bulk_case('bulk_pe_rand%d' % seed, feats, dg.make_records(seed + 200, 4000, feats, paired=True), True)
    feats = dg.make_features(13, 300)
    bulk_case('bulk_se_q0', feats, dg.make_records(313, 1500, feats), False, q=0)
    bulk_case('bulk_se_q255', feats, dg.make_records(314, 1500, feats), False, q=255)

    # ---------------- bulk failure modes
    enh = [{'chr': 'chr1', 'left': 100, 'right': 300, 'name': 'e1', 'type': 'enhancer', 'ensg': 'e1'},
           {'chr': 'chr1', 'left': 1000, 'right': 1300, 'strand': '+', 'name': 'g1', 'type': 'protein_coding', 'ensg': 'g1'}]
    hit_enh = [{'flag': 0, 'mapq': 255, 'qname': 'q', 'rname': 'chr1', 'pos': 150, 'end': 200}]
    hit_gene = [{'flag': 0, 'mapq': 255, 'qname': 'q', 'rname': 'chr1', 'pos': 1150, 'end': 1200}]
    error_case('bulk_err_enhancer', 'bulk', enh, hit_enh, {'paired': False, 'q': 20},
               lambda: rh.run_bulk(enh, hit_enh, False))
    error_case('bulk_err_strand', 'bulk', enh, hit_gene, {'paired': False, 'q': 20, 'strand': True},
               lambda: rh.run_bulk(enh, hit_gene, False, strand=True))
    error_case('bulk_err_qlist', 'bulk', enh, hit_gene, {'paired': False, 'q': [20]},
               lambda: rh.run_bulk(enh, hit_gene, False, quality_threshold=[20]))
    mism = [dict(hit_gene[0], qname='a/1'), dict(hit_gene[0], qname='b/2')]
    error_case('bulk_err_pe_mismatch', 'bulk', enh, mism, {'paired': True, 'q': 20},
               lambda: rh.run_bulk(enh, mism, True))

    # ---------------- single cell
    sc_types = ['protein_coding', 'lncRNA', 'TE', 'TE', 'snRNA', 'enhancer', 'miRNA']
    for seed, maxcells in ((21, 50), (22, 5)):
        feats = dg.make_features(seed, 300, types=sc_types)
        wl = dg.make_whitelist(seed, 200)
        recs = dg.make_sc_records(seed + 100, 3000, wl, feats, n_cells=14)
        for strand in (False, True):
            sc_case('sc_rand%d_%s' % (seed, 'strand' if strand else 'nostrand'), feats, recs, wl, maxcells, strand)
            sc_case('sc_rand%d_%s_multibundle' % (seed, 'strand' if strand else 'nostrand'),
                    feats, recs, wl, maxcells, strand, thr=150)
    # binding top-(maxcells+1000) selection: > 1003 observed cells
    feats = dg.make_features(23, 200, types=sc_types)
    wl = dg.make_whitelist(23, 1500)
    recs = dg.make_sc_records(323, 9000, wl, feats, n_cells=1400, reads_per_umi=1.2)
    sc_case('sc_topcells_binding', feats, recs, wl, 3, False)
    sc_case('sc_topcells_binding_multibundle', feats, recs, wl, 3, True, thr=1000)
    # a degenerate feature (right == left - 1) astride a 10 kb boundary is filed under no bucket
    # (genelist.py:371-373) although it passes the single-cell window test; inside a bucket it counts
    dfeats = [{'chr': 'chr1', 'left': 10000, 'right': 9999, 'strand': '+', 'name': 'a', 'type': 'protein_coding', 'ensg': 'STRADDLE'},
              {'chr': 'chr1', 'left': 10005, 'right': 10004, 'strand': '+', 'name': 'b', 'type': 'protein_coding', 'ensg': 'INSIDE'},
              {'chr': 'chr1', 'left': 20000, 'right': 19999, 'strand': '-', 'name': 'c', 'type': 'TE', 'ensg': 'LINE:L1:s'},
              {'chr': 'chr1', 'left': 500, 'right': 900, 'strand': '+', 'name': 'd', 'type': 'protein_coding', 'ensg': 'PLAIN'}]
    dwl = ['AAAA', 'CCCC']
    drecs = []
    for k, (pos, end) in enumerate([(9999, 10050), (9950, 10000), (10004, 10100), (9990, 10005), (19999, 20050),
                                    (19900, 20000), (600, 700), (9998, 9999), (10000, 10001)] * 3):
        drecs.append({'flag': 0, 'mapq': 255, 'rname': 'chr1', 'pos': pos, 'end': end,
                      'tags': {'CB': dwl[k % 2], 'UB': 'ACGT'[k % 4] + 'ACGT'[(k // 4) % 4] + 'ACGT'[(k // 16) % 4]}})
    sc_case('sc_degenerate_boundary', dfeats, drecs, dwl, 10, False)
    sc_case('sc_degenerate_boundary_strand', dfeats, drecs, dwl, 10, True)
    # sc failure modes
    wl = dg.make_whitelist(24, 20)
    recs = dg.make_sc_records(324, 200, wl, None, n_cells=4)
    nostrand = [dict((k, v) for k, v in f.items() if k != 'strand') for f in dg.make_features(24, 50, types=['enhancer'])]
    hitter = [{'flag': 0, 'mapq': 255, 'rname': 'chr' + nostrand[0]['chr'].replace('chr', ''),
               'pos': nostrand[0]['left'] + 1, 'end': nostrand[0]['left'] + 100,
               'tags': {'CB': wl[0], 'UB': u}} for u in ('AAAA', 'CCCC', 'GGGG')]
    error_case('sc_err_no_strand_key', 'sc', nostrand, hitter, {'maxcells': 5, 'strand': False, 'whitelist': wl},
               lambda: rh.run_sc(nostrand, hitter, wl, 5))
    feats = dg.make_features(25, 50, types=sc_types)
    error_case('sc_err_noumi', 'sc', feats, recs, {'maxcells': 5, 'strand': False, 'umis': False, 'whitelist': wl},
               lambda: rh.run_sc(feats, recs, wl, 5, umis=False))
    notag = [{'flag': 0, 'mapq': 255, 'rname': 'chr1', 'pos': 5, 'end': 50, 'tags': {}}]
    error_case('sc_err_no_cb', 'sc', feats, notag, {'maxcells': 5, 'strand': False, 'whitelist': wl},
               lambda: rh.run_sc(feats, notag, wl, 5))
    noub = [{'flag': 0, 'mapq': 255, 'rname': 'chr1', 'pos': 5, 'end': 50, 'tags': {'CB': wl[0]}}]
    error_case('sc_err_no_ub', 'sc', feats, noub, {'maxcells': 5, 'strand': False, 'whitelist': wl},
               lambda: rh.run_sc(feats, noub, wl, 5))


    # te_genome -m custom: a GTF and the feature rows the reference's make_custom builds from it
    gtf_text = dg.make_gtf_text(31, 80)
    gtf_path = os.path.join(HERE, 'custom_annotation.gtf')
    with open(gtf_path, 'w') as fh:
        fh.write(gtf_text)
    dump('custom_gtf_rows', {'kind': 'gtf', 'gtf_file': 'custom_annotation.gtf', 'expected': {'rows': rh.run_make_custom(gtf_path)}})

    # te_genome -m genes_tes from local files: GENCODE-like GTF + rmsk table and the reference's rows
    with gzip.GzipFile(os.path.join(HERE, 'genes_tes_gencode.gtf.gz'), 'wb', mtime=0) as fh:
        fh.write(dg.make_gencode_text(41, 50).encode())
    with gzip.GzipFile(os.path.join(HERE, 'genes_tes_rmsk.txt.gz'), 'wb', mtime=0) as fh:
        fh.write(dg.make_rmsk_text(41, 300).encode())
    for genome in ('hg38', 'mm10'):
        dump('genes_tes_rows_%s' % genome, {'kind': 'genes_tes', 'genome': genome, 'expected': {'rows': rh.run_make_genes_tes(
            genome, os.path.join(HERE, 'genes_tes_gencode.gtf.gz'), os.path.join(HERE, 'genes_tes_rmsk.txt.gz'))}})

    # te_genome -m enhancers from a local FANTOM5-style BED and the reference's rows
    import numpy as _np
    _rng = _np.random.default_rng(5)
    _chroms = ['chr1', 'chr2', 'chrX', 'chrY', 'chrM', 'chrUn_KI270742v1', 'chr10', 'chr22_KI270879v1_alt', 'chr29', 'chr30']
    _lines = []
    for _i in range(60):
        _c = _chroms[_i % len(_chroms)]
        _l = int(_rng.integers(1000, 100000))
        _r = _l + int(_rng.integers(50, 900))
        _lines.append('\t'.join([_c, str(_l), str(_r), '%s:%d-%d' % (_c, _l, _r), '0', '.', str(_l + 10), str(_l + 11), '0,0,0',
                                 '2', '20,30', '0,%d' % (_r - _l - 30)]))
    with gzip.GzipFile(os.path.join(HERE, 'enhancers_f5.bed.gz'), 'wb', mtime=0) as fh:
        fh.write(('\n'.join(_lines) + '\n').encode())
    dump('enhancers_rows_hg38', {'kind': 'enhancers', 'genome': 'hg38', 'expected': {'rows': rh.run_make_enh(
        'hg38', os.path.join(HERE, 'enhancers_f5.bed.gz'))}})
    with gzip.GzipFile(os.path.join(HERE, 'snrnps_gencode.gtf.gz'), 'wb', mtime=0) as fh:
        fh.write(dg.make_gencode_text(43, 120, named_frac=1.0).encode())  # make_snrnps needs gene_name everywhere
    dump('snrnps_rows_hg38', {'kind': 'snrnps', 'genome': 'hg38', 'expected': {'rows': rh.run_make_snrnps(
        'hg38', os.path.join(HERE, 'snrnps_gencode.gtf.gz'))}})

    # an index file written by the reference itself (a pickle): fixture for the .glb importer
    feats = dg.make_features(9, 200)
    rh.build_reference_index(feats, os.path.join(HERE, 'ref_built_index.glb'))
    with gzip.open(os.path.join(HERE, 'ref_built_index.features.json.gz'), 'wt') as fh:
        json.dump(feats, fh)


def end_to_end_index_cases():
    """Index files end to end through the reference: (1) a GTF -> the reference's make_custom -> .glb -> bulk SE and
    PE counts -> TSV; (2) the committed reference-built ref_built_index.glb -> bulk SE counts -> TSV."""
    gtf_path = os.path.join(HERE, 'custom_e2e.gtf')
    with open(gtf_path, 'w') as fh:
        fh.write(dg.make_gtf_text(37, 60, numeric_ids=False))
    rows = rh.run_make_custom(gtf_path)
    assert all(isinstance(r['ensg'], str) for r in rows)
    for paired in (False, True):
        recs = [r for r in dg.make_records(371 + paired, 3000, rows, span=160000, paired=paired) if r['end'] > r['pos']]
        if paired and len(recs) % 2:
            recs.pop()
        ref = rh.run_custom_end_to_end(gtf_path, recs, paired)
        assert sum(ref['result'].values()) > 0
        dump('custom_e2e_%s' % ('pe' if paired else 'se'),
             {'kind': 'custom_e2e', 'gtf_file': 'custom_e2e.gtf', 'records': recs, 'params': {'paired': paired, 'q': 20},
              'expected': {'rows': rows, 'result': ref['result'], 'total_reads': ref['total_reads'], 'tsv': ref['tsv']}})
    feats = json.load(gzip.open(os.path.join(HERE, 'ref_built_index.features.json.gz'), 'rt'))
    recs = [r for r in dg.make_records(91, 3000, feats) if r['end'] > r['pos']]
    ref = rh.run_bulk_on_glb(os.path.join(HERE, 'ref_built_index.glb'), recs, False)
    assert sum(ref['result'].values()) > 0
    dump('glb_e2e_se', {'kind': 'glb_e2e', 'glb_file': 'ref_built_index.glb', 'records': recs, 'params': {'paired': False, 'q': 20},
                        'expected': {'result': ref['result'], 'total_reads': ref['total_reads'], 'tsv': ref['tsv']}})


if __name__ == '__main__':
    if sys.argv[1:] == ['index_e2e']:  # only the index end-to-end fixtures (the others are byte-stable and stay)
        end_to_end_index_cases()
    else:
        main()
        end_to_end_index_cases()
