"""CPU-only checks: the C oracle against the golden vectors (through the product's host-side
index/unit code), the index file format, and the C ABI's symbol table."""

import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden_names, load_golden
from helpers import flat_from_golden, units_from_records
from oracle import c_oracle
from oracle import te_oracle as orc
from te_counter_amd import _lib
from te_counter_amd.index import FlatIndex, index_chrom_key

BULK = [n for n in golden_names('bulk_') if '_err_' not in n]


@pytest.mark.parametrize('name', BULK)
def test_c_oracle_golden(name):
    g = load_golden(name)
    flat = flat_from_golden(g)
    units, nrec = units_from_records(flat, g['records'], g['params']['paired'], g['params']['q'],
                                     batch_size=7 if 'quirks' in name else 1000)
    ci = c_oracle.COracleIndex.from_flat(flat)
    counts, stats = ci.bulk_count(*units, g['params']['q'])
    got = {n: int(c) for n, c in zip(flat.names, counts)}
    assert got == g['expected']['result']
    total = (nrec // 2 if g['params']['paired'] else nrec) + 1
    assert total == g['expected']['total_reads']
    assert orc.format_bulk_tsv(got, total) == g['expected']['tsv']
    # stats agree with the Python oracle
    _, _, st = orc.bulk_count(g['features'], g['records'], g['params']['paired'], g['params']['q'])
    assert stats == st


def test_c_oracle_enhancer_crash():
    g = load_golden('bulk_err_enhancer')
    flat = flat_from_golden(g)
    units, _ = units_from_records(flat, g['records'], False)
    with pytest.raises(NameError):
        c_oracle.COracleIndex.from_flat(flat).bulk_count(*units, 20)


def test_pe_carry_across_batches():
    g = load_golden('bulk_pe_rand11')
    flat = flat_from_golden(g)
    a, _ = units_from_records(flat, g['records'], True, batch_size=1 << 20)
    for bs in (1, 3, 7, 1001):
        b, _ = units_from_records(flat, g['records'], True, batch_size=bs)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_pe_name_mismatch_raises():
    g = load_golden('bulk_err_pe_mismatch')
    flat = flat_from_golden(g)
    with pytest.raises(AttributeError):
        units_from_records(flat, g['records'], True)


def test_chrom_keys():
    assert index_chrom_key('chr1') == '1' and index_chrom_key('chrX') == 'X'
    assert index_chrom_key('chr1_KI270706v1_random') == '1_KI270706V1_RANDOM'
    assert index_chrom_key('chrUn_GL000220') == 'UN_GL000220'
    flat = FlatIndex.from_features([{'chr': 'chrM', 'left': 1, 'right': 5, 'name': 'a', 'type': 'TE', 'ensg': 'a', 'strand': '+'},
                                    {'chr': 'chrUn_x', 'left': 1, 'right': 5, 'name': 'b', 'type': 'TE', 'ensg': 'b', 'strand': '-'}])
    assert flat.chrom_id_of_read('chrM') == 0 and flat.chrom_id_of_read('M') == 0
    assert flat.chrom_id_of_read('chrMT') == -1 and flat.chrom_id_of_read('chrm') == -1
    assert flat.chrom_id_of_read('chrUn_x') == -1  # index key is upper-cased, the read side is not
    assert flat.chrom_id_of_read('chrUN_X') == 1


def test_index_file_roundtrip(tmp_path):
    g = load_golden('bulk_se_rand11')
    flat = flat_from_golden(g)
    p = str(tmp_path / 'x.tecidx')
    flat.save(p)
    back = FlatIndex.load(p)
    assert back.names == flat.names and back.chrom_keys == flat.chrom_keys and back.strands == flat.strands
    for k in ('chrom_id', 'left', 'right', 'meta'):
        assert np.array_equal(getattr(back, k), getattr(flat, k))
    assert back.names == orc.feature_names(g['features'])


def test_cabi_exports_every_declared_symbol():
    """libtecount.so loads on a GPU-less box and exports exactly what include/tecount.h declares."""
    hdr = open(os.path.join(ROOT, 'include', 'tecount.h')).read()
    declared = sorted(set(re.findall(r'\b(tec_[a-z_0-9]+)\s*\(', hdr)))
    assert declared == _lib.exported_symbols()
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.tec_version() == 100


def test_glb_import_of_a_reference_built_index():
    """tests/golden/ref_built_index.glb was written by the reference's genelist.save()."""
    import gzip
    import json
    from conftest import GOLDEN_DIR
    feats = json.load(gzip.open(os.path.join(GOLDEN_DIR, 'ref_built_index.features.json.gz'), 'rt'))
    a = FlatIndex.from_glb(os.path.join(GOLDEN_DIR, 'ref_built_index.glb'))
    b = FlatIndex.from_features(feats)
    assert a.names == b.names and a.chrom_keys == b.chrom_keys and a.strands == b.strands
    for k in ('chrom_id', 'left', 'right', 'meta'):
        assert np.array_equal(getattr(a, k), getattr(b, k))


def test_glb_import_refuses_foreign_classes(tmp_path):
    import pickle
    p = tmp_path / 'evil.glb'
    p.write_bytes(pickle.dumps(os.system))
    with pytest.raises(pickle.UnpicklingError):
        FlatIndex.from_glb(str(p))


def test_custom_gtf_builder_matches_reference_rows(tmp_path):
    """te_genome -m custom (te_count/genome/make.py:173-226): same feature rows as the reference
    built from tests/golden/custom_annotation.gtf, and a loadable .tecidx."""
    from conftest import GOLDEN_DIR
    from te_counter_amd import genome
    g = load_golden('custom_gtf_rows')
    gtf = os.path.join(GOLDEN_DIR, g['gtf_file'])
    rows = genome.custom_features(gtf)
    assert rows == g['expected']['rows'] and len(rows) > 50
    usable = [r for r in rows if isinstance(r['ensg'], str)]
    out = genome.make_custom(gtf, str(tmp_path / 'mygenome')) if len(usable) == len(rows) else None
    if out:
        assert FlatIndex.load(out).names == sorted(set(r['ensg'] for r in rows))


@pytest.mark.parametrize('genome_name', ['hg38', 'mm10'])
def test_genes_tes_builder_matches_reference_rows(genome_name):
    """make_genes_tes (te_count/genome/make.py:15-127) from local files: same rows, same order as
    the reference built from the same GENCODE-like GTF and rmsk table (incl. its first-line drop)."""
    from conftest import GOLDEN_DIR
    from te_counter_amd import genome
    g = load_golden('genes_tes_rows_%s' % genome_name)
    rows = genome.genes_tes_features(genome_name, os.path.join(GOLDEN_DIR, 'genes_tes_gencode.gtf.gz'),
                                     os.path.join(GOLDEN_DIR, 'genes_tes_rmsk.txt.gz'))
    assert rows == g['expected']['rows'] and len(rows) > 100
    assert {r['type'] for r in rows} >= {'TE', 'protein_coding'}


def test_snrnps_builder_matches_reference_rows():
    """make_snrnps (make.py:228-288), including its substring test `not in ('snRNA')`."""
    from conftest import GOLDEN_DIR
    from te_counter_amd import genome
    g = load_golden('snrnps_rows_hg38')
    rows = genome.snrnps_features(os.path.join(GOLDEN_DIR, 'snrnps_gencode.gtf.gz'))
    assert rows == g['expected']['rows'] and len(rows) > 5
    assert all(r['name'].startswith('snRNA-') for r in rows)


def test_enhancers_builder_matches_reference_rows(tmp_path):
    """make_enh (make.py:129-171) over a local FANTOM5-style BED: rows identical to the reference's,
    and bin/te_genome -m enhancers --bed writes an index of them."""
    import subprocess
    import sys
    from conftest import GOLDEN_DIR, ROOT
    from te_counter_amd import genome
    from te_counter_amd.index import FlatIndex
    g = load_golden('enhancers_rows_hg38')
    bed = os.path.join(GOLDEN_DIR, 'enhancers_f5.bed.gz')
    rows = genome.enhancers_features('hg38', bed)
    assert rows == g['expected']['rows'] and len(rows) == 36
    assert all(r['type'] == 'enhancer' and 'strand' not in r for r in rows)
    env = dict(os.environ, TE_COUNT_GENOME_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bin', 'te_genome'), '-g', 'hg38', '-m', 'enhancers', '--bed', bed],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    built = [os.path.join(dp, f) for dp, _, fs in os.walk(str(tmp_path)) for f in fs if f.endswith('.tecidx')]
    assert len(built) == 1 and len(FlatIndex.load(built[0])) == 36


def test_whitelist_loader_equals_the_reference_expression(tmp_path):
    """te_count.py:330-339 is sorted(set(line.strip() for line in file)); the loader does it on bytes with numpy
    (uniform files as one array, anything else line by line) and must give the same list and the same packing."""
    import random
    from te_counter_amd import reads
    rnd = random.Random(3)
    uniform = [''.join(rnd.choice('ACGT') for _ in range(16)) for _ in range(5000)]
    cases = {
        'uniform': '\n'.join(uniform + uniform[:100]) + '\n',
        'no_final_newline': '\n'.join(uniform[:50]),
        'ragged': '\n'.join(uniform[:50] + ['  ACGT \t', '', 'ACGT', 'AC', 'ACGTACGTACGTACGTACGTACGT']) + '\n',
        'crlf': '\r\n'.join(uniform[:20]) + '\r\n',
        'blank_lines': 'AAAA\n\n\nCCCC\n',
        'empty': '',
    }
    for name, text in cases.items():
        p = tmp_path / (name + '.txt')
        p.write_text(text)
        want = sorted(set(line.strip() for line in open(p)))
        names, packed = reads.load_whitelist(str(p))
        assert list(names) == want, name
        assert len(names) == len(want) and (not want or names[len(want) // 2] == want[len(want) // 2])
        assert np.array_equal(packed, reads.pack_whitelist(want)), name
    p = tmp_path / 'long.txt'
    p.write_text('A' * 25 + '\n')
    with pytest.raises(ValueError):
        reads.load_whitelist(str(p))


def test_documented_environment_variables():
    """INTEGRATION.md section 4 lists exactly the variables the sources read: every getenv() of the library and every
    TEC_* lookup of the package, bench.py and the tools' rank set-up."""
    import glob
    read = set()
    for f in glob.glob(os.path.join(ROOT, 'te_counter_amd', 'csrc', '*')):
        read |= set(re.findall(r'getenv\("([A-Z_0-9]+)"\)', open(f).read()))
    for f in glob.glob(os.path.join(ROOT, 'te_counter_amd', '*.py')) + [os.path.join(ROOT, 'bench.py'), os.path.join(ROOT, 'tools', 'bench_sc.py')]:
        read |= set(re.findall(r"environ(?:\.get|\.setdefault)?[\[(]'((?:TEC|GPU)_[A-Z_0-9]+)'", open(f).read()))
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    table = doc[doc.index('## 4. Environment variables'):]
    documented = set(re.findall(r'`((?:TEC|GPU)_[A-Z_0-9]+)`', table.split('| variable |')[1]))
    assert read == documented, (sorted(read - documented), sorted(documented - read))


def test_no_torch_switch():
    """TEC_NO_TORCH=1: the library loads without importing torch (single-process runs)."""
    import subprocess
    import sys
    code = "import sys; from te_counter_amd import _lib; _lib.load(); print('torch' in sys.modules)"
    for value, want in (('1', 'False'), ('', 'True')):
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, cwd=ROOT,
                           env=dict(os.environ, TEC_NO_TORCH=value))
        assert r.returncode == 0 and r.stdout.strip() == want, (value, r.stdout, r.stderr[-500:])


def test_strand_split_index():
    """FlatIndex.strand_split: chromosome 2c = the '+' features of c, 2c + 1 = its '-' features, others dropped."""
    from te_counter_amd import FlatIndex
    feats = [dict(chr='chr1', left=10, right=20, name='a', type='protein_coding', ensg='A', strand='+'),
             dict(chr='chr1', left=30, right=40, name='b', type='LINE', ensg='B', strand='-'),
             dict(chr='chr2', left=5, right=9, name='c', type='LINE', ensg='C', strand='.'),
             dict(chr='chr2', left=50, right=90, name='d', type='LINE', ensg='D', strand='-')]
    flat = FlatIndex.from_features(feats)
    sp = flat.strand_split()
    assert sp.names == flat.names == ['A', 'B', 'C', 'D']
    assert len(sp.chrom_keys) == 4 and len(set(sp.chrom_keys)) == 4
    assert sp.chrom_id.tolist() == [0, 1, 3] and sp.left.tolist() == [10, 30, 50]
    assert (sp.meta & ((1 << 26) - 1)).tolist() == [0, 1, 3]
    assert all(flat._chrom_lookup.get(k) is None for k in sp.chrom_keys)
