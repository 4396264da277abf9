"""End to end on the GPU box: BAM/SAM file -> bin/te_count (the reference's flags) -> TSV bytes
identical to what the reference wrote for the same records (golden fixtures)."""

import os
import subprocess
import sys

import pytest

from conftest import ROOT, load_golden
from helpers import flat_from_golden
from te_counter_amd import bam, cli
from te_counter_amd.reads import ListSource

pytestmark = pytest.mark.gpu


def _prepare(g, tmp_path, sam=False):
    idx = str(tmp_path / 'custom.tecidx')
    flat_from_golden(g).save(idx)
    aln = str(tmp_path / ('in.sam' if sam else 'in.bam'))
    refs = ListSource(g['records']).reference_names
    (bam.write_sam if sam else bam.write_bam)(aln, refs, g['records'])
    return idx, aln


@pytest.mark.parametrize('name', ['bulk_se_rand11', 'bulk_pe_rand11', 'bulk_se_q0', 'bulk_pe_rand12'])
@pytest.mark.parametrize('sam', [False, True])
def test_bulk_cli(name, sam, tmp_path):
    g = load_golden(name)
    idx, aln = _prepare(g, tmp_path, sam)
    out = str(tmp_path / 'out.tsv')
    argv = ['-i', aln, '-o', out, '-g', idx, '-m', 'custom', '-q', str(g['params']['q'])]
    if not g['params']['paired']:
        argv.append('--se')
    cli.main(argv)
    assert open(out).read() == g['expected']['tsv']


@pytest.mark.parametrize('name', ['sc_rand21_nostrand', 'sc_rand21_strand', 'sc_rand22_strand'])
def test_sc_cli(name, tmp_path):
    g = load_golden(name)
    idx, aln = _prepare(g, tmp_path)
    wl = tmp_path / 'wl.txt'
    wl.write_text(''.join(b + '\n' for b in g['whitelist']))
    out = str(tmp_path / 'cells_out.tsv')
    argv = ['-i', aln, '-o', out, '-g', idx, '-m', 'custom', '--sc', '--se', '-w', str(wl), '--maxcells',
            str(g['params']['maxcells'])]
    if g['params']['strand']:
        argv.append('--strand')
    cli.main(argv)
    assert open(out).read() == g['expected']['tsv']
    assert open(out.replace('.tsv', '.barcode_freq.tsv')).read() == g['expected']['barcode_freq']


def test_script_and_refusals(tmp_path):
    g = load_golden('bulk_se_rand12')
    idx, aln = _prepare(g, tmp_path)
    out = str(tmp_path / 'o.tsv')
    exe = os.path.join(ROOT, 'bin', 'te_count')
    r = subprocess.run([sys.executable, exe, '-i', aln, '-o', out, '-g', idx, '-m', 'custom', '--se'],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert open(out).read() == g['expected']['tsv']
    with pytest.raises(NotImplementedError):  # bulk --strand: te_count.py:58-59
        cli.main(['-i', aln, '-o', out, '-g', idx, '-m', 'custom', '--se', '--strand'])
    with pytest.raises(AssertionError):       # --sc without --se: bin/te_count:109-110
        cli.main(['-i', aln, '-o', out, '-g', idx, '-m', 'custom', '--sc', '-w', 'x'])
    with pytest.raises(SystemExit):            # genome not built: bin/te_count:89-92
        cli.main(['-i', aln, '-o', out, '-g', 'hg38', '-m', 'genes_tes'])
