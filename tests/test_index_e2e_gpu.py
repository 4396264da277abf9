"""Index files end to end on the GPU box, against bytes the reference itself wrote:

  * `te_genome -g X -m custom --gtf F` (te_count/genome/make.py:173-226) -> `.tecidx` -> `te_count -g X.tecidx -m custom`
    (bin/te_count:80-84): the TSV equals the one the reference produced from ITS make_custom index of the same GTF;
  * `te_count -g ref_built_index.glb -m custom`: an index file pickled by the reference's te_genome
    (te_count/miniglbase/utils.py:21-59) is bound as it is, and the TSV equals the reference's on the same records.
Fixtures: tests/golden/custom_e2e_{se,pe}.json.gz, glb_e2e_se.json.gz (tests/golden/make_golden.py index_e2e)."""

import os
import subprocess
import sys

import pytest

from conftest import GOLDEN_DIR, ROOT, load_golden
from te_counter_amd import bam, cli
from te_counter_amd.reads import ListSource

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('layout', ['se', 'pe'])
def test_te_genome_custom_gtf_then_te_count(layout, tmp_path):
    g = load_golden('custom_e2e_%s' % layout)
    gtf = os.path.join(GOLDEN_DIR, g['gtf_file'])
    target = str(tmp_path / 'mygenome')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bin', 'te_genome'), '-g', target, '-m', 'custom', '--gtf', gtf],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    idx = target + '.tecidx'
    assert os.path.exists(idx)
    aln = str(tmp_path / 'in.bam')
    bam.write_bam(aln, ListSource(g['records']).reference_names, g['records'])
    out = str(tmp_path / 'out.tsv')
    cli.main(['-i', aln, '-o', out, '-g', idx, '-m', 'custom'] + (['--se'] if layout == 'se' else []))
    assert open(out).read() == g['expected']['tsv']
    assert sum(g['expected']['result'].values()) > 0


def test_te_count_binds_a_glb_written_by_the_reference(tmp_path):
    g = load_golden('glb_e2e_se')
    glb = os.path.join(GOLDEN_DIR, g['glb_file'])
    for sam in (False, True):
        aln = str(tmp_path / ('in.sam' if sam else 'in.bam'))
        (bam.write_sam if sam else bam.write_bam)(aln, ListSource(g['records']).reference_names, g['records'])
        out = str(tmp_path / ('o%d.tsv' % sam))
        cli.main(['-i', aln, '-o', out, '-g', glb, '-m', 'custom', '--se'])
        assert open(out).read() == g['expected']['tsv']
