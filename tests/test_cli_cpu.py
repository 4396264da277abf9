"""The command-line contract, on CPU: every `te_count` invocation of the reference's smoke script
(test/tests.sh:4-27) must parse into the same argparse namespace as the reference's parser
(bin/te_count:17-49) would produce: same dest names, arities and defaults.  Expected values were
read off the reference parser's definitions; the one deliberate difference is `-q` (the
reference declares it nargs=1 and then cannot use it; here it is a plain int)."""

import shlex

import pytest

from te_counter_amd import cli, genome_cli

DEFAULTS = dict(se=False, sc=False, noumi=False, strand=False, qual=20, maxcells=10000, w='')

# (command line as in test/tests.sh, overrides of DEFAULTS + the four required one-element lists)
TESTS_SH = [
    ('te_count -i data/bulk_rnaseq_PE.bam -o run_results/bulk_out_PE.tsv -g mm10 -m genes_tes',
     dict(inbam=['data/bulk_rnaseq_PE.bam'], outtsv=['run_results/bulk_out_PE.tsv'], genome=['mm10'], mode=['genes_tes'])),
    ('te_count --se -i data/bulk_rnaseq_PE.bam -o run_results/bulk_out_SE.tsv -g mm10 -m genes_tes',
     dict(se=True, inbam=['data/bulk_rnaseq_PE.bam'], outtsv=['run_results/bulk_out_SE.tsv'], genome=['mm10'], mode=['genes_tes'])),
    ('te_count -i data/single_cell_rnaseq.bam -w barcodes/version1.txt -g mm10 --se --sc -m genes_tes '
     '-o run_results/single_cell_out.tsv  --maxcells 3',
     dict(se=True, sc=True, maxcells=3, w='barcodes/version1.txt', inbam=['data/single_cell_rnaseq.bam'],
          outtsv=['run_results/single_cell_out.tsv'], genome=['mm10'], mode=['genes_tes'])),
    ('te_count -i data/single_cell_rnaseq.bam -w barcodes/version1.txt -g mm10 --se --sc -m genes_tes '
     '-o run_results/single_cell_strand_out.tsv -w barcodes/version1.txt --maxcells 3 --strand',
     dict(se=True, sc=True, strand=True, maxcells=3, w='barcodes/version1.txt', inbam=['data/single_cell_rnaseq.bam'],
          outtsv=['run_results/single_cell_strand_out.tsv'], genome=['mm10'], mode=['genes_tes'])),
    ('te_count -i data/bulk_rnaseq_PE.bam -o run_results/bulk_out_PE-enh.tsv -g mm10 -m enhancers',
     dict(inbam=['data/bulk_rnaseq_PE.bam'], outtsv=['run_results/bulk_out_PE-enh.tsv'], genome=['mm10'], mode=['enhancers'])),
    ('te_count -i data/single_cell_rnaseq.bam -w barcodes/version1.txt -g mm10 --sc -m enhancers '
     '-o run_results/single_cell_out-enh.tsv  --maxcells 3',
     dict(sc=True, maxcells=3, w='barcodes/version1.txt', inbam=['data/single_cell_rnaseq.bam'],
          outtsv=['run_results/single_cell_out-enh.tsv'], genome=['mm10'], mode=['enhancers'])),
    ('te_count -i data/bulk_rnaseq_PE.bam -o run_results/bulk_snrnpsiPE.tsv -g hg38 -m snrnps',
     dict(inbam=['data/bulk_rnaseq_PE.bam'], outtsv=['run_results/bulk_snrnpsiPE.tsv'], genome=['hg38'], mode=['snrnps'])),
    ('te_count --se -i data/bulk_rnaseq_PE.bam -o run_results/bulk_snrnpsSE.tsv -g hg38 -m snrnps',
     dict(se=True, inbam=['data/bulk_rnaseq_PE.bam'], outtsv=['run_results/bulk_snrnpsSE.tsv'], genome=['hg38'], mode=['snrnps'])),
]


@pytest.mark.parametrize('line,want', TESTS_SH, ids=[str(i) for i in range(len(TESTS_SH))])
def test_reference_smoke_script_lines_parse_to_the_reference_namespace(line, want):
    ns = vars(cli.prepare_parser().parse_args(shlex.split(line)[1:]))
    assert ns.pop('device') is None  # this build's extra flags: the GPU ordinal, and the stranded-bulk extension (off)
    assert ns.pop('bulk_strand_extension') is False
    assert ns == dict(DEFAULTS, **want)


def test_job_view_of_the_namespace():
    a = cli.prepare_parser().parse_args(shlex.split(TESTS_SH[3][0])[1:])
    job = cli.Job.from_args(a)
    assert (job.single_end, job.single_cell, job.strand, job.umis, job.maxcells) == (True, True, True, True, 3)
    assert job.label == 'single_cell_strand_out'  # bin/te_count:107
    assert job.whitelist == 'barcodes/version1.txt' and job.qual == 20


def test_required_flags_and_types():
    p = cli.prepare_parser()
    for missing in ('-i', '-o', '-g', '-m'):
        argv = ['-i', 'a', '-o', 'b', '-g', 'c', '-m', 'custom']
        k = argv.index(missing)
        with pytest.raises(SystemExit):
            p.parse_args(argv[:k] + argv[k + 2:])
    with pytest.raises(SystemExit):
        p.parse_args(['-i', 'a', '-o', 'b', '-g', 'c', '-m', 'custom', '--maxcells', 'many'])
    assert p.parse_args(['-i', 'a', '-o', 'b', '-g', 'c', '-m', 'custom', '-q', '0', '--noumi']).qual == 0


def test_te_genome_flags():
    """bin/te_genome:16-36: -g/-m required one-element lists, --gtf one-element list, -f switch."""
    a = genome_cli.prepare_parser().parse_args(['-g', 'mine', '-m', 'custom', '--gtf', 'x.gtf', '-f'])
    assert (a.genome, a.mode, a.gtf, a.f) == (['mine'], ['custom'], ['x.gtf'], True)
    with pytest.raises(SystemExit):
        genome_cli.prepare_parser().parse_args(['-g', 'mine'])
