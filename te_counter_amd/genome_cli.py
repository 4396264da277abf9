"""Command line of `te_genome` for the MI355X build: turns annotation files into `.tecidx` indices.

Flag names follow the reference's tool (bin/te_genome:16-36: -g, -m, --gtf, -f).  The reference
fetches GENCODE / rmsk / FANTOM5 files over the network for its prepackaged modes
(te_count/genome/make.py:41-43); this build takes local copies of the same files instead
(--gencode, --rmsk, --bed) because the target machines are offline.
"""

import argparse
import logging
import os
import sys

from . import common, genome


def prepare_parser():
    p = argparse.ArgumentParser(
        prog='te_genome', description='Annotation index builder for te_count (MI355X build).',
        epilog='e.g.  te_genome -g mygenome -m custom --gtf annotation.gtf')
    p.add_argument('-g', '--genome', nargs=1, required=True,
                   help='assembly the index is for (%s); free text with -m custom' % ', '.join(sorted(common.valid_assemblies)))
    p.add_argument('-m', '--mode', nargs=1, required=True, help='what to index: %s' % ', '.join(sorted(common.valid_modes)))
    p.add_argument('--gtf', nargs=1, help='-m custom: GTF whose exon rows become the features')
    p.add_argument('--gencode', nargs=1, help='-m genes_tes / snrnps: GENCODE annotation GTF (plain or .gz) on local disk')
    p.add_argument('--rmsk', nargs=1, help='-m genes_tes: UCSC RepeatMasker table rmsk.txt (plain or .gz) on local disk')
    p.add_argument('--bed', nargs=1, help='-m enhancers: FANTOM5 enhancer BED (plain or .gz) on local disk')
    p.add_argument('-f', action='store_true', help='rebuild even when an index of that name exists')
    logging.basicConfig(level=logging.DEBUG, format='%(levelname)-8s: %(message)s', datefmt='%m-%d %H:%M')
    p.log = logging.getLogger('te_genome')
    return p


def _packaged_target(assembly, mode):
    assert assembly in common.valid_assemblies, \
        '{0} genome assembly not in the list of valid assemblies: {1}'.format(assembly, common.valid_assemblies)
    os.makedirs(common.genome_dirs()[0], exist_ok=True)
    return os.path.join(common.genome_dirs()[0], '%s_%s' % (assembly, mode))


def main(argv=None):
    parser = prepare_parser()
    a = parser.parse_args(argv)
    log = parser.log
    assembly, mode = a.genome[0], a.mode[0]
    assert mode in common.valid_modes, f'{mode} mode not in the list of valid modes: {common.valid_modes}'
    one = lambda opt: opt[0] if opt else None  # noqa: E731
    log.info('te_genome: %s / %s', assembly, mode)

    if mode == 'custom':
        assert one(a.gtf), '--gtf must be a valid filename if mode=custom'
        if not os.path.exists(a.gtf[0]):
            raise AssertionError('--gtf file not found')
        written = genome.make_custom(a.gtf[0], assembly, log)
    else:
        have = common.genome_index_path(assembly, mode)
        if have and not a.f:
            log.error('%s already holds this index; pass -f to replace it', have)
            sys.exit()
        if mode == 'genes_tes' and one(a.gencode) and one(a.rmsk):
            written = genome.make_genes_tes(assembly, a.gencode[0], a.rmsk[0], _packaged_target(assembly, mode))
        elif mode == 'enhancers' and one(a.bed):
            written = genome.make_enhancers(assembly, a.bed[0], _packaged_target(assembly, mode))
        elif mode == 'snrnps' and one(a.gencode):
            written = genome.make_snrnps(a.gencode[0], _packaged_target(assembly, mode))
        else:
            log.error('this build does not download annotation files; supply them: -m genes_tes --gencode F --rmsk F | '
                      '-m snrnps --gencode F | -m enhancers --bed F | -m custom --gtf F')
            sys.exit(1)
    log.info('index written: %s', written)


def entry():
    try:
        main()
    except KeyboardInterrupt:
        sys.stderr.write('interrupted\n')
        sys.exit(0)
