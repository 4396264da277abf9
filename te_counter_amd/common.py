"""Constants of the reference's te_count/common.py:4-10 and the index lookup of
te_count/genome/__init__.py:15-32 (here: `.tecidx` files)."""

import os

valid_assemblies = {'mm10', 'hg38', 'macFas5'}
valid_modes = {'genes_tes', 'enhancers', 'custom', 'snrnps'}


def print_genomes(log=None):
    log.info('  Valid Genome assemblies are:')
    log.info('    hg38 - human')
    log.info('    mm10 - mouse')


def genome_dirs():
    dirs = []
    if os.environ.get('TE_COUNT_GENOME_DIR'):
        dirs.append(os.environ['TE_COUNT_GENOME_DIR'])
    dirs.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'genome'))
    return dirs


def genome_index_path(genome, mode):
    """Path of the built index for genome/mode, or None (check_genome_available)."""
    assert mode in valid_modes, f'{mode} mode not in the list of valid modes: {valid_modes}'
    if mode != 'custom':
        assert genome in valid_assemblies, '{0} genome assembly not in the list of valid assemblies: {1}'.format(genome, valid_assemblies)
    for d in genome_dirs():
        for ext in ('.tecidx', '.glb'):
            p = os.path.join(d, f'{genome}_{mode}{ext}')
            if os.path.exists(p):
                return p
    return None
