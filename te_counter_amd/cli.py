"""Command line of the MI355X build of `te_count`.

The flag surface (names, arity, types, defaults) is the contract of the reference's front end
(bin/te_count:17-49) and is kept; everything else -- help wording, log lines, control flow -- is
this build's own.  Deliberate differences:
  * `-q N` is usable (the reference parses it with nargs=1 and then compares an int with a list);
  * `-g/-m` resolve to this build's `.tecidx` files, or to a `.glb` written by the reference;
  * under `python -m torch.distributed.run` every rank decodes its own byte range of the BAM on
    its own GPU and the counts are merged over RCCL; rank 0 writes the output.
"""

import argparse
import logging
import os
import sys
from dataclasses import dataclass

from . import common

# (flags, argparse keywords).  Order = order in --help.
_SWITCHES = [
    (('--se',), dict(action='store_true', help='reads are single-end (without it, records are taken pairwise as mates)')),
    (('--sc',), dict(action='store_true', help='count per cell barcode (CB/CR tag) instead of per sample; needs --se and -w')),
    (('--noumi',), dict(action='store_true', help='with --sc: the library carries no UMI tag (UB/UR)')),
    (('--strand',), dict(action='store_true', help='respect read strand (stranded protocols); off unless given')),
    (('--bulk-strand-extension',), dict(action='store_true', help='bulk input with --strand: count features of the read\'s own '
                                        'strand only (read 1\'s for pairs) instead of refusing; an extension with no '
                                        'reference output to compare with')),
    (('-q', '--qual'), dict(type=int, default=20, help='drop records whose MAPQ is below this value [20]')),
    (('--maxcells',), dict(type=int, default=10000, help='with --sc: number of best-covered cells written [10000]')),
    (('-w',), dict(type=str, default='', help='with --sc: text file listing the accepted cell barcodes, one per line')),
    (('--device',), dict(type=int, default=None, help='GPU ordinal to run on [LOCAL_RANK, else 0]')),
]
_MANDATORY = [
    (('-i', '--inbam'), 'alignments to count (BAM, or SAM text)'),
    (('-o', '--outtsv'), 'where the table of counts goes'),
    (('-g', '--genome'), 'assembly name (%s), or the index file itself when -m custom' % ', '.join(sorted(common.valid_assemblies))),
    (('-m', '--mode'), 'annotation flavour: %s' % ', '.join(sorted(common.valid_modes))),
]


def prepare_parser():
    parser = argparse.ArgumentParser(
        prog='te_count', add_help=True,
        description='Overlap counting of aligned reads against gene / transposable-element annotations on an AMD MI355X.',
        epilog='e.g.  te_count -i sample.bam -o sample.tsv -g hg38 -m genes_tes')
    must = parser.add_argument_group('required arguments')
    for flags, text in _MANDATORY:
        must.add_argument(*flags, nargs=1, required=True, help=text)
    for flags, kw in _SWITCHES:
        parser.add_argument(*flags, required=False, **kw)
    logging.basicConfig(level=logging.DEBUG, format='%(levelname)-8s: %(message)s', datefmt='%m-%d %H:%M')
    parser.log = logging.getLogger('te_count')
    return parser


@dataclass
class Job:
    """What one invocation asks for, flattened out of the argparse namespace."""
    inbam: str
    outtsv: str
    genome: str
    mode: str
    single_end: bool
    single_cell: bool
    umis: bool
    strand: bool
    qual: int
    maxcells: int
    whitelist: str
    device: object
    bulk_strand_extension: bool = False

    @classmethod
    def from_args(cls, a):
        return cls(a.inbam[0], a.outtsv[0], a.genome[0], a.mode[0], a.se, a.sc, not a.noumi, a.strand, a.qual,
                   a.maxcells, a.w, a.device, a.bulk_strand_extension)

    @property
    def label(self):
        # the single-cell run is labelled by the output's base name (bin/te_count:107)
        return os.path.split(self.outtsv)[1].replace('.tsv', '')

    def describe(self, log):
        log.info('te_count (MI355X) job:')
        for key, val in (('alignments', self.inbam), ('output', self.outtsv), ('genome', self.genome),
                         ('annotation', self.mode), ('layout', 'single-end' if self.single_end else 'paired-end'),
                         ('stranded', self.strand), ('min MAPQ', self.qual)):
            log.info('    %-11s %s', key, val)
        if self.single_cell:
            for key, val in (('cells kept', self.maxcells), ('UMIs', 'collapsed' if self.umis else 'absent'),
                             ('whitelist', self.whitelist or '(none)')):
                log.info('    %-11s %s', key, val)


def _resolve_index(job, log):
    """Path of the annotation index this job counts against; exits like the reference when a
    prepackaged genome has not been built (bin/te_count:89-92)."""
    if job.mode == 'custom':
        if not os.path.exists(job.genome):
            raise AssertionError('genome index not file found')
        log.info('Custom annotation index: %s', job.genome)
        return job.genome
    path = common.genome_index_path(job.genome, job.mode)
    if path is None:
        log.error('no %s index for %s on this machine; build it with te_genome first', job.mode, job.genome)
        common.print_genomes(log=log)
        sys.exit()
    log.info('Annotation index: %s', path)
    return path


def _run_single_cell(mte, job, world, log):
    if not job.single_end:
        raise AssertionError('PE path for --sc is not implemented')
    table = mte.sc_parse_bamse(job.inbam, UMIS=job.umis, maxcells=job.maxcells, whitelistfilename=job.whitelist,
                               strand=job.strand, log=log, label=job.label)
    if world.rank == 0:
        mte.sc_save_result(table, job.outtsv, maxcells=job.maxcells, log=log)


def _run_bulk(mte, job, world, log):
    mte.load_genome()
    mte.bulk_strand_extension = job.bulk_strand_extension
    counter = mte.parse_bamse if job.single_end else mte.parse_bampe
    table = counter(job.inbam, strand=job.strand, log=log)
    if world.rank == 0:
        mte.save_result_bulk(table, job.outtsv, log=log)


def main(argv=None):
    assert sys.version_info >= (3, 6), 'Python >=3.6 is required'
    parser = prepare_parser()
    job = Job.from_args(parser.parse_args(argv))
    log = parser.log
    if job.single_cell and not job.whitelist:
        log.warning('--sc was given without -w: the run will stop when the whitelist is needed')
    job.describe(log)

    from . import dist
    from .measure import measureTE

    world = dist.World.from_env()
    try:
        device = job.device if job.device is not None else world.local_rank
        if world.size > 1 and world.backend == 'nccl' and device != world.local_rank:
            # RCCL reduces the count vector where it lives: the process group is bound to GPU LOCAL_RANK, and a
            # buffer on another GPU cannot take part in it
            raise SystemExit('te_count: --device %d conflicts with LOCAL_RANK %d (one process per GPU; leave --device '
                             'out under torch.distributed.run, or set TEC_DIST_BACKEND=gloo to share GPUs)'
                             % (device, world.local_rank))
        mte = measureTE(sys.argv[0], job.qual, device=device, world=world)
        mte.bind_genome(_resolve_index(job, log))
        (_run_single_cell if job.single_cell else _run_bulk)(mte, job, world, log)
    finally:
        world.close()


def entry():
    # A one-process run needs torch for nothing (it only carries the multi-GPU exchanges) and importing it costs
    # 2-3 s, more than counting a 100 M-record file: keep it out unless a launcher started several ranks.
    if int(os.environ.get('WORLD_SIZE', '1') or 1) <= 1:
        os.environ.setdefault('TEC_NO_TORCH', '1')
    try:
        main()
    except KeyboardInterrupt:
        sys.stderr.write('interrupted\n')
        sys.exit(0)
    # The outputs are written and closed.  Handing gigabytes of HBM and pinned memory back buffer by buffer (each
    # hipFree drains the device) takes ~0.1 s that the operating system's teardown of the process does not need.
    logging.shutdown()
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0)
