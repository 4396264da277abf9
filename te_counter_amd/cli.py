"""`te_count` command line: the reference's flags and dispatch (bin/te_count:17-120) over the
MI355X engine.

Differences, all deliberate and listed in DESIGN.md:
  * `-q N` works (the reference's `nargs=1` hands measureTE a list and dies at the first read);
  * `-g/-m` resolve to `.tecidx` files (this build's own index format) or, for `-m custom`, to a
    `.tecidx` / reference-built `.glb` path;
  * launched under torch.distributed.run (WORLD_SIZE > 1) the bulk path shards batches over the
    ranks' GPUs and sum-reduces the count vector over RCCL; rank 0 writes the TSV.
"""

import argparse
import logging
import os
import sys

from . import common


def prepare_parser():
    exmp = 'Example usage: te_count -i in.bam -o out.bam -g genome -m genes_tes'
    description = 'Counts up the number of reads that overlap some set of gene/TE or other features'
    parser = argparse.ArgumentParser(prog='te_count', description=description, epilog=exmp)
    optional = parser._action_groups.pop()
    optional.add_argument('--se', action='store_true', required=False, help='Set mode to SE (single-end) mode, default is paired-end mode')
    optional.add_argument('--sc', action='store_true', required=False, help='Set te_count to single-cell mode, default is bulk. Expects an CR:Z flag in the BAM/SAM file. Make sure you set --se appropriately')
    optional.add_argument('--noumi', action='store_true', required=False, help='If --sc is set, but this single cell data has no UMI in a UR:Z tag, set this switch')
    optional.add_argument('--strand', action='store_true', required=False, help='Use the strand information for the reads (protocol is strand-specific), default=False')
    optional.add_argument('-q', '--qual', type=int, default=20, required=False, help='q threshold for qulait filtering, default=20')
    optional.add_argument('--maxcells', action='store', type=int, default=10000, required=False, help='keep at most maxcells with the most reads, default=10,000')
    optional.add_argument('-w', action='store', type=str, default='', required=False, help='A whitelist of barcodes. Becomes a requried argument of --sc')
    optional.add_argument('--device', type=int, default=None, required=False, help='GPU ordinal (default: LOCAL_RANK or 0)')
    required = parser.add_argument_group('required arguments')
    required.add_argument('-i', '--inbam', nargs=1, required=True, help='the BAM alignment file containing the reads')
    required.add_argument('-o', '--outtsv', nargs=1, required=True, help='the TSV file to save the genes and count data to')
    required.add_argument('-g', '--genome', nargs=1, required=True, help='Genome assembly to use, valid genomes: {0}'.format(common.valid_assemblies))
    required.add_argument('-m', '--mode', nargs=1, required=True, help='Type of annotation to use, valid modes: {0}'.format(common.valid_modes))
    parser._action_groups.append(optional)
    logging.basicConfig(level=logging.DEBUG, format='%(levelname)-8s: %(message)s', datefmt='%m-%d %H:%M')
    parser.log = logging.getLogger('te_count')
    return parser


def main(argv=None):
    assert sys.version_info >= (3, 6), 'Python >=3.6 is required'
    parser = prepare_parser()
    args = parser.parse_args(argv)
    log = parser.log

    if args.sc and not args.w:
        log.warning('-w no whitelist of barcodes supplied')

    log.info('Arguments:')
    log.info('  inbam: %s' % args.inbam[0])
    log.info('  outtsv: %s' % args.outtsv[0])
    log.info('  genome: "%s"' % args.genome[0])
    log.info('  mode: "%s"' % args.mode[0])
    log.info('  strand: {0}'.format(args.strand))
    log.info('  single-end mode: {0} (default is PE)'.format(args.se))
    log.info('  quality thresold: {0} (default is 20)'.format(args.qual))
    if args.sc:
        log.info('  single cell mode: {0} (default is False/Bulk)'.format(args.sc))
        log.info('  maxcells: {0} (default is 10000)'.format(args.maxcells))
        log.info('  No UMI: {0} (default is False)'.format(args.noumi))
        log.info('  Whitelist: {0} (no default)'.format(args.w))

    from . import dist
    from .measure import measureTE

    world = dist.World.from_env()
    device = args.device if args.device is not None else world.local_rank
    mte = measureTE(sys.argv[0], args.qual, device=device, world=world)

    if args.mode[0] == 'custom':
        if not os.path.exists(args.genome[0]):
            raise AssertionError('genome index not file found')
        log.info('Loading {0} genome index'.format(args.genome[0]))
        mte.bind_genome(args.genome[0])
        log.info('Found and loaded {0} genome {1}'.format(args.genome[0], args.mode[0]))
    else:
        log.info('Finding Genome')
        genome = args.genome[0]
        path = common.genome_index_path(genome, args.mode[0])
        if path is None:
            log.error('{}/{} genome not found, please run te_genome'.format(genome, args.mode[0]))
            common.print_genomes(log=log)
            sys.exit()
        log.info('Found Genome {}/{}'.format(genome, args.mode[0]))
        mte.bind_genome(path)
        log.info('Found {} genome {}'.format(args.genome[0], args.mode[0]))

    if args.sc:
        if world.size > 1 and world.rank != 0:
            # the single-cell path needs the file-order epoch boundaries: it runs on one GPU
            # (DESIGN.md section 6); the other ranks of a multi-GPU launch stand by
            log.info('single-cell mode runs on rank 0 only; rank {} is idle'.format(world.rank))
            world.close()
            return
        if args.se:
            result = mte.sc_parse_bamse(args.inbam[0], UMIS=not args.noumi, maxcells=args.maxcells,
                                        whitelistfilename=args.w, strand=args.strand, log=log,
                                        label=os.path.split(args.outtsv[0])[1].replace('.tsv', ''))
            mte.sc_save_result(result, args.outtsv[0], maxcells=args.maxcells, log=log)
        else:
            raise AssertionError('PE path for --sc is not implemented')
    else:
        mte.load_genome()
        if args.se:
            result = mte.parse_bamse(args.inbam[0], strand=args.strand, log=log)
        else:
            result = mte.parse_bampe(args.inbam[0], strand=args.strand, log=log)
        if world.rank == 0:
            mte.save_result_bulk(result, args.outtsv[0], log=log)
    world.close()
    return


def entry():
    try:
        main()
    except KeyboardInterrupt:
        sys.stderr.write('User interrupt\n')
        sys.exit(0)
