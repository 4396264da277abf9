#!/usr/bin/env python3
"""End to end: a BAM file on disk -> bin/te_count's code path (on-GPU BGZF/BAM decode, count kernel, TSV)
on one GPU.  Prints one JSON line with the wall time per stage.

    python tools/bench_e2e.py [--records 12000000] [--layout pe|se] [--payload random|illumina] [--decoder gpu|cpu]

The BAM is synthetic (te_counter_amd.bam.write_bam_arrays): hg38-like annotation and units as in bench.py,
one 100-base record per read with an Illumina-style name and four small tags; payload 'random' = random
sequence and quality bytes (the hard case for an inflater), 'illumina' = ACGT + NovaSeq-binned qualities."""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def cleanup(result):
    import shutil
    shutil.rmtree(result.get('dir', ''), ignore_errors=True)


def run(records, layout='pe', payload='random', decoder='gpu', scale=1.0, reps=2, ann=None, reuse=None):
    """reuse: path of a BAM written by an earlier call with the same arguments (it is not rewritten).
    The temporary directory stays until cleanup(result)."""
    from te_counter_amd import bam, synth
    from te_counter_amd.measure import measureTE
    paired = layout == 'pe'
    ann = ann or synth.make_annotation('hg38', scale=scale)
    n_units = records // 2 if paired else records
    units = synth.make_units(ann, n_units, seed=4242, paired=paired)
    refs = ann.chrom_names + ['chrUn_1']
    out = {'layout': layout, 'payload': payload, 'decoder': decoder}
    prev = os.environ.get('TEC_DECODER')
    os.environ['TEC_DECODER'] = decoder
    td = os.path.dirname(reuse) if reuse else tempfile.mkdtemp(dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    out['dir'] = td
    try:
        bam_path, idx_path, tsv = os.path.join(td, 'in.bam'), os.path.join(td, 'ann.tecidx'), os.path.join(td, 'out.%s.tsv' % decoder)
        out['bam_path'] = bam_path
        if not reuse:
            tid, pos, fl, mq = bam.units_to_records(units, len(refs) - 1, paired)
            t = time.time()
            bam.write_bam_arrays(bam_path, refs, tid, pos, fl, mq, end=None if paired else units[2], pair_names=paired, payload=payload)
            out['write_bam_s'] = round(time.time() - t, 2)
            ann.flat.save(idx_path)
        out['bam_MB'] = round(os.path.getsize(bam_path) / 1e6, 1)
        n_rec = n_units * (2 if paired else 1)
        # one measureTE object for all passes, as a caller that counts several files against one annotation would
        # hold it: the first pass pays library loading and the allocation of the decoder's buffers (pinned staging,
        # HBM), the later ones find them parked on the context
        t0 = time.time()
        mte = measureTE('bench_e2e', 20)
        mte.bind_genome(idx_path)
        mte.load_genome()                                  # .tecidx -> host arrays -> fine-bin index in HBM
        load_s = time.time() - t0
        for rep in range(reps):
            t1 = time.time()
            result = (mte.parse_bampe if paired else mte.parse_bamse)(bam_path)   # decode + count
            t2 = time.time()
            mte.save_result_bulk(result, tsv)
            t3 = time.time()
            out['run%d_s' % rep] = {'load_index': round(load_s, 3), 'decode_and_count': round(t2 - t1, 3),
                                    'write_tsv': round(t3 - t2, 3), 'total': round(load_s + t3 - t1, 3)}
            if getattr(mte, 'decode_report', None):
                r = mte.decode_report
                out['decode_report'] = {k: r[k] for k in ('blocks', 'blocks_redone_on_host', 'compressed_bytes', 'inflated_bytes',
                                                          'segment_repairs', 'seconds')}
        mte._engine.close()
        out['tsv_lines'] = sum(1 for _ in open(tsv))
        # the TSV against the C oracle's counts in the reference writer's formatting (te_count.py:279-296)
        from oracle import c_oracle
        from oracle import te_oracle as orc
        want, _ = c_oracle.COracleIndex.from_flat(ann.flat).bulk_count(*[u[:2_000_000] for u in units], 20) if n_units <= 2_000_000 else (None, None)
        if want is not None:
            out['tsv_ok'] = open(tsv).read() == orc.format_bulk_tsv({n: int(c) for n, c in zip(ann.flat.names, want)}, n_units + 1)
        else:  # large runs: the count column must sum to what the engine reported as counted
            out['tsv_ok'] = sum(int(line.split('\t')[1]) for line in open(tsv)) == int(sum(result.values()))
    finally:
        if prev is None:
            os.environ.pop('TEC_DECODER', None)
        else:
            os.environ['TEC_DECODER'] = prev
    last = out['run%d_s' % (reps - 1)]
    out['records'] = int(n_rec)
    out['Mrecords_per_s_decode_and_count'] = round(n_rec / last['decode_and_count'] / 1e6, 2)
    out['Mrecords_per_s_end_to_end'] = round(n_rec / last['total'] / 1e6, 2)
    out['threads'] = int(os.environ.get('TEC_BAM_THREADS', 16))
    out['host_cores'] = os.cpu_count()
    return out


def run_sc(records, ann, n_cells=2000, maxcells=2000, reps=2):
    """What `te_count --sc --se -w WL` does on a 10x-style BAM (CB / UB tags): file -> on-GPU decode -> tec_sc_* -> result
    mapping, checked against the same records submitted as host arrays (tools/bench_sc.py's path)."""
    import numpy as np
    from te_counter_amd import bam, synth
    from te_counter_amd.measure import measureTE
    arrays, wl, tables = synth.make_sc_arrays(ann, records, n_cells=n_cells, n_whitelist=100_000, seed=77)
    n = len(arrays['pos'])
    td = tempfile.mkdtemp(dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    out = {'records': int(n), 'dir': td}
    try:
        bam_path, idx_path, wl_path = os.path.join(td, 'sc.bam'), os.path.join(td, 'ann.tecidx'), os.path.join(td, 'wl.txt')
        bam.write_bam_arrays(bam_path, tables[0], arrays['cls'], arrays['pos'], arrays['flag'], arrays['mapq'], end=arrays['end'] + 2,
                             cb=arrays['cb24'], ub=arrays['ub24'], payload='illumina')
        out['bam_MB'] = round(os.path.getsize(bam_path) / 1e6, 1)
        ann.flat.save(idx_path)
        with open(wl_path, 'w') as fh:
            fh.write(''.join(bytes(r).rstrip(b'\0').decode() + '\n' for r in wl))
        mte = measureTE('bench_e2e', 20)
        mte.bind_genome(idx_path)
        mte.load_genome()
        for rep in range(reps):
            t = time.time()
            result = mte.sc_parse_bamse(bam_path, whitelistfilename=wl_path, label='bench', maxcells=maxcells)
            out['run%d_s' % rep] = round(time.time() - t, 3)
        out['Mrecords_per_s'] = round(n / out['run%d_s' % (reps - 1)] / 1e6, 2)
        out['cells_with_hits'] = len(mte.barcodes)
        out['total_reads'] = int(mte.total_reads)
        out['ok'] = mte.total_reads == n + 1 and 0 < len(mte.barcodes) <= n_cells and sum(1 for _ in result) == ann.flat.n_names
        mte._engine.close()
    finally:
        cleanup(out)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--records', type=int, default=12_000_000)
    ap.add_argument('--layout', default='pe', choices=['pe', 'se'])
    ap.add_argument('--payload', default='random', choices=['random', 'illumina'])
    ap.add_argument('--decoder', default='gpu', choices=['gpu', 'cpu'])
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--reps', type=int, default=3)
    args = ap.parse_args()
    r = run(args.records, args.layout, args.payload, args.decoder, args.scale, args.reps)
    cleanup(r)
    print(json.dumps(r))


if __name__ == '__main__':
    main()
