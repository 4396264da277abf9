#!/usr/bin/env python3
"""Where the time of `te_count --sc --se` on a 10x-style BAM goes (development aid, run on the GPU box):
    python tools/time_sc_e2e.py [records]
Wraps the steps of measureTE.sc_parse_bamse with wall-clock timers; three passes over one synthetic file."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from te_counter_amd import bam, gbam, reads, scdist, synth  # noqa: E402
from te_counter_amd.engine import Engine  # noqa: E402
from te_counter_amd.measure import measureTE  # noqa: E402

T = {}


def timed(obj, name, label=None):
    fn = getattr(obj, name)

    def wrapper(*a, **k):
        t = time.time()
        try:
            return fn(*a, **k)
        finally:
            T[label or name] = T.get(label or name, 0.0) + time.time() - t
    setattr(obj, name, wrapper)


def main():
    records = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    ann = synth.make_annotation('hg38', scale=1.0)
    arrays, wl, tables = synth.make_sc_arrays(ann, records, n_cells=2000, n_whitelist=100_000, seed=77)
    td = tempfile.mkdtemp(dir='/dev/shm')
    path, idx, wlp = os.path.join(td, 'sc.bam'), os.path.join(td, 'a.tecidx'), os.path.join(td, 'wl.txt')
    bam.write_bam_arrays(path, tables[0], arrays['cls'], arrays['pos'], arrays['flag'], arrays['mapq'], end=arrays['end'] + 2,
                         cb=arrays['cb24'], ub=arrays['ub24'], payload='illumina')
    ann.flat.save(idx)
    open(wlp, 'w').write(''.join(bytes(r).rstrip(b'\0').decode() + '\n' for r in wl))
    timed(reads, 'load_whitelist')
    timed(reads, 'chrom_class_table')
    timed(Engine, 'sc_begin')
    timed(gbam.GpuBam, '__init__', 'gbam_open')
    timed(gbam.GpuBam, 'run_sc')
    timed(gbam.GpuBam, 'close', 'gbam_close')
    timed(scdist, 'run_parts')
    timed(Engine, 'sc_part1')
    timed(Engine, 'sc_part23') if hasattr(Engine, 'sc_part23') else None
    m = measureTE('t', 20)
    m.bind_genome(idx)
    m.load_genome()
    for rep in range(3):
        T.clear()
        t = time.time()
        m.sc_parse_bamse(path, whitelistfilename=wlp, label='x', maxcells=2000)
        total = time.time() - t
        print('pass %d: %.3f s  %s' % (rep, total, {k: round(v, 4) for k, v in T.items()}), m.decode_report['seconds'], flush=True)
    if os.environ.get('TEC_PROFILE'):
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        m.sc_parse_bamse(path, whitelistfilename=wlp, label='x', maxcells=2000)
        pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
    import shutil
    shutil.rmtree(td)


if __name__ == '__main__':
    main()
