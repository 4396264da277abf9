#!/usr/bin/env python3
"""Host-side profile of the single-cell pipeline on host batches (development aid, run on the GPU box):
    python tools/profile_sc_host.py [records]
cProfile of one warm tools/bench_sc.py run: which Engine calls the wall time of part 1 / parts 2+3 sits in."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_sc  # noqa: E402
from te_counter_amd import Engine, synth  # noqa: E402


def main():
    records = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    ann = synth.make_annotation('hg38', scale=1.0)
    eng = Engine(0)
    eng.upload_index(ann.flat)
    arrays, wl, tables = synth.make_sc_arrays(ann, records, n_cells=10000, seed=2004)
    bench_sc.run(eng, arrays, wl, tables, 10000, False, 10_000_000)
    print('---- second run', file=sys.stderr, flush=True)
    pr = cProfile.Profile()
    pr.enable()
    out, st, t, _ = bench_sc.run(eng, arrays, wl, tables, 10000, False, 10_000_000)
    pr.disable()
    print(t, {k: len(v) for k, v in out.items() if hasattr(v, '__len__')})
    pstats.Stats(pr).sort_stats('tottime').print_stats(14)


if __name__ == '__main__':
    main()
