"""Per-stage latency breakdown of the record kernel (diagnostic build only).

    make clean && make EXTRA=-DTEC_DIAGNOSTICS && python tools/stage_clocks.py [--order sorted] [--layout se]

Wave 0 of every workgroup sums the shader-clock cycles between full `s_waitcnt`s at the stage ends
(csrc/bulk.hip, TEC_CLK).  Prints cycles per loop trip and the share of each stage.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--records', type=int, default=125_000_000)
    ap.add_argument('--genome', default='hg38')
    ap.add_argument('--layout', default='pe')
    ap.add_argument('--order', default='shuffled')
    ap.add_argument('--launches', type=int, default=3)
    args = ap.parse_args()
    import torch
    from te_counter_amd import Engine, synth
    paired = args.layout == 'pe'
    n_units = args.records // 2 if paired else args.records
    ann = synth.make_custom_annotation() if args.genome == 'custom3m' else synth.make_annotation(args.genome)
    eng = Engine(0)
    eng.upload_index(ann.flat, 0)
    units = synth.make_units(ann, n_units, seed=2003, paired=paired, order=args.order)
    dev = [torch.from_numpy(u.view(np.int16) if u.dtype == np.uint16 else u).cuda() for u in units]
    ptrs = [d.data_ptr() for d in dev]
    torch.cuda.synchronize()
    fn = eng._lib.tec_diag_stage_clocks
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    out = (C.c_ulonglong * 8)()
    eng.bulk_begin(20)
    eng.bulk_submit_device(n_units, *ptrs)  # warm-up
    eng.bulk_sync()
    fn(eng._ctx, out)
    for _ in range(args.launches):
        eng.bulk_submit_device(n_units, *ptrs)
    ms, nl = eng.bulk_kernel_time()
    fn(eng._ctx, out)
    eng.bulk_finish()
    unroll = int(os.environ.get('TEC_UNROLL', 2))
    trips = n_units / (unroll * 1024) * args.launches  # loop trips summed over all workgroups = per wave-0 total
    names = ['stream', 'records', 'continuation', 'tests+count', 'loop+flush', 'final flush']
    tot = sum(out[i] for i in range(6))
    print('kernel %.3f ms/launch (diagnostic build: full waits at stage ends), %d launches' % (ms / nl, nl))
    for i, nm in enumerate(names):
        print('%-14s %10.1f cycles/trip  %5.1f %%' % (nm, out[i] / trips, 100.0 * out[i] / tot))
    print('total          %10.1f cycles/trip (s_memtime ticks)' % (tot / trips))


if __name__ == '__main__':
    main()
