#!/usr/bin/env python3
"""Headline benchmark: Mreads/s overlap-counted, bulk PE vs an hg38 genes_tes-like annotation
(BASELINE.json `metric`; north-star shape: 1 B PE reads over 8 GPUs = 125 M records per GPU).

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the count kernel over one HBM-resident batch of synthetic pair units
(default 62.5 M pairs = 125 M records per GPU).  Ranks hold disjoint batches and a full index
replica (weak scaling); the timed region ends with the RCCL sum-reduce of the count vector and
the copy of counts + stats to the host, i.e. it is a complete counting job.  One JSON line on
rank 0.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--records', type=int, default=125_000_000, help='BAM records per GPU per step')
    ap.add_argument('--genome', default='hg38', choices=['hg38', 'mm10', 'custom3m'],
                    help="custom3m: BASELINE config 5, 3 M unique-name intervals")
    ap.add_argument('--layout', default='pe', choices=['pe', 'se'])
    ap.add_argument('--order', default='shuffled', choices=['shuffled', 'sorted'])
    ap.add_argument('--scale', type=float, default=1.0, help='shrink the annotation (tests)')
    ap.add_argument('--bin-shift', type=int, default=0)
    ap.add_argument('--genes', type=int, default=40000, help='genes in the synthetic annotation (experiments)')
    ap.add_argument('--cpu-sample', type=int, default=25_000_000, help='units timed on the CPU oracle (0 = skip)')
    ap.add_argument('--traffic', default=None, help='HBM bytes per launch from rocprofv3 --pmc passes; default: profiles/traffic.json entry of this workload')
    ap.add_argument('--host-stream', action='store_true', help='(default at N=1) also time the pinned-host -> HBM -> kernel path (PCIe-inclusive; reported beside, never as value)')
    ap.add_argument('--no-host-stream', action='store_true', help='skip the PCIe-inclusive measurement')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ...' % (args.gpus, args.gpus))
        sys.exit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))

    import torch
    import torch.distributed as dist
    from te_counter_amd import Engine, synth

    # one process per GPU; TEC_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than
    # ranks (ranks then share devices and the reduce goes through gloo instead of RCCL)
    backend = os.environ.get('TEC_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)

    paired = args.layout == 'pe'
    n_units = args.records // 2 if paired else args.records
    rec_per_unit = 2 if paired else 1

    t0 = time.time()
    if args.genome == 'custom3m':
        ann = synth.make_custom_annotation(scale=args.scale)
    else:
        ann = synth.make_annotation(args.genome, scale=args.scale, n_genes=args.genes)
    eng = Engine(local)
    eng.upload_index(ann.flat, args.bin_shift)
    info = eng.index_info()
    units = synth.make_units(ann, n_units, seed=2003 + rank, paired=paired, order=args.order)
    dev = [torch.from_numpy(u.view(np.int16) if u.dtype == np.uint16 else u).cuda() for u in units]
    ptrs = [d.data_ptr() for d in dev]
    torch.cuda.synchronize()
    t_setup = time.time() - t0

    d_counts, _ = eng.bulk_device_buffers()

    class _Wrap:  # lets torch see libtecount's count vector without a copy
        __cuda_array_interface__ = {'shape': (eng.n_names,), 'typestr': '<i8', 'data': (d_counts, False), 'version': 2}
    counts_t = torch.as_tensor(_Wrap(), device=torch.device('cuda', local)) if world > 1 else None

    def job(steps):
        eng.bulk_begin(20)
        for _ in range(steps):
            eng.bulk_submit_device(n_units, *ptrs)
        if world > 1:
            eng.bulk_sync()
            dist.all_reduce(counts_t)  # RCCL sum over xGMI
            torch.cuda.synchronize()
        return eng.bulk_finish()

    job(args.warmup)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t1 = time.perf_counter()
    counts, stats = job(args.steps)
    fence()
    elapsed = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms, launches = eng.bulk_kernel_time()
    reduce_ms = eng.bulk_reduce_time()

    # size-independent property at full size: K passes over one batch = K x one pass
    single, sstats = job(1)
    if world == 1:
        assert np.array_equal(counts, single * args.steps), 'linearity violated'
        assert stats['assigned'] == sstats['assigned'] * args.steps

    if rank == 0:
        total_records = n_units * rec_per_unit * args.steps * world
        value = total_records / elapsed / 1e6
        # 15 B per unit + index (12 B per entry + 4 B per bin) + 8 B per name
        alg_bytes = 15.0 * n_units + 12.0 * info['n_entries'] + 4.0 * info['n_bins'] + 8.0 * eng.n_names
        avg_ms = kernel_ms / max(1, launches)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        out = {
            'metric': 'Mreads/s overlap-counted (bulk %s %s %s)' % (args.layout.upper(), args.genome, 'custom' if args.genome == 'custom3m' else 'genes_tes'),
            'value': round(value, 2), 'unit': 'Mreads/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'int32 coords / u64 counts', 'data': 'synthetic',
            'config': {'workload': 'bulk %s, synthetic %s-like annotation (%d intervals, %d names), '
                                   '%d records/GPU/step, %s order, -q 20' % (args.layout.upper(), args.genome, len(ann.flat),
                                                                            ann.flat.n_names, n_units * rec_per_unit, args.order),
                       'records_per_gpu_per_step': n_units * rec_per_unit, 'units_per_launch': n_units,
                       'index': info, 'parallelism': 'dp%d' % world,
                       'collective': ('%s all_reduce(sum) of u64 counts' % ('rccl' if backend == 'nccl' else backend)) if world > 1 else 'none',
                       'setup_s': round(t_setup, 1)},
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic_for(args, n_units),
                         'kernel': 'count_bulk_rec_kernel', 'avg_launch_ms': round(avg_ms, 4),
                         'algorithmic_bytes_per_launch': alg_bytes,
                         'reduce_slabs_kernel_avg_ms': round(reduce_ms / max(1, launches), 4)},
            'assigned_frac': round(sstats['assigned'] / max(1, n_units), 4),
        }
        tr = out['roofline']['traffic']
        if tr:  # the rate at which the measured L2<->fabric bytes move: what actually bounds the shuffled case
            out['roofline']['traffic_GBps'] = round(tr / (avg_ms * 1e-3) / 1e9, 1)
        if not args.no_host_stream and world == 1:
            out['pcie_inclusive'] = host_stream(eng, units, n_units, rec_per_unit)
        if args.cpu_sample and world == 1:
            out['cpu_baseline'] = cpu_baseline(ann, units, min(args.cpu_sample, n_units), rec_per_unit, eng)
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


def traffic_for(args, n_units):
    """HBM bytes per launch of the count kernel from the committed rocprofv3 --pmc passes
    (profiles/traffic.json, written by tools/traffic.py); null if this workload was not profiled."""
    if args.traffic:
        return float(args.traffic)
    try:
        table = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
    except (OSError, ValueError):
        return None
    key = '%s/%s/%s/%d' % (args.genome, args.layout, args.order, n_units)
    e = table.get(key)
    return float(e['hbm_bytes_per_launch']) if e else None


def host_stream(eng, units, n_units, rec_per_unit, batch=1 << 24, reps=3):
    """Pinned host SoA -> hipMemcpyAsync -> kernel through tec_bulk_submit (double buffered)."""
    import torch
    pinned = [torch.from_numpy(u.view(np.int16) if u.dtype == np.uint16 else u).pin_memory() for u in units]
    views = [p.numpy() for p in pinned]
    views[3] = views[3].view(np.uint16)
    best = None
    for _ in range(reps):
        eng.bulk_begin(20)
        t = time.perf_counter()
        for a in range(0, n_units, batch):
            eng.bulk_submit(*[v[a:a + batch] for v in views])
        eng.bulk_finish()
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    return {'value': round(n_units * rec_per_unit / best / 1e6, 1), 'unit': 'Mreads/s',
            'GB_per_s_over_pcie': round(15.0 * n_units / best / 1e9, 1), 'batch_units': batch,
            'note': 'pinned host batches streamed with hipMemcpyAsync, copy and compute streams overlapped'}


def cpu_baseline(ann, units, n_sample, rec_per_unit, eng):
    """The C oracle (oracle/te_oracle.c, the reference's loop restated) on one host core over a
    prefix of the same batch; also checks the GPU result on that prefix against it."""
    from oracle import c_oracle
    ci = c_oracle.COracleIndex.from_flat(ann.flat)
    sample = [u[:n_sample] for u in units]
    t = time.perf_counter()
    want, wst = ci.bulk_count(*sample, 20)
    dt = time.perf_counter() - t
    eng.bulk_begin(20)
    eng.bulk_submit(*sample)
    got, gst = eng.bulk_finish()
    ok = bool(np.array_equal(got, want) and all(gst[k] == v for k, v in wst.items()))
    if not ok:
        raise SystemExit('PARITY FAILURE: GPU counts differ from the oracle on the bench sample')
    return {'value': round(n_sample * rec_per_unit / dt / 1e6, 3), 'unit': 'Mreads/s', 'cores': 1, 'kind': 'port',
            'sample': 'first %d units (%d records) of rank 0\'s batch, C restatement of the reference loop, '
                      '%.1f s; GPU counts on the same sample are bit-identical' % (n_sample, n_sample * rec_per_unit, dt),
            'cpu_model': _cpu_model(), 'host_cores': os.cpu_count()}


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


if __name__ == '__main__':
    main()
