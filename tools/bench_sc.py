#!/usr/bin/env python3
"""Single-cell path benchmark (BASELINE config 4: 10 k cells x 5 k reads = 5e7 SE records,
737 280-barcode whitelist, 12-mer UMIs, ~3 reads per (CB,UB), bundle threshold 1e7 -> >= 2 spill
epochs).  Times tec_sc_* end to end from host SoA batches (H2D copies included) and checks a
reduced-size run of the same generator against the Python oracle.

    python tools/bench_sc.py [--records N] [--check 150000] [--pinned]

Multi-GPU: the single-cell path runs as REPLICAS (the spill-epoch boundaries need
the global running count of new (CB,UB) keys in file order, and the path is bound by the host side,
not the GPU).  Launched with `python -m torch.distributed.run --nproc-per-node N tools/bench_sc.py`,
every rank counts its own sample (seed + rank) on its own GPU; rank 0 prints the aggregate rate over
the slowest rank's time.  TEC_BENCH_BACKEND=gloo rehearses that on a box with fewer GPUs than ranks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(eng, arrays, wl, tables, maxcells, stranded, bundle_keys, batch=1 << 22):
    names, class_chrom, class_skip = tables
    t = {}
    t0 = time.perf_counter()
    eng.sc_begin(stranded, 20, wl, bundle_keys, class_chrom, class_skip)
    n = len(arrays['pos'])
    for a in range(0, n, batch):
        eng.sc_submit(*[arrays[k][a:a + batch] for k in ('cls', 'pos', 'end', 'flag', 'mapq', 'cb24', 'ub24', 'tagflags')])
    t['submit'] = time.perf_counter() - t0
    t1 = time.perf_counter()
    ids, counts, first = eng.sc_part1()
    t['part1'] = time.perf_counter() - t1
    t2 = time.perf_counter()
    order = np.lexsort((first, -counts.astype(np.int64)))
    selected = np.sort(ids[order[0:maxcells + 1000]])
    out = eng.sc_part23(selected)
    t['part23'] = time.perf_counter() - t2
    st = eng.sc_stats()
    eng.sc_end()
    t['total'] = time.perf_counter() - t0
    return out, st, t, (ids, counts)


def oracle_check(eng, ann, n_records, strand):
    """A reduced-size run of the config-4 generator (same annotation family at 1 % scale, several spill
    epochs) on the GPU against the Python oracle: result dict and per-cell totals must be identical."""
    from oracle import te_oracle as orc
    from te_counter_amd import synth
    small = synth.make_annotation(ann.genome if ann.genome in synth.GENOMES else 'hg38', scale=0.01)
    eng.upload_index(small.flat)
    arrays, wl, tables = synth.make_sc_arrays(small, n_records, n_cells=40, n_whitelist=5000, seed=11)
    thr = max(1000, n_records // 8)
    out, st, _, _ = run(eng, arrays, wl, tables, 25, strand, thr)
    wl_str = [bytes(r).rstrip(b'\0').decode() for r in wl]
    feats = [{'chr': small.flat.chrom_keys[c], 'left': int(l), 'right': int(r), 'strand': '+-'[(m >> 29) & 1],
              'name': 'x', 'type': 'TE' if ((m >> 26) & 7) == 2 else 'protein_coding', 'ensg': small.flat.names[m & ((1 << 26) - 1)]}
             for c, l, r, m in zip(small.flat.chrom_id.tolist(), small.flat.left.tolist(), small.flat.right.tolist(), small.flat.meta.tolist())]
    recs = synth.sc_arrays_to_records(arrays, tables[0])
    want, wbc, _ = orc.sc_count(feats, recs, wl_str, 25, strand=strand, bundle_threshold=thr)
    got = {}
    for c, nm, v in zip(out['coo_cell'].tolist(), out['coo_name'].tolist(), out['coo_count'].tolist()):
        got.setdefault(small.flat.names[nm], {})[wl_str[c]] = v
    want_nz = {k: v for k, v in want.items() if v}
    ok = got == want_nz and [(wl_str[c], h) for c, h in zip(out['cells'].tolist(), out['cell_hits'].tolist())] == list(wbc.items())
    return {'records': n_records, 'bundles': st['bundles'], 'identical': bool(ok)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--records', type=int, default=50_000_000)
    ap.add_argument('--cells', type=int, default=10000)
    ap.add_argument('--check', type=int, default=150000, help='records of the reduced-size oracle check (0 = skip)')
    ap.add_argument('--strand', action='store_true')
    ap.add_argument('--pinned', action='store_true', help='host batches in pinned memory (H2D at PCIe speed)')
    args = ap.parse_args()
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (('RANK', 0), ('WORLD_SIZE', 1), ('LOCAL_RANK', 0)))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        backend = os.environ.get('TEC_BENCH_BACKEND', 'nccl')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend != 'nccl':
            local = local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        dist.init_process_group(backend, **({'device_id': torch.device('cuda', local)} if backend == 'nccl' else {}))
        if rank:
            args.check = 0
    from te_counter_amd import Engine, synth
    ann = synth.make_annotation('hg38')
    eng = Engine(local)
    eng.upload_index(ann.flat)
    res = {}
    if args.check:
        res['oracle_check'] = oracle_check(eng, ann, args.check, args.strand)
        if not res['oracle_check']['identical']:
            print(json.dumps(res))
            sys.exit('PARITY FAILURE in the reduced-size single-cell check')
        eng.upload_index(ann.flat)
    t0 = time.time()
    arrays, wl, tables = synth.make_sc_arrays(ann, args.records, n_cells=args.cells, seed=2004 + rank)
    res['gen_s'] = round(time.time() - t0, 1)
    if args.pinned:
        import torch
        keep = {k: torch.from_numpy(v).pin_memory() for k, v in arrays.items() if isinstance(v, np.ndarray)}
        arrays = dict(arrays, **{k: t_.numpy() for k, t_ in keep.items()})
    n = len(arrays['pos'])
    run(eng, {k: v[:1 << 20] for k, v in arrays.items()}, wl, tables, args.cells, args.strand, 10_000_000)  # warm-up
    if dist:
        dist.barrier()
    out, st, t, _ = run(eng, arrays, wl, tables, args.cells, args.strand, 10_000_000)
    total = t['total']
    if dist:
        import torch
        tt = torch.tensor([total], dtype=torch.float64, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        total = float(tt.item())
    if rank == 0:
        res.update({'metric': 'Mreads/s single-cell (--sc --se) end to end from host SoA batches', 'records': n * world,
                    'n_gpus': world, 'parallelism': 'replicas' if world > 1 else 'single', 'pinned': bool(args.pinned),
                    'value': round(n * world / total / 1e6, 2), 'unit': 'Mreads/s',
                    'seconds': {k: round(v, 3) for k, v in t.items()},
                    'stats': st, 'cells_with_hits': int(len(out['cells'])), 'coo_nnz': int(len(out['coo_cell']))})
        print(json.dumps(res))
    eng.close()
    if dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
