#!/usr/bin/env python3
"""Headline benchmark: Mreads/s overlap-counted, bulk PE vs an hg38 genes_tes-like annotation
(BASELINE.json `metric`; north-star shape: 1 B PE reads over 8 GPUs = 125 M records per GPU).

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the count kernel over one HBM-resident batch of synthetic pair units
(default 62.5 M pairs = 125 M records per GPU).  Ranks hold disjoint batches and a full index
replica (weak scaling); the timed region ends with the RCCL sum-reduce of the count vector and
the copy of counts + stats to the host, i.e. it is a complete counting job.  One JSON line on
rank 0.

At N = 1 the line also carries what a user of te_count sees, all measured in this run:
  cpu_baseline              the pure-Python restatement of the reference loop on a bounded sample
  cpu_baseline_c_1core      the C restatement, one core
  cpu_baseline_c_allcores   the same, threads over disjoint unit ranges
  pcie_inclusive            pinned host batches -> hipMemcpyAsync -> kernel
  end_to_end_from_bam       a synthetic BAM file on disk -> measureTE.parse_bampe -> TSV (+ a 10x-style BAM -> sc_parse_bamse)
  single_cell               BASELINE config 4 (50 M records, 10 k cells) with an oracle sub-check
  secondary                 the headline workload coordinate-sorted, BASELINE configs 2 (100 M SE, shuffled and sorted) and 5 (3 M names)
(`--legs headline` skips them.)
"""

import argparse
import json
import os
import sys
import time

import numpy as np

if int(os.environ.get('GPU_MAX_HW_QUEUES') or 0) < 16:
    os.environ['GPU_MAX_HW_QUEUES'] = '16'  # before anything initialises HIP (see te_counter_amd/__init__.py)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--records', type=int, default=125_000_000, help='BAM records per GPU per step')
    ap.add_argument('--genome', default='hg38', choices=['hg38', 'mm10', 'custom3m'],
                    help="custom3m: BASELINE config 5, 3 M unique-name intervals")
    ap.add_argument('--layout', default='pe', choices=['pe', 'se'])
    ap.add_argument('--order', default='shuffled', choices=['shuffled', 'sorted'])
    ap.add_argument('--scale', type=float, default=1.0, help='shrink the annotation (tests)')
    ap.add_argument('--bin-shift', type=int, default=0)
    ap.add_argument('--genes', type=int, default=40000, help='genes in the synthetic annotation (experiments)')
    ap.add_argument('--cpu-sample', type=int, default=25_000_000, help='units timed on the C oracle (0 = skip the CPU baselines)')
    ap.add_argument('--traffic', default=None, help='HBM bytes per launch from rocprofv3 --pmc passes; default: profiles/traffic.json entry of this workload')
    ap.add_argument('--no-host-stream', action='store_true', help='skip the PCIe-inclusive measurement')
    ap.add_argument('--legs', default='all', help="'all' (default at N=1), 'headline', or a comma list of: cpu,pcie,e2e,sc,secondary")
    ap.add_argument('--details', default=None, help='also write the un-abridged result (every leg with its samples, stats and timings) to this file')
    ap.add_argument('--e2e-records', type=int, default=40_000_000, help='records of the end-to-end BAM (a launch of the inflate kernel holds 49 152 BGZF blocks = 13 M such records)')
    ap.add_argument('--sc-records', type=int, default=50_000_000)
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
            sys.exit(spawn_ranks(args.gpus))  # a bare `python bench.py --gpus N`: start the N ranks ourselves
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

    legs = set() if args.legs == 'headline' or world > 1 else \
        ({'cpu', 'pcie', 'e2e', 'sc', 'secondary'} if args.legs == 'all' else set(args.legs.split(',')))
    if args.no_host_stream:
        legs.discard('pcie')
    if not args.cpu_sample:
        legs.discard('cpu')

    t0 = time.time()
    if args.genome == 'custom3m':
        ann = synth.make_custom_annotation(scale=args.scale)
    else:
        ann = synth.make_annotation(args.genome, scale=args.scale, n_genes=args.genes)
    eng = Engine(local)
    run = BulkRun(eng, ann, args.layout, args.order, args.records, seed=2003 + rank, bin_shift=args.bin_shift)
    t_setup = time.time() - t0
    paired, n_units, rec_per_unit = run.paired, run.n_units, run.rec_per_unit

    d_counts, _ = eng.bulk_device_buffers()

    class _Wrap:  # lets torch see libtecount's count vector without a copy
        __cuda_array_interface__ = {'shape': (eng.n_names,), 'typestr': '<i8', 'data': (d_counts, False), 'version': 2}
    counts_t = torch.as_tensor(_Wrap(), device=torch.device('cuda', local)) if world > 1 else None

    def job(steps):
        eng.bulk_begin(20)
        for _ in range(steps):
            eng.bulk_submit_device(n_units, *run.ptrs)
        if world > 1:
            eng.bulk_sync()
            if backend == 'nccl':
                dist.all_reduce(counts_t)  # RCCL sum over xGMI
            else:  # (the rehearsal backend moves host memory only: synchronous copies out and back)
                host = counts_t.cpu()
                dist.all_reduce(host)
                counts_t.copy_(host)
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
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms, launches = eng.bulk_kernel_time()
    reduce_ms = eng.bulk_reduce_time()
    launch_ms = eng.bulk_launch_times()

    # size-independent property at full size: K passes over one batch = K x one pass
    single, sstats = job(1)
    if world == 1:
        assert np.array_equal(counts, single * args.steps), 'linearity violated'
        assert stats['assigned'] == sstats['assigned'] * args.steps

    if rank == 0:
        total_records = n_units * rec_per_unit * args.steps * world
        value = total_records / elapsed / 1e6
        out = {
            'metric': 'Mreads/s overlap-counted (bulk %s %s %s)' % (args.layout.upper(), args.genome, 'custom' if args.genome == 'custom3m' else 'genes_tes'),
            'value': round(value, 2), 'unit': 'Mreads/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'int32 coords / u64 counts', 'data': 'synthetic',
            'config': {'workload': 'bulk %s, synthetic %s-like annotation (%d intervals, %d names), '
                                   '%d records/GPU/step, %s order, -q 20' % (args.layout.upper(), args.genome, len(ann.flat),
                                                                            ann.flat.n_names, n_units * rec_per_unit, args.order),
                       'records_per_gpu_per_step': n_units * rec_per_unit, 'units_per_launch': n_units,
                       'index': run.info, 'parallelism': 'dp%d' % world,
                       'collective': ('%s all_reduce(sum) of the u64 count vector, once per timed job of %d steps (the CLI path reduces once per file)'
                                      % ('rccl' if backend == 'nccl' else backend, args.steps)) if world > 1 else 'none',
                       'setup_s': round(t_setup, 1)},
            'roofline': run.roofline(kernel_ms, launches, reduce_ms, launch_ms, traffic_for(args.traffic, args.genome, args.layout, args.order, n_units)),
            'assigned_frac': round(sstats['assigned'] / max(1, n_units), 4),
        }
        t_legs = time.time()
        leg_s = {}

        def timed(name, fn, *a):
            t = time.time()
            r = fn(*a)
            leg_s[name] = round(time.time() - t, 1)
            return r
        if 'pcie' in legs:
            out['pcie_inclusive'] = timed('pcie', host_stream, eng, run.units, n_units, rec_per_unit)
        if 'cpu' in legs:
            out['cpu_baseline'] = timed('cpu_python', cpu_baseline_python, ann, run.units, rec_per_unit, paired, eng)
            out['cpu_baseline_c_1core'] = timed('cpu_c_1core', cpu_baseline_c, ann, run.units, min(args.cpu_sample, n_units), rec_per_unit, eng)
            out['cpu_baseline_c_allcores'] = timed('cpu_c_allcores', lambda: cpu_baseline_c_threads(ann, run.units, n_units, rec_per_unit,
                                                                                                    counts_one_pass=single))
        run.release()
        if 'e2e' in legs:
            out['end_to_end_from_bam'] = timed('e2e', leg, end_to_end, ann, eng, args.e2e_records)
        if 'sc' in legs:
            out['single_cell'] = timed('sc', leg, single_cell, ann, eng, args.sc_records)
        if 'secondary' in legs:
            out['secondary'] = timed('secondary', leg, secondary, ann, eng, args)
        if legs:
            out['config']['legs_s'] = round(time.time() - t_legs, 1)
            out['config']['leg_seconds'] = leg_s
        if args.details:
            with open(args.details, 'w') as fh:
                json.dump(out, fh, indent=1)
        print(json.dumps(compact(out)))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: run the same command line under torch.distributed.run (one
    process per GPU) as a CHILD process -- this process has not touched the GPU yet and never will -- pass rank 0's
    JSON line through and return the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('MASTER_ADDR', '127.0.0.1')
    child = subprocess.run(cmd, env=env)
    return child.returncode


class BulkRun:
    """An index replica on the engine's GPU and one HBM-resident batch of synthetic units."""

    def __init__(self, eng, ann, layout, order, records, seed, bin_shift=0, stranded=False):
        import torch
        from te_counter_amd import synth
        self.eng, self.ann = eng, ann
        self.stranded = stranded  # the stranded-bulk extension (tec_bulk_set_stranded): a strand-split index
        self.paired = layout == 'pe'
        self.rec_per_unit = 2 if self.paired else 1
        self.n_units = records // 2 if self.paired else records
        eng.upload_index(ann.flat.strand_split() if stranded else ann.flat, bin_shift)
        self.info = eng.index_info()
        self.units = synth.make_units(ann, self.n_units, seed=seed, paired=self.paired, order=order)
        self.dev = [torch.from_numpy(u.view(np.int16) if u.dtype == np.uint16 else u).cuda() for u in self.units]
        self.ptrs = [d.data_ptr() for d in self.dev]
        torch.cuda.synchronize()

    def reference_index_shape(self):
        """The index as the reference holds it (te_count/miniglbase/genelist.py:367-378): one list entry per feature and
        10 kb bucket it touches, one bucket list per 10 kb of every chromosome that has features."""
        flat = self.ann.flat
        bs = 10000
        lo, hi = flat.left.astype(np.int64) // bs, flat.right.astype(np.int64) // bs
        entries = int((hi - lo + 1).sum())
        top = np.zeros(max(1, len(flat.chrom_keys)), np.int64)
        np.maximum.at(top, flat.chrom_id, hi + 1)
        return entries, int(top.sum())

    def algorithmic_bytes(self):
        # 15 B per unit + the index once (12 B per bucket entry + 4 B per bucket, in the REFERENCE's shape: 10 kb
        # buckets) + 8 B per name written
        entries, buckets = self.reference_index_shape()
        return 15.0 * self.n_units + 12.0 * entries + 4.0 * buckets + 8.0 * self.eng.n_names

    def roofline(self, kernel_ms, launches, reduce_ms, launch_ms, traffic):
        traffic, traffic_source = traffic
        alg = self.algorithmic_bytes()
        avg_ms = kernel_ms / max(1, launches)
        achieved = alg / (avg_ms * 1e-3) / 1e9
        entries, buckets = self.reference_index_shape()
        r = {'bound': 'hbm', 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
             'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic, 'traffic_source': traffic_source,
             'kernel': 'count_bulk_rec_kernel',
             'avg_launch_ms': round(avg_ms, 4), 'algorithmic_bytes_per_launch': alg,
             'algorithmic_bytes': {'units': 15.0 * self.n_units, 'reference_shaped_index': 12.0 * entries + 4.0 * buckets,
                                   'counts': 8.0 * self.eng.n_names,
                                   'this_build_index_bytes_for_comparison': 12.0 * self.info['n_entries'] + 4.0 * self.info['n_bins']},
             'reduce_slabs_kernel_avg_ms': round(reduce_ms / max(1, launches), 4)}
        if len(launch_ms):  # per-launch durations from the event pool: the sample is short, show its spread
            r['launch_ms_min_median_max'] = [round(float(np.min(launch_ms)), 4), round(float(np.median(launch_ms)), 4),
                                             round(float(np.max(launch_ms)), 4)]
        if traffic:  # the rate at which the measured L2<->fabric bytes move: what actually bounds the shuffled case
            r['traffic_GBps'] = round(traffic / (avg_ms * 1e-3) / 1e9, 1)
        return r

    def count(self, steps):
        self.eng.bulk_begin(20)
        if self.stranded:
            self.eng.bulk_set_stranded(True)
        for _ in range(steps):
            self.eng.bulk_submit_device(self.n_units, *self.ptrs)
        return self.eng.bulk_finish()

    def release(self):
        self.dev = self.ptrs = None
        import torch
        torch.cuda.empty_cache()


def compact(out):
    """The one line the driver parses stays short: every leg keeps its numbers, long texts and stat
    dumps go to --details."""
    o = dict(out)
    o['config'] = {k: v for k, v in out['config'].items() if k != 'index'}
    o['config']['index'] = {k: out['config']['index'][k] for k in ('n_entries', 'n_bins')}
    pick = lambda d, keys: {k: d[k] for k in keys if k in d}  # noqa: E731
    rl = ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms', 'gpu_stream_ms', 'launch_ms_min_median_max')
    if o['roofline'].get('traffic_source'):
        o['roofline'] = dict(o['roofline'], traffic_source={k: v for k, v in o['roofline']['traffic_source'].items() if k != 'collected'})
    for k in ('cpu_baseline', 'cpu_baseline_c_1core', 'cpu_baseline_c_allcores'):
        if k in o and 'error' not in o[k]:
            o[k] = pick(o[k], ('value', 'unit', 'cores', 'kind'))
            o[k]['sample'] = out[k]['sample'].split(',')[0][:90]
    if 'pcie_inclusive' in o:
        o['pcie_inclusive'] = pick(o['pcie_inclusive'], ('value', 'unit', 'GB_per_s_over_pcie'))
    e = out.get('end_to_end_from_bam')
    if e and 'error' not in e:
        c = {'value': e['value'], 'unit': e['unit'], 'host_decoder_16_threads': e['illumina_host_decoder']['Mrecords_per_s_decode_and_count']}
        for payload in ('illumina', 'random'):
            r = e[payload]
            c[payload] = {'records': r['records'], 'bam_MB': r['bam_MB'], 'Mrecords_per_s': r['Mrecords_per_s_decode_and_count'],
                          'first_run_s': r['first_run']['decode_and_count'], 'warm_run_s': r['warm_run']['decode_and_count'],
                          'with_index_load_and_tsv_s': r['warm_run']['total'], 'seconds': r['decode']['seconds'] if r.get('decode') else None,
                          'tsv_ok': r['tsv_identical_to_oracle_formatting']}
        if 'single_cell_from_bam' in e:
            c['single_cell_from_bam'] = pick(e['single_cell_from_bam'], ('records', 'Mrecords_per_s', 'ok'))
        o['end_to_end_from_bam'] = c
    sc = out.get('single_cell')
    if sc and 'error' not in sc:
        o['single_cell'] = {'value': sc['value'], 'unit': sc['unit'], 'records': sc['records'], 'seconds': sc['seconds'],
                            'bundles': sc['stats']['bundles'], 'oracle_check_identical': sc['oracle_check']['identical'],
                            'roofline': pick(sc['roofline'], rl)}
    sec = out.get('secondary')
    if sec and 'error' not in sec:
        o['secondary'] = {k: {'value': v['value'], 'unit': v['unit'], 'ms_per_step': v['ms_per_step'], 'gpu_ms_per_step': v['gpu_ms_per_step'],
                              'roofline': pick(v['roofline'], rl)}
                          for k, v in sec.items()}
    return o


def leg(fn, *a):
    """A side leg must not take the headline down with it -- except for parity failures, which are loud."""
    try:
        return fn(*a)
    except SystemExit:
        raise
    except Exception as e:  # noqa: BLE001
        return {'error': '%s: %s' % (type(e).__name__, e)}


def kernel_source_sha():
    """Fingerprint of the sources the count kernel is built from: a traffic figure measured on another version of the
    kernel is not reported."""
    import hashlib
    h = hashlib.sha256()
    for f in ('bulk.hip', 'index.hip', 'spill.hip', 'tecount_internal.h'):
        with open(os.path.join(ROOT, 'te_counter_amd', 'csrc', f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def sc_source_sha():
    """Fingerprint of the single-cell kernels' source."""
    import hashlib
    with open(os.path.join(ROOT, 'te_counter_amd', 'csrc', 'sc.hip'), 'rb') as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


def sc_traffic(n_records):
    """-> (HBM bytes of one whole single-cell job, source) from profiles/traffic.json (tools/prof_sc.sh + tools/traffic_sc.py);
    null if not profiled at this size or on another version of sc.hip."""
    try:
        e = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json'))).get('sc/config4/%d' % n_records)
    except (OSError, ValueError):
        e = None
    if not e:
        return None, None
    src = {'profile_tag': e.get('tag'), 'kernel_symbol': e.get('kernel_symbol'), 'kernel_source_sha': e.get('kernel_source_sha'),
           'collected': 'rocprofv3 --pmc, separate passes, summed over every kernel of one job; FETCH_SIZE*1024 + 64*TCC_EA0_RDREQ_128B + WRITE_SIZE*1024'}
    if e.get('kernel_source_sha') != sc_source_sha():
        src['stale'] = 'profiled on another version of sc.hip: not reported'
        return None, src
    return float(e['hbm_bytes_per_job']), src


def traffic_for(explicit, genome, layout, order, n_units):
    """-> (HBM bytes per launch of the count kernel, where the figure comes from).  The bytes are read from the
    committed rocprofv3 --pmc passes (profiles/traffic.json, written by tools/traffic.py from separate FETCH_SIZE /
    WRITE_SIZE / request-size passes over this very command line); null if this workload was not profiled, or was
    profiled on a different version of the kernel sources."""
    if explicit:
        return float(explicit), {'from': '--traffic'}
    try:
        table = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
    except (OSError, ValueError):
        return None, None
    e = table.get('%s/%s/%s/%d' % (genome, layout, order, n_units))
    if not e:
        return None, None
    src = {'profile_tag': e.get('tag'), 'kernel_symbol': e.get('kernel_symbol'), 'kernel_source_sha': e.get('kernel_source_sha'),
           'collected': 'rocprofv3 --pmc, separate passes; FETCH_SIZE*1024 + 64*TCC_EA0_RDREQ_128B + WRITE_SIZE*1024'}
    if e.get('kernel_source_sha') != kernel_source_sha():
        src['stale'] = 'profiled on another version of the kernel sources: not reported'
        return None, src
    return float(e['hbm_bytes_per_launch']), src


def host_stream(eng, units, n_units, rec_per_unit, batch=1 << 24, reps=3):
    """Pinned host SoA -> hipMemcpyAsync -> kernel through tec_bulk_submit (double buffered)."""
    views = []
    for u in units:  # page-locked batches from the library's own allocator (tec_alloc_pinned)
        v = eng.pinned_empty(len(u), u.dtype)
        v[:] = u
        views.append(v)
    best = None
    for _ in range(reps):
        eng.bulk_begin(20)
        t = time.perf_counter()
        for a in range(0, n_units, batch):
            eng.bulk_submit(*[v[a:a + batch] for v in views], stage=False)
        eng.bulk_finish()
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    return {'value': round(n_units * rec_per_unit / best / 1e6, 1), 'unit': 'Mreads/s',
            'GB_per_s_over_pcie': round(15.0 * n_units / best / 1e9, 1), 'batch_units': batch,
            'note': 'pinned host batches streamed with hipMemcpyAsync, copy and compute streams overlapped'}


def _host_cpus():
    """CPUs this process may actually use: the cgroup quota if there is one (the GPU boxes show 256
    logical CPUs and a quota of 16)."""
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            return max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    return os.cpu_count() or 1


def cpu_baseline_python(ann, units, rec_per_unit, paired, eng, n_sample=300_000):
    """The reference's own loop as restated in plain Python (oracle/te_oracle.py: byte-identical to
    te_count.py on the golden vectors) on ONE core -- all the reference can use -- over a bounded
    sample: the first units of the batch that map to one chromosome, against that chromosome's part of
    the annotation (the reference's 10 kb bucket lists are per chromosome, so the per-read work is that
    of the genome-wide run; building Python buckets for all 5.7 M features would take minutes)."""
    from oracle import te_oracle as orc
    from te_counter_amd import synth
    flat = ann.flat
    chrom = int(np.bincount(units[0][units[0] >= 0][:5_000_000]).argmax()) if (units[0] >= 0).any() else 0
    idx = np.nonzero(units[0] == chrom)[0][:n_sample]
    sample = [u[idx] for u in units]
    feats = synth.features_overlapping(flat, np.array([chrom]), np.array([0]), np.array([2 ** 31 - 2]))
    name = ann.chrom_names[chrom]
    recs = []
    for p1, p2, fl, mq in zip(sample[1].tolist(), sample[2].tolist(), sample[3].tolist(), sample[4].tolist()):
        if paired:
            recs.append({'flag': fl, 'mapq': mq, 'qname': 'q', 'rname': name, 'pos': p1, 'end': p1 + 100})
            recs.append({'flag': 0, 'mapq': mq, 'qname': 'q', 'rname': name, 'pos': p2, 'end': p2 + 100})
        else:
            recs.append({'flag': fl, 'mapq': mq, 'qname': 'q', 'rname': name, 'pos': p1, 'end': p2})
    t = time.perf_counter()
    buckets = orc.build_buckets(feats)
    t_index = time.perf_counter() - t
    t = time.perf_counter()
    res, total, _ = orc.bulk_count(feats, recs, paired, buckets=buckets)
    dt = time.perf_counter() - t
    eng.bulk_begin(20)
    eng.bulk_submit(*sample)
    got, _ = eng.bulk_finish()
    got = {n: int(c) for n, c in zip(flat.names, got) if c}
    if got != {k: v for k, v in res.items() if v}:
        raise SystemExit('PARITY FAILURE: GPU counts differ from the Python oracle on the bench sample')
    return {'value': round(len(recs) / dt / 1e6, 4), 'unit': 'Mreads/s', 'cores': 1, 'kind': 'port',
            'implementation': 'pure-Python restatement of te_count.py:76-152 / 201-263 (oracle/te_oracle.py)',
            'sample': 'first %d units (%d records) of rank 0\'s batch on %s against that chromosome\'s %d features, %.1f s '
                      '(+ %.1f s to build the Python buckets); GPU counts on the same sample are identical'
                      % (len(idx), len(recs), name, len(feats), dt, t_index),
            'cpu_model': _cpu_model(), 'host_cores': _host_cpus()}


def cpu_baseline_c(ann, units, n_sample, rec_per_unit, eng):
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
            'implementation': 'C restatement of the reference loop (oracle/te_oracle.c)',
            'sample': 'first %d units (%d records) of rank 0\'s batch, %.1f s; GPU counts on the same sample are bit-identical'
                      % (n_sample, n_sample * rec_per_unit, dt),
            'cpu_model': _cpu_model(), 'host_cores': _host_cpus()}


def cpu_baseline_c_threads(ann, units, n_units, rec_per_unit, counts_one_pass):
    """The C oracle on every core this process may use: threads over disjoint unit ranges (ctypes drops
    the GIL), counts summed -- the strongest CPU baseline SURVEY.md 8(d) asks for.  The sum over the whole
    batch must equal the GPU's one-pass counts."""
    import concurrent.futures as cf
    from oracle import c_oracle
    ci = c_oracle.COracleIndex.from_flat(ann.flat)
    threads = _host_cpus()
    per = (n_units + threads - 1) // threads
    ranges = [(a, min(n_units, a + per)) for a in range(0, n_units, per)]
    t = time.perf_counter()
    with cf.ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(lambda r: ci.bulk_count(*[u[r[0]:r[1]] for u in units], 20)[0], ranges))
    dt = time.perf_counter() - t
    total = np.sum(parts, axis=0).astype(np.uint64)
    if not np.array_equal(total, counts_one_pass):
        raise SystemExit('PARITY FAILURE: the threaded C oracle over the whole batch differs from the GPU counts')
    return {'value': round(n_units * rec_per_unit / dt / 1e6, 2), 'unit': 'Mreads/s', 'cores': threads, 'kind': 'port',
            'implementation': 'C restatement of the reference loop, one thread per disjoint unit range',
            'sample': 'the whole batch (%d units = %d records), %.1f s; summed counts equal the GPU\'s one-pass counts'
                      % (n_units, n_units * rec_per_unit, dt),
            'cpu_model': _cpu_model(), 'host_cores': threads}


def end_to_end(ann, eng, records):
    """What `te_count -i x.bam` does: a BAM file on disk -> on-GPU BGZF/BAM decode -> count kernel -> TSV
    (tools/bench_e2e.py).  Two payloads: 'illumina' (ACGT + binned qualities, what sequencers write) and
    'random' (incompressible sequence / quality bytes: the hard case for any inflater); the host decoder
    (16-thread CPU inflate) on the same file beside it."""
    import bench_e2e
    out = {}
    for payload, n in (('illumina', records), ('random', records // 4)):
        r = bench_e2e.run(n, 'pe', payload, 'gpu', reps=3, ann=ann)
        out[payload] = {'records': r['records'], 'bam_MB': r['bam_MB'],
                        'first_run': r['run0_s'], 'warm_run': r['run2_s'], 'decode': r.get('decode_report'),
                        'Mrecords_per_s_decode_and_count': r['Mrecords_per_s_decode_and_count'],
                        'Mrecords_per_s_end_to_end_with_index_load': r['Mrecords_per_s_end_to_end'],
                        'tsv_identical_to_oracle_formatting': r['tsv_ok']}
        if payload == 'illumina':
            h = bench_e2e.run(n, 'pe', payload, 'cpu', reps=1, ann=ann, reuse=r['bam_path'])
            out['illumina_host_decoder'] = {'Mrecords_per_s_decode_and_count': h['Mrecords_per_s_decode_and_count'],
                                            'threads': h['threads']}
        bench_e2e.cleanup(r)
    sc = bench_e2e.run_sc(records // 4, ann)
    out['single_cell_from_bam'] = {k: sc[k] for k in ('records', 'bam_MB', 'run0_s', 'run1_s', 'Mrecords_per_s', 'cells_with_hits', 'ok')}
    out['value'] = out['illumina']['Mrecords_per_s_decode_and_count']
    out['unit'] = 'Mrecords/s (BAM file -> counts, one GPU, index resident)'
    return out


def single_cell(ann, eng, records):
    """BASELINE config 4 through tec_sc_*: 10 k cells x 5 k reads, 737 280-barcode whitelist, 12-mer UMIs,
    bundle threshold 1e7; a reduced-size run of the same generator is compared with the Python oracle."""
    import bench_sc
    from te_counter_amd import synth
    res = bench_sc.oracle_check(eng, ann, 150_000, False)
    if not res['identical']:
        raise SystemExit('PARITY FAILURE in the reduced-size single-cell check')
    eng.upload_index(ann.flat)
    t0 = time.time()
    arrays, wl, tables = synth.make_sc_arrays(ann, records, n_cells=10000, seed=2004)
    gen_s = time.time() - t0
    n = len(arrays['pos'])
    first = bench_sc.run(eng, arrays, wl, tables, 10000, False, 10_000_000)[2]  # first run on this context: allocates the staging ring, survivor arrays and sorted views that the context then keeps
    out, st, t, _ = bench_sc.run(eng, arrays, wl, tables, 10000, False, 10_000_000)
    gpu_ms = eng.sc_gpu_ms
    alg = 32.0 * n  # 15 B + cb 8 + ub 8 + ub_len 1 per record
    traffic, traffic_src = sc_traffic(records)
    return {'metric': 'Mreads/s single-cell (--sc --se), host SoA batches -> COO counts', 'records': n,
            'value': round(n / t['total'] / 1e6, 2), 'unit': 'Mreads/s',
            'seconds': {k: round(v, 3) for k, v in t.items()}, 'first_run_seconds': {k: round(v, 3) for k, v in first.items()}, 'gen_s': round(gen_s, 1), 'stats': st,
            'cells_with_hits': int(len(out['cells'])), 'coo_nnz': int(len(out['coo_cell'])), 'oracle_check': res,
            'roofline': {'bound': 'hbm', 'achieved': round(alg / (gpu_ms * 1e-3) / 1e9, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(alg / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), 'traffic': traffic, 'traffic_source': traffic_src,
                         'kernel': 'all tec_sc_* kernels (filter, scatter, radix sorts, epochs, drop rule, query, RLE)',
                         'gpu_stream_ms': round(gpu_ms, 2), 'algorithmic_bytes': alg}}


def secondary(ann, eng, args):
    """The headline workload in coordinate-sorted order and BASELINE configs 2 and 5 on the count kernel, each with its own
    roofline; config 3 as named (bulk PE with --strand,
    which the reference refuses: te_count.py:58-59) through the stranded-bulk extension -- no reference output exists for
    it, so that entry says `"pinned": false`."""
    from te_counter_amd import synth
    out = {}
    for key, a, layout, order, genome in (('headline_125M_PE_hg38_sorted', ann, 'pe', 'sorted', 'hg38'),
                                          ('config2_100M_SE_hg38_shuffled', ann, 'se', 'shuffled', 'hg38'),
                                          ('config2_100M_SE_hg38_sorted', ann, 'se', 'sorted', 'hg38'),
                                          ('config5_100M_SE_3M_names', None, 'se', 'shuffled', 'custom3m'),
                                          ('config3_125M_PE_mm10_stranded_extension', None, 'pe', 'shuffled', 'mm10')):
        stranded = 'stranded' in key
        if a is None:
            a = synth.make_annotation('mm10') if genome == 'mm10' else synth.make_custom_annotation()
        run = BulkRun(eng, a, layout, order, 125_000_000 if layout == 'pe' else 100_000_000,
                      seed={'hg38': 2002, 'mm10': 2003}.get(genome, 2005), stranded=stranded)
        K = 20  # one job = begin, K launches over the resident batch, finish (D2H of the count vector: 24 MB for 3 M names)
        run.count(2)
        t = time.perf_counter()
        counts, st = run.count(K)
        dt = time.perf_counter() - t
        kernel_ms, launches = eng.bulk_kernel_time()
        reduce_ms, launch_ms = eng.bulk_reduce_time(), eng.bulk_launch_times()
        spill_ms, spill_names = eng.bulk_spill_time()
        single, _ = run.count(1)
        if not np.array_equal(counts, single * K):
            raise SystemExit('PARITY FAILURE: linearity violated in ' + key)
        out[key] = {'value': round(run.n_units * run.rec_per_unit * K / dt / 1e6, 1), 'unit': 'Mreads/s', 'steps': K,
                    'ms_per_step': round(dt / K * 1e3, 4),
                    'names': a.flat.n_names, 'assigned_frac': round(st['assigned'] / K / run.n_units, 4),
                    'roofline': run.roofline(kernel_ms, launches, reduce_ms, launch_ms,
                                             (None, None) if stranded else traffic_for(None, genome, layout, order, run.n_units))}
        if stranded:
            out[key]['pinned'] = False
            out[key]['note'] = 'extension: the reference raises NotImplementedError for bulk --strand; checked only against the oracle run per strand (tests/test_bulk_gpu.py)'

        if spill_names:  # names beyond the LDS histogram: listed by the count kernel, folded by spill_bucket_kernel behind it
            out[key]['spill_fold_ms_per_launch'] = round(spill_ms / max(1, launches), 3)
            out[key]['spilled_names_per_launch'] = int(spill_names // max(1, launches))
        out[key]['gpu_ms_per_step'] = round((kernel_ms + reduce_ms + spill_ms) / max(1, launches), 4)
        run.release()
    return out


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
