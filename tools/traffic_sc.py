#!/usr/bin/env python3
"""Condense a tools/prof_sc.sh output directory into profiles/<tag>_summary.json and the single-cell entry of
profiles/traffic.json (HBM bytes of ONE whole single-cell job: every kernel of tec_sc_* summed), read by bench.py.

    python tools/traffic_sc.py gpurun_out/prof_<tag> <tag> <records>

Same correction as tools/traffic.py: read = FETCH_SIZE * 1024 + 64 * TCC_EA0_RDREQ_128B, write = WRITE_SIZE * 1024."""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(path):
    """{counter: {kernel: sum over dispatches}}"""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(path, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r['Counter_Name']][r['Kernel_Name'].split('(')[0][-60:]] += float(r['Counter_Value'])
    return agg


def main():
    src, tag, records = sys.argv[1], sys.argv[2], int(sys.argv[3])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {'tag': tag, 'workload': 'sc/config4/%d' % records, 'bench_args': open(os.path.join(src, 'bench_args.txt')).read().strip(),
           'command': 'rocprofv3 <flags> -- python3 tools/bench_sc.py <bench_args>'}
    stats = glob.glob(os.path.join(src, 'trace', '**', '*kernel_stats.csv'), recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        out['kernel_stats'] = [{k: r[k] for k in ('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage')} for r in rows]
        out['kernel_sum_ms'] = round(sum(float(r['TotalDurationNs']) for r in rows if 'rocclr' not in r['Name']) / 1e6, 3)
        out['copy_and_fill_kernels_ms'] = round(sum(float(r['TotalDurationNs']) for r in rows if 'rocclr' in r['Name']) / 1e6, 3)
        open(os.path.join(root, 'profiles', '%s_kernel_stats.csv' % tag), 'w').write(open(stats[0]).read())
    agg = collections.defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(src, 'pmc_*'))):
        if os.path.isdir(d):
            for c, v in per_kernel(d).items():
                agg[c] = dict(v)
    total = {c: sum(v.values()) for c, v in agg.items()}
    out['pmc_sum_over_the_job'] = total
    if 'FETCH_SIZE' in total and 'WRITE_SIZE' in total:
        rd = total['FETCH_SIZE'] * 1024 + 64 * total.get('TCC_EA0_RDREQ_128B_sum', 0.0)
        wr = total['WRITE_SIZE'] * 1024
        out['hbm'] = {'read_bytes_per_job': rd, 'write_bytes_per_job': wr, 'hbm_bytes_per_job': rd + wr}
        top = sorted(((agg['FETCH_SIZE'].get(k, 0) + agg['WRITE_SIZE'].get(k, 0)) * 1024, k) for k in set(agg['FETCH_SIZE']) | set(agg['WRITE_SIZE']))
        out['hbm_top_kernels_uncorrected'] = [{'kernel': k, 'bytes': b} for b, k in reversed(top[-8:])]
        tpath = os.path.join(root, 'profiles', 'traffic.json')
        table = json.load(open(tpath))
        sys.path.insert(0, root)
        import bench
        table['sc/config4/%d' % records] = dict(out['hbm'], tag=tag, kernel_symbol='all tec_sc_* kernels of one job',
                                                 kernel_source_sha=bench.sc_source_sha(), hbm_bytes_per_launch=rd + wr)
        json.dump(table, open(tpath, 'w'), indent=1, sort_keys=True)
    json.dump(out, open(os.path.join(root, 'profiles', '%s_summary.json' % tag), 'w'), indent=1, sort_keys=True)
    print(json.dumps({k: out.get(k) for k in ('hbm', 'kernel_sum_ms', 'copy_and_fill_kernels_ms')}))


if __name__ == '__main__':
    main()
