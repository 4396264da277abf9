#!/usr/bin/env python3
"""Condense a tools/prof.sh output directory into profiles/<tag>_summary.json and update
profiles/traffic.json (HBM bytes per launch of the count kernel, read by bench.py).

    python tools/traffic.py gpurun_out/prof_<tag> <tag> <genome>/<layout>/<order>/<units>

HBM bytes (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB and come from
the L2's fabric request counters; on gfx950 FETCH_SIZE tallies every read request at 64 B, so a
128-B request is counted at half its size.  The request-size breakdown is collected in its own
pass and the missing half of each 128-B request is added back:
    read_bytes  = FETCH_SIZE * 1024 + 64 * TCC_EA0_RDREQ_128B
    write_bytes = WRITE_SIZE * 1024
"""
import collections
import csv
import glob
import json
import os
import sys


def kernel_means(path, needle='count_bulk'):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(path, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if needle in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    src, tag, key = sys.argv[1], sys.argv[2], sys.argv[3]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {'tag': tag, 'workload': key, 'bench_args': open(os.path.join(src, 'bench_args.txt')).read().strip(),
           'command': 'rocprofv3 <flags> -- python3 bench.py --legs headline --steps 5 --warmup 1 <bench_args>'}
    stats = glob.glob(os.path.join(src, 'trace', '**', '*kernel_stats.csv'), recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        out['kernel_stats'] = [{k: r[k] for k in ('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs')}
                               for r in rows]
        dst = os.path.join(root, 'profiles', '%s_kernel_stats.csv' % tag)
        open(dst, 'w').write(open(stats[0]).read())
    pmc, n = {}, {}
    for d in sorted(glob.glob(os.path.join(src, 'pmc_*'))):
        if os.path.isdir(d):
            m, c = kernel_means(d)
            pmc.update(m)
            n.update(c)
    out['pmc_mean_per_dispatch'] = pmc
    out['pmc_dispatches'] = n
    # config 5: the fold of the spill list (csrc/spill.hip) runs behind every count launch
    fold = {}
    for d in sorted(glob.glob(os.path.join(src, 'pmc_*'))):
        if os.path.isdir(d):
            fold.update(kernel_means(d, 'spill_bucket')[0])
    if fold:
        out['pmc_mean_per_dispatch_spill_bucket_kernel'] = fold
    if 'FETCH_SIZE' in pmc and 'WRITE_SIZE' in pmc:
        rd = pmc['FETCH_SIZE'] * 1024 + 64 * pmc.get('TCC_EA0_RDREQ_128B_sum', 0.0)
        wr = pmc['WRITE_SIZE'] * 1024
        out['hbm'] = {'read_bytes_per_launch': rd, 'write_bytes_per_launch': wr, 'hbm_bytes_per_launch': rd + wr,
                      'correction': 'FETCH_SIZE*1024 + 64*TCC_EA0_RDREQ_128B (128-B requests are tallied at 64 B on gfx950); WRITE_SIZE*1024'}
        tpath = os.path.join(root, 'profiles', 'traffic.json')
        try:
            table = json.load(open(tpath))
        except (OSError, ValueError):
            table = {}
        sys.path.insert(0, root)
        import bench
        symbol = next((r['Name'] for r in out.get('kernel_stats', []) if 'count_bulk' in r['Name']), None)
        table[key] = dict(out['hbm'], tag=tag, kernel_symbol=symbol, kernel_source_sha=bench.kernel_source_sha())
        json.dump(table, open(tpath, 'w'), indent=1, sort_keys=True)
    json.dump(out, open(os.path.join(root, 'profiles', '%s_summary.json' % tag), 'w'), indent=1, sort_keys=True)
    print(json.dumps({k: out[k] for k in ('hbm',) if k in out}))
    if 'kernel_stats' in out:
        for r in out['kernel_stats'][:3]:
            print(r['Name'][:70], r['Calls'], r['AverageNs'])


if __name__ == '__main__':
    main()
