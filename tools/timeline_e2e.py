#!/usr/bin/env python3
"""Condense a rocprofv3 kernel trace of tools/bench_e2e.py: the last pass over the file (everything behind the last gap
of more than 20 ms between kernels), per kernel family: launches, summed duration, first start and last end relative
to the pass's first kernel, and the share of the pass during which any kernel / an inflate kernel was running."""
import csv
import re
import sys


def union(spans):
    spans = sorted(spans)
    total, cur_a, cur_b = 0, None, None
    for a, b in spans:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                total += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        total += cur_b - cur_a
    return total


def main():
    rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
    rows.sort()
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - max(r[1] for r in rows[max(0, i - 50):i]) > 20_000_000:
            cut = i
    rows = rows[cut:]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    print('last pass: %d kernels over %.1f ms; some kernel running %.1f ms' % (len(rows), (t1 - t0) / 1e6, union([(a, b) for a, b, _ in rows]) / 1e6))
    fam = {}
    for a, b, n in rows:
        m = re.search(r'(\w+_kernel|__amd_rocclr_\w+|\w+)\s*(<|\(|$)', n.replace('(anonymous namespace)::', '').replace('void ', ''))
        short = (m.group(1) if m else n)[:40]
        f = fam.setdefault(short, [0, 0, a, b, []])
        f[0] += 1
        f[1] += b - a
        f[2] = min(f[2], a)
        f[3] = max(f[3], b)
        f[4].append((a, b))
    for k, f in sorted(fam.items(), key=lambda kv: -kv[1][1])[:14]:
        print('%-42s n=%4d  sum %7.2f ms  running %7.2f ms  first start %6.1f  last end %6.1f' %
              (k, f[0], f[1] / 1e6, union(f[4]) / 1e6, (f[2] - t0) / 1e6, (f[3] - t0) / 1e6))
    for k in ('ginflate_kernel', 'gresolve_kernel'):
        if k in fam:
            spans = sorted(fam[k][4])
            print(k, 'launches (start, duration ms):', ' '.join('%.0f+%.1f' % ((a - t0) / 1e6, (b - a) / 1e6) for a, b in spans))


if __name__ == '__main__':
    main()
