"""Worker of tests/test_dist_gloo.py: one rank of a world_size-N gloo job on CPU.

Exercises the product's multi-GPU plumbing without a GPU: te_counter_amd.dist.World (env rendezvous,
gather, sum all-reduce), byte-range sharded decoding of a BAM file (bam.BamSource(byte_range=...), range
tokens, dist.check_ranges), cross-range PE pairing (dist.plan_pairs, reads.RecordKeeper) and, for
sources that cannot be cut, round-robin batch ownership.  The per-rank counting is done by the CPU
oracle, since there is no GPU here.  Prints OK on rank 0 when the reduced result equals the
single-process result."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import datagen as dg  # noqa: E402
from oracle import c_oracle  # noqa: E402
from te_counter_amd import bam, dist  # noqa: E402
from te_counter_amd import reads as R  # noqa: E402
from te_counter_amd.index import FlatIndex  # noqa: E402


def round_robin(w, paired, batch):
    feats = dg.make_features(51, 800, span=300000)
    flat = FlatIndex.from_features(feats)
    recs = dg.make_records(52, 12001, feats, span=300000, paired=paired)
    ci = c_oracle.COracleIndex.from_flat(flat)
    src = R.ListSource(recs, batch_size=batch)
    tid_map = flat.tid_map(src.reference_names)
    pa = R.PairAssembler(tid_map, 20) if paired else None
    counts = np.zeros(flat.n_names, np.uint64)
    stats = np.zeros(4, np.int64)
    full_units = []
    for ordinal, b in enumerate(src.batches()):
        units = pa.units(b) if paired else R.se_units(b, tid_map)
        full_units.append(units)
        if w.mine(ordinal) and len(units[0]):
            c, st = ci.bulk_count(*units, 20)
            counts += c
            stats += [st['assigned'], st['lowq'], st['badchrom'], st['qcfail']]
    total, tstats = w.sum_counts(counts, stats)
    if w.rank == 0:
        allu = [np.concatenate([u[i] for u in full_units]) for i in range(5)]
        want, wst = ci.bulk_count(*allu, 20)
        ok = np.array_equal(total, want) and list(tstats) == [wst['assigned'], wst['lowq'], wst['badchrom'], wst['qcfail']]
        print('OK' if ok else 'MISMATCH', w.size, int(want.sum()))


def byte_ranges(w, paired, path, n_expected):
    """Every rank decodes its byte range of the BAM at `path` (written by the test) and counts it."""
    feats = dg.make_features(51, 800, span=300000)
    flat = FlatIndex.from_features(feats)
    ci = c_oracle.COracleIndex.from_flat(flat)
    size = os.path.getsize(path)
    src = bam.BamSource(path, batch_size=977, byte_range=bam.byte_range(size, w.rank, w.size), threads=2)
    tid_map = flat.tid_map(src.reference_names)
    keeper = R.RecordKeeper() if paired else None
    counts = np.zeros(flat.n_names, np.uint64)
    stats = np.zeros(4, np.int64)
    n_records = 0
    for b in src.batches():
        n_records += len(b)
        if paired:
            keeper.add(b)
        else:
            c, st = ci.bulk_count(*R.se_units(b, tid_map), 20)
            counts += c
            stats += [st['assigned'], st['lowq'], st['badchrom'], st['qcfail']]
    first, end, has = src.range_tokens()
    src.close()
    reports = w.gather({'n_records': n_records, 'has_records': has and n_records > 0, 'first_token': first, 'end_token': end,
                        'first_rec': keeper.record(0) if (paired and n_records) else None})
    dist.check_ranges(reports, size)
    if paired:
        skip, extra, unit0 = dist.plan_pairs(reports, w.rank)
        units = keeper.units(tid_map, 20, skip, extra)
        if len(units[0]):
            c, st = ci.bulk_count(*units, 20)
            counts += c
            stats += [st['assigned'], st['lowq'], st['badchrom'], st['qcfail']]
    total, tstats = w.sum_counts(counts, list(stats) + [n_records])
    if w.rank == 0:
        whole = bam.BamSource(path, batch_size=1 << 20)
        pa = R.PairAssembler(tid_map, 20) if paired else None
        parts = [pa.units(b) if paired else R.se_units(b, tid_map) for b in whole.batches()]
        whole.close()
        allu = [np.concatenate([u[i] for u in parts]) for i in range(5)]
        want, wst = ci.bulk_count(*allu, 20)
        ok = (np.array_equal(total, want) and list(tstats[:4]) == [wst['assigned'], wst['lowq'], wst['badchrom'], wst['qcfail']] and
              int(tstats[4]) == n_expected)
        split = [r['n_records'] for r in reports]
        print('OK' if ok else 'MISMATCH', w.size, int(want.sum()), 'records per range', split,
              'odd boundary' if any(sum(split[:k]) & 1 for k in range(1, len(split))) else 'even boundaries')


def main():
    w = dist.World.from_env(backend='gloo')
    if sys.argv[1] in ('se', 'pe'):
        round_robin(w, sys.argv[1] == 'pe', int(sys.argv[2]))
    else:
        byte_ranges(w, sys.argv[1] == 'bam_pe', sys.argv[2], int(sys.argv[3]))
    w.close()


if __name__ == '__main__':
    main()
