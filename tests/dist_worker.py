"""Worker of tests/test_dist_gloo.py: one rank of a world_size-N gloo job on CPU.

Exercises the product's multi-GPU plumbing (te_counter_amd.dist.World: env rendezvous, round-robin
batch ownership, sum all-reduce of counts + stats) with the per-rank counting done by the CPU
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
from te_counter_amd import reads as R  # noqa: E402
from te_counter_amd.dist import World  # noqa: E402
from te_counter_amd.index import FlatIndex  # noqa: E402


def main():
    paired = sys.argv[1] == 'pe'
    batch = int(sys.argv[2])
    w = World.from_env(backend='gloo')
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
    w.close()


if __name__ == '__main__':
    main()
