"""Shared test plumbing: golden record dicts -> unit arrays via the product's host code."""

import numpy as np

from te_counter_amd import reads as R
from te_counter_amd.index import FlatIndex


def units_from_records(flat, records, paired, q=20, batch_size=1 << 20):
    """Concatenated unit arrays (chrom, p1, p2, flag, mapq) and the record count."""
    src = R.ListSource(records, batch_size=batch_size)
    tid_map = flat.tid_map(src.reference_names)
    pa = R.PairAssembler(tid_map, q) if paired else None
    parts = []
    for b in src.batches():
        parts.append(pa.units(b) if paired else R.se_units(b, tid_map))
    if not parts:
        z = np.zeros(0, np.int32)
        return (z, z, z, np.zeros(0, np.uint16), np.zeros(0, np.uint8)), 0
    return tuple(np.concatenate([p[i] for p in parts]) for i in range(5)), len(records)


def flat_from_golden(g):
    return FlatIndex.from_features(g['features'])
