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


class FakeSegment:
    """The attributes of a pysam.AlignedSegment that the reference reads (te_count.py:81-98, 204-214, 394-438)."""

    def __init__(self, r, tid):
        self.reference_id = tid
        self.reference_name = r.get('rname')
        self.reference_start = r['pos']
        self.reference_end = r['end']  # (None: no CIGAR)
        self.flag = r['flag']
        self.mapping_quality = r['mapq']
        self.query_name = r.get('qname', 'q')
        self._tags = list(r.get('tags', {}).items())

    def get_tags(self):
        return list(self._tags)


class FakeAlignmentFile:
    """Stands in for an open pysam.AlignmentFile (pysam is not installed here): `references` and iteration."""

    def __init__(self, records):
        self.references = tuple(R.ListSource(records).reference_names)
        tid = {n: i for i, n in enumerate(self.references)}
        self._segments = [FakeSegment(r, tid.get(r.get('rname'), -1)) for r in records]

    def __iter__(self):
        return iter(self._segments)


class ThreadWorld:
    """Test double of te_counter_amd.dist.World for N ranks that are THREADS of one process (e.g. N engine
    contexts on one GPU): the collectives scdist / measure use, over a shared mailbox and a barrier."""

    class Shared:
        def __init__(self, size):
            import threading
            self.size = size
            self.barrier = threading.Barrier(size)
            self.box = [None] * size

    def __init__(self, rank, shared):
        self.rank, self.size, self.local_rank, self.backend = rank, shared.size, 0, 'threads'
        self._s = shared

    def gather(self, obj):
        self._s.box[self.rank] = obj
        self._s.barrier.wait()
        out = list(self._s.box)
        self._s.barrier.wait()
        return out

    def raise_together(self, exc):
        for r in self.gather(None if exc is None else exc):
            if r is not None:
                raise r

    def sum_int(self, v):
        return sum(self.gather(int(v)))

    def max_u8(self, a):
        import numpy as np
        return np.maximum.reduce(self.gather(a))

    def gather_array(self, a):
        return self.gather(a)

    def exchange(self, parts):
        import numpy as np
        allparts = self.gather(parts)
        return {c: np.concatenate([allparts[src][self.rank][c] for src in range(self.size)]) for c in parts[0]}

    def exchange_device(self, eng, sizes, out_off, in_off):
        """dist.exchange_survivors' transport for contexts of one process: every rank publishes the device addresses of
        its deal columns, every rank copies its groups out of its peers' columns (device to device), and nobody moves on
        before all copies are done."""
        addrs = self.gather([eng.sc_deal_column(c) for c in range(8)])
        offs = self.gather([int(x) for x in out_off])
        for src in range(self.size):
            n = sizes[src][self.rank]
            for c in range(8):
                a, elem = addrs[src][c]
                if n:
                    eng.sc_adopt_copy(c, int(in_off[src]), a + offs[src][self.rank] * elem, n)
        self._s.barrier.wait()

    def mine(self, ordinal):
        return ordinal % self.size == self.rank

    def sum_counts(self, counts, stats_vec):
        import numpy as np
        got = self.gather((counts, np.asarray(stats_vec, np.int64)))
        return np.sum([g[0] for g in got], axis=0).astype(np.uint64), np.sum([g[1] for g in got], axis=0)

    def close(self):
        pass
