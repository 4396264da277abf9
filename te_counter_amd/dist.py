"""Multi-GPU plumbing: one process per GPU (torch.distributed, backend nccl == RCCL on ROCm).

The bulk path shards naturally: units are independent and counts are commutative integer sums.
Rank r decodes and counts its own byte range of the alignment file (te_counter_amd/bam.py
byte_range; ownership rules in include/tecount.h) against a full index replica on its GPU; the
collectives are small:
  * one all-gather of every range's (record count, first / last record, range tokens) -- the tokens
    prove that the ranges tile the file exactly, the counts give the parity that decides which
    records are PE mates (the reference pairs records 2k, 2k+1 of the FILE, te_count.py:78-79);
  * one sum all-reduce of the u64 count vector and the log counters (~330 KB for genes_tes,
    latency-bound on xGMI).
The reference has no distributed path at all.
"""

import os

import numpy as np


class World:
    def __init__(self, rank=0, size=1, local_rank=0, backend=None):
        self.rank, self.size, self.local_rank = rank, size, local_rank
        self.backend = backend
        self._own = False

    @classmethod
    def from_env(cls, backend=None):
        size = int(os.environ.get('WORLD_SIZE', 1))
        if size <= 1:
            return cls()
        import torch
        import torch.distributed as dist
        rank = int(os.environ.get('RANK', 0))
        local = int(os.environ.get('LOCAL_RANK', 0))
        w = cls(rank, size, local)
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            # TEC_DIST_BACKEND=gloo: rehearsal on a box with fewer GPUs than ranks (the ranks share devices, --device N)
            backend = backend or os.environ.get('TEC_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
            if backend == 'nccl':
                torch.cuda.set_device(local)
                dist.init_process_group(backend, device_id=torch.device('cuda', local))
            else:
                dist.init_process_group(backend)
            w._own = True
        w.backend = dist.get_backend()
        return w

    def mine(self, batch_ordinal):
        """Does this rank count batch number `batch_ordinal` (round-robin sharding of a source that
        cannot be cut into byte ranges, e.g. SAM text)?"""
        return batch_ordinal % self.size == self.rank

    def _device(self):
        import torch
        return torch.device('cuda', self.local_rank) if self.backend == 'nccl' else torch.device('cpu')

    def sum_counts(self, counts, stats_vec, reduced=None):
        """All-reduce (sum) of the per-rank count vector and stats; returns numpy arrays.  counts None: the vector
        has been reduced on the device already (sum_device_counts) -- `reduced` is handed back and only the stats travel."""
        if self.size == 1:
            return (reduced if counts is None else counts), stats_vec
        import torch
        import torch.distributed as dist
        head = np.zeros(0, np.int64) if counts is None else counts.astype(np.int64)
        t = torch.from_numpy(np.concatenate([head, np.asarray(stats_vec, np.int64)])).to(self._device())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        out = t.cpu().numpy()
        if counts is None:
            return reduced, out
        return out[:len(counts)].astype(np.uint64), out[len(counts):]

    def sum_device_counts(self, d_counts_ptr, n_names):
        """In-place RCCL sum all-reduce of a count vector that lives in HBM (libtecount's own buffer, wrapped without
        a copy; the same call bench.py times at N > 1).  The caller reads the reduced vector through the engine."""
        import torch
        import torch.distributed as dist

        class _Wrap:
            __cuda_array_interface__ = {'shape': (int(n_names),), 'typestr': '<i8', 'data': (int(d_counts_ptr), False), 'version': 2}
        t = torch.as_tensor(_Wrap(), device=torch.device('cuda', self.local_rank))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()

    def gather(self, obj):
        """Every rank's small Python object, in rank order."""
        if self.size == 1:
            return [obj]
        import torch.distributed as dist
        out = [None] * self.size
        dist.all_gather_object(out, obj)
        return out

    def raise_together(self, exc):
        """A failure that only some ranks see (e.g. the record that kills the reference sits in one
        range) must stop every rank, or the others wait in the next collective forever: each rank
        reports (kind, message) or None; the first reported failure, in rank order, is raised by
        all."""
        kinds = {'TypeError': TypeError, 'AttributeError': AttributeError, 'NameError': NameError,
                 'IOError': IOError, 'OSError': IOError, 'RuntimeError': RuntimeError, 'AssertionError': AssertionError}
        mine = None if exc is None else (type(exc).__name__, str(exc))
        for r in self.gather(mine):
            if r is not None:
                raise kinds.get(r[0], RuntimeError)(r[1])

    def close(self):
        if self._own:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()


def check_ranges(reports, file_size):
    """reports: per rank {'has_records', 'first_token', 'end_token'} in rank order.  The ranges tile the
    file iff every range's end token (where the first record it does NOT own starts) is the first
    token of the next range that has records, and the last one ends at the end of the file.
    Range 0 knows where the file's first record starts (right behind the BAM header) and reports that as
    its first token even when it owns no record itself (a header larger than its share of a small file):
    the first range that does have records must start exactly there, so its speculative start is checked
    like every other one."""
    prev = None
    for r, rep in enumerate(reports):
        if not rep['has_records']:
            if r == 0 and tuple(rep['first_token']) != (0, 0):
                prev = (0, rep['first_token'])  # the header's token stands in for "range 0 ends here"
            continue
        if prev is not None and tuple(prev[1]) != tuple(rep['first_token']):
            raise RuntimeError('byte-range decoding is inconsistent: range %d ends at %r but range %d starts at %r '
                               '(a record-boundary guess was wrong; rerun on one GPU)'
                               % (prev[0], tuple(prev[1]), r, tuple(rep['first_token'])))
        prev = (r, rep['end_token'])
    if not any(rep['has_records'] for rep in reports):
        return
    if prev is not None and tuple(prev[1]) != (file_size, 0):
        raise RuntimeError('byte-range decoding is inconsistent: the last range ends at %r, not at the end of the file' % (tuple(prev[1]),))


def plan_pairs(reports, rank):
    """PE pairing across byte ranges.  The reference pairs records (2k, 2k+1) of the whole file
    (te_count.py:78-79), so whether a range's first record is a first or a second mate depends on
    the number of records before it.  reports: per rank {'n_records', 'first_rec'}.
    -> (skip_first, extra_mate, unit_ordinal0): this rank pairs its records from index skip_first on;
    if that leaves its last record alone, extra_mate (the next range's first record) completes the
    pair, or None at the end of the file (an odd trailing record is dropped, as StopIteration does in
    the reference)."""
    before = sum(rep['n_records'] for rep in reports[:rank])
    n = reports[rank]['n_records']
    skip = before & 1 if n else 0
    left = n - skip
    extra = None
    if left & 1:
        for rep in reports[rank + 1:]:
            if rep['n_records']:
                extra = rep['first_rec']
                break
    return skip, extra, (before + skip) // 2


# ---- collectives on numpy data (single cell across GPUs: te_counter_amd/scdist.py) --------------------

def _as_tensor(world, a):
    import torch
    a = np.ascontiguousarray(a)
    view = {np.dtype('uint64'): np.int64, np.dtype('uint32'): np.int32, np.dtype('uint16'): np.int16}.get(a.dtype)
    t = torch.from_numpy(a.view(view) if view else a)
    return t.to(world._device())


def _as_array(t, dtype):
    a = t.cpu().numpy()
    return a.view(dtype) if a.dtype != dtype else a


def world_sum(world, value):
    """Sum of one integer over the ranks."""
    if hasattr(world, 'sum_int'):
        return world.sum_int(value)
    if world.size == 1:
        return int(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(value)], dtype=torch.int64, device=world._device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def world_max_u8(world, a):
    """Element-wise max (= OR for 0/1 tables) of a uint8 array over the ranks."""
    if hasattr(world, 'max_u8'):
        return world.max_u8(a)
    if world.size == 1:
        return a
    import torch.distributed as dist
    t = _as_tensor(world, a)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return _as_array(t, np.dtype('uint8'))


def all_gather_array(world, a):
    """Every rank's 1-D array (lengths may differ), in rank order."""
    if hasattr(world, 'gather_array'):
        return world.gather_array(a)
    if world.size == 1:
        return [a]
    import torch
    import torch.distributed as dist
    sizes = world.gather(int(len(a)))
    cap = max(1, max(sizes))
    pad = np.zeros(cap, a.dtype)
    pad[:len(a)] = a
    t = _as_tensor(world, pad)
    outs = [torch.empty_like(t) for _ in range(world.size)]
    dist.all_gather(outs, t)
    return [_as_array(o, a.dtype)[:n].copy() for o, n in zip(outs, sizes)]


_COLUMN_TORCH = {4: 'int32', 8: 'int64'}  # (bit patterns travel; signedness does not matter)


def _wrap_device(address, n, elem_bytes, device_index):
    """A torch tensor over n elements of library-owned HBM (no copy)."""
    import torch

    class _Wrap:
        __cuda_array_interface__ = {'shape': (int(n),), 'typestr': '<i%d' % elem_bytes, 'data': (int(address), False), 'version': 2}
    return torch.as_tensor(_Wrap(), device=torch.device('cuda', device_index))


def exchange_survivors(world, eng):
    """The survivors of a single-cell run change GPUs without visiting host memory: eng.sc_deal groups them by
    destination in HBM, every pair of GPUs exchanges its group once (RCCL point-to-point on tensors wrapped over the
    library's deal / landing columns), eng.sc_adopt_commit orders what arrived by record ordinal.
    With the gloo backend -- ranks that share one GPU, or CPU-only rehearsals -- the groups are staged through host
    tensors, because gloo moves host memory only.  A test world may bring its own transport (`exchange_device`)."""
    counts = eng.sc_deal(world.size)
    sizes = world.gather(counts)  # sizes[src][dst]
    n_in = sum(sizes[src][world.rank] for src in range(world.size))
    eng.sc_adopt_begin(n_in)
    out_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    in_off = np.concatenate([[0], np.cumsum([sizes[src][world.rank] for src in range(world.size)])]).astype(np.int64)
    if hasattr(world, 'exchange_device'):
        world.exchange_device(eng, sizes, out_off, in_off)
    else:
        import torch
        import torch.distributed as dist
        on_host = world.backend != 'nccl'
        for col in range(8):
            src_addr, elem = eng.sc_deal_column(col)
            dst_addr, _ = eng.sc_adopt_column(col)
            send = _wrap_device(src_addr, out_off[-1], elem, eng.device) if out_off[-1] else None
            recv = _wrap_device(dst_addr, n_in, elem, eng.device) if n_in else None
            ops, keep, landed = [], [], []
            for peer in range(world.size):
                n_out, n_peer = sizes[world.rank][peer], sizes[peer][world.rank]
                if peer == world.rank:
                    if n_out:
                        eng.sc_adopt_copy(col, in_off[peer], src_addr + int(out_off[peer]) * elem, n_out)
                    continue
                if n_out:
                    t = send[out_off[peer]:out_off[peer] + n_out]
                    if on_host:
                        t = t.cpu()  # (synchronous copy: gloo takes host tensors only)
                    keep.append(t)
                    ops.append(dist.P2POp(dist.isend, t, peer))
                if n_peer:
                    r = recv[in_off[peer]:in_off[peer] + n_peer]
                    if on_host:
                        h = torch.empty(n_peer, dtype=r.dtype)
                        landed.append((r, h))
                        r = h
                    ops.append(dist.P2POp(dist.irecv, r, peer))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            for r, h in landed:
                r.copy_(h)
            if not on_host:
                torch.cuda.synchronize()
    eng.sc_adopt_commit()


def exchange_rows(world, parts):
    """All-to-all of row sets: parts[d] = {column: array} destined for rank d (same columns and dtypes on
    every rank).  Returns the columns received from all ranks, concatenated in source-rank order.
    Point-to-point sends underneath (RCCL over xGMI, or gloo): every pair of GPUs exchanges exactly its
    rows once."""
    if hasattr(world, 'exchange'):
        return world.exchange(parts)
    if world.size == 1:
        return parts[0]
    import torch
    import torch.distributed as dist
    cols = list(parts[0].keys())
    sizes = world.gather([int(len(p[cols[0]])) for p in parts])  # sizes[src][dst]
    out = {}
    for c in cols:
        dtype = parts[0][c].dtype
        recv = [None] * world.size
        ops, keep = [], []
        for peer in range(world.size):
            n_in = sizes[peer][world.rank]
            if peer == world.rank:
                recv[peer] = np.ascontiguousarray(parts[peer][c])
                continue
            if sizes[world.rank][peer]:
                t = _as_tensor(world, parts[peer][c])
                keep.append(t)
                ops.append(dist.P2POp(dist.isend, t, peer))
            if n_in:
                r = torch.empty(n_in, dtype=_as_tensor(world, np.zeros(1, dtype)).dtype, device=world._device())
                recv[peer] = r
                ops.append(dist.P2POp(dist.irecv, r, peer))
            else:
                recv[peer] = np.zeros(0, dtype)
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        out[c] = np.concatenate([x if isinstance(x, np.ndarray) else _as_array(x, dtype) for x in recv])
    return out
