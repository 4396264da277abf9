"""Thin object wrapper over the C ABI (include/tecount.h): one Engine per GPU."""

import ctypes as C

import numpy as np

from . import _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    def __init__(self, device=0):
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        rc = self._lib.tec_create(int(device), C.byref(self._ctx))
        if rc != _lib.TEC_OK:
            msg = self._lib.tec_last_error(None)
            raise _lib.TecError(rc, msg.decode() if msg else '?')
        self.device = int(device)
        self.n_names = 0
        self._keep = []  # host arrays of in-flight submits
        self._pinned = []  # (address, ctypes buffer) of every tec_alloc_pinned block: freed at close()
        self._ring, self._ring_cap, self._ring_slot = None, 0, 0
        self._ring_ticket = [0, 0, 0]  # ticket of the submit that last copied out of each pinned slot

    def close(self):
        if self._ctx:
            self._keep, self._ring = [], None
            for addr, _ in self._pinned:
                self._lib.tec_free_pinned(self._ctx, C.c_void_p(addr))
            self._pinned = []
            self._lib.tec_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def pinned_empty(self, n, dtype):
        """A numpy array over page-locked host memory (tec_alloc_pinned): hipMemcpyAsync from it really is
        asynchronous.  The memory belongs to the engine and is released by close()."""
        dtype = np.dtype(dtype)
        nbytes = max(1, int(n) * dtype.itemsize)
        p = C.c_void_p()
        self._check(self._lib.tec_alloc_pinned(self._ctx, nbytes, C.byref(p)))
        buf = (C.c_uint8 * nbytes).from_address(p.value)
        self._pinned.append((p.value, buf))
        return np.frombuffer(buf, np.uint8, int(n) * dtype.itemsize).view(dtype)

    def _staging_slot(self, n):
        """One of three pinned SoA slots of at least n units (the same ring depth as the device side)."""
        if n > self._ring_cap:
            cap = max(int(n), 1 << 16)
            if self._ring:  # (the caller has drained the copies out of the old slots)
                old = {a.ctypes.data for slot in self._ring for a in slot}
                self._ring = None
                for addr, _ in [e for e in self._pinned if e[0] in old]:
                    self._lib.tec_free_pinned(self._ctx, C.c_void_p(addr))
                self._pinned = [e for e in self._pinned if e[0] not in old]
            self._ring = [tuple(self.pinned_empty(cap, dt) for dt in (np.int32, np.int32, np.int32, np.uint16, np.uint8))
                          for _ in range(3)]
            self._ring_cap = cap
            self._ring_ticket = [0, 0, 0]
        k = self._ring_slot
        self._ring_slot = (k + 1) % 3
        if self._ring_ticket[k]:
            # the H2D copy that last read this pinned slot must have finished before the host writes into it again
            # (tec_bulk_wait = hipEventSynchronize on that batch's copy event: a host-side wait, unlike the
            # stream-side dependency that protects the device slot)
            self._check(self._lib.tec_bulk_wait(self._ctx, self._ring_ticket[k]))
            self._ring_ticket[k] = 0
        return self._ring[k], k

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        return _lib.check(self._ctx, rc)

    # ------------------------------------------------------------------ index
    def upload_index(self, flat, bin_shift=0):
        self._check(self._lib.tec_index_build(self._ctx, len(flat), _ptr(flat.chrom_id), _ptr(flat.left),
                                              _ptr(flat.right), _ptr(flat.meta), max(1, len(flat.chrom_keys)),
                                              max(1, flat.n_names), int(bin_shift)))
        self.n_names = max(1, flat.n_names)
        self._real_names = flat.n_names

    def index_info(self):
        ne, nb, by, sh = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int()
        self._check(self._lib.tec_index_info(self._ctx, C.byref(ne), C.byref(nb), C.byref(sh), C.byref(by)))
        return {'n_entries': ne.value, 'n_bins': nb.value, 'bin_shift': sh.value, 'device_bytes': by.value}

    # ------------------------------------------------------------------ bulk
    def bulk_begin(self, mapq_min):
        self._keep = []
        self._ring_ticket = [0, 0, 0]  # (tec_bulk_begin drains both streams)
        self._check(self._lib.tec_bulk_begin(self._ctx, int(mapq_min)))

    def bulk_set_stranded(self, on=True):
        """tec_bulk_set_stranded: this run counts against a strand-split index (FlatIndex.strand_split)."""
        self._check(self._lib.tec_bulk_set_stranded(self._ctx, int(bool(on))))

    def bulk_submit(self, chrom, p1, p2, flag, mapq, stage=True):
        """Host SoA batch (numpy).  stage=True: the batch is copied into one of three page-locked slots
        (tec_alloc_pinned) first, so that the call returns while the copy to HBM is still running and the
        caller may reuse its arrays at once; stage=False: the arrays are handed over as they are (callers
        that build their batches in pinned_empty() arrays) and kept alive until their copy has finished."""
        arrs = (np.ascontiguousarray(chrom, np.int32), np.ascontiguousarray(p1, np.int32),
                np.ascontiguousarray(p2, np.int32), np.ascontiguousarray(flag, np.uint16),
                np.ascontiguousarray(mapq, np.uint8))
        n = len(arrs[0])
        if any(len(a) != n for a in arrs):
            raise ValueError('unit arrays differ in length')
        if stage and n:
            if n > self._ring_cap:  # the slots are replaced: nothing may still be copying out of the old ones
                self._check(self._lib.tec_bulk_sync(self._ctx))
                self._keep = []
            slot, ring_k = self._staging_slot(n)
            for dst, src in zip(slot, arrs):
                dst[:n] = src
            arrs = tuple(dst[:n] for dst in slot)
        else:
            ring_k = None
        t = C.c_uint64()
        self._check(self._lib.tec_bulk_submit(self._ctx, n, *[_ptr(a) for a in arrs], C.byref(t)))
        if ring_k is not None:
            self._ring_ticket[ring_k] = t.value
        else:
            # caller-owned arrays: kept alive until their copy has finished (the device ring is three deep, so
            # by the time ticket t is queued the copy of ticket t - 2 is waited for and its arrays let go)
            self._keep.append(arrs)
            if len(self._keep) > 2:
                self._check(self._lib.tec_bulk_wait(self._ctx, t.value - 2))
                self._keep.pop(0)
        return t.value

    def bulk_submit_device(self, n, d_chrom, d_p1, d_p2, d_flag, d_mapq):
        """Device-resident batch: raw HBM addresses (ints)."""
        self._check(self._lib.tec_bulk_submit_device(self._ctx, int(n), C.c_void_p(d_chrom), C.c_void_p(d_p1),
                                                     C.c_void_p(d_p2), C.c_void_p(d_flag), C.c_void_p(d_mapq)))

    def bulk_count_noalloc(self, n, d_chrom, d_p1, d_p2, d_flag, d_mapq, mapq_min, d_counts_out, d_stats_out=0, stream=0):
        """tec_bulk_count_noalloc: raw HBM addresses (ints) in and out; only queues work (synchronise the stream
        before reading the outputs)."""
        self._check(self._lib.tec_bulk_count_noalloc(self._ctx, int(n), C.c_void_p(d_chrom), C.c_void_p(d_p1), C.c_void_p(d_p2),
                                                     C.c_void_p(d_flag), C.c_void_p(d_mapq), int(mapq_min),
                                                     C.c_void_p(d_counts_out), C.c_void_p(d_stats_out or None),
                                                     C.c_void_p(stream or None)))

    def bulk_sync(self):
        self._check(self._lib.tec_bulk_sync(self._ctx))
        self._keep = []
        self._ring_ticket = [0, 0, 0]

    def bulk_device_buffers(self):
        dc, ds = C.c_void_p(), C.c_void_p()
        self._check(self._lib.tec_bulk_device_buffers(self._ctx, C.byref(dc), C.byref(ds)))
        return dc.value, ds.value

    def bulk_finish(self, raise_enhancer=True):
        """-> (counts uint64[n_names], stats dict).  A read that hit only 'enhancer' features kills the
        reference with NameError (te_count.py:145-147, 258-260): raised here, or -- with
        raise_enhancer=False, for callers that first have to tell the other ranks -- returned as a
        third value."""
        counts = np.zeros(self.n_names, np.uint64)
        st = _lib.Stats()
        rc = self._lib.tec_bulk_finish(self._ctx, _ptr(counts), C.byref(st))
        self._keep = []
        self._ring_ticket = [0, 0, 0]
        enhancer_only = rc == _lib.TEC_EENHANCER
        if enhancer_only and raise_enhancer:
            raise NameError("name 'barcode' is not defined")
        if not enhancer_only:
            self._check(rc)
        stats = {k: int(getattr(st, k)) for k, _ in _lib.Stats._fields_}
        if raise_enhancer:
            return counts[:self._real_names], stats
        return counts[:self._real_names], stats, enhancer_only

    def bulk_kernel_time(self):
        ms, nl = C.c_double(), C.c_uint64()
        self._check(self._lib.tec_bulk_kernel_time(self._ctx, C.byref(ms), C.byref(nl)))
        return ms.value, nl.value

    def bulk_launch_times(self):
        """Per-launch durations (ms) of the count kernel since bulk_begin (HIP events on its stream)."""
        n = C.c_uint64()
        self._check(self._lib.tec_bulk_launch_times(self._ctx, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.float32)
        if n.value:
            self._check(self._lib.tec_bulk_launch_times(self._ctx, _ptr(out), n.value, C.byref(n)))
        return out

    def bulk_spill_time(self):
        ms, n = C.c_double(), C.c_uint64()
        self._check(self._lib.tec_bulk_spill_time(self._ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def bulk_reduce_time(self):
        ms = C.c_double()
        self._check(self._lib.tec_bulk_reduce_time(self._ctx, C.byref(ms)))
        return ms.value

    # ------------------------------------------------------------------ single cell
    def sc_begin(self, stranded, mapq_min, whitelist24, bundle_keys, class_index_chrom, class_skip):
        wl = np.ascontiguousarray(whitelist24, np.uint8)
        cic = np.ascontiguousarray(class_index_chrom, np.int32)
        csk = np.ascontiguousarray(class_skip, np.uint8)
        self._check(self._lib.tec_sc_begin(self._ctx, int(bool(stranded)), int(mapq_min), wl.shape[0], _ptr(wl),
                                           int(bundle_keys), len(cic), _ptr(cic), _ptr(csk)))
        self._n_wl = int(wl.shape[0])

    def sc_submit(self, cls, pos, end, flag, mapq, cb24, ub24, tagflags):
        a = (np.ascontiguousarray(cls, np.int32), np.ascontiguousarray(pos, np.int32), np.ascontiguousarray(end, np.int32),
             np.ascontiguousarray(flag, np.uint16), np.ascontiguousarray(mapq, np.uint8),
             np.ascontiguousarray(cb24, np.uint8), np.ascontiguousarray(ub24, np.uint8),
             np.ascontiguousarray(tagflags, np.uint8))
        n = len(a[0])
        if a[5].shape != (n, 24) or a[6].shape != (n, 24) or any(len(x) != n for x in a):
            raise ValueError('single-cell batch arrays differ in length')
        self._check(self._lib.tec_sc_submit(self._ctx, n, *[_ptr(x) for x in a]))

    def sc_part1(self):
        """-> (barcode ids ascending, raw counts, first-seen ordinals).  Raises AssertionError where
        the reference asserts on a missing CB|CR / UB|UR tag (te_count.py:409, 426)."""
        n = C.c_uint64()
        rc = self._lib.tec_sc_part1(self._ctx, C.byref(n))
        if rc == _lib.TEC_ENOCB:
            raise AssertionError('CB or CR tag not found!')
        if rc == _lib.TEC_ENOUB:
            raise AssertionError('UB or UR tag not found!')
        self._check(rc)
        ids = np.zeros(n.value, np.uint32)
        cnt = np.zeros(n.value, np.uint64)
        first = np.zeros(n.value, np.uint64)
        self._check(self._lib.tec_sc_part1_fetch(self._ctx, _ptr(ids), _ptr(cnt), _ptr(first)))
        return ids, cnt, first

    # ---- part 1 in steps and the multi-GPU hooks (include/tecount.h; orchestrated by scdist.run_parts)
    _SURVIVOR_COLUMNS = (('bcid', np.uint32), ('u0', np.uint64), ('u1', np.uint64), ('u2', np.uint64), ('rec', np.uint64),
                         ('klass', np.int32), ('left', np.int32), ('rite', np.int32))

    def sc_set_record_base(self, base, total_records):
        self._check(self._lib.tec_sc_set_record_base(self._ctx, int(base), int(total_records)))

    def sc_survivors(self):
        n = C.c_uint64()
        self._check(self._lib.tec_sc_survivors(self._ctx, C.byref(n)))
        rows = {k: np.zeros(n.value, dt) for k, dt in self._SURVIVOR_COLUMNS}
        self._check(self._lib.tec_sc_survivors_fetch(self._ctx, *[_ptr(rows[k]) for k, _ in self._SURVIVOR_COLUMNS]))
        return rows

    def sc_survivors_replace(self, rows):
        cols = [np.ascontiguousarray(rows[k], dt) for k, dt in self._SURVIVOR_COLUMNS]
        self._check(self._lib.tec_sc_survivors_replace(self._ctx, len(cols[0]), *[_ptr(c) for c in cols]))

    # ---- the exchange on the device (include/tecount.h, tec_sc_deal ...): what dist.exchange_survivors drives
    def sc_deal(self, world_size):
        """Groups the survivors by destination GPU (barcode id % world) in HBM -> rows per destination."""
        counts = np.zeros(int(world_size), np.uint64)
        self._check(self._lib.tec_sc_deal(self._ctx, int(world_size), _ptr(counts)))
        return [int(c) for c in counts]

    def _sc_column(self, fn, column):
        p, e = C.c_void_p(), C.c_uint32()
        self._check(fn(self._ctx, int(column), C.byref(p), C.byref(e)))
        return int(p.value or 0), int(e.value)

    def sc_deal_column(self, column):
        """-> (device address, element bytes) of deal column 0..7 (bcid u0 u1 u2 rec klass left rite)."""
        return self._sc_column(self._lib.tec_sc_deal_column, column)

    def sc_adopt_begin(self, n_rows):
        self._check(self._lib.tec_sc_adopt_begin(self._ctx, int(n_rows)))

    def sc_adopt_column(self, column):
        return self._sc_column(self._lib.tec_sc_adopt_column, column)

    def sc_adopt_copy(self, column, dst_row, src_address, n_rows):
        self._check(self._lib.tec_sc_adopt_copy(self._ctx, int(column), int(dst_row), C.c_void_p(int(src_address)), int(n_rows)))

    def sc_adopt_commit(self):
        self._check(self._lib.tec_sc_adopt_commit(self._ctx))

    def sc_host_row_bytes(self):
        b = C.c_uint64()
        self._check(self._lib.tec_sc_host_row_bytes(self._ctx, C.byref(b)))
        return int(b.value)

    def sc_part1_sort(self):
        n = C.c_uint64()
        self._check(self._lib.tec_sc_part1_sort(self._ctx, C.byref(n)))
        return int(n.value)

    def sc_sort_key_bits(self):
        """Bits of part 1's sort key string (0: the columns were sorted one by one); after the sort."""
        b = C.c_uint32()
        self._check(self._lib.tec_sc_sort_key_bits(self._ctx, C.byref(b)))
        return int(b.value)

    def sc_missing_tags(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self._lib.tec_sc_missing_tags(self._ctx, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def sc_long_umi(self):
        a = C.c_uint64()
        self._check(self._lib.tec_sc_long_umi(self._ctx, C.byref(a)))
        return int(a.value)

    def sc_epoch_marks(self, lower):
        n = C.c_uint64()
        self._check(self._lib.tec_sc_epoch_marks(self._ctx, int(lower), C.byref(n)))
        return int(n.value)

    def sc_epoch_rank(self, v):
        n = C.c_uint64()
        self._check(self._lib.tec_sc_epoch_rank(self._ctx, int(v), C.byref(n)))
        return int(n.value)

    def sc_epoch_kth(self, k):
        v = C.c_uint64()
        self._check(self._lib.tec_sc_epoch_kth(self._ctx, int(k), C.byref(v)))
        return int(v.value)

    def sc_part1_finish(self, bounds):
        b = np.ascontiguousarray(bounds, np.uint64)
        n = C.c_uint64()
        self._check(self._lib.tec_sc_part1_finish(self._ctx, len(b), _ptr(b) if len(b) else None, C.byref(n)))
        ids = np.zeros(n.value, np.uint32)
        cnt = np.zeros(n.value, np.uint64)
        first = np.zeros(n.value, np.uint64)
        self._check(self._lib.tec_sc_part1_fetch(self._ctx, _ptr(ids), _ptr(cnt), _ptr(first)))
        return ids, cnt, first

    def sc_presence(self, n_bundles):
        out = np.zeros(int(n_bundles) * self._n_wl, np.uint8)
        self._check(self._lib.tec_sc_presence(self._ctx, int(n_bundles), _ptr(out)))
        return out

    def sc_part23(self, selected_ids, n_bundles=None, presence=None):
        """-> dict(kept_lines, cells, cell_hits, coo_cell, coo_name, coo_count).  presence: the OR of all
        GPUs' (bundle, barcode) tables (multi-GPU), None: this GPU's own."""
        sel = np.ascontiguousarray(selected_ids, np.uint32)
        kept, nc, nz = C.c_uint64(), C.c_uint64(), C.c_uint64()
        if presence is None:
            rc = self._lib.tec_sc_part23(self._ctx, len(sel), _ptr(sel), C.byref(kept), C.byref(nc), C.byref(nz))
        else:
            pres = np.ascontiguousarray(presence, np.uint8)
            rc = self._lib.tec_sc_part23_global(self._ctx, len(sel), _ptr(sel), int(n_bundles), _ptr(pres), C.byref(kept),
                                                C.byref(nc), C.byref(nz))
        if rc == _lib.TEC_ENOSTRAND:
            raise KeyError('strand')
        self._check(rc)
        out = {'kept_lines': kept.value, 'cells': np.zeros(nc.value, np.uint32), 'cell_hits': np.zeros(nc.value, np.uint64),
               'coo_cell': np.empty(nz.value, np.uint32), 'coo_name': np.empty(nz.value, np.uint32),
               'coo_count': np.empty(nz.value, np.uint64)}  # (tec_sc_fetch fills every element)
        self._check(self._lib.tec_sc_fetch(self._ctx, _ptr(out['cells']), _ptr(out['cell_hits']), _ptr(out['coo_cell']),
                                           _ptr(out['coo_name']), _ptr(out['coo_count'])))
        return out

    def sc_stats(self):
        st = _lib.ScStats()
        self._check(self._lib.tec_sc_stats(self._ctx, C.byref(st)))
        self.sc_gpu_ms = int(st.gpu_us) / 1000.0  # GPU stream time of the run's phases (not part of the log counters)
        return {k: int(getattr(st, k)) for k, _ in _lib.ScStats._fields_ if k != 'gpu_us'}

    def sc_end(self):
        self._check(self._lib.tec_sc_end(self._ctx))
