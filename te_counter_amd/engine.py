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

    def close(self):
        if self._ctx:
            self._lib.tec_destroy(self._ctx)
            self._ctx = C.c_void_p()

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
        self._check(self._lib.tec_bulk_begin(self._ctx, int(mapq_min)))

    def bulk_submit(self, chrom, p1, p2, flag, mapq):
        """Host SoA batch (numpy).  Arrays are kept alive until bulk_finish / bulk_sync."""
        arrs = (np.ascontiguousarray(chrom, np.int32), np.ascontiguousarray(p1, np.int32),
                np.ascontiguousarray(p2, np.int32), np.ascontiguousarray(flag, np.uint16),
                np.ascontiguousarray(mapq, np.uint8))
        n = len(arrs[0])
        if any(len(a) != n for a in arrs):
            raise ValueError('unit arrays differ in length')
        t = C.c_uint64()
        self._check(self._lib.tec_bulk_submit(self._ctx, n, *[_ptr(a) for a in arrs], C.byref(t)))
        self._keep.append(arrs)
        if len(self._keep) > 2:  # a slot ring of 3: the oldest copy has certainly been consumed
            self._check(self._lib.tec_bulk_wait(self._ctx, t.value - 2))
            self._keep.pop(0)
        return t.value

    def bulk_submit_device(self, n, d_chrom, d_p1, d_p2, d_flag, d_mapq):
        """Device-resident batch: raw HBM addresses (ints)."""
        self._check(self._lib.tec_bulk_submit_device(self._ctx, int(n), C.c_void_p(d_chrom), C.c_void_p(d_p1),
                                                     C.c_void_p(d_p2), C.c_void_p(d_flag), C.c_void_p(d_mapq)))

    def bulk_sync(self):
        self._check(self._lib.tec_bulk_sync(self._ctx))
        self._keep = []

    def bulk_device_buffers(self):
        dc, ds = C.c_void_p(), C.c_void_p()
        self._check(self._lib.tec_bulk_device_buffers(self._ctx, C.byref(dc), C.byref(ds)))
        return dc.value, ds.value

    def bulk_finish(self):
        """-> (counts uint64[n_names], stats dict).  Raises NameError like the reference if a
        read hit only 'enhancer' features (te_count.py:145-147, 258-260)."""
        counts = np.zeros(self.n_names, np.uint64)
        st = _lib.Stats()
        rc = self._lib.tec_bulk_finish(self._ctx, _ptr(counts), C.byref(st))
        self._keep = []
        if rc == _lib.TEC_EENHANCER:
            raise NameError("name 'barcode' is not defined")
        self._check(rc)
        stats = {k: int(getattr(st, k)) for k, _ in _lib.Stats._fields_}
        return counts[:self._real_names], stats

    def bulk_kernel_time(self):
        ms, nl = C.c_double(), C.c_uint64()
        self._check(self._lib.tec_bulk_kernel_time(self._ctx, C.byref(ms), C.byref(nl)))
        return ms.value, nl.value
