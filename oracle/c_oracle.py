"""TEST INFRASTRUCTURE ONLY: ctypes wrapper of oracle/libteoracle.so (oracle/te_oracle.c)."""

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libteoracle.so')
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(['make', '-C', _HERE, '-s'])
        l = C.CDLL(_SO)
        l.orc_index_build.restype = C.c_void_p
        l.orc_index_build.argtypes = [C.c_uint64] + [C.c_void_p] * 5 + [C.c_uint32, C.c_uint32]
        l.orc_index_free.argtypes = [C.c_void_p]
        l.orc_bulk_count.restype = C.c_int
        l.orc_bulk_count.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p]
        _lib = l
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class COracleIndex:
    """chrom_id/left/right/name_id/cls: one row per feature (same numbering as FlatIndex)."""

    def __init__(self, chrom_id, left, right, name_id, cls, n_chrom, n_names):
        a = [np.ascontiguousarray(chrom_id, np.int32), np.ascontiguousarray(left, np.int32),
             np.ascontiguousarray(right, np.int32), np.ascontiguousarray(name_id, np.uint32),
             np.ascontiguousarray(cls, np.uint32)]
        self.n_names = int(n_names)
        self._h = lib().orc_index_build(len(a[0]), *[_p(x) for x in a], int(n_chrom), int(n_names))

    @classmethod
    def from_flat(cls, flat):
        meta = flat.meta
        return cls(flat.chrom_id, flat.left, flat.right, meta & ((1 << 26) - 1), (meta >> 26) & 7,
                   max(1, len(flat.chrom_keys)), max(1, flat.n_names))

    def bulk_count(self, chrom, p1, p2, flag, mapq, q):
        """-> (counts uint64[n_names], stats dict); raises NameError on the enhancer crash."""
        a = [np.ascontiguousarray(chrom, np.int32), np.ascontiguousarray(p1, np.int32),
             np.ascontiguousarray(p2, np.int32), np.ascontiguousarray(flag, np.uint16),
             np.ascontiguousarray(mapq, np.uint8)]
        counts = np.zeros(self.n_names, np.uint64)
        st = np.zeros(4, np.uint64)
        rc = lib().orc_bulk_count(self._h, len(a[0]), *[_p(x) for x in a], int(q), _p(counts), _p(st))
        if rc:
            raise NameError("name 'barcode' is not defined")
        return counts, {'assigned': int(st[0]), 'lowq': int(st[1]), 'badchrom': int(st[2]), 'qcfail': int(st[3])}

    def __del__(self):
        try:
            lib().orc_index_free(self._h)
        except Exception:
            pass
