"""The on-GPU BAM decoder (csrc/gbam.hip) as the alignment source of the counting paths.

Stands where the reference has `pysam.AlignmentFile(filename, 'r')` and `next(sam)`
(te_count/te_count.py:65,78-79 / 190,203 / 351,393).  The host moves compressed bytes only; BGZF
inflation, record boundaries, field extraction and PE pairing run on the MI355X, and the rows go
straight into the count kernel / the single-cell ingest (include/tecount.h, tec_gbam_*)."""

import ctypes as C

import numpy as np

from . import _lib

NONE64 = 2 ** 64 - 1


class GbamRec(C.Structure):
    _fields_ = [('chrom', C.c_int32), ('pos', C.c_int32), ('flag', C.c_uint16), ('mapq', C.c_uint8), ('qkey', C.c_uint64)]

    def as_tuple(self):
        return (int(self.chrom), int(self.pos), int(self.flag), int(self.mapq), int(self.qkey))

    @classmethod
    def from_tuple(cls, t):
        return cls(*t)


class GbamReport(C.Structure):
    _fields_ = [('n_records', C.c_uint64), ('has_records', C.c_int), ('first_token', C.c_uint64 * 2),
                ('end_token', C.c_uint64 * 2), ('first_noend_live', C.c_uint64), ('first_name_mismatch', C.c_uint64),
                ('units_counted', C.c_uint64), ('odd_record_left', C.c_int), ('first_rec', GbamRec), ('last_rec', GbamRec),
                ('blocks', C.c_uint64), ('blocks_redone_on_host', C.c_uint64), ('compressed_bytes', C.c_uint64),
                ('inflated_bytes', C.c_uint64), ('segment_repairs', C.c_uint64), ('seconds_stage_wait', C.c_double),
                ('seconds_inflate', C.c_double), ('seconds_segments', C.c_double), ('seconds_rows', C.c_double),
                ('seconds_stage_alloc', C.c_double), ('seconds_stage_read', C.c_double)]

    def as_dict(self):
        return {'n_records': int(self.n_records), 'has_records': bool(self.has_records),
                'first_token': (int(self.first_token[0]), int(self.first_token[1])),
                'end_token': (int(self.end_token[0]), int(self.end_token[1])),
                'first_noend_live': int(self.first_noend_live), 'first_name_mismatch': int(self.first_name_mismatch),
                'units_counted': int(self.units_counted), 'odd_record_left': bool(self.odd_record_left),
                'first_rec': self.first_rec.as_tuple(), 'last_rec': self.last_rec.as_tuple(),
                'blocks': int(self.blocks), 'blocks_redone_on_host': int(self.blocks_redone_on_host),
                'compressed_bytes': int(self.compressed_bytes), 'inflated_bytes': int(self.inflated_bytes),
                'segment_repairs': int(self.segment_repairs),
                'seconds': {'stage_wait': round(self.seconds_stage_wait, 4), 'inflate': round(self.seconds_inflate, 4),
                            'segments': round(self.seconds_segments, 4), 'rows': round(self.seconds_rows, 4),
                            'stage_alloc': round(self.seconds_stage_alloc, 4), 'stage_read': round(self.seconds_stage_read, 4)}}


def is_bgzf(path):
    """BGZF-compressed (a BAM file) rather than SAM text?"""
    try:
        with open(path, 'rb') as fh:
            m = fh.read(4)
    except OSError:
        return False
    return len(m) == 4 and m[0] == 31 and m[1] == 139 and m[2] == 8 and (m[3] & 4) != 0


class GpuBam:
    """One reader over a byte range of a BAM file, decoding on the engine's GPU."""

    def __init__(self, engine, path, byte_range=None):
        self._lib = _lib.load()
        self._eng = engine
        self._h = C.c_void_p()
        a, b = (0, NONE64) if byte_range is None else (int(byte_range[0]), int(byte_range[1]))
        rc = self._lib.tec_gbam_open(engine._ctx, str(path).encode(), a, b, C.byref(self._h))
        if rc != _lib.TEC_OK:
            msg = self._lib.tec_last_error(engine._ctx)
            raise IOError('cannot read %s: %s' % (path, msg.decode() if msg else '?'))
        n = self._lib.tec_gbam_n_refs(self._h)
        self.reference_names = [self._lib.tec_gbam_ref_name(self._h, i).decode() for i in range(n)]
        self.file_size = int(self._lib.tec_gbam_file_size(self._h))

    def _check(self, rc):
        if rc != _lib.TEC_OK:
            msg = self._lib.tec_gbam_last_error(self._h)
            text = msg.decode() if msg else ''
            if not text:
                m2 = self._lib.tec_last_error(self._eng._ctx)
                text = m2.decode() if m2 else '?'
            if rc == _lib.TEC_EINVAL:
                raise IOError(text)
            raise _lib.TecError(rc, text)

    def run_bulk(self, paired, mapq_min, tid_map, defer_pairing=False):
        """Decode this reader's records and count them (between bulk_begin and bulk_finish)."""
        tm = np.ascontiguousarray(tid_map, np.int32)
        rep = GbamReport()
        self._check(self._lib.tec_gbam_run_bulk(self._h, int(bool(paired)), int(mapq_min), tm.ctypes.data_as(C.c_void_p),
                                                len(tm), int(bool(defer_pairing)), C.byref(rep)))
        return rep.as_dict()

    def finish_pairs(self, mapq_min, skip_first, extra_mate, unit_ordinal0):
        n_units, mism = C.c_uint64(), C.c_uint64()
        extra = GbamRec.from_tuple(extra_mate) if extra_mate is not None else None
        self._check(self._lib.tec_gbam_finish_pairs(self._h, int(mapq_min), int(bool(skip_first)),
                                                    C.byref(extra) if extra is not None else None, int(unit_ordinal0),
                                                    C.byref(n_units), C.byref(mism)))
        return int(n_units.value), int(mism.value)

    def run_sc(self, mapq_min, tid_class):
        tc = np.ascontiguousarray(tid_class, np.int32)
        rep = GbamReport()
        self._check(self._lib.tec_gbam_run_sc(self._h, int(mapq_min), tc.ctypes.data_as(C.c_void_p), len(tc), C.byref(rep)))
        return rep.as_dict()

    def decode(self):
        """Decode only -> (report, dict of numpy arrays with every field the host decoder delivers)."""
        rep = GbamReport()
        self._check(self._lib.tec_gbam_decode(self._h, C.byref(rep)))
        n = int(rep.n_records)
        a = {'tid': np.zeros(n, np.int32), 'pos': np.zeros(n, np.int32), 'end': np.zeros(n, np.int32),
             'flag': np.zeros(n, np.uint16), 'mapq': np.zeros(n, np.uint8), 'qkey': np.zeros(n, np.uint64),
             'cb24': np.zeros((n, 24), np.uint8), 'ub24': np.zeros((n, 24), np.uint8), 'tagflags': np.zeros(n, np.uint8)}
        p = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
        self._check(self._lib.tec_gbam_fetch(self._h, 0, n, p(a['tid']), p(a['pos']), p(a['end']), p(a['flag']), p(a['mapq']),
                                             p(a['qkey']), p(a['cb24']), p(a['ub24']), p(a['tagflags'])))
        return rep.as_dict(), a

    def close(self):
        if self._h:
            self._lib.tec_gbam_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
