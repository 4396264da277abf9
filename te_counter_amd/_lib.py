"""ctypes binding of libtecount.so (include/tecount.h).

The HIP library is the product: there is no CPU fallback.  If the shared object is missing or
does not load, importing anything that needs it fails loudly.
"""

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libtecount.so')

TEC_OK = 0
TEC_EINVAL, TEC_EHIP, TEC_ENOMEM, TEC_ESTATE, TEC_EENHANCER, TEC_EOVERFLOW = -1, -2, -3, -4, -5, -6
TEC_ENOCB, TEC_ENOUB, TEC_ENOSTRAND = -7, -8, -9
STRAND_MISSING = 7

CLASS_OTHER, CLASS_GENE, CLASS_TE, CLASS_SNRNA, CLASS_ENHANCER = 0, 1, 2, 3, 4
META_NAME_BITS, META_CLASS_SHIFT, META_STRAND_SHIFT = 26, 26, 29


class TecError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('libtecount error %d: %s' % (code, msg))
        self.code = code


class Stats(C.Structure):
    _fields_ = [('units', C.c_uint64), ('assigned', C.c_uint64), ('lowq', C.c_uint64),
                ('badchrom', C.c_uint64), ('qcfail', C.c_uint64)]


class ScStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ('records', 'qcfail', 'lowq', 'invalid_barcode', 'skipped_chrom',
                                          'survivors', 'keys', 'bundles', 'kept_lines', 'assigned', 'gpu_us')]


_p = C.c_void_p
_u64 = C.c_uint64
_SIGNATURES = {
    # name: (restype, argtypes)  -- one row per symbol declared in include/tecount.h
    'tec_version': (C.c_int, []),
    'tec_last_error': (C.c_char_p, [_p]),
    'tec_create': (C.c_int, [C.c_int, C.POINTER(_p)]),
    'tec_destroy': (None, [_p]),
    'tec_alloc_pinned': (C.c_int, [_p, C.c_size_t, C.POINTER(_p)]),
    'tec_free_pinned': (C.c_int, [_p, _p]),
    'tec_index_build': (C.c_int, [_p, _u64, _p, _p, _p, _p, C.c_uint32, C.c_uint32, C.c_int]),
    'tec_index_info': (C.c_int, [_p, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(C.c_int), C.POINTER(_u64)]),
    'tec_bulk_begin': (C.c_int, [_p, C.c_int]),
    'tec_bulk_submit': (C.c_int, [_p, _u64, _p, _p, _p, _p, _p, C.POINTER(_u64)]),
    'tec_bulk_wait': (C.c_int, [_p, _u64]),
    'tec_bulk_submit_device': (C.c_int, [_p, _u64, _p, _p, _p, _p, _p]),
    'tec_bulk_sync': (C.c_int, [_p]),
    'tec_bulk_set_stranded': (C.c_int, [_p, C.c_int]),
    'tec_bulk_count_noalloc': (C.c_int, [_p, _u64, _p, _p, _p, _p, _p, C.c_int, _p, _p, _p]),
    'tec_bulk_device_buffers': (C.c_int, [_p, C.POINTER(_p), C.POINTER(_p)]),
    'tec_bulk_finish': (C.c_int, [_p, _p, C.POINTER(Stats)]),
    'tec_bulk_kernel_time': (C.c_int, [_p, C.POINTER(C.c_double), C.POINTER(_u64)]),
    'tec_bulk_launch_times': (C.c_int, [_p, _p, _u64, C.POINTER(_u64)]),
    'tec_bulk_spill_time': (C.c_int, [_p, C.POINTER(C.c_double), C.POINTER(_u64)]),
    'tec_bulk_reduce_time': (C.c_int, [_p, C.POINTER(C.c_double)]),
    'tec_sc_begin': (C.c_int, [_p, C.c_int, C.c_int, C.c_uint32, _p, _u64, C.c_uint32, _p, _p]),
    'tec_sc_submit': (C.c_int, [_p, _u64, _p, _p, _p, _p, _p, _p, _p, _p]),
    'tec_sc_part1': (C.c_int, [_p, C.POINTER(_u64)]),
    'tec_sc_set_record_base': (C.c_int, [_p, _u64, _u64]),
    'tec_sc_survivors': (C.c_int, [_p, C.POINTER(_u64)]),
    'tec_sc_survivors_fetch': (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'tec_sc_survivors_replace': (C.c_int, [_p, _u64, _p, _p, _p, _p, _p, _p, _p, _p]),
    'tec_sc_part1_sort': (C.c_int, [_p, C.POINTER(_u64)]),
    'tec_sc_missing_tags': (C.c_int, [_p, C.POINTER(_u64), C.POINTER(_u64)]),
    'tec_sc_long_umi': (C.c_int, [_p, C.POINTER(_u64)]),
    'tec_sc_epoch_marks': (C.c_int, [_p, C.c_int64, C.POINTER(_u64)]),
    'tec_sc_epoch_rank': (C.c_int, [_p, _u64, C.POINTER(_u64)]),
    'tec_sc_epoch_kth': (C.c_int, [_p, _u64, C.POINTER(_u64)]),
    'tec_sc_part1_finish': (C.c_int, [_p, C.c_uint32, _p, C.POINTER(_u64)]),
    'tec_sc_presence': (C.c_int, [_p, C.c_uint32, _p]),
    'tec_sc_part23_global': (C.c_int, [_p, C.c_uint32, _p, C.c_uint32, _p, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    'tec_sc_part1_fetch': (C.c_int, [_p, _p, _p, _p]),
    'tec_sc_part23': (C.c_int, [_p, C.c_uint32, _p, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    'tec_sc_fetch': (C.c_int, [_p, _p, _p, _p, _p, _p]),
    'tec_sc_stats': (C.c_int, [_p, C.POINTER(ScStats)]),
    'tec_sc_end': (C.c_int, [_p]),
    'tec_write_dense_tsv': (C.c_int, [C.c_char_p, C.c_char_p, _u64, C.c_uint32, C.c_uint32, C.c_char_p, _p, _u64, _p, _p, _p]),
    'tec_inflate_raw': (C.c_int, [_p, _u64, _p, _u64, C.c_int]),
    'tec_bam_open': (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_p)]),
    'tec_bam_open_range': (C.c_int, [C.c_char_p, C.c_int, _u64, _u64, C.POINTER(_p)]),
    'tec_bam_range_tokens': (C.c_int, [_p, _p, _p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'tec_bam_file_size': (_u64, [_p]),
    'tec_bam_close': (None, [_p]),
    'tec_bam_last_error': (C.c_char_p, [_p]),
    'tec_bam_n_refs': (C.c_int, [_p]),
    'tec_bam_ref_len': (C.c_uint32, [_p, C.c_int]),
    'tec_gbam_open': (C.c_int, [_p, C.c_char_p, _u64, _u64, C.POINTER(_p)]),
    'tec_gbam_close': (None, [_p]),
    'tec_gbam_last_error': (C.c_char_p, [_p]),
    'tec_gbam_n_refs': (C.c_int, [_p]),
    'tec_gbam_ref_name': (C.c_char_p, [_p, C.c_int]),
    'tec_gbam_file_size': (_u64, [_p]),
    'tec_gbam_run_bulk': (C.c_int, [_p, C.c_int, C.c_int, _p, C.c_uint32, C.c_int, _p]),
    'tec_gbam_finish_pairs': (C.c_int, [_p, C.c_int, C.c_int, _p, _u64, C.POINTER(_u64), C.POINTER(_u64)]),
    'tec_gbam_run_sc': (C.c_int, [_p, C.c_int, _p, C.c_uint32, _p]),
    'tec_gbam_decode': (C.c_int, [_p, _p]),
    'tec_gbam_fetch': (C.c_int, [_p, _u64, _u64, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'tec_sc_deal': (C.c_int, [_p, C.c_uint32, _p]),
    'tec_sc_deal_column': (C.c_int, [_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]),
    'tec_sc_adopt_begin': (C.c_int, [_p, _u64]),
    'tec_sc_adopt_column': (C.c_int, [_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]),
    'tec_sc_adopt_copy': (C.c_int, [_p, C.c_int, _u64, C.c_void_p, _u64]),
    'tec_sc_adopt_commit': (C.c_int, [_p]),
    'tec_sc_host_row_bytes': (C.c_int, [_p, C.POINTER(_u64)]),
    'tec_sc_sort_key_bits': (C.c_int, [_p, C.POINTER(C.c_uint32)]),
    'tec_sc_submit_device': (C.c_int, [_p, _u64, _p, _p, _p, _p, _p, _p, _p, _p]),
    'tec_bam_ref_name': (C.c_char_p, [_p, C.c_int]),
    'tec_bam_read': (C.c_int, [_p, _u64, _p, _p, _p, _p, _p, _p, _p, _p, _p, C.POINTER(_u64)]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGNATURES)


def load():
    """dlopen libtecount.so and type its entry points.  Raises if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.environ.get('TEC_NO_TORCH'):
        # torch wheels ship their own libamdhip64.so.7 / libhsa-runtime64; two HIP runtimes in one
        # process cannot both own the GPU.  Importing torch first makes the loader resolve our
        # DT_NEEDED libamdhip64.so.7 to the copy torch has already mapped.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise ImportError('%s is missing: build it with `make` (or __graft_entry__.build()); '
                          'te_counter_amd has no CPU fallback' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.tec_version() < 100:
        raise ImportError('libtecount.so is too old')
    _lib = lib
    return lib


def check(ctx, rc):
    if rc != TEC_OK:
        msg = load().tec_last_error(ctx)
        raise TecError(rc, msg.decode() if msg else '?')
    return rc
