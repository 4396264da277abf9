"""The BAM reader's own DEFLATE decoders against zlib -- csrc/inflate_fast.h (host decoder) and the host
build of csrc/ginflate.h (what one GPU lane runs in the on-GPU front end): every stream zlib can
produce for blocks up to 64 KiB must decode to the same bytes, and damaged streams must be
refused (or decode to something zlib agrees with) without touching memory outside the buffers."""

import ctypes as C
import zlib

import numpy as np
import pytest

from te_counter_amd import _lib


HOST, ZLIB, LANE = 0, 1, 2   # tec_inflate_raw's decoders
DECODERS = pytest.mark.parametrize('decoder', [HOST, LANE], ids=['host', 'gpu_lane'])


def _inflate(comp, n, decoder=HOST, shift=0):
    """shift: misalignment of the output address (the lane decoder copies matches in 8-byte groups)."""
    lib = _lib.load()
    out = (C.c_uint8 * (n + 128 + shift))()
    guard = bytes(range(64))
    C.memmove(C.addressof(out) + shift, guard, 64)
    C.memmove(C.addressof(out) + shift + 64 + n, guard, 64)
    dst = C.cast(C.addressof(out) + shift + 64, C.c_void_p)
    rc = lib.tec_inflate_raw(comp, len(comp), dst, n, decoder)
    assert bytes(out[shift:shift + 64]) == guard and bytes(out[shift + 64 + n:shift + 128 + n]) == guard, 'wrote outside the output buffer'
    return rc, bytes(out[shift + 64:shift + 64 + n])


def _payloads():
    rng = np.random.default_rng(5)
    yield b''
    yield b'a'
    yield b'abc' * 7
    yield bytes(65280)                                   # one long run (dist 1 matches)
    yield bytes(rng.integers(0, 256, 65280, dtype=np.uint8))   # incompressible -> stored blocks
    yield bytes(rng.integers(0, 4, 65280, dtype=np.uint8))     # 2-bit alphabet: short codes, long matches
    yield bytes(rng.integers(65, 69, 30000, dtype=np.uint8)) + bytes(rng.integers(0, 256, 30000, dtype=np.uint8))
    yield (b'ACGT' * 25 + b'\xff' * 100 + b'NH:i:1\x00') * 300  # BAM-like records
    # BAM-like with real entropy in seq/qual
    recs = bytearray()
    for i in range(260):
        recs += b'\x10\x01\x00\x00' + int(i * 37).to_bytes(4, 'little') + b'A00123:45:HXX:1:%d:%d\x00' % (1100 + i, i * 7)
        recs += bytes(rng.integers(0, 256, 50, dtype=np.uint8)) + bytes(rng.integers(2, 41, 100, dtype=np.uint8)) + b'NHC\x01HIC\x01'
    yield bytes(recs)
    for n in (1, 2, 3, 7, 8, 9, 255, 256, 257, 258, 259, 1000, 32768, 32769, 65535):
        yield bytes(rng.integers(0, 7, n, dtype=np.uint8))
    for _ in range(40):
        n = int(rng.integers(1, 65281))
        k = int(rng.integers(1, 257))
        yield bytes(rng.integers(0, k, n, dtype=np.uint8))


@pytest.mark.parametrize('level,strategy', [(1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                            (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE), (0, zlib.Z_DEFAULT_STRATEGY)])
@DECODERS
def test_matches_zlib(level, strategy, decoder):
    for k, data in enumerate(_payloads()):
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 1 if k % 5 == 4 else 9, strategy)
        comp = co.compress(data) + co.flush()
        rc, got = _inflate(comp, len(data), decoder, shift=k % 16)
        assert rc == 0 and got == data, (level, strategy, len(data))
        # wrong expected sizes are refused
        if len(data) > 1:
            assert _inflate(comp, len(data) - 1, decoder)[0] != 0
        assert _inflate(comp, len(data) + 1, decoder)[0] != 0


@DECODERS
def test_every_history_window_and_memory_level(decoder):
    """Streams from compressors with a smaller history window (2^9 .. 2^15 bytes) and every memLevel (which sets how
    many symbols a deflate block holds: memLevel 1 cuts a 64 KiB payload into some 250 deflate blocks, each with its
    own code tables), every strategy: what other BGZF writers may be configured to emit."""
    rng = np.random.default_rng(21)
    recs = bytearray()
    for i in range(330):
        recs += b'\x10\x01\x00\x00' + int(i * 37).to_bytes(4, 'little') + b'A00123:45:HXX:1:%d:%d\x00' % (1100 + i, i * 7)
        recs += bytes(rng.integers(0, 4, 50, dtype=np.uint8) * 17) + bytes(rng.choice([2, 12, 23, 37], 100).astype(np.uint8)) + b'NHC\x01HIC\x01'
    payloads = [bytes(recs)[:65280], bytes(rng.integers(0, 256, 20000, dtype=np.uint8)) + bytes(recs)[:30000], b'ab' * 30000]
    k = 0
    for wbits in range(9, 16):
        for memlevel in (1, 2, 5, 8, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
                data = payloads[k % len(payloads)]
                co = zlib.compressobj(1 + k % 9, zlib.DEFLATED, -wbits, memlevel, strategy)
                comp = co.compress(data) + co.flush()
                rc, got = _inflate(comp, len(data), decoder, shift=k % 16)
                assert rc == 0 and got == data, (wbits, memlevel, strategy)
                k += 1


@DECODERS
def test_multi_block_streams(decoder):
    rng = np.random.default_rng(9)
    for _ in range(20):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        parts, comp = [], b''
        for _ in range(int(rng.integers(2, 6))):
            d = bytes(rng.integers(0, int(rng.integers(2, 200)), int(rng.integers(1, 12000)), dtype=np.uint8))
            parts.append(d)
            comp += co.compress(d) + co.flush(zlib.Z_FULL_FLUSH if rng.random() < 0.5 else zlib.Z_SYNC_FLUSH)
        comp += co.flush()
        data = b''.join(parts)
        rc, got = _inflate(comp, len(data), decoder)
        assert rc == 0 and got == data


@DECODERS
def test_damaged_streams_never_disagree_with_zlib(decoder):
    rng = np.random.default_rng(13)
    base = bytes(rng.integers(0, 20, 20000, dtype=np.uint8))
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(base) + co.flush()
    accepted = 0
    for trial in range(600):
        b = bytearray(comp)
        kind = trial % 3
        if kind == 0:
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            del b[int(rng.integers(1, len(b))):]
        else:
            for _ in range(4):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        rc, got = _inflate(bytes(b), len(base), decoder, shift=trial % 16)
        zrc, zgot = _inflate(bytes(b), len(base), ZLIB)
        if rc == 0:  # whatever the fast decoder accepts, zlib decodes to the same bytes
            accepted += 1
            assert zrc == 0 and zgot == got
    assert accepted < 600
