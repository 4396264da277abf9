"""Alignment files: the native BGZF/BAM/SAM decoder of libtecount (csrc/bamio.hip) as a batch
source, and a small BAM writer used to make synthetic inputs and test fixtures.

`open_alignment` stands where the reference has `pysam.AlignmentFile(filename, 'r')`
(te_count/te_count.py:65, 190, 351)."""

import ctypes as C
import struct
import zlib

import numpy as np

from . import _lib
from .reads import TAG_BYTES, RecordBatch

NO_END = -2 ** 31  # reference_end of a record without CIGAR (pysam gives None)


class BamSource:
    def __init__(self, path, batch_size=1 << 22, threads=0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        rc = self._lib.tec_bam_open(str(path).encode(), int(threads), C.byref(self._h))
        if rc != _lib.TEC_OK:
            msg = self._lib.tec_bam_last_error(None)
            raise IOError('cannot read %s: %s' % (path, msg.decode() if msg else '?'))
        self.batch_size = int(batch_size)
        n = self._lib.tec_bam_n_refs(self._h)
        self.reference_names = [self._lib.tec_bam_ref_name(self._h, i).decode() for i in range(n)]

    def batches(self, want_tags=False, prefetch=2):
        """Batches in file order.  With prefetch > 0 a reader thread decodes up to that many
        batches ahead (tec_bam_read runs without the GIL), so decoding overlaps with the caller's
        unit assembly and GPU submission."""
        if prefetch <= 0:
            yield from self._batches(want_tags)
            return
        import queue
        import threading
        q = queue.Queue(maxsize=prefetch)
        stop = threading.Event()

        def reader():
            try:
                for b in self._batches(want_tags):
                    while not stop.is_set():
                        try:
                            q.put(b, timeout=0.2)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        return
                q.put(None)
            except BaseException as e:  # hand decoding errors to the consumer
                q.put(e)

        t = threading.Thread(target=reader, daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()
            t.join(timeout=5)

    def _batches(self, want_tags=False):
        n = self.batch_size
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        while True:
            tid, pos, end = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.int32)
            flag, mapq, qkey = np.empty(n, np.uint16), np.empty(n, np.uint8), np.empty(n, np.uint64)
            cb = ub = tf = None
            if want_tags:
                cb, ub, tf = np.empty((n, TAG_BYTES), np.uint8), np.empty((n, TAG_BYTES), np.uint8), np.empty(n, np.uint8)
            got = C.c_uint64()
            rc = self._lib.tec_bam_read(self._h, n, p(tid), p(pos), p(end), p(flag), p(mapq), p(qkey),
                                        p(cb) if want_tags else None, p(ub) if want_tags else None,
                                        p(tf) if want_tags else None, C.byref(got))
            if rc != _lib.TEC_OK:
                raise IOError(self._lib.tec_bam_last_error(self._h).decode())
            k = got.value
            if k:
                b = RecordBatch(tid[:k], pos[:k], end[:k], flag[:k], mapq[:k])
                b.qkey = qkey[:k]
                if want_tags:
                    b.cb24, b.ub24, b.tagflags = cb[:k], ub[:k], tf[:k]
                yield b
            if k < n:
                return

    def close(self):
        if self._h:
            self._lib.tec_bam_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def open_alignment(path, batch_size=1 << 22, threads=0):
    return BamSource(path, batch_size=batch_size, threads=threads)


# --------------------------------------------------------------------------- writer (tests, synthetic data)

def _bgzf_block(data):
    comp = zlib.compressobj(1, zlib.DEFLATED, -15)
    c = comp.compress(data) + comp.flush()
    bsize = len(c) + 25
    return (b'\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00' + struct.pack('<H', bsize) + c +
            struct.pack('<II', zlib.crc32(data) & 0xffffffff, len(data)))


_BGZF_EOF = bytes.fromhex('1f8b08040000000000ff0600424302001b0003000000000000000000')


def _encode_record(tid, pos, end, flag, mapq, qname, tags):
    name = qname.encode() + b'\0'
    reflen = max(0, end - pos)
    if reflen > 0:
        # 100M (or what fits), an N gap for the rest: reference_end = pos + sum(M, N)
        m = min(reflen, 100)
        ops = [(m, 0)] + ([(reflen - m, 3)] if reflen > m else [])
    else:
        ops = []
    l_seq = sum(n for n, op in ops if op == 0)
    cigar = b''.join(struct.pack('<I', (n << 4) | op) for n, op in ops)
    aux = b''
    for k, v in (tags or {}).items():
        aux += k.encode() + b'Z' + v.encode() + b'\0'
    body = struct.pack('<iiBBHHHIiii', tid, pos, len(name), mapq, 0, len(ops), flag, l_seq, -1, -1, 0)
    body += name + cigar + bytes((l_seq + 1) // 2) + b'\xff' * l_seq + aux
    return struct.pack('<I', len(body)) + body


def write_bam(path, reference_names, records, reference_length=300000000):
    """records: dicts of the schema in oracle/ref_harness.py (flag, mapq, qname, rname, pos, end, tags).
    `end` is honoured through the CIGAR (an M run plus an N gap)."""
    tid_of = {n: i for i, n in enumerate(reference_names)}
    text = ''.join('@SQ\tSN:%s\tLN:%d\n' % (n, reference_length) for n in reference_names).encode()
    head = b'BAM\x01' + struct.pack('<I', len(text)) + text + struct.pack('<I', len(reference_names))
    for n in reference_names:
        nm = n.encode() + b'\0'
        head += struct.pack('<I', len(nm)) + nm + struct.pack('<I', reference_length)
    with open(path, 'wb') as fh:
        buf = bytearray(head)
        for r in records:
            buf += _encode_record(tid_of.get(r.get('rname'), -1), r['pos'], r['end'], r['flag'], r['mapq'],
                                  r.get('qname', 'q'), r.get('tags'))
            while len(buf) >= 0xff00:
                fh.write(_bgzf_block(bytes(buf[:0xff00])))
                del buf[:0xff00]
        if buf:
            fh.write(_bgzf_block(bytes(buf)))
        fh.write(_BGZF_EOF)


def write_sam(path, reference_names, records, reference_length=300000000):
    with open(path, 'w') as fh:
        for n in reference_names:
            fh.write('@SQ\tSN:%s\tLN:%d\n' % (n, reference_length))
        for r in records:
            reflen = max(0, r['end'] - r['pos'])
            m = min(reflen, 100)
            cigar = ('%dM' % m + ('%dN' % (reflen - m) if reflen > m else '')) if reflen else '*'
            tags = ''.join('\t%s:Z:%s' % kv for kv in (r.get('tags') or {}).items())
            fh.write('%s\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t%s\t%s%s\n' % (
                r.get('qname', 'q'), r['flag'], r.get('rname') or '*', r['pos'] + 1, r['mapq'], cigar,
                'A' * m if m else '*', 'I' * m if m else '*', tags))
