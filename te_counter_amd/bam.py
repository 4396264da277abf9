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


def byte_range(file_size, rank, world):
    """The slice of the compressed file rank `rank` of `world` decodes: equal byte shares; a reader owns
    the records that start in BGZF blocks starting inside its share (include/tecount.h)."""
    return (file_size * rank) // world, (file_size * (rank + 1)) // world


class BamSource:
    """Host-side decoder (csrc/bamio.hip).  byte_range = (begin, end): deliver only the records owned by
    that slice of the compressed file (multi-GPU sharded decoding); None: the whole file."""

    def __init__(self, path, batch_size=1 << 22, threads=0, byte_range=None):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        if byte_range is None:
            rc = self._lib.tec_bam_open(str(path).encode(), int(threads), C.byref(self._h))
        else:
            rc = self._lib.tec_bam_open_range(str(path).encode(), int(threads), int(byte_range[0]), int(byte_range[1]),
                                              C.byref(self._h))
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

        def put(item):
            """False once the consumer has gone (it then never empties the queue: a plain put would block for ever
            and the consumer's join with it)."""
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.2)
                    return True
                except queue.Full:
                    continue
            return False

        def reader():
            try:
                for b in self._batches(want_tags):
                    if not put(b):
                        return
                put(None)
            except BaseException as e:  # hand decoding errors to the consumer
                put(e)

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
            t.join()  # tec_bam_read is bounded; closing the handle under a running read would free its buffers

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

    def range_tokens(self):
        """(first, end, has_first): where this reader's first record and the first record it does NOT own
        start, each as (BGZF block file offset, offset inside the inflated block).  Valid once the last
        record has been delivered; the multi-rank job checks end[k] == first[k+1] (dist.check_ranges)."""
        first, end = (C.c_uint64 * 2)(), (C.c_uint64 * 2)()
        hf, fin = C.c_int(), C.c_int()
        self._lib.tec_bam_range_tokens(self._h, first, end, C.byref(hf), C.byref(fin))
        if not fin.value:
            raise RuntimeError('range tokens are only known after the last record has been read')
        return (first[0], first[1]), (end[0], end[1]), bool(hf.value)

    def close(self):
        if self._h:
            self._lib.tec_bam_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def open_alignment(path, batch_size=1 << 22, threads=0, byte_range=None):
    return BamSource(path, batch_size=batch_size, threads=threads, byte_range=byte_range)


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


def write_bam(path, reference_names, records, reference_length=300000000, block_bytes=0xff00):
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
            while len(buf) >= block_bytes:
                fh.write(_bgzf_block(bytes(buf[:block_bytes])))
                del buf[:block_bytes]
        if buf:
            fh.write(_bgzf_block(bytes(buf)))
        fh.write(_BGZF_EOF)


def units_to_records(units, spare_tid, paired):
    """Counting units (synth.make_units) -> per-record arrays (tid, pos, flag, mapq[, end]) of an
    alignment file that decodes back to exactly those units: PE pairs become two consecutive records
    (the unit's flags on the first mate), chrom -1 becomes reference `spare_tid` (a contig that is
    not in the index)."""
    chrom, p1, p2, flag, mapq = units
    if paired:
        tid = np.repeat(chrom, 2)
        pos = np.stack([p1, p2], 1).reshape(-1)
        fl = np.stack([flag, np.zeros_like(flag)], 1).reshape(-1)
        mq = np.repeat(mapq, 2)
    else:
        tid, pos, fl, mq = chrom, p1, flag, mapq
    tid = np.where(tid < 0, spare_tid, tid).astype(np.int32)
    return tid, pos.astype(np.int32), fl.astype(np.uint16), mq.astype(np.uint8)


def write_bam_arrays(path, reference_names, tid, pos, flag, mapq, end=None, pair_names=False, cb=None, ub=None,
                     payload='random', seed=1, threads=0, level=1, reference_length=300000000, block_bytes=0xff00):
    """Vectorised BAM writer for large synthetic inputs (numpy + zlib, BGZF blocks of 0xff00 bytes
    compressed on a thread pool).  One fixed-width record per read: a 28-character Illumina-style
    name (mates share it up to '/1' '/2' when pair_names), CIGAR 50M <gap>N 50M (gap = end - pos - 100,
    0 when `end` is None), 100 bases + qualities, NH/HI/AS/nM tags and, if given, CB / UB 'Z' tags
    (uint8[n, 24] zero padded, all of one length).
    payload: 'random'   sequence and quality bytes uniformly random -- the hard case for an inflater;
             'illumina' ACGT sequence, NovaSeq-binned qualities {2,12,23,37} -- what real files hold."""
    import concurrent.futures as cf
    import os
    n = len(tid)
    rng = np.random.default_rng(seed)
    name_len = 29  # 28 characters + NUL
    fields = [('bs', '<u4'), ('tid', '<i4'), ('pos', '<i4'), ('l_name', 'u1'), ('mapq', 'u1'), ('bin', '<u2'),
              ('n_cig', '<u2'), ('flag', '<u2'), ('l_seq', '<u4'), ('ntid', '<i4'), ('npos', '<i4'), ('tlen', '<i4'),
              ('name', 'S%d' % name_len), ('cigar', '<u4', 3), ('seq', 'u1', 50), ('qual', 'u1', 100), ('aux', 'S16')]
    for tag, arr in (('cb', cb), ('ub', ub)):
        if arr is not None:
            ln = int((arr[0] != 0).sum()) if n else 1
            fields += [(tag + '_hdr', 'S3'), (tag, 'u1', ln), (tag + '_nul', 'u1')]
    rec = np.zeros(n, dtype=fields)
    rec['bs'] = rec.dtype.itemsize - 4
    rec['tid'], rec['pos'], rec['flag'], rec['mapq'] = tid, pos, flag, mapq
    rec['l_name'] = name_len
    rec['n_cig'] = 3
    rec['l_seq'] = 100
    rec['ntid'] = rec['npos'] = -1
    gap = np.zeros(n, np.int64) if end is None else np.asarray(end, np.int64) - np.asarray(pos, np.int64) - 100
    if (gap < 0).any():
        raise ValueError('end must be >= pos + 100')
    rec['cigar'][:, 0] = (50 << 4) | 0
    rec['cigar'][:, 1] = (gap << 4) | 3
    rec['cigar'][:, 2] = (50 << 4) | 0
    if payload not in ('random', 'illumina'):
        raise ValueError(payload)
    workers = threads or min(16, os.cpu_count() or 1)
    # names, sequence and quality bytes are made in slabs on the thread pool (numpy releases the GIL inside its
    # generators and ufuncs): 'A00123:45:HXX:1:<10 digits>/<1|2>' + NUL
    slab = 1 << 20
    seeds = np.random.SeedSequence(seed).spawn((n + slab - 1) // slab or 1)
    prefix = np.frombuffer(b'A00123:45:HXX:1:', np.uint8)
    seq_lut = np.array([(a << 4) | b for a in (1, 2, 4, 8) for b in (1, 2, 4, 8)], np.uint8)
    qual_lut = np.repeat(np.array([2, 12, 23, 37], np.uint8), [3, 10, 26, 217])  # ~1 %, 4 %, 10 %, 85 % of 256

    def fill(k):
        lo, hi = k * slab, min(n, (k + 1) * slab)
        r = np.random.default_rng(seeds[k])
        view = rec[lo:hi]
        m = hi - lo
        nm = np.zeros((m, name_len), np.uint8)
        nm[:, :16] = prefix
        ordinal = (np.arange(lo, hi) // 2) if pair_names else np.arange(lo, hi)
        for d in range(10):
            nm[:, 16 + d] = (ordinal // 10 ** (9 - d)) % 10 + 48
        nm[:, 26] = ord('/')
        nm[:, 27] = (np.arange(lo, hi) % 2 + 49) if pair_names else 49
        view['name'] = nm.view('S%d' % name_len).reshape(m)
        if payload == 'random':
            view['seq'] = r.integers(0, 256, (m, 50), dtype=np.uint8)
            view['qual'] = r.integers(2, 41, (m, 100), dtype=np.uint8)
        else:
            view['seq'] = seq_lut[r.integers(0, 16, (m, 50), dtype=np.uint8)]
            view['qual'] = qual_lut[r.integers(0, 256, (m, 100), dtype=np.uint8)]

    with cf.ThreadPoolExecutor(workers) as ex:
        list(ex.map(fill, range(len(seeds))))
    rec['aux'] = b'NHC\x01HIC\x01ASC\x62nMC\x00'
    for tag, arr, hdr in (('cb', cb, b'CBZ'), ('ub', ub, b'UBZ')):
        if arr is not None:
            ln = rec[tag].shape[1]
            if ((arr[:, :ln] == 0).any()) or (arr.shape[1] > ln and (arr[:, ln:] != 0).any()):
                raise ValueError('%s tags must all have the same length' % tag.upper())
            rec[tag + '_hdr'] = hdr
            rec[tag] = arr[:, :ln]
    text = ''.join('@SQ\tSN:%s\tLN:%d\n' % (r, reference_length) for r in reference_names).encode()
    head = b'BAM\x01' + struct.pack('<I', len(text)) + text + struct.pack('<I', len(reference_names))
    for r in reference_names:
        nm = r.encode() + b'\0'
        head += struct.pack('<I', len(nm)) + nm + struct.pack('<I', reference_length)
    payload_bytes = head + rec.tobytes()
    del rec

    def pack(off):
        data = payload_bytes[off:off + block_bytes]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        c = co.compress(data) + co.flush()
        return (b'\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00' + struct.pack('<H', len(c) + 25) + c +
                struct.pack('<II', zlib.crc32(data) & 0xffffffff, len(data)))

    offs = range(0, len(payload_bytes), block_bytes)
    with open(path, 'wb') as fh, cf.ThreadPoolExecutor(workers) as ex:
        for blk in ex.map(pack, offs, chunksize=64):
            fh.write(blk)
        fh.write(_BGZF_EOF)
    return len(payload_bytes)


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
