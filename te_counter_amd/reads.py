"""Alignment records as structure-of-arrays batches, and their reduction to counting units.

This is the boundary the reference crosses with `next(sam)` on a pysam.AlignmentFile
(te_count/te_count.py:65,78-79 / 190,203 / 351,393): every source below yields RecordBatch
objects carrying exactly the per-record fields the reference reads (te_count.py:81-98,
204-214, 394-438).
"""

import numpy as np

from .index import read_chrom_key

FLAG_FILTER = 0x4 | 0x200 | 0x400
NO_END = -2 ** 31  # reference_end of a record without CIGAR (pysam gives None); the same value as bam.NO_END


class RecordBatch:
    """tid/pos/end/flag/mapq arrays (+ optional qname list, cb/ub object arrays for --sc)."""
    __slots__ = ('tid', 'pos', 'end', 'flag', 'mapq', 'qname', 'cb', 'ub', 'qkey', 'cb24', 'ub24', 'tagflags')

    def __init__(self, tid, pos, end, flag, mapq, qname=None, cb=None, ub=None):
        self.tid = np.ascontiguousarray(tid, np.int32)
        self.pos = np.ascontiguousarray(pos, np.int32)
        self.end = np.ascontiguousarray(end, np.int32)
        self.flag = np.ascontiguousarray(flag, np.uint16)
        self.mapq = np.ascontiguousarray(mapq, np.uint8)
        self.qname = qname
        self.cb = cb
        self.ub = ub
        self.qkey = None      # native decoder: hash of the mate-name key (te_count.py:92)
        self.cb24 = self.ub24 = self.tagflags = None  # native decoder: tags already packed

    def __len__(self):
        return len(self.tid)


class ListSource:
    """In-memory source over record dicts (schema: oracle/ref_harness.py); used by tests and as
    the adapter for any pysam-like iterator."""

    def __init__(self, records, batch_size=1 << 20):
        self.records = records
        self.batch_size = batch_size
        names = []
        seen = {}
        for r in records:
            n = r.get('rname')
            if n is not None and n not in seen:
                seen[n] = len(names)
                names.append(n)
        self.reference_names = names
        self._tid = seen

    def batches(self, want_tags=False):
        recs = self.records
        for a in range(0, len(recs), self.batch_size):
            chunk = recs[a:a + self.batch_size]
            tid = [self._tid.get(r.get('rname'), -1) for r in chunk]
            cb = ub = None
            if want_tags:
                cb = [_first_tag(r, 'CB', 'CR') for r in chunk]
                ub = [_first_tag(r, 'UB', 'UR') for r in chunk]
            yield RecordBatch(tid, [r['pos'] for r in chunk], [r['end'] for r in chunk],
                              [r['flag'] for r in chunk], [r['mapq'] for r in chunk],
                              qname=[r.get('qname', 'q') for r in chunk], cb=cb, ub=ub)

    def close(self):
        pass


class SegmentSource:
    """Adapter over an open pysam-style alignment file: any iterable of segment objects with the
    attributes the reference reads from a `pysam.AlignedSegment` -- `flag`, `mapping_quality`,
    `query_name`, `reference_id`, `reference_start`, `reference_end` (None without a CIGAR),
    `get_tags()` -- plus the file's `references` tuple (te_count.py:65,78-98 / 190,203-214 /
    351,393-438).  It is the way in for inputs the native decoders do not read (CRAM through pysam,
    a filtered iterator, records made in memory); the records take the host path record by record,
    so this is the slow door, not the product path.  The caller keeps ownership of the file."""

    def __init__(self, segments, references=None, batch_size=1 << 20):
        self.segments = segments
        refs = references if references is not None else getattr(segments, 'references', None)
        if refs is None:
            raise TypeError('SegmentSource: the alignment file has no `references` and none were given')
        self.reference_names = list(refs)
        self.batch_size = batch_size

    def batches(self, want_tags=False):
        tid, pos, end, flag, mapq, qname, cb, ub = [], [], [], [], [], [], [], []

        def batch():
            return RecordBatch(tid, pos, end, flag, mapq, qname=qname, cb=cb if want_tags else None,
                               ub=ub if want_tags else None)

        for seg in self.segments:
            tid.append(seg.reference_id)
            pos.append(seg.reference_start)
            e = seg.reference_end
            end.append(NO_END if e is None else e)
            flag.append(seg.flag)
            mapq.append(seg.mapping_quality)
            qname.append(seg.query_name)
            if want_tags:
                tags = dict(seg.get_tags())  # te_count.py:403: a later tag of the same name wins
                cb.append(tags['CB'] if 'CB' in tags else tags.get('CR'))
                ub.append(tags['UB'] if 'UB' in tags else tags.get('UR'))
            if len(tid) == self.batch_size:
                yield batch()
                tid, pos, end, flag, mapq, qname, cb, ub = [], [], [], [], [], [], [], []
        if tid:
            yield batch()

    def close(self):
        pass


def _first_tag(r, a, b):
    t = r.get('tags', {})
    if a in t:
        return t[a]
    return t.get(b)


def _mate_key(q):
    return '_'.join(q.split('/')[0:-1])  # te_count.py:92


def se_units(batch, tid_map):
    """SE unit = one record: chrom, p1 = reference_start, p2 = reference_end (te_count.py:212-214)."""
    chrom = np.where(batch.tid >= 0, tid_map[np.clip(batch.tid, 0, max(0, len(tid_map) - 1))], -1).astype(np.int32) \
        if len(tid_map) else np.full(len(batch), -1, np.int32)
    return chrom, batch.pos, batch.end, batch.flag, batch.mapq


class PairAssembler:
    """PE unit = records (2k, 2k+1) of the file (te_count.py:78-79): chrom and mapq of read 1,
    p1/p2 = the two reference_starts (te_count.py:96-98), flags OR-ed for the 0x4/0x200/0x400
    filter (te_count.py:81-86).  Carries an odd record across batch boundaries; a trailing
    unpaired record is dropped, as `next(sam)` raising StopIteration does in the reference."""

    def __init__(self, tid_map, quality_threshold):
        self.tid_map = tid_map
        self.q = quality_threshold
        self._carry = None  # (tid, pos, flag, mapq, qname)

    def units(self, batch):
        tid, pos, flag, mapq = batch.tid, batch.pos, batch.flag, batch.mapq
        qn = batch.qname
        if qn is None and batch.qkey is not None:
            qn = batch.qkey  # hashes compare like the strings do
        if self._carry is not None:
            c = self._carry
            tid = np.concatenate([[c[0]], tid]).astype(np.int32)
            pos = np.concatenate([[c[1]], pos]).astype(np.int32)
            flag = np.concatenate([[c[2]], flag]).astype(np.uint16)
            mapq = np.concatenate([[c[3]], mapq]).astype(np.uint8)
            if qn is not None:
                qn = np.concatenate([[c[4]], qn]) if isinstance(qn, np.ndarray) else [c[4]] + list(qn)
            self._carry = None
        n = len(tid)
        if n % 2:
            self._carry = (tid[-1], pos[-1], flag[-1], mapq[-1], qn[-1] if qn is not None else None)
            n -= 1
        t1, p1, p2 = tid[0:n:2], pos[0:n:2], pos[1:n:2]
        # (the reverse bit is read 1's alone: nothing but the stranded-bulk extension looks at it)
        fl = (((flag[0:n:2] | flag[1:n:2]) & ~np.uint16(0x10)) | (flag[0:n:2] & np.uint16(0x10))).astype(np.uint16)
        mq = mapq[0:n:2]
        if qn is not None:
            self._check_names(qn, n, fl, mq)
        tm = self.tid_map
        chrom = np.where(t1 >= 0, tm[np.clip(t1, 0, max(0, len(tm) - 1))], -1).astype(np.int32) \
            if len(tm) else np.full(n // 2, -1, np.int32)
        return chrom, np.ascontiguousarray(p1), np.ascontiguousarray(p2), fl, np.ascontiguousarray(mq)

    def _check_names(self, qn, n, fl, mq):
        # The reference compares mate names only for pairs that passed the flag and mapq filters
        # (te_count.py:81-94) and dies on a mismatch (sys.quit does not exist -> AttributeError).
        live = ((fl & FLAG_FILTER) == 0) & (mq.astype(np.int64) >= self.q)
        if isinstance(qn, np.ndarray):
            if (live & (qn[0:n:2] != qn[1:n:2])).any():
                raise AttributeError("module 'sys' has no attribute 'quit'")
            return
        if not any('/' in q for q in qn[:n]):
            return  # names without '/' always compare '' == ''
        for k in np.nonzero(live)[0]:
            if _mate_key(qn[2 * k]) != _mate_key(qn[2 * k + 1]):
                raise AttributeError("module 'sys' has no attribute 'quit'")


class RecordKeeper:
    """PE over byte ranges on the host decoder: the records of this rank's range are kept (tid, pos,
    flag, mapq, mate key) until the number of records in the ranges before it is known, because that
    parity decides which records are mates (te_count.py:78-79; dist.plan_pairs)."""

    def __init__(self):
        self._parts = []

    def add(self, batch):
        if batch.qkey is None:
            raise ValueError('byte-range pairing needs the native decoder\'s mate-name keys')
        qk = batch.qkey
        self._parts.append((batch.tid.copy(), batch.pos.copy(), batch.flag.copy(), batch.mapq.copy(), np.asarray(qk, np.uint64).copy()))

    def _all(self):
        if not self._parts:
            return (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.uint16), np.zeros(0, np.uint8), np.zeros(0, np.uint64))
        if len(self._parts) > 1:
            self._parts = [tuple(np.concatenate([p[i] for p in self._parts]) for i in range(5))]
        return self._parts[0]

    def record(self, i):
        tid, pos, flag, mapq, qk = self._all()
        return (int(tid[i]), int(pos[i]), int(flag[i]), int(mapq[i]), int(qk[i]))

    def units(self, tid_map, q, skip_first, extra_mate):
        """Unit arrays of this range: records from index skip_first on, pairwise; extra_mate = the next
        range's first record (tid, pos, flag, mapq, qkey), the mate of an otherwise unpaired last record."""
        tid, pos, flag, mapq, qk = self._all()
        if extra_mate is not None:
            tid = np.concatenate([tid, [extra_mate[0]]]).astype(np.int32)
            pos = np.concatenate([pos, [extra_mate[1]]]).astype(np.int32)
            flag = np.concatenate([flag, [extra_mate[2]]]).astype(np.uint16)
            mapq = np.concatenate([mapq, [extra_mate[3]]]).astype(np.uint8)
            qk = np.concatenate([qk, np.array([extra_mate[4]], np.uint64)])
        b = RecordBatch(tid[skip_first:], pos[skip_first:], np.zeros(len(tid) - skip_first, np.int32), flag[skip_first:], mapq[skip_first:])
        b.qkey = qk[skip_first:]
        return PairAssembler(tid_map, q).units(b)


def chrom_class_table(reference_names):
    """--sc: per tid, (skip, class id).  skip = '_' or 'alt' in the read-side chromosome string
    (te_count.py:431-433); the class id is that string's identity (two header names that
    reduce to the same string are the same class)."""
    keys, lookup = [], {}
    skip = np.zeros(len(reference_names), bool)
    cls = np.zeros(len(reference_names), np.int32)
    for i, n in enumerate(reference_names):
        k = read_chrom_key(n)
        skip[i] = ('_' in k) or ('alt' in k)
        if k not in lookup:
            lookup[k] = len(keys)
            keys.append(k)
        cls[i] = lookup[k]
    return skip, cls, keys


TAG_BYTES = 24  # fixed width of the CB / UB fields handed to libtecount (include/tecount.h)


def pack_tags(values, what):
    """list of str|None -> (uint8[n,24] zero padded, present bool[n], too_long bool[n])."""
    n = len(values)
    out = np.zeros((n, TAG_BYTES), np.uint8)
    present = np.zeros(n, bool)
    too_long = np.zeros(n, bool)
    for i, v in enumerate(values):
        if v is None:
            continue
        present[i] = True
        b = v.encode('utf-8') if isinstance(v, str) else bytes(v)
        if len(b) > TAG_BYTES:
            too_long[i] = True
            continue
        out[i, :len(b)] = np.frombuffer(b, np.uint8)
    return out, present, too_long


def pack_whitelist(sorted_barcodes):
    """sorted(set(whitelist lines)) -> uint8[n,24]; the row number is the barcode id
    (te_count.py:338).  Python's str order equals bytewise order of the UTF-8 encoding."""
    n = len(sorted_barcodes)
    if n == 0:
        return np.zeros((0, TAG_BYTES), np.uint8)
    arr = np.array([v.encode('utf-8') for v in sorted_barcodes], dtype='S')
    return _pack_fixed(arr, sorted_barcodes)


def _pack_fixed(arr, names):
    """numpy 'S' array (one barcode per element) -> uint8[n,24], zero padded."""
    n, width = len(arr), arr.dtype.itemsize
    raw = arr.view(np.uint8).reshape(n, width)
    if width > TAG_BYTES and raw[:, TAG_BYTES:].any():
        bad = int(np.nonzero(raw[:, TAG_BYTES:].any(axis=1))[0][0])
        raise ValueError('whitelist barcode %r is longer than %d bytes' % (names[bad], TAG_BYTES))
    out = np.zeros((n, TAG_BYTES), np.uint8)
    out[:, :min(width, TAG_BYTES)] = raw[:, :TAG_BYTES]
    return out


class BarcodeNames:
    """whitelist id -> barcode string over a sorted numpy 'S' array; strings are made when they are asked for
    (a run touches a few thousand of the millions of whitelist entries)."""

    def __init__(self, arr):
        self._arr = arr

    def __len__(self):
        return len(self._arr)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [b.decode('utf-8') for b in self._arr[i].tolist()]
        return self._arr[i].decode('utf-8')

    def __iter__(self):
        return (b.decode('utf-8') for b in self._arr.tolist())

    def __eq__(self, other):
        return list(self) == list(other)


def load_whitelist(path):
    """The reference's whitelist: set(line.strip() for line in file), sorted (te_count.py:330-339) -> (barcode id ->
    string sequence, uint8[n,24]).  A 10x v3 whitelist has 6.8 M lines: a file of equal-length lines without stray
    whitespace (the normal case) is handled as one numpy array; anything else line by line on bytes.
    (bytes.strip removes ASCII whitespace; barcodes are ASCII.)"""
    with open(path, 'rb') as fh:
        data = fh.read()
    if b'\0' in data:
        raise ValueError('whitelist file with a NUL byte')
    arr = None
    first = data.find(b'\n')
    if first > 0 and len(data) % (first + 1) == 0:
        raw = np.frombuffer(data, np.uint8).reshape(-1, first + 1)
        body = raw[:, :first]
        if (raw[:, first] == 10).all() and not ((body == 32) | ((body >= 9) & (body <= 13))).any():
            arr = np.unique(np.ascontiguousarray(body).view('S%d' % first).reshape(-1))
    if arr is None:
        lines = data.split(b'\n')
        if lines and lines[-1] == b'':
            lines.pop()  # iterating a file yields nothing after the final newline
        uniq = list(set(map(bytes.strip, lines)))
        if not uniq:
            return BarcodeNames(np.zeros(0, 'S1')), np.zeros((0, TAG_BYTES), np.uint8)
        arr = np.array(uniq, dtype='S')
        arr.sort()
    names = BarcodeNames(arr)
    return names, _pack_fixed(arr, names)


def sc_batch_arrays(batch, tid_class):
    """RecordBatch -> the arrays tec_sc_submit takes."""
    n = len(batch)
    if batch.tagflags is not None:
        cb, ub, tagflags = batch.cb24, batch.ub24, batch.tagflags  # (bit 3, an over-long UMI, is judged on the GPU)
    else:
        if batch.cb is None or batch.ub is None:
            raise ValueError('this alignment source does not carry CB/UB tags')
        cb, cb_present, cb_long = pack_tags(batch.cb, 'CB')
        ub, ub_present, ub_long = pack_tags(batch.ub, 'UB')
        # an over-long UMI (bit 3) only matters on a record that survives every filter: the GPU decides (sc_filter_kernel)
        tagflags = (cb_present.astype(np.uint8) | (ub_present.astype(np.uint8) << 1) | (cb_long.astype(np.uint8) << 2) |
                    (ub_long.astype(np.uint8) << 3))
    cls = np.where(batch.tid >= 0, tid_class[np.clip(batch.tid, 0, max(0, len(tid_class) - 1))], -1).astype(np.int32) \
        if len(tid_class) else np.full(n, -1, np.int32)
    return cls, batch.pos, batch.end, batch.flag, batch.mapq, cb, ub, tagflags
