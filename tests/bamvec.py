"""Hand-assembled BAM test vectors: bytes built field by field from the record layout of the SAM/BAM
specification (SAMv1 section 4.2), NOT with the product's own writer, so that they pin the decoders
(host: csrc/bamio.hip, GPU: csrc/gbam.hip) independently.  Expected values are what pysam exposes for
such records: reference_start = pos, reference_end = pos + reference bases consumed by M/D/N/=/X (htslib
bam_endpos: at least one base when a CIGAR is present), None without CIGAR; get_tags() values."""

import struct
import zlib

OPS = 'MIDNSHP=X'
NO_END = -2 ** 31


def cigar_ops(text):
    out, num = [], ''
    for ch in text:
        if ch.isdigit():
            num += ch
        else:
            out.append((int(num), OPS.index(ch)))
            num = ''
    return out


def aux(tag, typ, value, sub=None):
    """One aux field.  typ: A c C s S i I f Z H B (B: sub = element type, value = list)."""
    t = tag.encode() + typ.encode()
    if typ == 'A':
        return t + value.encode()
    if typ in 'cCsSiIf':
        return t + struct.pack('<' + {'c': 'b', 'C': 'B', 's': 'h', 'S': 'H', 'i': 'i', 'I': 'I', 'f': 'f'}[typ], value)
    if typ in 'ZH':
        return t + value.encode() + b'\0'
    if typ == 'B':
        fmt = {'c': 'b', 'C': 'B', 's': 'h', 'S': 'H', 'i': 'i', 'I': 'I', 'f': 'f'}[sub]
        return t + sub.encode() + struct.pack('<I', len(value)) + b''.join(struct.pack('<' + fmt, v) for v in value)
    raise ValueError(typ)


def record(tid, pos, mapq, flag, qname, cigar='', l_seq=None, aux_bytes=b'', next_tid=-1, next_pos=-1, tlen=0):
    ops = cigar_ops(cigar) if isinstance(cigar, str) else list(cigar)
    if l_seq is None:
        l_seq = sum(n for n, op in ops if op in (0, 1, 4, 7, 8))  # M I S = X consume the query
    name = qname.encode() + b'\0'
    n_cig = len(ops)
    body = struct.pack('<iiBBHHHIiii', tid, pos, len(name), mapq, 4680, n_cig & 0xFFFF, flag, l_seq, next_tid, next_pos, tlen)
    body += name + b''.join(struct.pack('<I', (n << 4) | op) for n, op in ops)
    body += bytes([0x12] * ((l_seq + 1) // 2)) + bytes([30] * l_seq) + aux_bytes
    return struct.pack('<I', len(body)) + body


def header(refs):
    text = ''.join('@SQ\tSN:%s\tLN:%d\n' % r for r in refs).encode()
    h = b'BAM\x01' + struct.pack('<I', len(text)) + text + struct.pack('<I', len(refs))
    for name, ln in refs:
        nm = name.encode() + b'\0'
        h += struct.pack('<I', len(nm)) + nm + struct.pack('<I', ln)
    return h


def bgzf_block(data, level=6, wbits=15, memlevel=8, strategy=0, extra_before=b'', extra_after=b''):
    """One BGZF block.  level < 0: a stored deflate block, by hand.  extra_before / extra_after: other gzip extra
    subfields (SI1 SI2 SLEN data) around the BC subfield -- the BGZF section of the SAM specification allows them."""
    assert len(data) <= 65536, 'a BGZF block holds at most 64 KiB'
    if level < 0:
        c = b'\x01' + struct.pack('<HH', len(data), len(data) ^ 0xFFFF) + data
    else:
        co = zlib.compressobj(level, zlib.DEFLATED, -wbits, memlevel, strategy)
        c = co.compress(data) + co.flush()
    xlen = 6 + len(extra_before) + len(extra_after)
    bsize = 12 + xlen + len(c) + 8
    if bsize > 65536:
        raise ValueError('BGZF block larger than 64 KiB')
    return (b'\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff' + struct.pack('<H', xlen) + extra_before + b'BC\x02\x00' +
            struct.pack('<H', bsize - 1) + extra_after + c + struct.pack('<II', zlib.crc32(data) & 0xffffffff, len(data)))


def subfield(si, data):
    return si + struct.pack('<H', len(data)) + data


BGZF_EOF = bytes.fromhex('1f8b08040000000000ff0600424302001b0003000000000000000000')


def write_blocks(path, payload, cuts, level=6, empty_after=()):
    """payload split at the byte offsets in `cuts` into BGZF blocks; an empty block follows block k for k in empty_after."""
    edges = [0] + sorted(cuts) + [len(payload)]
    with open(path, 'wb') as fh:
        for k in range(len(edges) - 1):
            fh.write(bgzf_block(payload[edges[k]:edges[k + 1]], level))
            if k in empty_after:
                fh.write(bgzf_block(b'', level))
        fh.write(BGZF_EOF)


def write_blocks_varied(path, payload, cuts, seed, empty_after=()):
    """As write_blocks, but every block is compressed with its own zlib settings -- level 0..9, history window 2^9..2^15,
    memLevel 1..9, every strategy -- and a third of the blocks carry foreign extra subfields in front of and/or behind BC."""
    import random
    rnd = random.Random(seed)
    edges = [0] + sorted(cuts) + [len(payload)]
    with open(path, 'wb') as fh:
        for k in range(len(edges) - 1):
            kind = rnd.randrange(6)
            before = subfield(b'XA', bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 9)))) if kind in (1, 3) else b''
            after = subfield(b'RA', bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 20)))) if kind in (2, 3) else b''
            try:
                blk = bgzf_block(payload[edges[k]:edges[k + 1]], rnd.randrange(10), rnd.randrange(9, 16), rnd.randrange(1, 10),
                                 rnd.randrange(5), before, after)
            except ValueError:  # (a full 0xFF00-byte payload that these settings do not shrink below 64 KiB)
                blk = bgzf_block(payload[edges[k]:edges[k + 1]], 6, extra_before=before)
            fh.write(blk)
            if k in empty_after:
                fh.write(bgzf_block(b'', rnd.randrange(10), extra_before=before))
        fh.write(BGZF_EOF)


# ---- the example alignment of SAMv1 section 1.1 (positions are 1-based in SAM, 0-based in BAM)
SPEC_REFS = [('ref', 45)]
SPEC_RECORDS = [
    # qname, flag, pos0, mapq, cigar, aux, expected reference_end
    ('r001', 99, 6, 30, '8M2I4M1D3M', b'', 22),
    ('r002', 0, 8, 30, '3S6M1P1I4M', b'', 18),
    ('r003', 0, 8, 30, '5S6M', aux('SA', 'Z', 'ref,29,-,6H5M,17,0;'), 14),
    ('r004', 0, 15, 30, '6M14N5M', b'', 40),
    ('r003', 2064, 28, 17, '6H5M', aux('SA', 'Z', 'ref,9,+,5S6M,30,1;'), 33),
    ('r001', 147, 36, 30, '9M', aux('NM', 'i', 1), 45),
]


def spec_example():
    payload = header(SPEC_REFS)
    want = []
    for qn, flag, pos, mapq, cig, ax, end in SPEC_RECORDS:
        payload += record(0, pos, mapq, flag, qn, cig, aux_bytes=ax, next_tid=0 if flag & 1 else -1)
        want.append({'tid': 0, 'pos': pos, 'end': end, 'flag': flag, 'mapq': mapq})
    return payload, want


def every_cigar_op_and_aux_type():
    """Records that carry every CIGAR op and every aux type IN FRONT OF the CB / UB tags, the CR / UR
    fallbacks, a repeated tag, an over-long barcode, a CIGAR that consumes no reference, no CIGAR at
    all, an unmapped read and a CG:B,I long-CIGAR placeholder."""
    refs = [('chr1', 100000), ('chr2', 50000)]
    zoo = (aux('XA', 'A', 'q') + aux('Xc', 'c', -5) + aux('XC', 'C', 200) + aux('Xs', 's', -3000) + aux('XS', 'S', 60000) +
           aux('NH', 'i', -7) + aux('XI', 'I', 4000000000) + aux('Xf', 'f', 1.5) + aux('RG', 'Z', 'grp:1') + aux('XH', 'H', '1AE301') +
           aux('Bc', 'B', [-1, 2], 'c') + aux('BC', 'B', [1, 2, 3], 'C') + aux('Bs', 'B', [-300], 's') + aux('BS', 'B', [300, 4], 'S') +
           aux('Bi', 'B', [-70000], 'i') + aux('BI', 'B', [70000, 1], 'I') + aux('Bf', 'B', [0.5, 2.0], 'f') + aux('Be', 'B', [], 'C'))
    rows = [
        # (record bytes, expected dict)
        (record(0, 100, 60, 0, 'all_ops/1', '5S3M2I4M1D2=1X10N3M2P4H', aux_bytes=zoo + aux('CB', 'Z', 'ACGTACGTACGTACGT-1') + aux('UB', 'Z', 'AACCGGTTAACC')),
         {'tid': 0, 'pos': 100, 'end': 100 + 3 + 4 + 1 + 2 + 1 + 10 + 3, 'flag': 0, 'mapq': 60, 'cb': 'ACGTACGTACGTACGT-1', 'ub': 'AACCGGTTAACC', 'tf': 3}),
        (record(1, 7, 3, 16, 'fallbacks/2', '10M', aux_bytes=aux('CR', 'Z', 'TTTT') + zoo + aux('UR', 'Z', 'GG')),
         {'tid': 1, 'pos': 7, 'end': 17, 'flag': 16, 'mapq': 3, 'cb': 'TTTT', 'ub': 'GG', 'tf': 3}),
        (record(0, 50, 255, 0x400, 'cb_beats_cr', '4M', aux_bytes=aux('CR', 'Z', 'RAW') + aux('CB', 'Z', 'FIXED') + aux('UR', 'Z', 'RAWU') + aux('UB', 'Z', 'FIXU')),
         {'tid': 0, 'pos': 50, 'end': 54, 'flag': 0x400, 'mapq': 255, 'cb': 'FIXED', 'ub': 'FIXU', 'tf': 3}),
        (record(0, 60, 20, 0, 'repeat_last_wins', '4M', aux_bytes=aux('CB', 'Z', 'FIRST') + aux('CB', 'Z', 'SECOND') + aux('UB', 'Z', 'U')),
         {'tid': 0, 'pos': 60, 'end': 64, 'flag': 0, 'mapq': 20, 'cb': 'SECOND', 'ub': 'U', 'tf': 3}),
        (record(0, 70, 20, 0, 'long_barcode', '4M', aux_bytes=aux('CB', 'Z', 'A' * 25) + aux('UB', 'Z', 'U')),
         {'tid': 0, 'pos': 70, 'end': 74, 'flag': 0, 'mapq': 20, 'cb': '', 'ub': 'U', 'tf': 1 | 2 | 4}),
        (record(0, 80, 20, 0, 'no_tags', '4M', aux_bytes=zoo),
         {'tid': 0, 'pos': 80, 'end': 84, 'flag': 0, 'mapq': 20, 'cb': '', 'ub': '', 'tf': 0}),
        (record(0, 90, 20, 0, 'cb_not_Z', '4M', aux_bytes=aux('CB', 'i', 5) + aux('UB', 'Z', 'U')),
         {'tid': 0, 'pos': 90, 'end': 94, 'flag': 0, 'mapq': 20, 'cb': '', 'ub': 'U', 'tf': 2}),
        (record(0, 200, 20, 0, 'no_ref_consumed', '5S3I2H', aux_bytes=b''),
         {'tid': 0, 'pos': 200, 'end': 201, 'flag': 0, 'mapq': 20, 'cb': '', 'ub': '', 'tf': 0}),   # htslib: rlen 0 -> 1
        (record(0, 300, 20, 0, 'no_cigar', '', l_seq=8, aux_bytes=b''),
         {'tid': 0, 'pos': 300, 'end': NO_END, 'flag': 0, 'mapq': 20, 'cb': '', 'ub': '', 'tf': 0}),
        (record(-1, -1, 0, 4, 'unmapped', '', l_seq=6, aux_bytes=aux('CB', 'Z', 'X') + aux('UB', 'Z', 'Y')),
         {'tid': -1, 'pos': -1, 'end': NO_END, 'flag': 4, 'mapq': 0, 'cb': 'X', 'ub': 'Y', 'tf': 3}),
        # the real CIGAR (70000 ops would not fit n_cigar_op) lives in CG:B,I behind the placeholder <l_seq>S<ref_len>N
        (record(1, 1000, 40, 0, 'long_cigar', [(12, 4), (999, 3)], l_seq=12,
                aux_bytes=aux('NM', 'i', 2) + aux('CG', 'B', [(5 << 4) | 0, (20 << 4) | 3, (4 << 4) | 1, (3 << 4) | 0, (7 << 4) | 2], 'I') + aux('CB', 'Z', 'LC') + aux('UB', 'Z', 'LU')),
         {'tid': 1, 'pos': 1000, 'end': 1000 + 5 + 20 + 3 + 7, 'flag': 0, 'mapq': 40, 'cb': 'LC', 'ub': 'LU', 'tf': 3}),
        (record(0, 99990, 20, 0, 'last', '10M', aux_bytes=aux('CB', 'Z', 'Z') + aux('UB', 'Z', 'Z')),
         {'tid': 0, 'pos': 99990, 'end': 100000, 'flag': 0, 'mapq': 20, 'cb': 'Z', 'ub': 'Z', 'tf': 3}),
        # htslib's bam_tag2cigar asks only for a first op <l_seq>S: a one-op placeholder is one too ...
        (record(1, 2000, 40, 0, 'cg_one_op_placeholder', [(12, 4)], l_seq=12, aux_bytes=aux('CG', 'B', [(12 << 4) | 0, (30 << 4) | 3], 'I')),
         {'tid': 1, 'pos': 2000, 'end': 2042, 'flag': 0, 'mapq': 40, 'cb': '', 'ub': '', 'tf': 0}),
        # ... a CG array shorter than the placeholder is ignored (12S 999N stays: 999 reference bases) ...
        (record(1, 3000, 40, 0, 'cg_shorter_than_fake', [(12, 4), (999, 3)], l_seq=12, aux_bytes=aux('CG', 'B', [(5 << 4) | 0], 'I')),
         {'tid': 1, 'pos': 3000, 'end': 3999, 'flag': 0, 'mapq': 40, 'cb': '', 'ub': '', 'tf': 0}),
        # ... an unplaced read keeps its CIGAR as it stands, and a UMI of more than 24 bytes is flagged (bit 3), not fatal here
        (record(-1, -1, 0, 4, 'cg_unplaced_long_umi', [(12, 4), (7, 3)], l_seq=12,
                aux_bytes=aux('CG', 'B', [(5 << 4) | 0, (3 << 4) | 0], 'I') + aux('CB', 'Z', 'Q') + aux('UB', 'Z', 'U' * 25)),
         {'tid': -1, 'pos': -1, 'end': 6, 'flag': 4, 'mapq': 0, 'cb': 'Q', 'ub': '', 'tf': 1 | 2 | 8}),
    ]
    payload = header(refs)
    hdr_len = len(payload)
    starts = []
    for rb, _ in rows:
        starts.append(len(payload))
        payload += rb
    return payload, [w for _, w in rows], hdr_len, starts


def files_never_seen():
    """Shapes a decoder meets in files from other writers: a 254-character read name, records without sequence
    (l_seq = 0, with and without CIGAR), the smallest record there is, a mapped record without CIGAR, aux fields
    of every type BEHIND CB / UB, CB / UB as the very last bytes of a record, a read long enough to fill a maximal
    0xFF00-byte BGZF block on its own, and names that end in '/' or are nothing but '/'."""
    refs = [('chr1', 100000), ('chrUn_KI270742v1', 186739), ('chr2', 50000)]
    zoo = (aux('XA', 'A', 'q') + aux('Xc', 'c', -5) + aux('XC', 'C', 200) + aux('Xs', 's', -3000) + aux('XS', 'S', 60000) +
           aux('NH', 'i', -7) + aux('XI', 'I', 4000000000) + aux('Xf', 'f', 1.5) + aux('RG', 'Z', 'grp:1') + aux('XH', 'H', '1AE301') +
           aux('Bc', 'B', [-1, 2], 'c') + aux('BS', 'B', [300, 4], 'S') + aux('Bf', 'B', [0.5, 2.0], 'f') + aux('Be', 'B', [], 'C'))
    long_name = 'n' * 250 + '/1xy'  # 254 characters: l_read_name = 255
    rows = [
        (record(0, 10, 30, 0, long_name, '10M', aux_bytes=aux('CB', 'Z', 'NAMECB') + aux('UB', 'Z', 'NAMEUB')),
         {'tid': 0, 'pos': 10, 'end': 20, 'flag': 0, 'mapq': 30, 'cb': 'NAMECB', 'ub': 'NAMEUB', 'tf': 3}),
        (record(0, 30, 30, 0, 'noseq_cigar', '7M2D3M', l_seq=0, aux_bytes=aux('CB', 'Z', 'A') + aux('UB', 'Z', 'C')),
         {'tid': 0, 'pos': 30, 'end': 42, 'flag': 0, 'mapq': 30, 'cb': 'A', 'ub': 'C', 'tf': 3}),
        (record(0, 40, 30, 0, 'noseq_nocigar', '', l_seq=0),
         {'tid': 0, 'pos': 40, 'end': NO_END, 'flag': 0, 'mapq': 30, 'cb': '', 'ub': '', 'tf': 0}),
        (record(-1, -1, 0, 4, 'x', '', l_seq=0),  # 34 bytes behind the size field: nothing is smaller
         {'tid': -1, 'pos': -1, 'end': NO_END, 'flag': 4, 'mapq': 0, 'cb': '', 'ub': '', 'tf': 0}),
        (record(2, 49990, 0, 16, 'mapped_nocigar', '', l_seq=20, aux_bytes=aux('UB', 'Z', 'ONLYUB')),
         {'tid': 2, 'pos': 49990, 'end': NO_END, 'flag': 16, 'mapq': 0, 'cb': '', 'ub': 'ONLYUB', 'tf': 2}),
        (record(1, 500, 60, 0, 'tags_behind', '50M', aux_bytes=aux('CB', 'Z', 'ACGTACGTACGTACGT-1') + aux('UB', 'Z', 'AACCGGTTAACC') + zoo),
         {'tid': 1, 'pos': 500, 'end': 550, 'flag': 0, 'mapq': 60, 'cb': 'ACGTACGTACGTACGT-1', 'ub': 'AACCGGTTAACC', 'tf': 3}),
        (record(1, 600, 60, 0, 'tags_between', '50M', aux_bytes=aux('UB', 'Z', 'UFIRST') + zoo + aux('CB', 'Z', 'CLAST')),
         {'tid': 1, 'pos': 600, 'end': 650, 'flag': 0, 'mapq': 60, 'cb': 'CLAST', 'ub': 'UFIRST', 'tf': 3}),
        (record(0, 1000, 44, 0, 'fills_a_block/2', '43000M', aux_bytes=aux('CB', 'Z', 'BIG') + aux('UB', 'Z', 'READ')),
         {'tid': 0, 'pos': 1000, 'end': 44000, 'flag': 0, 'mapq': 44, 'cb': 'BIG', 'ub': 'READ', 'tf': 3}),
        (record(0, 2000, 20, 0, 'ends_in_slash/', '5M'),
         {'tid': 0, 'pos': 2000, 'end': 2005, 'flag': 0, 'mapq': 20, 'cb': '', 'ub': '', 'tf': 0}),
        (record(0, 2100, 20, 0, '/', '5M', aux_bytes=aux('CB', 'Z', '') + aux('UB', 'Z', '')),  # empty tag strings are present tags
         {'tid': 0, 'pos': 2100, 'end': 2105, 'flag': 0, 'mapq': 20, 'cb': '', 'ub': '', 'tf': 3}),
        (record(2, 0, 255, 0x900, 'first_base', '1M', aux_bytes=aux('CR', 'Z', 'R') + aux('UR', 'Z', 'S')),
         {'tid': 2, 'pos': 0, 'end': 1, 'flag': 0x900, 'mapq': 255, 'cb': 'R', 'ub': 'S', 'tf': 3}),
    ]
    payload = header(refs)
    hdr_len = len(payload)
    starts = []
    for rb, _ in rows:
        starts.append(len(payload))
        payload += rb
    return payload, [w for _, w in rows], hdr_len, starts


def check_rows(arrays, want):
    """arrays: dict tid/pos/end/flag/mapq[/cb24/ub24/tagflags] (numpy) as the decoders deliver them."""
    assert len(arrays['tid']) == len(want)
    for i, w in enumerate(want):
        got = {k: int(arrays[k][i]) for k in ('tid', 'pos', 'end', 'flag', 'mapq')}
        assert got == {k: w[k] for k in got}, (i, got, w)
        if 'cb' in w and arrays.get('tagflags') is not None:
            assert int(arrays['tagflags'][i]) == w['tf'], (i, int(arrays['tagflags'][i]), w)
            if w['tf'] & 1 and not w['tf'] & 4:
                assert bytes(arrays['cb24'][i]).rstrip(b'\0').decode() == w['cb'], i
            if w['tf'] & 2 and not w['tf'] & 8:
                assert bytes(arrays['ub24'][i]).rstrip(b'\0').decode() == w['ub'], i
