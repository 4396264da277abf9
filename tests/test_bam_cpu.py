"""The native BGZF/BAM/SAM decoder (csrc/bamio.hip, host code) against files written by
te_counter_amd.bam.write_bam / write_sam -- runs without a GPU."""

import os

import numpy as np
import pytest

import datagen as dg
from te_counter_amd import bam
from te_counter_amd.reads import ListSource, PairAssembler, SegmentSource, sc_batch_arrays, chrom_class_table, se_units


def _records(n, seed=3, sc=False):
    if sc:
        wl = dg.make_whitelist(seed, 50)
        recs = dg.make_sc_records(seed, n, wl, None, n_cells=8)
    else:
        recs = dg.make_records(seed, n, paired=True)
    for i, r in enumerate(recs):
        if i % 97 == 0 and not sc:
            r['end'] = r['pos']  # no CIGAR -> reference_end is None in pysam
    return recs


def _refs(recs):
    return ListSource(recs).reference_names


def _collect(src, want_tags=False):
    out = []
    for b in src.batches(want_tags=want_tags):
        out.append(b)
    return out


@pytest.mark.parametrize('writer,ext', [(bam.write_bam, 'bam'), (bam.write_sam, 'sam')])
@pytest.mark.parametrize('batch', [1, 7, 1000, 1 << 20])
def test_roundtrip_fields(tmp_path, writer, ext, batch):
    recs = _records(5000)
    refs = _refs(recs)
    path = str(tmp_path / ('x.' + ext))
    writer(path, refs, recs)
    src = bam.open_alignment(path, batch_size=batch, threads=3)
    assert src.reference_names == refs
    got = _collect(src)
    src.close()
    tid = np.concatenate([b.tid for b in got])
    assert len(tid) == len(recs)
    assert [refs[t] if t >= 0 else None for t in tid] == [r['rname'] for r in recs]
    assert np.array_equal(np.concatenate([b.pos for b in got]), [r['pos'] for r in recs])
    want_end = [r['end'] if r['end'] > r['pos'] else bam.NO_END for r in recs]
    assert np.array_equal(np.concatenate([b.end for b in got]), want_end)
    assert np.array_equal(np.concatenate([b.flag for b in got]), [r['flag'] for r in recs])
    assert np.array_equal(np.concatenate([b.mapq for b in got]), [r['mapq'] for r in recs])
    qk = np.concatenate([b.qkey for b in got])
    assert len(set(qk.tolist())) == 1  # names without '/' all reduce to the empty key


def test_units_match_list_source(tmp_path):
    """SE and PE units from the decoded BAM == units from the in-memory records."""
    recs = [r for r in _records(4001) if r['end'] > r['pos']]
    refs = _refs(recs)
    path = str(tmp_path / 'x.bam')
    bam.write_bam(path, refs, recs)
    tid_map = np.arange(len(refs), dtype=np.int32)
    for paired in (False, True):
        outs = []
        for src in (bam.open_alignment(path, batch_size=333), ListSource(recs, batch_size=1000)):
            pa = PairAssembler(tid_map, 20)
            parts = [pa.units(b) if paired else se_units(b, tid_map) for b in src.batches()]
            outs.append([np.concatenate([p[i] for p in parts]) for i in range(5)])
            src.close()
        assert all(np.array_equal(a, b) for a, b in zip(*outs))


def test_mate_name_keys(tmp_path):
    recs = [{'flag': 0, 'mapq': 255, 'qname': q, 'rname': 'chr1', 'pos': 5, 'end': 50}
            for q in ('a/1', 'a/2', 'x/y/1', 'x_y/2', 'b/1', 'c/1', 'plain', 'other')]
    path = str(tmp_path / 'n.bam')
    bam.write_bam(path, ['chr1'], recs)
    qk = _collect(bam.open_alignment(path))[0].qkey
    assert qk[0] == qk[1] and qk[2] == qk[3] and qk[4] != qk[5] and qk[6] == qk[7]
    with pytest.raises(AttributeError):
        PairAssembler(np.zeros(1, np.int32), 20).units(_collect(bam.open_alignment(path))[0])


@pytest.mark.parametrize('writer,ext', [(bam.write_bam, 'bam'), (bam.write_sam, 'sam')])
def test_sc_tags(tmp_path, writer, ext):
    recs = _records(3000, sc=True)
    recs[5]['tags'] = {}
    recs[6]['tags'] = {'CB': 'A' * 30, 'UB': 'ACGT'}
    recs[7]['tags'] = {'CR': 'ACGTACGT', 'UR': 'TTTT', 'NH': 'x'}
    refs = _refs(recs)
    path = str(tmp_path / ('s.' + ext))
    writer(path, refs, recs)
    _, cls, _ = chrom_class_table(refs)
    a = [np.concatenate(x) for x in zip(*[sc_batch_arrays(b, cls) for b in bam.open_alignment(path, 701).batches(True)])]
    b = [np.concatenate(x) for x in zip(*[sc_batch_arrays(b, cls) for b in ListSource(recs, 500).batches(True)])]
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert a[7][5] == 0 and a[7][6] == (1 | 2 | 4) and a[7][7] == 3


def test_not_a_bam(tmp_path):
    p = tmp_path / 'junk.bam'
    p.write_bytes(b'\x1f\x8b' + b'junkjunkjunkjunkjunkjunk')
    with pytest.raises(IOError):
        bam.open_alignment(str(p))


def test_chunked_reader_threads_and_batches(tmp_path):
    """A file of several 512-block chunks: every thread count and batch size must give the same rows
    (records straddle BGZF blocks, thread shares and chunks; batches end in the middle of chunks)."""
    n = 330_001
    rng = np.random.default_rng(8)
    tid = rng.integers(0, 3, n).astype(np.int32)
    pos = rng.integers(0, 1 << 28, n).astype(np.int32)
    flag = rng.integers(0, 4096, n).astype(np.uint16)
    mapq = rng.integers(0, 256, n).astype(np.uint8)
    path = str(tmp_path / 'big.bam')
    bam.write_bam_arrays(path, ['chr1', 'chr2', 'chrUn'], tid, pos, flag, mapq, pair_names=True)
    assert os.path.getsize(path) > 40_000_000  # > 512 blocks of 64 KiB
    ref = None
    for threads, batch in [(1, 1 << 22), (3, 77_777), (8, 1000), (8, 1 << 22), (5, 130_000)]:
        src = bam.BamSource(path, batch_size=batch, threads=threads)
        cols = [[], [], [], [], [], []]
        for b in src.batches(prefetch=0):
            for c, v in zip(cols, (b.tid, b.pos, b.end, b.flag, b.mapq, b.qkey)):
                c.append(v.copy())
        src.close()
        got = [np.concatenate(c) for c in cols]
        assert len(got[0]) == n
        assert (got[0] == tid).all() and (got[1] == pos).all() and (got[2] == pos + 100).all()
        assert (got[3] == flag).all() and (got[4] == mapq).all()
        if ref is None:
            ref = got[5]
        assert (got[5] == ref).all()
    # a file cut in the middle of a block, and one cut in the middle of a record, are refused
    data = open(path, 'rb').read()
    cut = str(tmp_path / 'cut.bam')
    open(cut, 'wb').write(data[:len(data) // 2])
    with pytest.raises(IOError):
        for _ in bam.BamSource(cut, threads=4).batches(prefetch=0):
            pass


# ---------------------------------------------------------------------------------------------
# vectors that were NOT produced by this repo's writer: bytes assembled from the BAM specification

def _decode_all(path, want_tags=True, **kw):
    src = bam.BamSource(str(path), **kw)
    parts = list(src._batches(want_tags=want_tags))
    names = src.reference_names
    src.close()
    if not parts:
        z = np.zeros(0, np.int32)
        return {'tid': z, 'pos': z, 'end': z, 'flag': z, 'mapq': z, 'tagflags': None}, names
    out = {k: np.concatenate([getattr(b, k) for b in parts]) for k in ('tid', 'pos', 'end', 'flag', 'mapq', 'qkey')}
    if want_tags:
        out.update({k: np.concatenate([getattr(b, k) for b in parts]) for k in ('cb24', 'ub24', 'tagflags')})
    return out, names


def test_sam_spec_example_alignment(tmp_path):
    """The example of SAMv1 section 1.1: POS / CIGAR -> the reference_end the spec's picture shows."""
    import bamvec
    payload, want = bamvec.spec_example()
    p = tmp_path / 'spec.bam'
    bamvec.write_blocks(str(p), payload, [])
    got, names = _decode_all(p)
    assert names == ['ref']
    bamvec.check_rows(got, want)
    assert len(set(got['qkey'].tolist())) == 1  # names without '/': every mate key is the empty string (te_count.py:92)


@pytest.mark.parametrize('level', [-1, 0, 1, 6, 9])
def test_every_cigar_op_every_aux_type_and_block_layouts(tmp_path, level):
    """Every CIGAR op, every aux type in front of CB/UB, CR/UR fallbacks, CG:B,I, no-CIGAR and
    reference-less CIGARs -- as one block, cut inside records (a record spanning two and three BGZF
    blocks, a cut inside the 4-byte size field), with empty blocks in between, stored / fixed / dynamic
    deflate blocks."""
    import bamvec
    payload, want, hdr_len, starts = bamvec.every_cigar_op_and_aux_type()
    layouts = [([], ()), ([hdr_len], (0,)), ([starts[0] + 2], ()), ([starts[1] + 40, starts[1] + 90], (1,)),
               ([starts[3] + 1, starts[5] + 3, starts[10] + 50], (0, 2)), (list(range(hdr_len, len(payload), 37)), (3, 4, 5))]
    for cuts, empties in layouts:
        p = tmp_path / 'zoo.bam'
        bamvec.write_blocks(str(p), payload, cuts, level=level, empty_after=empties)
        for threads in (1, 3):
            got, names = _decode_all(p, threads=threads, batch_size=5)
            assert names == ['chr1', 'chr2']
            bamvec.check_rows(got, want)
        # '/'-names: mates share a key, different names do not
        assert got['qkey'][0] != got['qkey'][1]


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_files_from_other_writers(tmp_path, seed):
    """Record shapes and BGZF blocks this repo's writer never produces (tests/bamvec.py, files_never_seen and
    write_blocks_varied): 254-character names, l_seq = 0, the 34-byte record, mapped without CIGAR, every aux type
    behind CB/UB, a read that fills a maximal block; every block with its own zlib level / window / memLevel / strategy
    and foreign gzip extra subfields around BC; htslib's 0xFF00-byte blocks and tiny ones with empty blocks between."""
    import bamvec
    from te_counter_amd import dist
    payload, want, hdr_len, starts = bamvec.files_never_seen()
    layouts = [(list(range(0xff00, len(payload), 0xff00)), ()), ([hdr_len, hdr_len + 30000], (0,)), (sorted(set(starts[1:])), (2, 3)),
               (list(range(hdr_len, len(payload), 4001)), (1, 5, 6)), ([starts[3] + 2, starts[3] + 33, starts[4] + 1], (0, 1, 2))]
    for cuts, empties in layouts:
        p = tmp_path / 'unseen.bam'
        bamvec.write_blocks_varied(str(p), payload, cuts, seed, empty_after=empties)
        for threads in (1, 3):
            got, names = _decode_all(p, threads=threads, batch_size=4)
            assert names == ['chr1', 'chrUn_KI270742v1', 'chr2']
            bamvec.check_rows(got, want)
        assert got['qkey'][9] == got['qkey'][3]          # '/' and a name without '/': the key is the (empty) text before the last '/'
        assert got['qkey'][8] != got['qkey'][9] and got['qkey'][0] != got['qkey'][8]
        # byte ranges: a range start is found by the general BGZF header form too (extra subfields around BC)
        size = os.path.getsize(str(p))
        for world in (2, 3):
            rows, reports = [], []
            for r in range(world):
                src = bam.BamSource(str(p), byte_range=bam.byte_range(size, r, world), batch_size=4, threads=2)
                parts = list(src._batches(want_tags=True))
                first, end, has = src.range_tokens()
                src.close()
                n = sum(len(b) for b in parts)
                reports.append({'has_records': has and n > 0, 'first_token': first, 'end_token': end, 'n_records': n})
                rows += parts
            if rows:
                ranged = {k: np.concatenate([getattr(b, k) for b in rows]) for k in ('tid', 'pos', 'end', 'flag', 'mapq', 'cb24', 'ub24', 'tagflags')}
                bamvec.check_rows(ranged, want)
            dist.check_ranges(reports, size)


def test_hand_assembled_vectors_byte_ranges(tmp_path):
    """The same records cut into 37-byte BGZF blocks and read as 1..9 byte ranges: every range owns the
    records that start in its blocks, nothing is lost or doubled, tokens chain."""
    import bamvec
    from te_counter_amd import dist
    payload, want, hdr_len, starts = bamvec.every_cigar_op_and_aux_type()
    p = tmp_path / 'zoo.bam'
    bamvec.write_blocks(str(p), payload, list(range(hdr_len, len(payload), 37)), level=1, empty_after=(2, 9))
    size = os.path.getsize(str(p))
    for world in (1, 2, 3, 5, 9):
        rows, reports = [], []
        for r in range(world):
            src = bam.BamSource(str(p), byte_range=bam.byte_range(size, r, world), batch_size=4, threads=2)
            parts = list(src._batches(want_tags=True))
            first, end, has = src.range_tokens()
            src.close()
            n = sum(len(b) for b in parts)
            reports.append({'has_records': has and n > 0, 'first_token': first, 'end_token': end, 'n_records': n})
            rows += parts
        got = {k: np.concatenate([getattr(b, k) for b in rows]) for k in ('tid', 'pos', 'end', 'flag', 'mapq', 'cb24', 'ub24', 'tagflags')}
        bamvec.check_rows(got, want)
        dist.check_ranges(reports, size)


def test_first_range_without_records_still_pins_the_first_record(tmp_path):
    """A header larger than range 0's share of the file: range 0 owns no record, and the first range that does
    starts from a guess.  Range 0 reports the header's first-record token, so check_ranges verifies that guess too
    -- and refuses a forged one."""
    import bamvec
    from te_counter_amd import dist
    refs = [('contig_with_a_long_name_%04d' % i, 100000 + i) for i in range(120)]
    payload = bamvec.header(refs)
    hdr_len = len(payload)
    n_rec = 60
    for i in range(n_rec):
        payload += bamvec.record(i % 120, 10 * i, 30, 0, 'read%03d/1' % i, '20M', aux_bytes=bamvec.aux('NH', 'i', 1))
    p = tmp_path / 'bighdr.bam'
    bamvec.write_blocks(str(p), payload, list(range(300, len(payload), 300)), level=1)
    size = os.path.getsize(str(p))
    seen_empty_first = False
    for world in (2, 3, 4):
        reports, pos = [], []
        for r in range(world):
            src = bam.BamSource(str(p), byte_range=bam.byte_range(size, r, world), batch_size=7, threads=1)
            got = [b.pos for b in src._batches()]
            first, end, has = src.range_tokens()
            src.close()
            n = sum(len(b) for b in got)
            reports.append({'has_records': has and n > 0, 'first_token': first, 'end_token': end, 'n_records': n})
            pos += got
        assert np.concatenate(pos).tolist() == [10 * i for i in range(n_rec)]
        dist.check_ranges(reports, size)
        if not reports[0]['has_records']:
            seen_empty_first = True
            assert tuple(reports[0]['first_token']) != (0, 0)
            k = next(i for i, rep in enumerate(reports) if rep['has_records'])
            forged = [dict(rep) for rep in reports]
            forged[k]['first_token'] = (forged[k]['first_token'][0], forged[k]['first_token'][1] + 1)
            with pytest.raises(RuntimeError):
                dist.check_ranges(forged, size)
    assert seen_empty_first and hdr_len > size // 4


def test_thread_count_from_the_environment(tmp_path, monkeypatch):
    """TEC_BAM_THREADS sets the host decoder's threads when the caller passes none: same rows with 1, 3 and the default."""
    recs = _records(3000)
    path = str(tmp_path / 't.bam')
    bam.write_bam(path, _refs(recs), recs, block_bytes=900)
    ref = None
    for value in (None, '1', '3'):
        if value is None:
            monkeypatch.delenv('TEC_BAM_THREADS', raising=False)
        else:
            monkeypatch.setenv('TEC_BAM_THREADS', value)
        got, _ = _decode_all(path, threads=0, batch_size=257)
        cols = [got[k].tolist() for k in ('tid', 'pos', 'end', 'flag', 'mapq', 'qkey')]
        ref = ref or cols
        assert cols == ref and len(cols[0]) == len(recs)


@pytest.mark.parametrize('sc', [False, True])
def test_pysam_style_file_through_the_adapter(sc):
    """An open pysam-style alignment file (te_count.py:65,190,351 hold one) gives the same batches as the record dicts."""
    from helpers import FakeAlignmentFile
    recs = _records(3001, seed=8, sc=sc)
    if sc:  # (a record whose only barcode / UMI tags are the raw ones: te_count.py:405-406, 423-424)
        recs[5]['tags'] = {'CR': recs[5]['tags']['CB'], 'UR': recs[5]['tags']['UB']}
    a = _collect(SegmentSource(FakeAlignmentFile(recs), batch_size=700), want_tags=sc)
    b = _collect(ListSource(recs, batch_size=700), want_tags=sc)
    assert len(a) == len(b) == 5
    assert SegmentSource(FakeAlignmentFile(recs)).reference_names == ListSource(recs).reference_names
    for x, y in zip(a, b):
        for f in ('tid', 'pos', 'end', 'flag', 'mapq'):
            assert np.array_equal(getattr(x, f), getattr(y, f)), f
        assert x.qname == y.qname and x.cb == y.cb and x.ub == y.ub
    # reference_end is None without a CIGAR
    f = FakeAlignmentFile(recs[:3])
    f._segments[1].reference_end = None
    (batch,) = _collect(SegmentSource(f))
    assert batch.end[1] == bam.NO_END and batch.end[0] == recs[0]['end']
    with pytest.raises(TypeError):
        SegmentSource(iter(()))  # no reference names to be had


def test_cram_path_without_pysam_says_so(tmp_path):
    from te_counter_amd.measure import measureTE
    p = tmp_path / 'x.CRAM'
    p.write_bytes(b'CRAM')
    try:
        import pysam  # noqa: F401
        pytest.skip('pysam is installed')
    except ImportError:
        pass
    with pytest.raises(ImportError, match='CRAM input needs pysam'):
        measureTE._as_source(str(p))
    assert measureTE._as_source('reads.bam') == 'reads.bam'
