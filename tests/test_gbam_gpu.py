"""The on-GPU BAM decoder (csrc/gbam.hip: BGZF inflate, record boundaries, field extraction, PE
pairing on the MI355X) against the host decoder, the hand-assembled specification vectors and the
oracle."""

import os

import numpy as np
import pytest

import bamvec
from oracle import c_oracle
from te_counter_amd import Engine, bam, dist, synth
from te_counter_amd.gbam import NONE64, GpuBam
from te_counter_amd.reads import PairAssembler, se_units

pytestmark = pytest.mark.gpu

FIELDS = ('tid', 'pos', 'end', 'flag', 'mapq', 'qkey', 'cb24', 'ub24', 'tagflags')


@pytest.fixture(scope='module')
def eng():
    e = Engine(0)
    yield e
    e.close()


def host_decode(path, byte_range=None):
    src = bam.BamSource(str(path), byte_range=byte_range, threads=4)
    parts = list(src._batches(want_tags=True))
    src.close()
    if not parts:
        return {k: np.zeros((0, 24) if k in ('cb24', 'ub24') else 0) for k in FIELDS}
    return {k: np.concatenate([getattr(b, k) for b in parts]) for k in FIELDS}


def gpu_decode(eng, path, byte_range=None):
    g = GpuBam(eng, str(path), byte_range)
    rep, arr = g.decode()
    g.close()
    return rep, arr


def same(a, b):
    for k in FIELDS:
        assert np.array_equal(a[k], b[k]), k


@pytest.fixture
def small_chunks(monkeypatch):
    """Many chunks per file, several pieces of the pinned staging ring per chunk: record tails carried between
    chunks, a chunk boundary inside a record, BGZF blocks cut by the end of a piece."""
    monkeypatch.setenv('TEC_GBAM_CHUNK_MB', '1')
    monkeypatch.setenv('TEC_GBAM_CHUNK_BLOCKS', '37')
    monkeypatch.setenv('TEC_GBAM_PIECE_KB', '192')


def _synthetic(tmp_path, payload, n_units, level=1, sc=False):
    ann = synth.make_annotation('hg38', scale=0.02)
    path = str(tmp_path / 'x.bam')
    if sc:
        arrays, wl, tables = synth.make_sc_arrays(ann, n_units, n_cells=50, n_whitelist=4000, seed=5)
        bam.write_bam_arrays(path, tables[0], arrays['cls'], arrays['pos'], arrays['flag'], arrays['mapq'], end=arrays['end'] + 2,
                             cb=arrays['cb24'], ub=arrays['ub24'], payload=payload, level=level)
        return ann, path, arrays
    u = synth.make_units(ann, n_units, seed=3, paired=True)
    tid, pos, fl, mq = bam.units_to_records(u, len(ann.chrom_names), True)
    bam.write_bam_arrays(path, ann.chrom_names + ['chrUn_1'], tid, pos, fl, mq, pair_names=True, payload=payload, level=level)
    return ann, path, u


@pytest.mark.parametrize('payload,level', [('random', 1), ('illumina', 1), ('illumina', 6), ('random', 9), ('illumina', 0)])
def test_every_field_equals_the_host_decoder(eng, tmp_path, payload, level):
    """Dynamic, fixed and stored deflate blocks; literal-heavy and match-heavy payloads."""
    _, path, _ = _synthetic(tmp_path, payload, 150_000, level)
    rep, got = gpu_decode(eng, path)
    want = host_decode(path)
    assert rep['n_records'] == 300_000 == len(want['tid'])
    same(got, want)
    assert rep['blocks'] > 100 and rep['end_token'] == (os.path.getsize(path), 0)


@pytest.mark.parametrize('world', [1, 3])
def test_header_spanning_several_bgzf_blocks(eng, tmp_path, small_chunks, world):
    """An assembly with thousands of contigs: the BAM header (text + reference table) fills several BGZF blocks
    and the first record starts in the middle of one; byte ranges whose start falls inside the header."""
    ann = synth.make_annotation('hg38', scale=0.02)
    refs = ann.chrom_names + ['chrUn_KI%06dv1_decoy' % i for i in range(9000)]
    u = synth.make_units(ann, 20_000, seed=8, paired=True)
    tid, pos, fl, mq = bam.units_to_records(u, len(ann.chrom_names), True)
    path = str(tmp_path / 'h.bam')
    bam.write_bam_arrays(path, refs, tid, pos, fl, mq, pair_names=True, payload='illumina')
    size = os.path.getsize(path)
    want = host_decode(path)
    assert len(want['tid']) == 40_000
    parts = []
    for r in range(world):
        rng = bam.byte_range(size, r, world) if world > 1 else None
        rep, arr = gpu_decode(eng, path, rng)
        same(arr, host_decode(path, rng))
        parts.append(arr)
    same({k: np.concatenate([p[k] for p in parts]) for k in FIELDS}, want)
    g = GpuBam(eng, path)
    assert g.reference_names == refs
    g.close()


def test_many_chunks_and_single_cell_tags(eng, tmp_path, small_chunks):
    _, path, arrays = _synthetic(tmp_path, 'illumina', 120_000, sc=True)
    rep, got = gpu_decode(eng, path)
    want = host_decode(path)
    same(got, want)
    assert np.array_equal(got['cb24'], arrays['cb24']) and np.array_equal(got['ub24'], arrays['ub24'])
    assert rep['blocks'] > 4 * 37  # several chunks


@pytest.mark.parametrize('world', [2, 3, 7, 40])
def test_byte_ranges_tile_the_file(eng, tmp_path, small_chunks, world):
    _, path, _ = _synthetic(tmp_path, 'random', 40_000)
    size = os.path.getsize(path)
    whole = host_decode(path)
    parts, reports = [], []
    for r in range(world):
        rep, arr = gpu_decode(eng, path, bam.byte_range(size, r, world))
        h = host_decode(path, bam.byte_range(size, r, world))
        same(arr, h)  # the two decoders own the same records of every range
        parts.append(arr)
        reports.append(rep)
    cat = {k: np.concatenate([p[k] for p in parts]) for k in FIELDS}
    same(cat, whole)
    dist.check_ranges(reports, size)


@pytest.mark.parametrize('chunked,decoys', [(False, False), (True, False), (True, True)])
def test_long_reads_spanning_many_segments(eng, tmp_path, monkeypatch, chunked, decoys):
    """Long-read records (up to 400 kb of sequence = 600 KB records) between short ones: whole 64 KiB segments and
    whole BGZF blocks hold no record start.  decoys: the long records carry, in a B-array tag, three complete
    plausible BAM records in a row -- exactly what the start-of-range guess looks for.  The whole-file decode is
    immune (its first record start is known and every later one is reached by the chain); a byte range that
    begins inside such a tag may guess wrong, and then the range tokens must give it away (dist.check_ranges)."""
    if chunked:
        monkeypatch.setenv('TEC_GBAM_CHUNK_BLOCKS', '5')
        monkeypatch.setenv('TEC_GBAM_PIECE_KB', '192')
    import random
    rnd = random.Random(7)
    refs = [('chr1', 5_000_000), ('chr2', 3_000_000)]
    payload = bamvec.header(refs)
    want = []
    decoy = bamvec.record(1, 1234, 30, 0, 'decoy/1', '50M')
    for i in range(260):
        kind = rnd.random()
        l = 40 if kind < 0.55 else (rnd.randrange(60_000, 90_000) if kind < 0.85 else rnd.randrange(150_000, 400_000))
        pos = rnd.randrange(0, 2_000_000)
        aux = bamvec.aux('XD', 'B', list(decoy * 3), 'C') if (decoys and kind >= 0.85) else bamvec.aux('NM', 'i', i)
        payload += bamvec.record(i % 2, pos, 20 + i % 40, 16 * (i % 2), 'long%d/1' % i, '%dM' % l, aux_bytes=aux)
        want.append((i % 2, pos, pos + l, 16 * (i % 2), 20 + i % 40))
    p = str(tmp_path / 'long.bam')
    bamvec.write_blocks(p, payload, list(range(0xff00, len(payload), 0xff00)), level=1)  # htslib's block size
    rep, got = gpu_decode(eng, p)
    assert rep['n_records'] == len(want)
    for k, col in enumerate(('tid', 'pos', 'end', 'flag', 'mapq')):
        assert got[col].tolist() == [w[k] for w in want], col
    same(got, host_decode(p))
    size = os.path.getsize(p)
    parts, reports = [], []
    for r in range(5):
        rng = bam.byte_range(size, r, 5)
        rp, arr = gpu_decode(eng, p, rng)
        same(arr, host_decode(p, rng))  # the two decoders take the same decision, right or wrong
        parts.append(arr)
        reports.append(rp)
    tiled = sum(len(q['tid']) for q in parts) == len(want)
    if tiled:
        dist.check_ranges(reports, size)
        same({k: np.concatenate([q[k] for q in parts]) for k in FIELDS}, got)
    else:
        assert decoys
        with pytest.raises(RuntimeError, match='inconsistent'):
            dist.check_ranges(reports, size)


def test_specification_vectors(eng, tmp_path):
    """The SAM spec's example alignment and the every-CIGAR-op / every-aux-type records, in block layouts
    that cut records (and the size field) across two and more BGZF blocks, with empty blocks."""
    payload, want = bamvec.spec_example()
    p = str(tmp_path / 'spec.bam')
    bamvec.write_blocks(p, payload, [])
    _, got = gpu_decode(eng, p)
    bamvec.check_rows(got, want)
    payload, want, hdr_len, starts = bamvec.every_cigar_op_and_aux_type()
    for level in (-1, 1, 6):
        for cuts, empties in [([], ()), ([hdr_len], (0,)), ([starts[0] + 2], ()), ([starts[1] + 40, starts[1] + 90], (1,)),
                              (list(range(hdr_len, len(payload), 37)), (3, 4, 5))]:
            bamvec.write_blocks(p, payload, cuts, level=level, empty_after=empties)
            rep, got = gpu_decode(eng, p)
            bamvec.check_rows(got, want)
            same(got, host_decode(p))
            size = os.path.getsize(p)
            for world in (2, 5):
                parts = [gpu_decode(eng, p, bam.byte_range(size, r, world)) for r in range(world)]
                bamvec.check_rows({k: np.concatenate([a[k] for _, a in parts]) for k in FIELDS}, want)
                dist.check_ranges([r for r, _ in parts], size)


@pytest.mark.parametrize('seed', [1, 2])
def test_files_from_other_writers(eng, tmp_path, seed):
    """tests/bamvec.py files_never_seen / write_blocks_varied (254-character names, l_seq = 0, the 34-byte record, every
    aux type behind CB/UB, a read that fills a block; per-block zlib level / window / memLevel / strategy, foreign gzip
    extra subfields): the GPU decoder delivers the expected rows, agrees with the host decoder, and so do byte ranges."""
    payload, want, hdr_len, starts = bamvec.files_never_seen()
    p = str(tmp_path / 'unseen.bam')
    for cuts, empties in [(list(range(0xff00, len(payload), 0xff00)), ()), (sorted(set(starts[1:])), (2, 3)),
                          (list(range(hdr_len, len(payload), 4001)), (1, 5, 6)), ([starts[3] + 2, starts[3] + 33, starts[4] + 1], (0, 1, 2))]:
        bamvec.write_blocks_varied(p, payload, cuts, seed, empty_after=empties)
        rep, got = gpu_decode(eng, p)
        bamvec.check_rows(got, want)
        same(got, host_decode(p))
        size = os.path.getsize(p)
        for world in (2, 3):
            parts = [gpu_decode(eng, p, bam.byte_range(size, r, world)) for r in range(world)]
            bamvec.check_rows({k: np.concatenate([a[k] for _, a in parts]) for k in FIELDS}, want)
            dist.check_ranges([r for r, _ in parts], size)


def test_corrupt_and_truncated_files(eng, tmp_path):
    _, path, _ = _synthetic(tmp_path, 'illumina', 20_000)
    raw = open(path, 'rb').read()
    bad = str(tmp_path / 'bad.bam')
    open(bad, 'wb').write(raw[:len(raw) // 2])                 # cut inside a block
    with pytest.raises(IOError):
        gpu_decode(eng, bad)
    cut = bytearray(raw)
    cut[len(raw) // 2] ^= 0x55                                  # one damaged byte inside a deflate stream
    open(bad, 'wb').write(bytes(cut))
    try:
        _, got = gpu_decode(eng, bad)
        same(got, host_decode(bad))                             # if zlib accepts it, both decoders deliver the same bytes
    except IOError:
        with pytest.raises(IOError):
            host_decode(bad)
    open(bad, 'wb').write(b'not a bam at all' * 10)
    with pytest.raises(IOError):
        GpuBam(eng, bad)


@pytest.mark.parametrize('where', [0.1, 0.5, 0.95])
def test_failure_in_a_later_chunk_leaves_the_engine_usable(eng, tmp_path, small_chunks, where):
    """Many chunks, two inflate launches in flight: a truncated file / a damaged block somewhere in the middle must
    surface as an error (or as what zlib makes of it), never hang, and the next file on the same engine -- which
    takes over the parked buffers -- must decode correctly."""
    _, good, _ = _synthetic(tmp_path, 'illumina', 60_000)
    raw = open(good, 'rb').read()
    bad = str(tmp_path / 'bad.bam')
    open(bad, 'wb').write(raw[:int(len(raw) * where)])
    with pytest.raises(IOError):
        gpu_decode(eng, bad)
    cut = bytearray(raw)
    for k in range(40):
        cut[int(len(raw) * where) + 7 * k] ^= 0xA5  # a burst inside deflate streams / block headers
    open(bad, 'wb').write(bytes(cut))
    try:
        _, got = gpu_decode(eng, bad)
        same(got, host_decode(bad))
    except IOError:
        with pytest.raises(IOError):
            host_decode(bad)
    rep, got = gpu_decode(eng, good)
    assert rep['n_records'] == 120_000
    same(got, host_decode(good))


@pytest.mark.parametrize('paired', [False, True])
def test_counts_from_a_bam_equal_the_oracle(eng, tmp_path, small_chunks, paired):
    """BAM file -> GPU decode -> count kernel, nothing on the host in between: counts, log counters
    and the record total against the C oracle on the units the file was written from."""
    ann = synth.make_annotation('hg38', scale=0.02)
    eng.upload_index(ann.flat)
    u = synth.make_units(ann, 90_001, seed=9, paired=paired)
    refs = ann.chrom_names + ['chrUn_1']
    path = str(tmp_path / 'c.bam')
    if paired:
        tid, pos, fl, mq = bam.units_to_records(u, len(refs) - 1, True)
        bam.write_bam_arrays(path, refs, tid, pos, fl, mq, pair_names=True, payload='illumina')
    else:
        tid, pos, fl, mq = bam.units_to_records(u, len(refs) - 1, False)
        bam.write_bam_arrays(path, refs, tid, pos, fl, mq, end=u[2], payload='illumina')
    want, wst = c_oracle.COracleIndex.from_flat(ann.flat).bulk_count(*u, 20)
    g = GpuBam(eng, path)
    eng.bulk_begin(20)
    rep = g.run_bulk(paired, 20, ann.flat.tid_map(g.reference_names))
    got, gst = eng.bulk_finish()
    g.close()
    assert rep['n_records'] == (2 if paired else 1) * 90_001
    assert rep['first_noend_live'] == NONE64 and rep['first_name_mismatch'] == NONE64
    assert np.array_equal(got, want) and all(gst[k] == v for k, v in wst.items())


@pytest.mark.parametrize('world', [2, 3, 5])
@pytest.mark.parametrize('n_records', [60_000, 60_001, 59_999])
def test_pe_pairs_across_byte_ranges(eng, tmp_path, small_chunks, world, n_records):
    """The multi-GPU PE decomposition on one device: every range keeps its records, the parity of the
    records before it decides which are mates, an unpaired last record borrows the next range's first
    record; the summed counts equal the single-reader result (odd totals drop the trailing record)."""
    ann = synth.make_annotation('hg38', scale=0.02)
    eng.upload_index(ann.flat)
    u = synth.make_units(ann, 30_001, seed=11, paired=True)
    refs = ann.chrom_names + ['chrUn_1']
    tid, pos, fl, mq = bam.units_to_records(u, len(refs) - 1, True)
    path = str(tmp_path / 'p.bam')
    bam.write_bam_arrays(path, refs, tid[:n_records], pos[:n_records], fl[:n_records], mq[:n_records], pair_names=True, payload='random')
    size = os.path.getsize(path)
    g = GpuBam(eng, path)
    tm = ann.flat.tid_map(g.reference_names)
    eng.bulk_begin(20)
    rep1 = g.run_bulk(True, 20, tm)
    want, wst = eng.bulk_finish()
    g.close()
    assert rep1['units_counted'] == n_records // 2 and rep1['odd_record_left'] == bool(n_records & 1)
    feeds = [GpuBam(eng, path, bam.byte_range(size, r, world)) for r in range(world)]
    eng.bulk_begin(20)
    reports = [f.run_bulk(True, 20, tm, defer_pairing=True) for f in feeds]
    dist.check_ranges(reports, size)
    total_units = 0
    for r, f in enumerate(feeds):
        skip, extra, unit0 = dist.plan_pairs(reports, r)
        n_units, mism = f.finish_pairs(20, skip, extra, unit0)
        assert mism == NONE64 and unit0 == total_units
        total_units += n_units
        f.close()
    got, gst = eng.bulk_finish()
    assert total_units == n_records // 2
    assert np.array_equal(got, want) and gst == wst
