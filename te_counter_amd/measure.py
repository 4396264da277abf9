"""`measureTE`: the reference's counting class (te_count/te_count.py:14-754) with the same
method names, arguments, return values and failure modes, running on the MI355X through
libtecount.so.

What stays on the host, as in the reference: opening the alignment file, `total_reads`, name
sorting and every byte of text formatting (TSV float repr must match Python's).
"""

import os

import numpy as np

import ctypes

from . import _lib
from . import reads as _reads
from .bam import NO_END
from .engine import Engine
from .index import FlatIndex
from .scresult import ScResult


class _NullLog:
    def info(self, *a, **k):
        pass

    error = warning = debug = info


class measureTE:
    def __init__(self, base_path, quality_threshold, device=0, batch_size=1 << 22, bin_shift=0, world=None):
        self.base_path = base_path
        self.total_reads = 0
        self.quality_threshold = quality_threshold
        self.device = device
        self.batch_size = batch_size
        self.bin_shift = bin_shift
        self.genome = None
        self.stats = {}
        self._engine = None
        # EXTENSION, off unless set: parse_bamse / parse_bampe(strand=True) count with the single-cell path's strand
        # rule (te_count.py:654, 665) instead of raising what the reference raises (te_count.py:58-59, 183-184).
        # No reference output exists for it: nothing pins its results.
        self.bulk_strand_extension = False
        self._split_for = None  # the FlatIndex whose strand-split form is resident on the GPU instead of itself
        if world is None:
            from .dist import World
            world = World()
        self.world = world  # rank / size of the multi-GPU job (1 process per GPU)

    # ------------------------------------------------------------------ index
    def bind_genome(self, genelist_glb_filename):
        """te_count.py:37-40.  Accepts a `.tecidx` file or a ready FlatIndex."""
        if isinstance(genelist_glb_filename, FlatIndex):
            self.genelist_glb_filename = genelist_glb_filename
            return
        assert os.path.isfile(genelist_glb_filename), f'{genelist_glb_filename} not found'
        self.genelist_glb_filename = genelist_glb_filename

    def load_genome(self):
        """te_count.py:31-35: loads the index and derives all_feature_names.  Loading the genome that is already
        resident on the GPU again (sc_parse_bamse does it on every call, like the reference's part 3) is free."""
        assert getattr(self, 'genelist_glb_filename', None) is not None, 'You need to bind the genome first'
        src = self.genelist_glb_filename
        # (a FlatIndex is identified by the object itself -- a reference is kept, ids are recycled -- a file by path, mtime, size)
        key = (src if isinstance(src, FlatIndex) else (str(src), os.path.getmtime(src), os.path.getsize(src)), self.bin_shift)
        held = getattr(self, '_resident', None)
        if self._engine is not None and held is not None and held[1] == key[1] and \
                (held[0] is key[0] if isinstance(src, FlatIndex) else held[0] == key[0]):
            return
        if isinstance(src, FlatIndex):
            self.genome = src
        elif str(src).endswith('.glb'):
            self.genome = FlatIndex.from_glb(src)  # an index built by the reference's te_genome
        else:
            self.genome = FlatIndex.load(src)
        self.all_feature_names = list(self.genome.names)
        if self._engine is None:
            self._engine = Engine(self.device)
        self._engine.upload_index(self.genome, self.bin_shift)
        self._resident = key
        self._split_for = None

    def _index_for(self, stranded):
        """The plain index, or (stranded-bulk extension) its strand-split form, resident on the GPU."""
        if stranded:
            if self._split_for is not self.genome:
                self._engine.upload_index(self.genome.strand_split(), self.bin_shift)
                self._split_for = self.genome
                self._resident = None
        elif self._split_for is not None:
            self._resident = None
            self.load_genome()

    # ------------------------------------------------------------------ bulk
    def _open(self, filename, byte_range=None):
        if hasattr(filename, 'batches'):
            return filename
        from .bam import open_alignment
        return open_alignment(filename, batch_size=self.batch_size, byte_range=byte_range)

    @staticmethod
    def _as_source(filename):
        """A path stays a path (the native decoders read BAM and SAM).  An OPEN pysam-style alignment
        file -- what the reference itself holds after `pysam.AlignmentFile(filename, 'r')`
        (te_count.py:65,190,351) -- and a CRAM path (read through pysam when it is installed; neither native
        decoder reads CRAM) go through the record-by-record adapter."""
        if hasattr(filename, 'batches'):
            return filename
        if isinstance(filename, (str, bytes, os.PathLike)):
            if os.fsdecode(filename).lower().endswith('.cram'):
                try:
                    import pysam
                except ImportError as e:
                    raise ImportError('%s: CRAM input needs pysam (it is not installed); convert to BAM with '
                                      '`samtools view -b`' % os.fsdecode(filename)) from e
                return _reads.SegmentSource(pysam.AlignmentFile(os.fsdecode(filename), 'r'))
            return filename
        if hasattr(filename, 'references'):
            return _reads.SegmentSource(filename)
        return filename

    @staticmethod
    def _decoder_for(filename):
        """'gpu': BGZF-compressed BAM files are decoded on the device (csrc/gbam.hip); 'cpu': the host
        decoder (csrc/bamio.hip) -- SAM text, in-memory sources, or TEC_DECODER=cpu."""
        from .gbam import is_bgzf
        if hasattr(filename, 'batches') or os.environ.get('TEC_DECODER', 'gpu') == 'cpu':
            return 'cpu'
        return 'gpu' if is_bgzf(filename) else 'cpu'

    def _bulk(self, filename, paired, strand, log):
        assert filename, 'You must specify a filename'
        if strand and not self.bulk_strand_extension:
            raise NotImplementedError()  # te_count.py:58-59, 183-184
        filename = self._as_source(filename)
        log = log or _NullLog()
        assert self.genome is not None, 'call load_genome() first'
        self._index_for(bool(strand))
        eng, world = self._engine, self.world
        q = self.quality_threshold
        q_int = int(q) if isinstance(q, (int, np.integer)) else None
        n_records, failure, stats, counts = 0, None, None, None
        try:
            eng.bulk_begin(q_int or 0)
            if strand:
                eng.bulk_set_stranded(True)
            if self._decoder_for(filename) == 'gpu':
                n_records, failure = self._bulk_gpu_decode(filename, paired, q_int or 0)
            else:
                n_records, failure = self._bulk_host_decode(filename, paired, q_int or 0)
        except Exception as e:  # noqa: BLE001 -- the other ranks must hear about it before the next collective
            if world.size == 1:
                raise
            failure = e
        world.raise_together(failure)
        on_device = world.size > 1 and getattr(world, 'backend', None) == 'nccl'
        if on_device:
            # RCCL sum all-reduce of the count vector where it lives (libtecount's HBM buffer, in the index's internal
            # name order, the same on every rank): what bulk_finish then copies out is already the job's total
            eng.bulk_sync()
            world.sum_device_counts(eng.bulk_device_buffers()[0], eng.n_names)
        counts, stats, enhancer_only = eng.bulk_finish(raise_enhancer=False)
        keys = ('units', 'assigned', 'lowq', 'badchrom', 'qcfail')
        if world.size > 1:  # the log counters (and, without RCCL, the count vector through host memory)
            counts, sv = world.sum_counts(None if on_device else counts, [stats[k] for k in keys] + [int(enhancer_only), n_records], counts)
            stats = {k: int(v) for k, v in zip(keys, sv)}
            enhancer_only, n_records = bool(sv[5]), int(sv[6])
        if q_int is None and stats['units'] - stats['qcfail'] > 0:
            # `-q N` on the reference CLI hands over a list and the first record that passes the flag
            # filter dies in `int(mapq) < [N]` (bin/te_count:30,77; te_count.py:88,208)
            raise TypeError("'<' not supported between instances of 'int' and '%s'" % type(q).__name__)
        if enhancer_only:  # te_count.py:145-147, 258-260 -- on every rank alike
            raise NameError("name 'barcode' is not defined")
        units_total = n_records // 2 if paired else n_records
        self.total_reads = units_total + 1  # idx is bumped before next(sam): te_count.py:77,202 / 163,275
        self.stats = stats
        what = 'reads' if paired else 'SE reads'
        log.info('Processed {:,} {}'.format(self.total_reads, what))
        log.info('{:,} Reads were assigned to a gene'.format(stats['assigned']))
        log.info('{:,} Read quality is too low (<{})'.format(stats['lowq'], self.quality_threshold))
        log.info('{:,} Reads mapped to an invalid chromosome'.format(stats['badchrom']))
        log.info('{:,} Reads are QC fails'.format(stats['qcfail']))
        return {name: int(c) for name, c in zip(self.all_feature_names, counts)}

    _NAME_MISMATCH = "module 'sys' has no attribute 'quit'"            # te_count.py:94 calls sys.quit()
    _NO_END = "unsupported operand type(s) for +: 'NoneType' and 'int'"  # te_count.py:223 with reference_end None

    def _bulk_gpu_decode(self, filename, paired, q):
        """BAM file -> (this rank's byte range) -> on-GPU decode -> count kernel.  Returns (records owned
        by this rank, the exception the reference would have died with or None)."""
        from . import dist as _dist
        from .bam import byte_range
        from .gbam import NONE64, GpuBam
        world = self.world
        size = os.path.getsize(filename)
        feed = GpuBam(self._engine, filename, byte_range(size, world.rank, world.size) if world.size > 1 else None)
        self.decode_report = None
        try:
            tid_map = self.genome.tid_map(feed.reference_names)
            defer = paired and world.size > 1
            rep = feed.run_bulk(paired, q, tid_map, defer_pairing=defer)
            self.decode_report = rep
            failure = None
            if not paired and rep['first_noend_live'] != NONE64:
                failure = TypeError(self._NO_END)
            if paired and not defer and rep['first_name_mismatch'] != NONE64:
                failure = AttributeError(self._NAME_MISMATCH)
            if world.size > 1:
                reports = world.gather({k: rep[k] for k in ('n_records', 'has_records', 'first_token', 'end_token', 'first_rec')})
                _dist.check_ranges(reports, size)
                if paired:
                    skip, extra, unit0 = _dist.plan_pairs(reports, world.rank)
                    _, mism = feed.finish_pairs(q, skip, extra, unit0)
                    if mism != NONE64:
                        failure = AttributeError(self._NAME_MISMATCH)
            return rep['n_records'], failure
        finally:
            feed.close()

    def _bulk_host_decode(self, filename, paired, q):
        """Host decoder (SAM text, in-memory record sources, TEC_DECODER=cpu): SoA batches -> units ->
        tec_bulk_submit.  A BAM file is cut into byte ranges over the ranks like the GPU decoder does;
        other sources are read by every rank and their batches dealt round-robin."""
        from . import dist as _dist
        from .gbam import is_bgzf
        from .bam import byte_range
        world, eng = self.world, self._engine
        ranged = world.size > 1 and not hasattr(filename, 'batches') and is_bgzf(filename)
        src = self._open(filename, byte_range(os.path.getsize(filename), world.rank, world.size) if ranged else None)
        failure, n_records = None, 0
        try:
            tid_map = self.genome.tid_map(src.reference_names)
            keeper = _reads.RecordKeeper() if (paired and ranged) else None
            pairs = _reads.PairAssembler(tid_map, q) if (paired and not ranged) else None
            for ordinal, batch in enumerate(src.batches()):
                n_records += len(batch)
                if keeper is not None:
                    keeper.add(batch)
                    continue
                try:
                    units = pairs.units(batch) if paired else _reads.se_units(batch, tid_map)
                except AttributeError as e:
                    failure = failure or e
                    continue
                if not paired and (units[2] == NO_END).any() and self._has_live_no_end(units, q):
                    failure = failure or TypeError(self._NO_END)
                if len(units[0]) and (ranged or world.mine(ordinal)):
                    eng.bulk_submit(*units)
            if ranged:
                first, end, has = src.range_tokens()
                rep = {'n_records': n_records, 'has_records': has and n_records > 0, 'first_token': first, 'end_token': end,
                       'first_rec': keeper.record(0) if (keeper is not None and n_records) else None}
                reports = world.gather(rep)
                _dist.check_ranges(reports, os.path.getsize(filename))
                if paired:
                    skip, extra, _ = _dist.plan_pairs(reports, world.rank)
                    try:
                        units = keeper.units(tid_map, q, skip, extra)
                        if len(units[0]):
                            eng.bulk_submit(*units)
                    except AttributeError as e:
                        failure = failure or e
            elif world.size > 1:
                n_records = n_records if world.rank == 0 else 0  # every rank read the whole source: count it once
        finally:
            src.close()
        return n_records, failure

    @staticmethod
    def _has_live_no_end(units, q):
        # a mapped SE record without CIGAR has reference_end None: `loc2+1` dies (te_count.py:223)
        chrom, _, p2, flag, mapq = units
        return bool(((p2 == NO_END) & ((flag & _reads.FLAG_FILTER) == 0) & (mapq >= q) & (chrom >= 0)).any())

    @staticmethod
    def _reaches_mapq_test(batch, paired):
        fl = batch.flag
        if paired:
            n = len(fl) - (len(fl) % 2)
            ok = ((fl[0:n:2] | fl[1:n:2]) & _reads.FLAG_FILTER) == 0
        else:
            ok = (fl & _reads.FLAG_FILTER) == 0
        return bool(ok.any())

    def parse_bamse(self, filename, strand=False, log=None):
        """te_count.py:167-277."""
        return self._bulk(filename, False, strand, log)

    def parse_bampe(self, filename, strand=False, log=None):
        """te_count.py:42-165."""
        return self._bulk(filename, True, strand, log)

    # ------------------------------------------------------------------ single cell
    BUNDLE_KEYS = 10000000  # the spill threshold literal at te_count.py:377 -- a semantic constant
    CELL_PAD = 1000         # te_count.py:502

    def sc_parse_bamse(self, filename, UMIS=True, whitelistfilename=None, strand=False, log=None, label=None,
                       maxcells=None):
        """te_count.py:298-707.  Returns {ensg: {barcode: count}}; sets self.barcodes (barcode ->
        number of UMIs that mapped to a feature, in the reference's insertion order) and
        self.total_reads.  The reference's temporary bundle files are not written: the (CB,UB)
        table lives in HBM."""
        assert filename, 'You must specify a filename'
        assert whitelistfilename, 'You must specify a whitelist of barcodes'
        assert label, 'You must specify a label'
        log = log or _NullLog()
        if not os.path.exists(whitelistfilename):
            raise AssertionError(f'{whitelistfilename} -w whitelist file not found')
        whitelist, packed_whitelist = _reads.load_whitelist(whitelistfilename)  # te_count.py:330-339
        filename = self._as_source(filename)
        self.barcodes = {}
        q = self.quality_threshold
        q_int = int(q) if isinstance(q, (int, np.integer)) else None
        log.info('Part 1: Collapsing UMI/CB combinations')
        self.load_genome()  # the reference loads it in part 3 (te_count.py:581); same index either way
        eng, world = self._engine, self.world
        from . import dist as _dist
        from . import scdist
        from .bam import byte_range
        from .gbam import GpuBam, is_bgzf
        # --noumi and a non-int -q end in the reference's crashes: they take the host decoder, which
        # looks at every batch on the way
        on_gpu = self._decoder_for(filename) == 'gpu' and UMIS and q_int is not None
        ranged = world.size > 1 and not hasattr(filename, 'batches') and is_bgzf(filename)
        rng = byte_range(os.path.getsize(filename), world.rank, world.size) if ranged else None
        n_records, failure, report = 0, None, None
        src = None
        try:
            src = GpuBam(eng, filename, rng) if on_gpu else self._open(filename, rng)
            skip, cls_of_tid, keys = _reads.chrom_class_table(src.reference_names)
            key_skip = np.zeros(len(keys), np.uint8)
            for t, c in enumerate(cls_of_tid):
                key_skip[c] = skip[t]
            key_chrom = np.array([self.genome._chrom_lookup.get(k, -1) for k in keys], np.int32)
            eng.sc_begin(strand, q_int or 0, packed_whitelist, self.BUNDLE_KEYS, key_chrom, key_skip)
            if on_gpu:
                report = src.run_sc(q_int, cls_of_tid)
                self.decode_report = report
                n_records = report['n_records']
            else:
                for ordinal, batch in enumerate(src.batches(want_tags=True)):
                    if world.size > 1 and not ranged and not world.mine(0):
                        continue  # a source that cannot be cut into ranges is read by rank 0 alone
                    n_records += len(batch)
                    if q_int is None and self._reaches_mapq_test(batch, False):
                        raise TypeError("'<' not supported between instances of 'int' and '%s'" % type(q).__name__)
                    arrays = _reads.sc_batch_arrays(batch, cls_of_tid)
                    if not UMIS:
                        # --noumi: the UB|UR tags are never looked at and nothing is ever stored
                        # (te_count.py:428-442); only the CB|CR assertion can still fire
                        arrays = arrays[:7] + (arrays[7] | 2,)
                    eng.sc_submit(*arrays)
                if ranged:
                    first, end, has = src.range_tokens()
                    report = {'n_records': n_records, 'has_records': has and n_records > 0, 'first_token': first, 'end_token': end}
        except Exception as e:  # noqa: BLE001 -- the other ranks must hear about it before the next collective
            if world.size == 1:
                raise
            failure = e
        finally:
            if src is not None:
                src.close()
        world.raise_together(failure)
        if ranged:
            _dist.check_ranges(world.gather({k: report[k] for k in ('n_records', 'has_records', 'first_token', 'end_token')}),
                               os.path.getsize(filename))
        if not UMIS:
            eng.sc_part1()  # (the CB|CR assertion)
            eng.sc_end()
            self.total_reads = _dist.world_sum(world, n_records) + 1
            raise ZeroDivisionError('division by zero')  # te_count.py:703: no UMI was ever stored
        log.info(f'Part 2: Get the best {maxcells} barcodes and remove dupes')
        out = scdist.run_parts(eng, world, n_records, self.BUNDLE_KEYS, maxcells, self.CELL_PAD)
        self.total_reads = out['total_records'] + 1  # te_count.py:373,705
        st = out['stats']
        self.stats = st
        log.info(f"  Observed {len(out['raw_ids']):,} raw barcodes")
        log.info(f"  Preserved {out['kept_lines']:,}/{self.total_reads:,} "
                 f"({out['kept_lines'] / self.total_reads * 100:.1f}%) of the reads")
        log.info('Part 3: Mapping the remaining UMIs to features')

        names = self.all_feature_names
        for b, h in zip(out['cells'].tolist(), out['cell_hits'].tolist()):
            self.barcodes[whitelist[b]] = h
        final_results = ScResult(names, whitelist, out['coo_cell'], out['coo_name'], out['coo_count'])
        self._barcode_id = {whitelist[b]: b for b in out['cells'].tolist()}
        log.info('  In the total pipeling, processed {:,} SE reads'.format(self.total_reads))
        log.info('  {:,} invalid barcode reads'.format(st['invalid_barcode']))
        log.info('  {:,} Read quality is too low (<{})'.format(st['lowq'], self.quality_threshold))
        log.info('  {:,} Reads QC failed'.format(st['qcfail']))
        log.info('  {:,} total valid reads'.format(out['kept_lines']))
        if out['kept_lines'] == 0:
            raise ZeroDivisionError('division by zero')  # te_count.py:703
        log.info('  Assigned {:,} ({:.1f}%) of total valid reads to features'.format(
            st['assigned'], st['assigned'] / out['kept_lines'] * 100.0))
        return final_results

    def sc_save_result(self, result, out_filename, maxcells=None, log=None):
        """te_count.py:709-754: `<out>.barcode_freq.tsv` and the dense cell x feature TSV."""
        assert out_filename, 'You must specify a filename'
        assert maxcells, 'You must specify maxcells'
        log = log or _NullLog()
        log.info('Densifying and saving "{0}"'.format(out_filename))
        log.info('Found {0:,} barcodes'.format(len(self.barcodes)))
        ranked = sorted(self.barcodes.items(), key=lambda kv: kv[1], reverse=True)  # stable, like itemgetter(1)
        cells = [b for b, _ in ranked]
        if len(self.barcodes) > maxcells:
            log.info(f'Keeping the best {maxcells:,} barcodes')
            cells = cells[0:maxcells]
        elif maxcells > len(self.barcodes):
            log.warning('Asked for {0:,} maxcells, but only {1:,} barcodes found'.format(maxcells, len(self.barcodes)))
        if '.tsv' not in out_filename:
            out_filename = f'{out_filename}.tsv'
        freq_filename = out_filename.replace('.tsv', '.barcode_freq.tsv')
        with open(freq_filename, 'w') as oh:
            for b in cells:
                oh.write('{0}\t{1}\n'.format(b, self.barcodes[b]))
        log.info('Saving barcode read frequency file to {0}'.format(freq_filename))
        features = list(result.keys())
        if isinstance(result, ScResult) and all(b in getattr(self, '_barcode_id', {}) for b in cells):
            # native writer over the GPU's triplets (same bytes; include/tecount.h tec_write_dense_tsv)
            header = '{}\t{}\n'.format('name', '\t'.join(features)).encode()
            labels = b''.join(b.encode() + b'\0' for b in cells)
            row_cell = np.array([self._barcode_id[b] for b in cells], np.uint32)
            lib = _lib.load()
            p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
            rc = lib.tec_write_dense_tsv(out_filename.encode(), header, len(header), len(features), len(cells), labels,
                                         p(row_cell), len(result.coo_cell), p(result.coo_cell), p(result.coo_name),
                                         p(result.coo_count))
            _lib.check(None, rc)
            return
        with open(out_filename, 'w') as oh:
            oh.write('{}\t{}\n'.format('name', '\t'.join(features)))
            for b in cells:
                row = [b]
                for f in features:
                    row.append(str(result[f].get(b, 0)))
                oh.write('{}\n'.format('\t'.join(row)))

    def save_result_bulk(self, result, out_filename, log=None):
        """te_count.py:279-296: `name<TAB>count<TAB>CPM`, CPM = count / (total_reads / 1e6)."""
        assert out_filename, 'You must specify a filename'
        total_reads = self.total_reads / 1e6
        with open(out_filename, 'w') as oh:
            for k in sorted(result.keys()):
                oh.write('{0}\t{1}\t{2}\n'.format(k, result[k], result[k] / total_reads))
        (log or _NullLog()).info('Saved {0}'.format(out_filename))
