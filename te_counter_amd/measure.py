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
        """te_count.py:31-35: loads the index and derives all_feature_names."""
        assert getattr(self, 'genelist_glb_filename', None) is not None, 'You need to bind the genome first'
        src = self.genelist_glb_filename
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

    # ------------------------------------------------------------------ bulk
    def _open(self, filename):
        if hasattr(filename, 'batches'):
            return filename
        from .bam import open_alignment
        return open_alignment(filename, batch_size=self.batch_size)

    def _bulk(self, filename, paired, strand, log):
        assert filename, 'You must specify a filename'
        if strand:
            raise NotImplementedError()  # te_count.py:58-59, 183-184
        log = log or _NullLog()
        assert self.genome is not None, 'call load_genome() first'
        src = self._open(filename)
        tid_map = self.genome.tid_map(src.reference_names)
        eng = self._engine
        q = self.quality_threshold
        q_int = int(q) if isinstance(q, (int, np.integer)) else None
        n_records = 0
        pairs = _reads.PairAssembler(tid_map, q_int or 0) if paired else None
        try:
            eng.bulk_begin(q_int or 0)
            for ordinal, batch in enumerate(src.batches()):
                n_records += len(batch)
                if q_int is None and self._reaches_mapq_test(batch, paired):
                    # `-q N` on the reference CLI hands over a list and the first record that
                    # passes the flag filter dies in `int(mapq) < [N]` (bin/te_count:30,77;
                    # te_count.py:88,208)
                    raise TypeError("'<' not supported between instances of 'int' and '%s'" % type(q).__name__)
                units = pairs.units(batch) if paired else _reads.se_units(batch, tid_map)
                if not paired and (units[2] == NO_END).any():
                    self._check_reference_end(units, q_int or 0)
                if len(units[0]) and self.world.mine(ordinal):
                    eng.bulk_submit(*units)
            counts, stats = eng.bulk_finish()
            if self.world.size > 1:  # RCCL sum-reduce of the per-GPU count vectors
                keys = ('units', 'assigned', 'lowq', 'badchrom', 'qcfail')
                counts, sv = self.world.sum_counts(counts, [stats[k] for k in keys])
                stats = {k: int(v) for k, v in zip(keys, sv)}
        finally:
            src.close()
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

    @staticmethod
    def _check_reference_end(units, q):
        # a mapped SE record without CIGAR has reference_end None: `loc2+1` dies (te_count.py:223)
        chrom, _, p2, flag, mapq = units
        bad = (p2 == NO_END) & ((flag & _reads.FLAG_FILTER) == 0) & (mapq >= q) & (chrom >= 0)
        if bad.any():
            raise TypeError("unsupported operand type(s) for +: 'NoneType' and 'int'")

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
        with open(whitelistfilename, 'r') as fh:
            whitelist = sorted(set(line.strip() for line in fh))  # te_count.py:330-339
        self.barcodes = {}
        src = self._open(filename)
        q = self.quality_threshold
        q_int = int(q) if isinstance(q, (int, np.integer)) else None
        log.info('Part 1: Collapsing UMI/CB combinations')
        self.load_genome()  # the reference loads it in part 3 (te_count.py:581); same index either way
        eng = self._engine
        skip, cls_of_tid, keys = _reads.chrom_class_table(src.reference_names)
        key_skip = np.zeros(len(keys), np.uint8)
        for t, c in enumerate(cls_of_tid):
            key_skip[c] = skip[t]
        key_chrom = np.array([self.genome._chrom_lookup.get(k, -1) for k in keys], np.int32)
        n_records = 0
        try:
            eng.sc_begin(strand, q_int or 0, _reads.pack_whitelist(whitelist), self.BUNDLE_KEYS, key_chrom, key_skip)
            for batch in src.batches(want_tags=True):
                n_records += len(batch)
                if q_int is None and self._reaches_mapq_test(batch, False):
                    raise TypeError("'<' not supported between instances of 'int' and '%s'" % type(q).__name__)
                arrays = _reads.sc_batch_arrays(batch, cls_of_tid)
                if not UMIS:
                    # --noumi: the UB|UR tags are never looked at and nothing is ever stored
                    # (te_count.py:428-442); only the CB|CR assertion can still fire
                    arrays = arrays[:7] + (arrays[7] | 2,)
                eng.sc_submit(*arrays)
            ids, counts, first = eng.sc_part1()
        finally:
            src.close()
        self.total_reads = n_records + 1  # te_count.py:373,705
        if not UMIS:
            eng.sc_end()
            raise ZeroDivisionError('division by zero')  # te_count.py:703: no UMI was ever stored

        # ---- part 2 selection (te_count.py:502): best maxcells+1000 raw counts, ties first-seen first
        log.info(f'Part 2: Get the best {maxcells} barcodes and remove dupes')
        log.info(f'  Observed {len(ids):,} raw barcodes')
        order = np.lexsort((first, -counts.astype(np.int64)))
        selected = np.sort(ids[order[0:maxcells + self.CELL_PAD]])
        out = eng.sc_part23(selected)
        st = eng.sc_stats()
        eng.sc_end()
        self.stats = st
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
