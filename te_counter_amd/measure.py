"""`measureTE`: the reference's counting class (te_count/te_count.py:14-754) with the same
method names, arguments, return values and failure modes, running on the MI355X through
libtecount.so.

What stays on the host, as in the reference: opening the alignment file, `total_reads`, name
sorting and every byte of text formatting (TSV float repr must match Python's).
"""

import os

import numpy as np

from . import reads as _reads
from .engine import Engine
from .index import FlatIndex


class _NullLog:
    def info(self, *a, **k):
        pass

    error = warning = debug = info


class measureTE:
    def __init__(self, base_path, quality_threshold, device=0, batch_size=1 << 22, bin_shift=0):
        self.base_path = base_path
        self.total_reads = 0
        self.quality_threshold = quality_threshold
        self.device = device
        self.batch_size = batch_size
        self.bin_shift = bin_shift
        self.genome = None
        self.stats = {}
        self._engine = None

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
        self.genome = src if isinstance(src, FlatIndex) else FlatIndex.load(src)
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
            for batch in src.batches():
                n_records += len(batch)
                if q_int is None and self._reaches_mapq_test(batch, paired):
                    # `-q N` on the reference CLI hands over a list and the first record that
                    # passes the flag filter dies in `int(mapq) < [N]` (bin/te_count:30,77;
                    # te_count.py:88,208)
                    raise TypeError("'<' not supported between instances of 'int' and '%s'" % type(q).__name__)
                units = pairs.units(batch) if paired else _reads.se_units(batch, tid_map)
                if len(units[0]):
                    eng.bulk_submit(*units)
            counts, stats = eng.bulk_finish()
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

    def save_result_bulk(self, result, out_filename, log=None):
        """te_count.py:279-296: `name<TAB>count<TAB>CPM`, CPM = count / (total_reads / 1e6)."""
        assert out_filename, 'You must specify a filename'
        total_reads = self.total_reads / 1e6
        with open(out_filename, 'w') as oh:
            for k in sorted(result.keys()):
                oh.write('{0}\t{1}\t{2}\n'.format(k, result[k], result[k] / total_reads))
        (log or _NullLog()).info('Saved {0}'.format(out_filename))
