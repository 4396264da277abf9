"""
TEST INFRASTRUCTURE ONLY -- never imported by the product path (te_counter_amd/, bin/).

Drives the *real* reference (oaxiom/te_counter, mounted read-only at /root/reference)
on in-memory inputs, so that golden vectors can be captured from it.  It only works in
a container that has /root/reference; it is never run on the GPU box and nothing from
the reference is copied into this repository.

How: the reference's only hot-path dependency that is absent
here is pysam (te_count/te_count.py:11).  A stand-in module object named `pysam` is
registered in sys.modules before `import te_count`; it exposes AlignmentFile(filename,
mode) with __next__/close and yields records carrying the ten attributes the reference
touches (te_count.py:81-98, 204-214, 394-438).

Record schema used everywhere in this repo's tests (one dict per BAM record):
    {"flag": int, "mapq": int, "qname": str, "rname": str, "pos": int, "end": int,
     "tags": {"CB": "...", "UB": "..."} (optional)}
Feature schema: {"chr": str, "left": int, "right": int, "strand": str (optional),
                 "name": str, "type": str, "ensg": str}
"""

import io
import logging
import os
import sys
import tempfile
import types

REFERENCE_ROOT = '/root/reference'


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'te_count'))


class _Rec:
    __slots__ = ('is_unmapped', 'is_duplicate', 'is_qcfail', 'is_reverse', 'mapping_quality',
                 'query_name', 'reference_name', 'reference_start', 'reference_end', '_tags')

    def __init__(self, r):
        f = r['flag']
        self.is_unmapped = bool(f & 0x4)
        self.is_duplicate = bool(f & 0x400)
        self.is_qcfail = bool(f & 0x200)
        self.is_reverse = bool(f & 0x10)
        self.mapping_quality = r['mapq']
        self.query_name = r.get('qname', 'q')
        self.reference_name = r['rname']
        self.reference_start = r['pos']
        self.reference_end = r['end']
        self._tags = list(r.get('tags', {}).items())

    def get_tags(self):
        return self._tags


class _FakeAlignmentFile:
    """Stand-in for pysam.AlignmentFile; `records` is injected per run."""
    records = ()

    def __init__(self, filename, mode='r'):
        self._it = iter(type(self).records)

    def __iter__(self):
        return self

    def __next__(self):
        return _Rec(next(self._it))

    def close(self):
        pass


_te_count = None


def load_reference():
    """Import the reference package once, with the stand-in pysam in place."""
    global _te_count
    if _te_count is not None:
        return _te_count
    if not reference_available():
        raise RuntimeError('reference not mounted at %s' % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    mod = types.ModuleType('pysam')
    mod.AlignmentFile = _FakeAlignmentFile
    sys.modules['pysam'] = mod
    import te_count  # noqa: E402  (the reference)
    logging.getLogger('glbase3').setLevel(logging.ERROR)
    _te_count = te_count
    return te_count


def _quiet_log():
    log = logging.getLogger('ref_harness')
    log.handlers[:] = [logging.NullHandler()]
    log.propagate = False
    return log


def build_reference_index(features, path):
    """features -> .glb through the reference's own genelist.load_list/_optimiseData/save."""
    tc = load_reference()
    mg = tc.miniglbase
    newl = []
    for f in features:
        e = {'loc': mg.location(chr=f['chr'], left=f['left'], right=f['right']),
             'name': f['name'], 'type': f['type'], 'ensg': f['ensg']}
        if 'strand' in f:
            e['strand'] = f['strand']
        newl.append(e)
    gl = mg.genelist()
    gl.load_list(newl)
    gl.save(path)
    return gl


def run_bulk(features, records, paired, quality_threshold=20, strand=False):
    """Returns dict(result, total_reads, tsv) from measureTE.parse_bamse/parse_bampe +
    save_result_bulk (te_count.py:42-296)."""
    tc = load_reference()
    with tempfile.TemporaryDirectory() as td:
        glb = os.path.join(td, 'idx.glb')
        build_reference_index(features, glb)
        m = tc.measureTE('harness', quality_threshold)
        m.bind_genome(glb)
        m.load_genome()
        _FakeAlignmentFile.records = records
        log = _quiet_log()
        if paired:
            res = m.parse_bampe('x.bam', strand=strand, log=log)
        else:
            res = m.parse_bamse('x.bam', strand=strand, log=log)
        out = os.path.join(td, 'out.tsv')
        m.save_result_bulk(res, out, log=log)
        with open(out, 'rb') as fh:
            tsv = fh.read().decode()
    return {'result': res, 'total_reads': m.total_reads, 'tsv': tsv}


def _patch_bundle_threshold(func, new_threshold):
    """Return a copy of sc_parse_bamse whose literal 1e7 spill threshold
    (te_count.py:377, `len(umis) >= 1e7`) is replaced, so that the multi-bundle rules can
    be pinned with small fixtures.  The progress-log modulus at :374 shares the constant;
    it only affects logging.  The reference's code object is re-used as is, nothing is
    copied."""
    code = func.__code__
    consts = tuple(float(new_threshold) if (isinstance(c, float) and c == 1e7) else c
                   for c in code.co_consts)
    assert consts != code.co_consts, 'threshold literal not found'
    return types.FunctionType(code.replace(co_consts=consts), func.__globals__,
                              func.__name__, func.__defaults__, func.__closure__)


def run_sc(features, records, whitelist, maxcells, quality_threshold=20, strand=False,
           umis=True, label='sc_out', bundle_threshold=None):
    """Returns dict(result, barcodes (ordered pairs), total_reads, tsv, barcode_freq) from
    measureTE.sc_parse_bamse + sc_save_result (te_count.py:298-754)."""
    tc = load_reference()
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)  # spill files land in CWD (te_count.py:381)
        try:
            glb = os.path.join(td, 'idx.glb')
            build_reference_index(features, glb)
            wl = os.path.join(td, 'whitelist.txt')
            with open(wl, 'w') as fh:
                for b in whitelist:
                    fh.write(b + '\n')
            m = tc.measureTE('harness', quality_threshold)
            m.bind_genome(glb)
            _FakeAlignmentFile.records = records
            log = _quiet_log()
            fn = tc.measureTE.sc_parse_bamse
            if bundle_threshold is not None:
                fn = _patch_bundle_threshold(fn, bundle_threshold)
            res = fn(m, 'x.bam', UMIS=umis, whitelistfilename=wl, strand=strand, log=log,
                     label=label, maxcells=maxcells)
            out = os.path.join(td, label + '.tsv')
            m.sc_save_result(res, out, maxcells=maxcells, log=log)
            with open(out, 'rb') as fh:
                tsv = fh.read().decode()
            with open(out.replace('.tsv', '.barcode_freq.tsv'), 'rb') as fh:
                freq = fh.read().decode()
        finally:
            os.chdir(cwd)
    return {'result': res, 'barcodes': list(m.barcodes.items()), 'total_reads': m.total_reads,
            'tsv': tsv, 'barcode_freq': freq}


def run_make_custom(gtf_path):
    """The reference's te_genome custom builder (te_count/genome/make.py:173-226) on a GTF file;
    returns the feature rows of the index it writes."""
    tc = load_reference()
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, 'custom.glb')
        stdout, stderr = sys.stdout, sys.stderr
        sys.stdout = io.StringIO()  # the builder prints progress bars
        sys.stderr = io.StringIO()
        try:
            tc.genome.make_custom(gtf_path, out, _quiet_log())
        finally:
            sys.stdout, sys.stderr = stdout, stderr
        gl = tc.miniglbase.glload(out)
        rows = []
        for r in gl.linearData:
            rows.append({'chr': r['loc']['chr'], 'left': r['loc']['left'], 'right': r['loc']['right'],
                         'strand': r['strand'], 'name': r['name'], 'type': r['type'], 'ensg': r['ensg']})
    return rows


def run_custom_end_to_end(gtf_path, records, paired, quality_threshold=20):
    """What a user of the reference does for a custom annotation, in one go: te_genome -m custom --gtf F
    (te_count/genome/make.py:173-226) writes the .glb, te_count -g that.glb -m custom binds and loads it
    (bin/te_count:80-84,112) and counts (te_count.py:42-296).  Returns dict(result, total_reads, tsv)."""
    tc = load_reference()
    with tempfile.TemporaryDirectory() as td:
        glb = os.path.join(td, 'custom.glb')
        stdout, stderr = sys.stdout, sys.stderr
        sys.stdout = io.StringIO()  # the builder prints progress bars
        sys.stderr = io.StringIO()
        try:
            tc.genome.make_custom(gtf_path, glb, _quiet_log())
        finally:
            sys.stdout, sys.stderr = stdout, stderr
        m = tc.measureTE('harness', quality_threshold)
        m.bind_genome(glb)
        m.load_genome()
        _FakeAlignmentFile.records = records
        log = _quiet_log()
        res = (m.parse_bampe if paired else m.parse_bamse)('x.bam', strand=False, log=log)
        out = os.path.join(td, 'out.tsv')
        m.save_result_bulk(res, out, log=log)
        with open(out, 'rb') as fh:
            tsv = fh.read().decode()
    return {'result': res, 'total_reads': m.total_reads, 'tsv': tsv}


def run_bulk_on_glb(glb_path, records, paired, quality_threshold=20):
    """te_count -g FILE.glb -m custom on an index file that already exists (bin/te_count:80-84)."""
    tc = load_reference()
    with tempfile.TemporaryDirectory() as td:
        m = tc.measureTE('harness', quality_threshold)
        m.bind_genome(glb_path)
        m.load_genome()
        _FakeAlignmentFile.records = records
        log = _quiet_log()
        res = (m.parse_bampe if paired else m.parse_bamse)('x.bam', strand=False, log=log)
        out = os.path.join(td, 'out.tsv')
        m.save_result_bulk(res, out, log=log)
        with open(out, 'rb') as fh:
            tsv = fh.read().decode()
    return {'result': res, 'total_reads': m.total_reads, 'tsv': tsv}


def run_make_snrnps(genome, gencode_gz):
    """The reference's make_snrnps (make.py:228-288) on a local GENCODE file (same os proxy as
    run_make_genes_tes)."""
    import shutil
    tc = load_reference()
    mk = sys.modules['te_count.genome.make']
    names = {'mm10': 'gencode.vM23.annotation.gtf.gz', 'hg38': 'gencode.v42.annotation.gtf.gz'}
    with tempfile.TemporaryDirectory() as td:
        shutil.copy(gencode_gz, os.path.join(td, names[genome]))
        proxy = types.SimpleNamespace(system=lambda cmd: 0,
                                      path=types.SimpleNamespace(dirname=lambda p: td, realpath=os.path.realpath))
        real_os, stdout, stderr = mk.os, sys.stdout, sys.stderr
        mk.os = proxy
        sys.stdout, sys.stderr = io.StringIO(), io.StringIO()
        try:
            mk.make_snrnps(genome, _quiet_log())
        finally:
            mk.os = real_os
            sys.stdout, sys.stderr = stdout, stderr
        gl = tc.miniglbase.glload(os.path.join(td, f'{genome}_snrnps.glb'))
        return [{'chr': r['loc']['chr'], 'left': r['loc']['left'], 'right': r['loc']['right'], 'strand': r['strand'],
                 'name': r['name'], 'type': r['type'], 'ensg': r['ensg']} for r in gl.linearData]


def run_make_enh(genome, bed_gz):
    """The reference's make_enh (te_count/genome/make.py:129-171) on a LOCAL copy of the FANTOM5
    enhancer BED it would download (same os proxy as run_make_genes_tes; make_enh also joins paths)."""
    import shutil
    tc = load_reference()
    mk = sys.modules['te_count.genome.make']
    with tempfile.TemporaryDirectory() as td:
        shutil.copy(bed_gz, os.path.join(td, f'enh_{genome}.bed.gz'))
        proxy = types.SimpleNamespace(system=lambda cmd: 0,
                                      path=types.SimpleNamespace(dirname=lambda p: td, realpath=os.path.realpath, join=os.path.join))
        real_os, stdout, stderr = mk.os, sys.stdout, sys.stderr
        mk.os = proxy
        sys.stdout, sys.stderr = io.StringIO(), io.StringIO()
        try:
            mk.make_enh(genome, _quiet_log())
        finally:
            mk.os = real_os
            sys.stdout, sys.stderr = stdout, stderr
        gl = tc.miniglbase.glload(os.path.join(td, f'{genome}_enhancers.glb'))
        return [{'chr': r['loc']['chr'], 'left': r['loc']['left'], 'right': r['loc']['right'],
                 'name': r['name'], 'type': r['type'], 'ensg': r['ensg']} for r in gl.linearData]


def run_make_genes_tes(genome, gencode_gz, rmsk_gz):
    """The reference's make_genes_tes (te_count/genome/make.py:15-127) on LOCAL copies of the two
    files it would download: its module-level `os` is swapped for a proxy whose system() does
    nothing and whose path.dirname() points at a scratch directory holding the files under the
    names the function expects.  Returns the feature rows of the index it writes."""
    import shutil
    tc = load_reference()
    mk = sys.modules['te_count.genome.make']
    names = {'mm10': 'gencode.vM23.annotation.gtf.gz', 'hg38': 'gencode.v42.annotation.gtf.gz',
             'macFas5': 'Macaca_fascicularis.Macaca_fascicularis_6.0.115.gtf.gz'}
    with tempfile.TemporaryDirectory() as td:
        shutil.copy(gencode_gz, os.path.join(td, names[genome]))
        shutil.copy(rmsk_gz, os.path.join(td, f'{genome}_rmsk.txt.gz'))
        proxy = types.SimpleNamespace(system=lambda cmd: 0,
                                      path=types.SimpleNamespace(dirname=lambda p: td, realpath=os.path.realpath))
        real_os, stdout, stderr = mk.os, sys.stdout, sys.stderr
        mk.os = proxy
        sys.stdout = io.StringIO()
        sys.stderr = io.StringIO()  # progress bars
        try:
            mk.make_genes_tes(genome, _quiet_log())
        finally:
            mk.os = real_os
            sys.stdout, sys.stderr = stdout, stderr
        gl = tc.miniglbase.glload(os.path.join(td, f'{genome}_genes_tes.glb'))
        return [{'chr': r['loc']['chr'], 'left': r['loc']['left'], 'right': r['loc']['right'], 'strand': r['strand'],
                 'name': r['name'], 'type': r['type'], 'ensg': r['ensg']} for r in gl.linearData]
