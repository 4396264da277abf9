"""Host side of the annotation index: feature table -> flat structure-of-arrays.

Replaces what the reference keeps as a pickled `genelist` (te_count/miniglbase/genelist.py,
loaded by te_count/te_count.py:31-35): `linearData` (a list of feature dicts), `buckets` and
`genome['ensg']`.  Here the index is five flat arrays plus three small string tables; the
10 kb bucket lists are never materialised (libtecount evaluates the bucket-candidate rule in
closed form, see csrc/bulk.hip).

Own file format (`.tecidx`):  magic, JSON header, 64-byte aligned raw little-endian arrays.
"""

import io
import json
import os
import pickle

import numpy as np

from . import _lib

MAGIC = b'TECIDX01'
GENE_TYPES = ('protein_coding', 'lincRNA', 'lncRNA')  # te_count.py:134, 247, 662
NAME_MASK = (1 << _lib.META_NAME_BITS) - 1


def index_chrom_key(chrom):
    """Index-side chromosome key: te_count/miniglbase/location.py:31."""
    return str(chrom).strip('chr').rstrip().upper()


def read_chrom_key(reference_name):
    """Read-side chromosome key: te_count/te_count.py:96, 212, 431 (case is kept)."""
    return reference_name.replace('chr', '')


def type_class(type_string):
    if type_string in GENE_TYPES:
        return _lib.CLASS_GENE
    if type_string == 'TE':
        return _lib.CLASS_TE
    if type_string == 'snRNA':
        return _lib.CLASS_SNRNA
    if type_string == 'enhancer':
        return _lib.CLASS_ENHANCER
    return _lib.CLASS_OTHER


class FlatIndex:
    """names: sorted distinct `ensg` values (= all_feature_names, te_count.py:35);
    chrom_keys: index-side chromosome keys; per feature: chrom_id, left, right,
    meta = name id | class << 26 | strand id << 29."""

    def __init__(self, names, chrom_keys, strands, chrom_id, left, right, meta, has_strand=True):
        self.names = list(names)
        self.chrom_keys = list(chrom_keys)
        self.strands = list(strands)
        self.chrom_id = np.ascontiguousarray(chrom_id, dtype=np.int32)
        self.left = np.ascontiguousarray(left, dtype=np.int32)
        self.right = np.ascontiguousarray(right, dtype=np.int32)
        self.meta = np.ascontiguousarray(meta, dtype=np.uint32)
        self.has_strand = bool(has_strand)  # False: some feature lacks 'strand' (sc -> KeyError)
        n = len(self.chrom_id)
        if not (len(self.left) == len(self.right) == len(self.meta) == n):
            raise ValueError('feature arrays differ in length')
        if len(self.names) > (1 << _lib.META_NAME_BITS):
            raise ValueError('too many feature names')
        if len(self.strands) > 7:
            raise ValueError('more than 7 distinct strand strings')
        self._chrom_lookup = {k: i for i, k in enumerate(self.chrom_keys)}

    def __len__(self):
        return len(self.chrom_id)

    @property
    def n_names(self):
        return len(self.names)

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_features(cls, features):
        """features: iterable of dicts {chr,left,right,name,type,ensg[,strand]} -- the rows the
        reference's builders hand to genelist.load_list (te_count/genome/make.py:65-71, 106-112)."""
        features = list(features)
        names = sorted(set(f['ensg'] for f in features))
        name_id = {n: i for i, n in enumerate(names)}
        chrom_keys, chrom_lookup = [], {}
        strands, strand_lookup = ['+', '-'], {'+': 0, '-': 1}
        has_strand = True
        n = len(features)
        chrom_id = np.empty(n, np.int32)
        left = np.empty(n, np.int64)
        right = np.empty(n, np.int64)
        meta = np.empty(n, np.uint32)
        for i, f in enumerate(features):
            key = index_chrom_key(f['chr'])
            c = chrom_lookup.get(key)
            if c is None:
                c = chrom_lookup[key] = len(chrom_keys)
                chrom_keys.append(key)
            if 'strand' in f:
                s = strand_lookup.get(f['strand'])
                if s is None:
                    s = strand_lookup[f['strand']] = len(strands)
                    strands.append(f['strand'])
                    if s > 6:
                        raise ValueError('more than 7 distinct strand strings')
            else:
                has_strand = False
                s = _lib.STRAND_MISSING  # --sc: a hit on such a feature is a KeyError (te_count.py:661)
            chrom_id[i] = c
            left[i] = int(f['left'])  # location.py:31 stores int(left), int(right)
            right[i] = int(f['right'])
            meta[i] = name_id[f['ensg']] | (type_class(f['type']) << _lib.META_CLASS_SHIFT) | (s << _lib.META_STRAND_SHIFT)
        if n and (left.min() < 0 or max(left.max(), right.max()) >= 2 ** 31 - 2 or right.min() < -2 ** 31 + 2):
            raise ValueError('feature coordinates outside the supported int32 range [0, 2^31-2)')
        return cls(names, chrom_keys, strands, chrom_id, left, right, meta, has_strand)

    @classmethod
    def from_glb(cls, path):
        """Import an index built by the reference's te_genome: a `.glb` is a pickle of the whole
        genelist object (te_count/miniglbase/base_genelist.py:302-305, utils.py:21-59).  Only
        `linearData` is read, through a restricted unpickler that maps the reference's classes
        (te_count.miniglbase.genelist.genelist, ...location.location) to inert stand-ins -- no
        reference code is imported or executed."""
        with open(path, 'rb') as fh:
            obj = _GlbUnpickler(io.BytesIO(fh.read())).load()
        rows = getattr(obj, 'linearData', None)
        if rows is None:
            raise ValueError('%s holds no linearData' % path)
        feats = []
        for r in rows:
            loc = r['loc']
            d = loc.loc if hasattr(loc, 'loc') else loc
            f = {'chr': d['chr'], 'left': d['left'], 'right': d['right'], 'name': r.get('name'), 'type': r['type'],
                 'ensg': r['ensg']}
            if 'strand' in r:
                f['strand'] = r['strand']
            feats.append(f)
        return cls.from_features(feats)

    def strand_split(self):
        """The index of the stranded-bulk EXTENSION (tec_bulk_set_stranded; the reference refuses bulk `--strand`,
        te_count.py:58-59): chromosome 2c holds the '+' features of chromosome c, 2c + 1 its '-' features; features
        of any other (or no) strand can match no read strand (te_count.py:654: '-' if read.is_reverse else '+')
        and are left out.  Names and name ids are unchanged; reads keep the plain chromosome id c (`tid_map` of
        the unsplit index)."""
        sid = (self.meta >> _lib.META_STRAND_SHIFT) & 7
        plus = self.strands.index('+') if '+' in self.strands else -1
        minus = self.strands.index('-') if '-' in self.strands else -1
        keep = (sid == plus) | (sid == minus)
        keys = [k + s for k in self.chrom_keys for s in ('\x00+', '\x00-')]  # (never equal to a read's chromosome key)
        chrom = 2 * self.chrom_id[keep] + (sid[keep] == minus)
        return FlatIndex(self.names, keys, self.strands, chrom, self.left[keep], self.right[keep], self.meta[keep], True)

    # ------------------------------------------------------------------ queries the host needs
    def chrom_id_of_read(self, reference_name):
        """-1 unless reference_name.replace('chr','') is an index key (te_count.py:96-102)."""
        return self._chrom_lookup.get(read_chrom_key(reference_name), -1)

    def tid_map(self, reference_names):
        """BAM header tid -> index chrom id table (int32, -1 = 'invalid chromosome')."""
        return np.array([self.chrom_id_of_read(n) for n in reference_names], dtype=np.int32).reshape(-1)

    # ------------------------------------------------------------------ file format
    def save(self, path):
        names_blob = '\n'.join(self.names).encode('utf-8')
        arrays = [('chrom_id', self.chrom_id), ('left', self.left), ('right', self.right), ('meta', self.meta)]
        header = {'n_features': len(self), 'n_names': len(self.names), 'chrom_keys': self.chrom_keys,
                  'strands': self.strands, 'has_strand': self.has_strand, 'names_bytes': len(names_blob),
                  'bucket_size': 10000}
        hdr = json.dumps(header).encode('utf-8')
        with open(path, 'wb') as fh:
            fh.write(MAGIC)
            fh.write(np.uint64(len(hdr)).tobytes())
            fh.write(hdr)
            for _, a in arrays:
                fh.write(b'\0' * (-fh.tell() % 64))
                fh.write(a.tobytes())
            fh.write(b'\0' * (-fh.tell() % 64))
            fh.write(names_blob)

    @classmethod
    def load(cls, path):
        with open(path, 'rb') as fh:
            if fh.read(8) != MAGIC:
                raise ValueError('%s is not a .tecidx index' % path)
            hlen = int(np.frombuffer(fh.read(8), np.uint64)[0])
            header = json.loads(fh.read(hlen).decode('utf-8'))
            n = header['n_features']
            out = []
            for dt in (np.int32, np.int32, np.int32, np.uint32):
                fh.seek(-fh.tell() % 64, os.SEEK_CUR)
                out.append(np.frombuffer(fh.read(4 * n), dtype=dt))
            fh.seek(-fh.tell() % 64, os.SEEK_CUR)
            blob = fh.read(header['names_bytes']).decode('utf-8')
        names = blob.split('\n') if header['n_names'] else []
        return cls(names, header['chrom_keys'], header['strands'], out[0], out[1], out[2], out[3],
                   header.get('has_strand', True))


class _GlbStub:
    """Inert stand-in for any class of the reference's miniglbase package found in a .glb."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):
            self.__dict__.update(state[1] or {})
            if isinstance(state[0], dict):
                self.__dict__.update(state[0])


class _GlbUnpickler(pickle.Unpickler):
    _SAFE = {('builtins', 'set'), ('builtins', 'frozenset'), ('builtins', 'list'), ('builtins', 'dict'),
             ('builtins', 'tuple'), ('builtins', 'object'), ('collections', 'OrderedDict'),
             ('collections', 'defaultdict'), ('copyreg', '_reconstructor')}

    def find_class(self, module, name):
        if module.startswith('te_count.miniglbase') or module.startswith('glbase3') or module.startswith('miniglbase'):
            return _GlbStub
        if (module, name) in self._SAFE:
            import importlib
            return getattr(importlib.import_module(module), name)
        if module == 'logging' or module.startswith('numpy'):
            return _GlbStub
        raise pickle.UnpicklingError('refusing to load %s.%s from a .glb' % (module, name))
