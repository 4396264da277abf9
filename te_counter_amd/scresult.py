"""The value sc_parse_bamse returns: `{ensg: {barcode: count}}` (te_count/te_count.py:583,707) as a
read-only mapping over the COO triplets that come back from the GPU.  Inner dicts are built on
demand, so a 1e4-cell x 4e4-feature result does not cost 1e7 Python dict insertions up front, and
sc_save_result can hand the triplets straight to the native TSV writer."""

from collections.abc import Mapping

import numpy as np


class ScResult(Mapping):
    def __init__(self, names, barcodes, coo_cell, coo_name, coo_count):
        self.names = list(names)            # every feature name, sorted (all_feature_names)
        self.barcodes = barcodes            # whitelist id -> barcode string
        self.coo_cell = np.ascontiguousarray(coo_cell, np.uint32)    # sorted by (cell, name)
        self.coo_name = np.ascontiguousarray(coo_name, np.uint32)
        self.coo_count = np.ascontiguousarray(coo_count, np.uint64)
        self._index = None      # name -> row, and the triplets grouped by name: built when the mapping is first
        self._by_name = None    # looked into (sc_save_result's native writer takes the triplets as they are)
        self._starts = None
        self._cache = {}

    def _group(self):
        if self._index is None:
            self._index = {n: i for i, n in enumerate(self.names)}
        if self._by_name is None:
            n = len(self.names)
            if n <= 1 << 16:   # (numpy's stable sort of 16-bit keys is a radix sort: 10x faster than of 32-bit keys)
                order = np.argsort(self.coo_name.astype(np.uint16), kind='stable')
            else:
                order = np.argsort(self.coo_name, kind='stable')     # by name, cells ascending inside
            self._starts = np.searchsorted(self.coo_name[order], np.arange(n + 1))
            self._by_name = order

    def __len__(self):
        return len(self.names)

    def __iter__(self):
        return iter(self.names)

    def __contains__(self, key):
        if self._index is None:
            self._index = {n: i for i, n in enumerate(self.names)}
        return key in self._index

    def __getitem__(self, key):
        self._group()
        i = self._index[key]
        d = self._cache.get(i)
        if d is None:
            sel = self._by_name[self._starts[i]:self._starts[i + 1]]
            bc = self.barcodes
            d = {bc[c]: v for c, v in zip(self.coo_cell[sel].tolist(), self.coo_count[sel].tolist())}
            self._cache[i] = d
        return d

    def __eq__(self, other):
        if isinstance(other, Mapping):
            return len(self) == len(other) and all(k in other and self[k] == other[k] for k in self)
        return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None
