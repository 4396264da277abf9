"""TEST INFRASTRUCTURE: a numpy stand-in for te_counter_amd.engine.Engine's single-cell methods, so that
the multi-GPU orchestration (te_counter_amd/scdist.py: record ordinals, survivor exchange, distributed
bundle boundaries, global selection, presence-table drop rule, result merge) can run on CPU ranks under
gloo.  It restates what csrc/sc.hip computes per GPU, step for step, on numpy arrays; part 3's feature
lookup is the oracle's (oracle/te_oracle.py).  Checked against the reference's golden vectors with one
rank (tests/test_scdist_cpu.py) before it is trusted with several."""

import numpy as np

from oracle import te_oracle as orc
from te_counter_amd.index import read_chrom_key

NONE64 = 2 ** 64 - 1
COLS = ('bcid', 'u0', 'u1', 'u2', 'rec', 'klass', 'left', 'rite')


def pack_umi(s):
    b = s.encode()[:24].ljust(24, b'\0')
    return [int.from_bytes(b[8 * k:8 * k + 8], 'big') for k in range(3)]


def unpack_umi(u0, u1, u2):
    return b''.join(int(x).to_bytes(8, 'big') for x in (u0, u1, u2)).rstrip(b'\0').decode()


class NpScEngine:
    def __init__(self, features, whitelist_sorted, class_names, stranded, q, bundle_keys):
        self.features = features
        self.buckets = orc.build_buckets(features)
        self.names = orc.feature_names(features)
        self.name_id = {n: i for i, n in enumerate(self.names)}
        self.wl = whitelist_sorted
        self.wl_id = {b: i for i, b in enumerate(whitelist_sorted)}
        self._n_wl = len(whitelist_sorted)
        self.class_names = class_names  # read-side chromosome strings, the same table on every rank
        self.class_id = {c: i for i, c in enumerate(class_names)}
        self.stranded, self.q, self.bundle_keys = stranded, q, bundle_keys
        self.rows = {k: [] for k in COLS}
        self.stats = dict(records=0, qcfail=0, lowq=0, invalid_barcode=0, skipped_chrom=0, survivors=0, keys=0, bundles=0,
                          kept_lines=0, assigned=0)
        self.no_cb = self.no_ub = NONE64
        self.base = 0

    # ---- ingest: the filters of part 1 (te_count.py:394-433) in the reference's order
    def ingest(self, records):
        for r in records:
            i = self.stats['records']
            self.stats['records'] += 1
            if r['flag'] & 0x604:
                self.stats['qcfail'] += 1
                continue
            if int(r['mapq']) < self.q:
                self.stats['lowq'] += 1
                continue
            tags = r.get('tags', {})
            bc = tags.get('CB', tags.get('CR'))
            if bc is None:
                self.no_cb = min(self.no_cb, i)
                continue
            if bc not in self.wl_id:
                self.stats['invalid_barcode'] += 1
                continue
            ub = tags.get('UB', tags.get('UR'))
            if ub is None:
                self.no_ub = min(self.no_ub, i)
                continue
            chrom = read_chrom_key(r['rname'])
            if '_' in chrom or 'alt' in chrom:
                self.stats['skipped_chrom'] += 1
                continue
            u = pack_umi(ub)
            for k, v in zip(COLS, (self.wl_id[bc], u[0], u[1], u[2], i, (self.class_id[chrom] << 1) | (1 if (self.stranded and r['flag'] & 0x10) else 0),
                                   r['pos'], r['end'])):
                self.rows[k].append(v)

    def _arrays(self):
        dts = dict(bcid=np.uint32, u0=np.uint64, u1=np.uint64, u2=np.uint64, rec=np.uint64, klass=np.int32, left=np.int32, rite=np.int32)
        return {k: np.array(self.rows[k], dts[k]) for k in COLS}

    # ---- the Engine interface used by scdist.run_parts
    def sc_set_record_base(self, base, total):
        self.base = base
        self.a = self._arrays()
        self.a['rec'] = self.a['rec'] + np.uint64(base)

    def sc_survivors(self):
        return {k: v.copy() for k, v in self.a.items()}

    def sc_survivors_replace(self, rows):
        o = np.argsort(rows['rec'], kind='stable')
        self.a = {k: np.asarray(rows[k])[o] for k in COLS}

    def sc_part1_sort(self):
        a = self.a
        self.stats['survivors'] = len(a['rec'])
        o = np.lexsort((a['rec'], a['u2'], a['u1'], a['u0'], a['bcid']))
        self.s = {k: v[o] for k, v in a.items()}
        s = self.s
        n = len(o)
        self.ghead = np.ones(n, bool)
        if n > 1:
            self.ghead[1:] = (s['bcid'][1:] != s['bcid'][:-1]) | (s['u0'][1:] != s['u0'][:-1]) | (s['u1'][1:] != s['u1'][:-1]) | \
                             (s['u2'][1:] != s['u2'][:-1])
        self.gid = np.cumsum(self.ghead) - 1
        self.stats['keys'] = int(self.ghead.sum())
        return self.stats['keys']

    def sc_missing_tags(self):
        f = lambda v: NONE64 if v == NONE64 else v + self.base  # noqa: E731
        return f(self.no_cb), f(self.no_ub)

    def sc_long_umi(self):
        return NONE64  # (the stand-in keeps UMIs of any length)

    def sc_epoch_marks(self, lower):
        rec = self.s['rec'].astype(np.int64)
        prev = np.concatenate([[-1], rec[:-1]])
        m = (rec > lower) & (self.ghead | (prev <= lower))
        self.marks = np.sort(rec[m])
        return int(len(self.marks))

    def sc_epoch_rank(self, v):
        return int(np.searchsorted(self.marks, v, 'right'))

    def sc_epoch_kth(self, k):
        return int(self.marks[k - 1])

    def sc_part1_finish(self, bounds):
        s = self.s
        n = len(s['rec'])
        self.n_bundles = len(bounds) + 1
        self.epoch = np.searchsorted(np.array(bounds, np.uint64), s['rec'], 'left').astype(np.int64) if n else np.zeros(0, np.int64)
        self.ihead = self.ghead.copy()
        if n > 1:
            self.ihead[1:] |= self.epoch[1:] != self.epoch[:-1]
        self.headidx = np.maximum.accumulate(np.where(self.ihead, np.arange(n), 0)) if n else np.zeros(0, np.int64)
        self.counted = self.ihead | (s['klass'] != s['klass'][self.headidx]) if n else np.zeros(0, bool)
        ids = np.unique(s['bcid'][self.counted])
        cnt = np.array([int((self.counted & (s['bcid'] == b)).sum()) for b in ids], np.uint64)
        first = np.array([int(s['rec'][self.counted & (s['bcid'] == b)].min()) for b in ids], np.uint64)
        return ids.astype(np.uint32), cnt, first

    def _lines(self):
        idx = np.nonzero(self.ihead)[0]
        return idx[np.argsort(self.epoch[idx], kind='stable')]  # (bundle, barcode, UMI) order

    def sc_presence(self, n_bundles):
        pres = np.zeros((n_bundles, self._n_wl), np.uint8)
        for i in self._lines():
            pres[self.epoch[i], self.s['bcid'][i]] = 1
        return pres.reshape(-1)

    def sc_part23(self, selected, n_bundles=None, presence=None):
        s = self.s
        n = len(s['rec'])
        if presence is None:
            n_bundles = self.n_bundles
            presence = self.sc_presence(n_bundles)
        pres = np.asarray(presence, np.uint8).reshape(n_bundles, self._n_wl)
        sel = np.zeros(self._n_wl, bool)
        sel[np.asarray(selected, np.int64)] = True
        selprefix = np.concatenate([[0], np.cumsum(sel)[:-1]])
        lines = self._lines()
        kept = np.zeros(n, bool)
        prev = None
        for i in lines:
            b, e = int(s['bcid'][i]), int(self.epoch[i])
            k = False
            if sel[b]:
                if prev is not None and self.epoch[prev] == e and s['bcid'][prev] == b:
                    k = True
                else:
                    before = np.nonzero(pres[e, :b])[0]
                    k = len(before) > 0 and selprefix[b] == selprefix[before[-1]]
            kept[i] = k
            prev = i
        self.stats['kept_lines'] = int(kept.sum())
        firstkept = {}
        for i in np.nonzero(kept)[0]:
            firstkept.setdefault(int(self.gid[i]), int(i))
        # records that feed part 3
        cellhits, coo = {}, {}
        assigned = 0
        for i in range(n):
            h = int(self.headidx[i])
            if not (self.counted[i] and kept[h] and firstkept.get(int(self.gid[h])) == h):
                continue
            if not self.ihead[i]:
                x = i + 1
                over = False
                while x < n and not self.ihead[x]:
                    if self.counted[x] and s['klass'][x] == s['klass'][i]:
                        over = True
                        break
                    x += 1
                if over:
                    continue
            kl = int(s['klass'][i])
            chrom = self.class_names[kl >> 1]
            if chrom not in self.buckets:
                continue
            hits = orc._sc_hits(self.features, self.buckets, chrom, int(s['left'][i]), int(s['rite'][i]))
            if not hits:
                continue
            b = int(s['bcid'][i])
            cellhits[b] = cellhits.get(b, 0) + 1
            types = set(self.features[j]['type'] for j in hits)
            pairs = set((self.features[j]['ensg'], self.features[j]['strand']) for j in hits)
            rstrand = '-' if kl & 1 else '+'
            if types & set(orc.GENE_TYPES):
                assigned += 1
                for e_, st_ in pairs:
                    if self.stranded and rstrand != st_:
                        continue
                    coo[(b, self.name_id[e_])] = coo.get((b, self.name_id[e_]), 0) + 1
            elif 'TE' in types or 'enhancer' in types:
                assigned += 1
                for e_, st_ in pairs:
                    coo[(b, self.name_id[e_])] = coo.get((b, self.name_id[e_]), 0) + 1
        self.stats['assigned'] = assigned
        cells = sorted(cellhits)
        keys = sorted(coo)
        return {'kept_lines': self.stats['kept_lines'], 'cells': np.array(cells, np.uint32),
                'cell_hits': np.array([cellhits[c] for c in cells], np.uint64),
                'coo_cell': np.array([k[0] for k in keys], np.uint32), 'coo_name': np.array([k[1] for k in keys], np.uint32),
                'coo_count': np.array([coo[k] for k in keys], np.uint64)}

    def sc_stats(self):
        st = dict(self.stats)
        st['bundles'] = self.n_bundles if st['survivors'] else 0
        return st

    def sc_end(self):
        pass


def result_views(out, names, whitelist):
    """(result dict {ensg: {barcode: count}} with every name, barcodes as ordered (barcode, hits) pairs) from run_parts' output."""
    result = {n: {} for n in names}
    for c, nm, v in zip(out['coo_cell'].tolist(), out['coo_name'].tolist(), out['coo_count'].tolist()):
        result[names[nm]][whitelist[c]] = v
    return result, [[whitelist[c], int(h)] for c, h in zip(out['cells'].tolist(), out['cell_hits'].tolist())]
