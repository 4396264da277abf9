"""Seeded synthetic annotations and read batches of the shapes BASELINE.json names.

Real hg38/mm10 `genes_tes` indices need network downloads (te_count/genome/make.py:41-43), so the
benchmark and the full-size tests use annotations with the same statistics, as specified in
SURVEY.md section 8(d): UCSC primary-assembly chromosome lengths, ~4.5 M (hg38) / 3.7 M (mm10)
TE intervals with Zipf-distributed family names, ~1.2 M exon rows grouped into 40 k genes.
Everything is numpy-vectorised and deterministic in (seed, chunk).
"""

import numpy as np

from . import _lib
from .index import FlatIndex

HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
        133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285,
        58617616, 64444167, 46709983, 50818468, 156040895, 57227415, 16569]
HG38_NAMES = ['chr%d' % i for i in range(1, 23)] + ['chrX', 'chrY', 'chrM']
MM10 = [195471971, 182113224, 160039680, 156508116, 151834684, 149736546, 145441459, 129401213, 124595110,
        130694993, 122082543, 120129022, 120421639, 124902244, 104043685, 98207768, 94987271, 90702639,
        61431566, 171031299, 91744698, 16299]
MM10_NAMES = ['chr%d' % i for i in range(1, 20)] + ['chrX', 'chrY', 'chrM']

GENOMES = {'hg38': (HG38_NAMES, HG38, 1001, 4500000), 'mm10': (MM10_NAMES, MM10, 1002, 3700000)}

TE_CLASSES = [('LINE', 'L1'), ('LINE', 'L2'), ('SINE', 'Alu'), ('SINE', 'MIR'), ('LTR', 'ERVL'), ('LTR', 'ERVK'),
              ('DNA', 'hAT'), ('DNA', 'TcMar'), ('Retroposon', 'SVA'), ('tRNA', 'tRNA')]


class Annotation:
    """A FlatIndex plus the tables the read generator anchors on."""

    def __init__(self, genome, flat, chrom_names, chrom_len, exon_gene, n_genes, n_exon_rows):
        self.genome = genome
        self.flat = flat
        self.chrom_names = chrom_names      # BAM header names ('chr1', ...); index keys are '1', ...
        self.chrom_len = np.asarray(chrom_len, np.int64)
        self.exon_gene = exon_gene          # gene ordinal of each exon row (exon rows come first in flat)
        self.n_genes = n_genes
        self.n_exon_rows = n_exon_rows


def make_annotation(genome='hg38', n_te=None, n_genes=40000, seed=None, scale=1.0):
    """genes_tes-like annotation.  `scale` < 1 shrinks chromosome lengths and interval counts
    together (same density) for tests."""
    names, lens, dseed, dte = GENOMES[genome]
    seed = dseed if seed is None else seed
    n_te = int((dte if n_te is None else n_te) * scale)
    n_genes = max(8, int(n_genes * scale))
    lens = np.maximum(20000, (np.asarray(lens, np.float64) * scale)).astype(np.int64)
    rng = np.random.default_rng(seed)
    w = lens / lens.sum()

    # ---- exon rows: 1-60 exons per gene (geometric), span <= 200 kb, duplicated across "transcripts"
    g_chrom = rng.choice(len(lens), n_genes, p=w)
    n_ex = np.clip(rng.geometric(1.0 / 21.0, n_genes), 1, 60)
    span = np.minimum(200000, n_ex * rng.integers(1000, 8000, n_genes))
    g_start = (rng.random(n_genes) * np.maximum(1, lens[g_chrom] - span - 2000)).astype(np.int64) + 1
    gene_of = np.repeat(np.arange(n_genes), n_ex)
    off = (rng.random(len(gene_of)) * span[gene_of]).astype(np.int64)
    e_left = g_start[gene_of] + off
    e_len = (rng.lognormal(5.0, 0.7, len(gene_of)) + 1).astype(np.int64)
    dup = 1 + (rng.random(len(gene_of)) < 0.4) + (rng.random(len(gene_of)) < 0.16)
    gene_of = np.repeat(gene_of, dup)
    e_left = np.repeat(e_left, dup)
    e_len = np.repeat(e_len, dup)
    e_chrom = g_chrom[gene_of]
    e_right = e_left + e_len
    g_strand = rng.integers(0, 2, n_genes)
    g_class = np.where(np.arange(n_genes) < n_genes // 2, _lib.CLASS_GENE, _lib.CLASS_GENE)  # protein_coding | lncRNA
    n_exon_rows = len(gene_of)

    # ---- TE rows
    n_fam = max(4, int(1300 * min(1.0, max(scale, 0.02) * 4)))
    t_chrom = rng.choice(len(lens), n_te, p=w)
    t_left = (rng.random(n_te) * (lens[t_chrom] - 10)).astype(np.int64)
    t_right = t_left + (rng.lognormal(5.5, 0.8, n_te) + 1).astype(np.int64)
    ranks = np.arange(1, n_fam + 1, dtype=np.float64)
    pz = ranks ** -1.1
    pz /= pz.sum()
    t_fam = rng.choice(n_fam, n_te, p=pz)
    t_strand = rng.integers(0, 2, n_te)

    gene_names = ['ENSG%011d' % i for i in range(n_genes)]
    fam_names = ['%s:%s:%s_%04d' % (TE_CLASSES[i % len(TE_CLASSES)] + (TE_CLASSES[i % len(TE_CLASSES)][1], i))
                 for i in range(n_fam)]
    all_names = gene_names + fam_names
    order = sorted(range(len(all_names)), key=lambda i: all_names[i])
    rank = np.empty(len(all_names), np.int64)
    rank[order] = np.arange(len(all_names))
    sorted_names = [all_names[i] for i in order]

    chrom_id = np.concatenate([e_chrom, t_chrom]).astype(np.int32)
    left = np.concatenate([e_left, t_left])
    right = np.concatenate([e_right, t_right])
    name_id = np.concatenate([rank[gene_of], rank[n_genes + t_fam]])
    cls = np.concatenate([g_class[gene_of], np.full(n_te, _lib.CLASS_TE)])
    strand = np.concatenate([g_strand[gene_of], t_strand])
    meta = (name_id | (cls << _lib.META_CLASS_SHIFT) | (strand << _lib.META_STRAND_SHIFT)).astype(np.uint32)
    keys = [n.replace('chr', '') for n in names]  # == location.py:31 for these names
    flat = FlatIndex(sorted_names, keys, ['+', '-'], chrom_id, left, right, meta)
    return Annotation(genome, flat, list(names), lens, gene_of, n_genes, n_exon_rows)


def make_custom_annotation(n_intervals=3000000, seed=2005, scale=1.0):
    """BASELINE config 5: `custom` GTF-like index of unique lncRNA intervals, length 200-2000,
    chr1-22,X,Y (hg38 lengths)."""
    names, lens = HG38_NAMES[:24], np.asarray(HG38[:24], np.int64)
    n = int(n_intervals * scale)
    lens = np.maximum(20000, (lens * scale)).astype(np.int64)
    rng = np.random.default_rng(seed)
    chrom = rng.choice(len(lens), n, p=lens / lens.sum())
    left = (rng.random(n) * (lens[chrom] - 2100)).astype(np.int64) + 1
    right = left + rng.integers(200, 2001, n)
    feat_names = ['ENH%07d' % i for i in range(n)]  # already sorted
    meta = (np.arange(n) | (_lib.CLASS_GENE << _lib.META_CLASS_SHIFT) |
            (rng.integers(0, 2, n) << _lib.META_STRAND_SHIFT)).astype(np.uint32)
    flat = FlatIndex(feat_names, [x.replace('chr', '') for x in names], ['+', '-'], chrom.astype(np.int32), left, right, meta)
    return Annotation('custom', flat, list(names), lens, np.zeros(0, np.int64), 0, 0)


_MAPQ_VALUES = np.array([255, 3, 1, 0, 30], np.uint8)
_MAPQ_P = np.array([0.80, 0.08, 0.05, 0.04, 0.03])


def make_units(ann, n_units, seed, paired, order='shuffled', chunk=1 << 22):
    """Counting units (chrom, p1, p2, flag, mapq) as the host decoder would hand them over.

    SE (config 2): 100 bp reads, 15 % spliced (one N gap, lognormal(7,1)); p1/p2 = start/end.
    PE (configs 1/3 and the headline): insert ~ N(300,50); p1/p2 = the two mates' starts, read 1
    being the reverse mate half of the time.  70 % of reads are anchored on a feature (half of
    them on exons with Zipf gene expression, the rest on TEs), 30 % uniform; 3 % sit on contigs
    that are not in the index; mapq and flag mixes as in SURVEY.md 8(d).
    order: 'shuffled' (name-grouped aligner output, what the PE path needs) or 'sorted'
    (coordinate-sorted)."""
    flat = ann.flat
    n_feat = len(flat)
    lens = ann.chrom_len
    n_exon = ann.n_exon_rows
    if ann.n_genes:
        gw = (np.arange(1, ann.n_genes + 1, dtype=np.float64)) ** -1.0
        gperm = np.random.default_rng(seed ^ 0x5EED).permutation(ann.n_genes)
        gene_cum = np.cumsum(gw / gw.sum())
        # exon rows grouped by gene: first row and count per gene
        first = np.searchsorted(ann.exon_gene, np.arange(ann.n_genes), 'left')
        cnt = np.searchsorted(ann.exon_gene, np.arange(ann.n_genes), 'right') - first
    outs = [[] for _ in range(5)]
    done = 0
    ci = 0
    while done < n_units:
        m = min(chunk, n_units - done)
        rng = np.random.default_rng([seed, ci])
        kind = rng.random(m)
        f = np.empty(m, np.int64)
        if ann.n_genes:
            g = gperm[np.minimum(np.searchsorted(gene_cum, rng.random(m)), ann.n_genes - 1)]
            f_exon = first[g] + (rng.random(m) * cnt[g]).astype(np.int64)
            f_te = n_exon + (rng.random(m) * max(1, n_feat - n_exon)).astype(np.int64)
            f = np.where(kind < 0.35, f_exon, np.minimum(f_te, n_feat - 1))
        else:
            f = (rng.random(m) * n_feat).astype(np.int64)
        fl_, fr_ = flat.left[f].astype(np.int64), flat.right[f].astype(np.int64)
        anchored = kind < 0.70
        a_pos = fl_ - 60 + (rng.random(m) * (fr_ - fl_ + 70)).astype(np.int64)
        u_chrom = rng.choice(len(lens), m, p=lens / lens.sum())
        u_pos = (rng.random(m) * (lens[u_chrom] - 1000)).astype(np.int64)
        chrom = np.where(anchored, flat.chrom_id[f], u_chrom).astype(np.int32)
        pos = np.maximum(0, np.where(anchored, a_pos, u_pos))
        if paired:
            insert = np.maximum(100, rng.normal(300, 50, m)).astype(np.int64)
            mate = pos + insert - 100
            swap = rng.random(m) < 0.5
            p1 = np.where(swap, mate, pos)
            p2 = np.where(swap, pos, mate)
        else:
            gap = np.where(rng.random(m) < 0.15, rng.lognormal(7.0, 1.0, m), 0).astype(np.int64)
            p1 = pos
            p2 = pos + 100 + gap
        chrom = np.where(rng.random(m) < 0.03, -1, chrom).astype(np.int32)  # chrUn_*, *_random, chrEBV
        n_mates = 2 if paired else 1
        flag = np.zeros(m, np.uint16)
        for _ in range(n_mates):
            x = rng.random(m)
            flag |= np.where(x < 0.01, 0x4, np.where(x < 0.03, 0x400, np.where(x < 0.035, 0x200, 0))).astype(np.uint16)
            flag |= np.where(rng.random(m) < 0.10, 0x100, 0).astype(np.uint16)
        flag |= np.where(rng.random(m) < 0.5, 0x10, 0).astype(np.uint16)
        mapq = _MAPQ_VALUES[rng.choice(len(_MAPQ_VALUES), m, p=_MAPQ_P)]
        for o, a in zip(outs, (chrom, p1.astype(np.int32), p2.astype(np.int32), flag, mapq)):
            o.append(a)
        done += m
        ci += 1
    units = [np.concatenate(o) for o in outs]
    if order == 'sorted':
        key = (units[0].astype(np.int64) << 32) | np.minimum(units[1], units[2]).astype(np.int64)
        perm = np.argsort(key, kind='stable')
        units = [u[perm] for u in units]
    elif order != 'shuffled':
        raise ValueError(order)
    return tuple(units)


# --------------------------------------------------------------------------- single cell (config 4)

_BASES = np.frombuffer(b'ACGT', np.uint8)


def make_whitelist_bytes(n, length=16, seed=7):
    """n distinct random `length`-mers, sorted bytewise; uint8[n, 24] zero padded (te_count.py:338
    gives a barcode the id of its row in this order)."""
    rng = np.random.default_rng(seed)
    codes = np.unique(rng.integers(0, 4 ** length, int(n * 1.05), dtype=np.int64))[:n]
    while len(codes) < n:
        codes = np.unique(np.concatenate([codes, rng.integers(0, 4 ** length, n, dtype=np.int64)]))[:n]
    out = np.zeros((n, 24), np.uint8)
    for j in range(length):  # most significant base first: numeric order == bytewise order (A<C<G<T)
        out[:, j] = _BASES[(codes >> (2 * (length - 1 - j))) & 3]
    return out


def make_sc_arrays(ann, n_records, n_cells=10000, n_whitelist=737280, seed=2004, reads_per_umi=3.0, umi_len=12,
                   off_whitelist=0.02, order='shuffled'):
    """10x-style SE records as the arrays tec_sc_submit takes (BASELINE config 4: 10 k cells x 5 k
    reads, whitelist of 737 280 16-mers, 12-mer UMIs, ~3 reads per (CB,UB)).  Every (CB,UB) key stays
    on one (chromosome, strand) class, i.e. inside the reference's determinism envelope
    (SURVEY.md 3.4).  Returns (arrays dict, whitelist uint8[n,24], class tables)."""
    rng = np.random.default_rng(seed)
    wl = make_whitelist_bytes(n_whitelist)
    cell_rows = rng.choice(n_whitelist, n_cells, replace=False)
    n_keys = max(1, int(n_records / reads_per_umi))
    reps = rng.geometric(1.0 / reads_per_umi, n_keys)
    reps = reps[np.cumsum(reps) <= n_records]
    n_keys = len(reps)
    n = int(reps.sum())
    key_cell = cell_rows[rng.integers(0, n_cells, n_keys)]
    key_umi = rng.integers(0, 4, (n_keys, umi_len), dtype=np.int64)
    # positions per key, with the bulk generator's placement
    chrom, p1, _, _, _ = make_units(ann, n_keys, seed=seed + 1, paired=False)
    key_rev = rng.integers(0, 2, n_keys)
    rec_key = np.repeat(np.arange(n_keys), reps)
    if order == 'shuffled':
        rec_key = rec_key[rng.permutation(n)]
    pos = np.maximum(0, p1[rec_key] + rng.integers(-3, 4, n)).astype(np.int32)
    cb = wl[key_cell[rec_key]].copy()
    bad = rng.random(n) < off_whitelist
    cb[bad, 15] = np.where(cb[bad, 15] == ord('A'), ord('C'), ord('A'))  # mostly off the whitelist
    ub = np.zeros((n, 24), np.uint8)
    ub[:, :umi_len] = _BASES[key_umi[rec_key]]
    flag = np.where(key_rev[rec_key] == 1, 0x10, 0).astype(np.uint16)
    x = rng.random(n)
    flag |= np.where(x < 0.01, 0x4, np.where(x < 0.03, 0x400, np.where(x < 0.035, 0x200, 0))).astype(np.uint16)
    mapq = _MAPQ_VALUES[rng.choice(len(_MAPQ_VALUES), n, p=_MAPQ_P)]
    # read-side chromosome classes: one per header name; a few records on contigs that part 1 skips
    names = list(ann.chrom_names) + ['chrUn_KI270742v1', 'chr1_KI270706v1_random']
    cls = chrom[rec_key].astype(np.int32)
    stray = rng.random(n) < 0.02
    cls[stray] = len(ann.chrom_names) + rng.integers(0, 2, int(stray.sum()))
    cls[chrom[rec_key] < 0] = len(ann.chrom_names)
    arrays = {'cls': cls, 'pos': pos, 'end': (pos + 98).astype(np.int32), 'flag': flag, 'mapq': mapq, 'cb24': cb,
              'ub24': ub, 'tagflags': np.full(n, 3, np.uint8)}
    class_chrom = np.array([ann.flat._chrom_lookup.get(nm.replace('chr', ''), -1) for nm in names], np.int32)
    class_skip = np.array([('_' in nm.replace('chr', '')) or ('alt' in nm.replace('chr', '')) for nm in names], np.uint8)
    return arrays, wl, (names, class_chrom, class_skip)


def features_overlapping(flat, chrom, lo, hi):
    """Feature dicts (oracle schema) of every index row that overlaps one of the windows
    [lo[i], hi[i]] on chromosome id chrom[i] (negative ids are skipped): lets the Python oracle check a
    sample of records against a multi-million-row annotation without materialising all of it."""
    order = np.argsort((flat.chrom_id.astype(np.int64) << 32) + flat.left, kind='stable')
    starts = ((flat.chrom_id.astype(np.int64) << 32) + flat.left)[order]
    rights = flat.right[order].astype(np.int64)
    maxlen = int((flat.right.astype(np.int64) - flat.left).max()) + 1
    chrom = np.asarray(chrom, np.int64)
    lo, hi = np.asarray(lo, np.int64), np.asarray(hi, np.int64)
    ok = chrom >= 0
    a = np.searchsorted(starts, (chrom << 32) + np.maximum(0, lo - maxlen))
    b = np.searchsorted(starts, (chrom << 32) + hi, 'right')
    near = np.zeros(len(flat), bool)
    for i in np.nonzero(ok)[0].tolist():
        sl = slice(a[i], b[i])
        near[order[sl][rights[sl] >= lo[i]]] = True
    sel = np.nonzero(near)[0]
    strands = flat.strands
    return [{'chr': flat.chrom_keys[c], 'left': int(l), 'right': int(r), 'strand': strands[(m >> 29) & 7], 'name': 'x',
             'type': 'TE' if ((m >> 26) & 7) == 2 else 'protein_coding', 'ensg': flat.names[m & ((1 << 26) - 1)]}
            for c, l, r, m in zip(flat.chrom_id[sel].tolist(), flat.left[sel].tolist(), flat.right[sel].tolist(),
                                  flat.meta[sel].tolist())]


def sc_arrays_to_records(arrays, names):
    """The same records as dicts (oracle / reference schema) -- small inputs only."""
    recs = []
    for i in range(len(arrays['pos'])):
        recs.append({'flag': int(arrays['flag'][i]), 'mapq': int(arrays['mapq'][i]), 'rname': names[arrays['cls'][i]],
                     'pos': int(arrays['pos'][i]), 'end': int(arrays['end'][i]),
                     'tags': {'CB': bytes(arrays['cb24'][i]).rstrip(b'\0').decode(),
                              'UB': bytes(arrays['ub24'][i]).rstrip(b'\0').decode()}})
    return recs
