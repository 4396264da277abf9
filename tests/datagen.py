"""Seeded generators of boundary-heavy test inputs (features, BAM-record dicts, whitelists).

Schemas: see oracle/ref_harness.py.  Coordinates are deliberately concentrated on the
10 kb bucket boundaries (k*bs, k*bs +- 1) and on the lengths {1,2,50,300,5000,bs-1,bs,
bs+1,25000} that the survey's probes used.
"""

import random

BS = 10000
LENGTHS = [1, 2, 3, 50, 100, 300, 5000, BS - 1, BS, BS + 1, 25000]
TYPES = ['protein_coding', 'lncRNA', 'lincRNA', 'TE', 'TE', 'TE', 'snRNA', 'miRNA']
INDEX_CHROMS = ['chr1', 'chr2', 'chrX', 'chrM', '7']
READ_CHROMS = ['chr1', 'chr2', 'chrX', 'chrM', '7', 'chr7', 'chrUn_x', 'chr1_random', 'chr2', 'chr1']
MAPQS = [0, 1, 3, 19, 20, 21, 60, 255, 255, 255, 255, 255]


def _coord(rng, span):
    k = rng.randrange(0, span // BS)
    mode = rng.random()
    if mode < 0.45:
        return max(0, k * BS + rng.choice([-2, -1, 0, 1, 2]))
    return rng.randrange(0, span)


def make_features(seed, n, span=200000, types=None, chroms=None, with_strand=True):
    rng = random.Random(seed)
    types = types or TYPES
    chroms = chroms or INDEX_CHROMS
    n_genes = max(3, n // 6)
    n_tes = max(3, n // 10)
    feats = []
    for _ in range(n):
        t = rng.choice(types)
        left = _coord(rng, span)
        right = left + rng.choice(LENGTHS) - (1 if rng.random() < 0.3 else 0)
        if t == 'TE':
            k = rng.randrange(n_tes)
            ensg = 'LINE:L1:L1Md_%d' % k if k % 2 else 'SINE:Alu:AluY%d' % k
            name = ensg
        else:
            k = rng.randrange(n_genes)
            ensg = 'ENSG%011d' % k
            name = 'Gene%d' % k
        f = {'chr': rng.choice(chroms), 'left': left, 'right': right, 'name': name, 'type': t, 'ensg': ensg}
        if with_strand:
            f['strand'] = rng.choice(['+', '-'])
        feats.append(f)
    return feats


def _flag(rng):
    f = 0
    x = rng.random()
    if x < 0.03:
        f |= 0x4
    elif x < 0.06:
        f |= 0x400
    elif x < 0.08:
        f |= 0x200
    if rng.random() < 0.1:
        f |= 0x100
    if rng.random() < 0.05:
        f |= 0x800
    if rng.random() < 0.5:
        f |= 0x10
    return f


def make_records(seed, n, features=None, span=200000, paired=False):
    """n BAM records.  ~60 % are anchored on a feature edge (if features given)."""
    rng = random.Random(seed)
    recs = []
    for i in range(n):
        if features and rng.random() < 0.6:
            f = rng.choice(features)
            anchor = rng.choice([f['left'], f['right']]) + rng.choice([-2, -1, 0, 1, 2])
            ln = rng.choice(LENGTHS)
            pos = max(0, anchor - (ln if rng.random() < 0.5 else 0))
            rname = 'chr' + str(f['chr']).replace('chr', '') if rng.random() < 0.9 else rng.choice(READ_CHROMS)
        else:
            pos = _coord(rng, span)
            ln = rng.choice(LENGTHS)
            rname = rng.choice(READ_CHROMS)
        qn = 'r%d' % (i // 2 if paired else i)
        recs.append({'flag': _flag(rng), 'mapq': rng.choice(MAPQS), 'qname': qn, 'rname': rname,
                     'pos': pos, 'end': pos + ln})
    return recs


def make_whitelist(seed, n, length=16):
    rng = random.Random(seed)
    s = set()
    while len(s) < n:
        s.add(''.join(rng.choice('ACGT') for _ in range(length)))
    wl = sorted(s)
    rng.shuffle(wl)
    return wl


def make_sc_records(seed, n, whitelist, features=None, span=200000, n_cells=12, umi_len=10,
                    reads_per_umi=3, second_class_frac=0.05, stranded=False):
    """10x-style SE records inside the reference's determinism envelope:
    every (CB,UB) key stays on one (chrom[,strand]) class, except that a fraction receives
    exactly one read of a second class as its LAST read."""
    rng = random.Random(seed)
    cells = [rng.choice(whitelist) for _ in range(n_cells)]
    weights = [1.0 / (i + 1) for i in range(n_cells)]
    groups = []
    made = 0
    while made < n:
        cb = rng.choices(cells, weights)[0]
        if rng.random() < 0.04:
            cb = cb[:-1] + ('A' if cb[-1] != 'A' else 'C')  # mostly off-whitelist
        ub = ''.join(rng.choice('ACGT' if rng.random() < 0.97 else 'ACGTN') for _ in range(umi_len))
        if features and rng.random() < 0.7:
            f = rng.choice(features)
            anchor = rng.choice([f['left'], f['right']]) + rng.choice([-2, -1, 0, 1, 2])
            ln = rng.choice([50, 98, 100, 300, 5000])
            pos = max(0, anchor - (ln if rng.random() < 0.5 else 0))
            rname = 'chr' + str(f['chr']).replace('chr', '')
        else:
            pos = _coord(rng, span)
            ln = rng.choice([50, 98, 100])
            rname = rng.choice(READ_CHROMS)
        rev = rng.random() < 0.5
        k = 1 + int(rng.expovariate(1.0 / reads_per_umi))
        grp = []
        for _ in range(k):
            p = max(0, pos + rng.choice([-3, -2, -1, 0, 1, 2, 3]))
            fl = (_flag(rng) & ~0x10) | (0x10 if rev else 0)
            grp.append({'flag': fl, 'mapq': rng.choice(MAPQS), 'rname': rname, 'pos': p, 'end': p + ln,
                        'tags': {'CB': cb, 'UB': ub}})
        if rng.random() < second_class_frac:
            other = rng.choice([c for c in ['chr1', 'chr2', 'chrX'] if c != rname] or ['chr2'])
            p = _coord(rng, span)
            grp.append({'flag': (0x10 if rev else 0), 'mapq': 255, 'rname': other, 'pos': p, 'end': p + 100,
                        'tags': {'CB': cb, 'UB': ub}, '_last': True})
        groups.append(grp)
        made += len(grp)
    # interleave groups while keeping each group's internal order (and '_last' reads last)
    cursors = [0] * len(groups)
    live = list(range(len(groups)))
    recs = []
    while live:
        gi = rng.choice(live[:50])
        g = groups[gi]
        r = dict(g[cursors[gi]])
        r.pop('_last', None)
        r['qname'] = 'q%d' % len(recs)
        # use CR/UR instead of CB/UB now and then (te_count.py:404-410, 421-427)
        if rng.random() < 0.05:
            r['tags'] = {'CR': r['tags']['CB'], 'UR': r['tags']['UB']}
        recs.append(r)
        cursors[gi] += 1
        if cursors[gi] == len(g):
            live.remove(gi)
    return recs[:n] if len(recs) > n else recs


def make_gtf_text(seed, n_genes=40, numeric_ids=True):
    """A GTF with the quirks the reference's reader cares about: comment lines, non-exon rows,
    contigs outside chr1-29/X/Y, dotted / numeric-looking ids, quoted and unquoted values, a
    trailing ';' on the last attribute, extra attributes.  numeric_ids=False leaves out the gene ids
    that the reference's reader turns into ints (an index holding both ints and strings cannot be
    loaded by the reference: sorted() of mixed types, te_count.py:35)."""
    rng = random.Random(seed)
    lines = ['##description: synthetic', '#!genome-build test']
    chroms = ['chr1', 'chr2', 'chrX', 'chrY', 'chrM', 'chrUn_KI270742v1', 'chr1_KI270706v1_random', 'chr7', '12']
    types = ['lncRNA', 'protein_coding', 'lincRNA', 'snRNA', 'miRNA']
    for g in range(n_genes):
        c = rng.choice(chroms)
        start = rng.randrange(1, 150000)
        gid = rng.choice(['ENSG%011d.%d' % (g, rng.randrange(1, 9)), 'ENH%07d' % g, 'gene_%d' % g, '%d' % (1000 + g)])
        if not numeric_ids and gid.isdigit():
            gid = 'ENH%07d' % g
        gname = rng.choice(['Gene%d' % g, 'GENE-%d' % g, 'G%d.1' % g])
        gtype = rng.choice(types)
        strand = rng.choice('+-')
        attrs = 'gene_id "%s"; gene_type "%s"; gene_name "%s"; ensg "%s"; level %d; tag "basic";' % (
            gid, gtype, gname, gid, rng.randrange(1, 4))
        lines.append('\t'.join([c, 'SYN', 'gene', str(start), str(start + 5000), '.', strand, '.', attrs]))
        for e in range(rng.randrange(1, 5)):
            s = start + rng.randrange(0, 4000)
            lines.append('\t'.join([c, 'SYN', rng.choice(['exon', 'exon', 'exon', 'CDS']), str(s), str(s + rng.randrange(1, 900)),
                                    '.', strand, '.', attrs.replace('level', 'exon_number %d; level' % (e + 1))]))
    return '\n'.join(lines) + '\n'


def make_gencode_text(seed, n_genes=40, named_frac=0.8):
    """GENCODE-like GTF: genes, transcripts and exons with gene_type / transcript_type."""
    rng = random.Random(seed)
    lines = ['##description: synthetic gencode', '##provider: test']
    chroms = ['chr1', 'chr2', 'chrX', 'chrY', 'chrM', 'chrUn_KI270742v1', 'chr17']
    gtypes = ['protein_coding', 'lncRNA', 'lincRNA', 'snRNA', 'snRNA', 'processed_pseudogene', 'miRNA', 'RNA']
    for g in range(n_genes):
        c = rng.choice(chroms)
        start = rng.randrange(1, 150000)
        gid = 'ENSG%011d.%d' % (g, rng.randrange(1, 19))
        gt = rng.choice(gtypes)
        strand = rng.choice('+-')
        named = rng.random() < named_frac
        base = 'gene_id "%s"; gene_type "%s";%s level 2;' % (gid, gt, (' gene_name "GN%d";' % g) if named else '')
        lines.append('\t'.join([c, 'HAVANA', 'gene', str(start), str(start + 6000), '.', strand, '.', base]))
        for t in range(rng.randrange(1, 4)):
            tt = rng.choice([gt, gt, 'retained_intron', 'protein_coding'])
            tx = base.replace(' level', ' transcript_id "ENST%08d.%d"; transcript_type "%s"; level' % (g * 10 + t, t + 1, tt))
            lines.append('\t'.join([c, 'HAVANA', 'transcript', str(start), str(start + 6000), '.', strand, '.', tx]))
            for e in range(rng.randrange(1, 4)):
                s = start + rng.randrange(0, 5000)
                ex = tx.replace(' level', ' exon_number %d; exon_id "ENSE%08d.1"; level' % (e + 1, g * 100 + t * 10 + e))
                lines.append('\t'.join([c, 'HAVANA', 'exon', str(s), str(s + rng.randrange(5, 700)), '.', strand, '.', ex]))
    return '\n'.join(lines) + '\n'


def make_rmsk_text(seed, n=300):
    """UCSC rmsk.txt rows (17 tab-separated columns, no header)."""
    rng = random.Random(seed)
    classes = [('LINE', 'L1', 'L1Md_F2'), ('LINE', 'L2', 'L2a'), ('SINE', 'Alu', 'AluY'), ('SINE', 'B2', 'B3'),
               ('LTR', 'ERVK', 'IAPEz-int'), ('DNA', 'hAT-Charlie', 'MER5A'), ('Retroposon', 'SVA', 'SVA_D'),
               ('tRNA', 'tRNA', 'tRNA-Ser-TCA'), ('Simple_repeat', 'Simple_repeat', '(CA)n'), ('Low_complexity', 'Low_complexity', 'GA-rich'),
               ('Satellite', 'centr', 'GSAT_MM'), ('rRNA', 'rRNA', '5S'), ('LTR?', 'LTR?', 'LTR90A'), ('snRNA', 'snRNA', 'U6')]
    chroms = ['chr1', 'chr2', 'chrX', 'chrY', 'chrM', 'chr4_GL456216_random', 'chrUn_JH584304', 'chr19']
    lines = []
    for i in range(n):
        cl, fam, name = rng.choice(classes)
        c = rng.choice(chroms)
        s = rng.randrange(0, 180000)
        e = s + rng.randrange(20, 6000)
        lines.append('\t'.join(map(str, [585 + s // 131072, rng.randrange(200, 30000), rng.randrange(0, 300), rng.randrange(0, 80),
                                         rng.randrange(0, 80), c, s, e, -rng.randrange(1, 10 ** 8), rng.choice('+-'), name, cl, fam,
                                         rng.randrange(1, 500), rng.randrange(500, 6000), -rng.randrange(0, 900), i % 9])))
    return '\n'.join(lines) + '\n'
