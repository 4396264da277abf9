"""
TEST INFRASTRUCTURE ONLY -- the CPU oracle.

A plain-Python restatement of the reference's read-vs-annotation counting path
(oaxiom/te_counter; citations are /root/reference-relative file:line).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product
path (te_counter_amd/) must never call it.

Parity status: PINNED.  Every function below is checked bit-for-bit against golden vectors
captured from the real reference by tests/golden/make_golden.py (stand-in pysam, see
oracle/ref_harness.py), and, when /root/reference is mounted, against the live reference
on freshly randomised inputs (tests/test_oracle_vs_reference.py).

Record / feature schemas: see oracle/ref_harness.py.
"""

from collections import OrderedDict

BUCKET_SIZE = 10000  # te_count/miniglbase/config.py:36

GENE_TYPES = ('protein_coding', 'lincRNA', 'lncRNA')


def index_chrom_key(chrom):
    """Index-side chromosome key, te_count/miniglbase/location.py:31
    (strip() of the *characters* c,h,r from both ends, rstrip, upper)."""
    return str(chrom).strip('chr').rstrip().upper()


def read_chrom_key(rname):
    """Read-side chromosome key, te_count/te_count.py:96,212,431 (not upper-cased)."""
    return rname.replace('chr', '')


def build_buckets(features):
    """buckets[chrom][k*bs] = [feature index, ...]; te_count/miniglbase/genelist.py:367-380.
    A feature is listed in every 10 kb bin from floor(left/bs) to floor(right/bs)."""
    bs = BUCKET_SIZE
    buckets = {}
    for n, f in enumerate(features):
        chrom = index_chrom_key(f['chr'])
        per = buckets.setdefault(chrom, {})
        lo = (int(f['left']) // bs) * bs
        hi = ((int(f['right']) + bs) // bs) * bs
        for b in range(lo, hi, bs):
            per.setdefault(b, []).append(n)
    return buckets


def feature_names(features):
    """all_feature_names, te_count/te_count.py:35."""
    return sorted(set(f['ensg'] for f in features))


class EnhancerBulkCrash(NameError):
    """The reference dies with NameError('barcode') when a bulk read's only countable
    type is 'enhancer' (te_count/te_count.py:145-147, 258-260)."""


def _bulk_hits(features, buckets, chrom, p1, p2):
    """Candidate lookup in exactly two 10 kb bins + the two 2-bp edge tests;
    te_count/te_count.py:106-126 (PE) / 222-241 (SE)."""
    bs = BUCKET_SIZE
    per = buckets[chrom]
    cand = set()
    for b in (((p1 - 1) // bs) * bs, ((p2 + 1) // bs) * bs):
        if b in per:
            cand.update(per[b])
    hits = []
    for i in cand:
        gl = int(features[i]['left'])
        gr = int(features[i]['right'])
        if p1 >= gl and p1 + 1 <= gr:
            hits.append(i)
        elif p2 - 1 >= gl and p2 <= gr:
            hits.append(i)
    return hits


def bulk_count(features, records, paired, quality_threshold=20, strand=False, buckets=None):
    """measureTE.parse_bamse (te_count.py:167-277) / parse_bampe (te_count.py:42-165).
    Returns (result dict ensg->int, total_reads, stats dict)."""
    if strand:
        raise NotImplementedError()  # te_count.py:58-59, 183-184
    buckets = buckets if buckets is not None else build_buckets(features)  # (pre-built: lets a caller time the loop alone)
    result = {n: 0 for n in feature_names(features)}
    stats = {'assigned': 0, 'lowq': 0, 'badchrom': 0, 'qcfail': 0}
    idx = 0
    it = iter(records)
    while True:
        idx += 1  # incremented before the read is fetched -> total = N+1 (te_count.py:202,77)
        try:
            r1 = next(it)
            r2 = next(it) if paired else None
        except StopIteration:
            break
        if r1['flag'] & 0x604 or (paired and r2['flag'] & 0x604):
            stats['qcfail'] += 1
            continue
        if int(r1['mapq']) < quality_threshold:
            stats['lowq'] += 1
            continue
        if paired:
            a = '_'.join(r1.get('qname', 'q').split('/')[0:-1])
            b = '_'.join(r2.get('qname', 'q').split('/')[0:-1])
            if a != b:
                raise AttributeError("module 'sys' has no attribute 'quit'")  # te_count.py:92-94
        chrom = read_chrom_key(r1['rname'])
        p1 = r1['pos']
        p2 = r2['pos'] if paired else r1['end']
        if chrom not in buckets:
            stats['badchrom'] += 1
            continue
        hits = _bulk_hits(features, buckets, chrom, p1, p2)
        if hits:
            types = set(features[i]['type'] for i in hits)
            ensgs = set(features[i]['ensg'] for i in hits)
            # the three countable branches do the same thing in bulk; `':' in ensgs` at
            # te_count.py:136,249 is a set-membership test of the string ':' -> dead code.
            if types & set(GENE_TYPES) or 'TE' in types or 'snRNA' in types:
                for e in ensgs:
                    result[e] += 1
            elif 'enhancer' in types:
                raise EnhancerBulkCrash("name 'barcode' is not defined")
            stats['assigned'] += 1
    return result, idx, stats


def format_bulk_tsv(result, total_reads):
    """measureTE.save_result_bulk, te_count.py:290-295."""
    per_m = total_reads / 1e6
    return ''.join('{0}\t{1}\t{2}\n'.format(k, result[k], result[k] / per_m) for k in sorted(result))


# --------------------------------------------------------------------------- single cell

def _sc_hits(features, buckets, chrom, left, rite):
    """All 10 kb bins from (left-1)//bs to rite//bs, then the two inclusive 2-bp window
    tests; te_count.py:619-649."""
    bs = BUCKET_SIZE
    per = buckets[chrom]
    cand = set()
    for b in range(((left - 1) // bs) * bs, (rite // bs) * bs + bs, bs):
        if b in per:
            cand.update(per[b])
    hits = []
    for i in cand:
        gl = int(features[i]['left'])
        gr = int(features[i]['right'])
        if left + 1 >= gl and left <= gr:
            hits.append(i)
        elif rite >= gl and rite - 1 <= gr:
            hits.append(i)
    return hits


def sc_count(features, records, whitelist, maxcells, quality_threshold=20, strand=False,
             umis=True, bundle_threshold=10000000, pad=1000):
    """measureTE.sc_parse_bamse, te_count.py:298-707.

    Returns (result {ensg: {barcode: int}}, barcodes OrderedDict barcode->int (Part-3 counts,
    insertion order preserved), total_reads).

    Where the reference is hash-seed dependent (a (CB,UB) key that already holds two
    strings, te_count.py:452) this restatement takes the FIRST-INSERTED string as
    next(iter(set)); inside the determinism envelope of SURVEY.md section 3.4 the two agree.
    """
    if not umis:
        raise ZeroDivisionError('division by zero')  # te_count.py:428-442,703: nothing is stored
    wl_sorted = sorted(set(b.strip() for b in whitelist))  # te_count.py:330-339
    wl_id = {b: i for i, b in enumerate(wl_sorted)}

    # ---- Part 1 (te_count.py:371-476): per-epoch (CB,UB) collapse
    part1_counts = OrderedDict()
    bundles = []
    cur = OrderedDict()  # (bcid, umi) -> list of strings, insertion ordered, set semantics
    idx = 0
    it = iter(records)
    while True:
        idx += 1
        if len(cur) >= bundle_threshold:
            bundles.append(cur)
            cur = OrderedDict()
        try:
            r = next(it)
        except StopIteration:
            break
        if r['flag'] & 0x604:
            continue
        if int(r['mapq']) < quality_threshold:
            continue
        tags = r.get('tags', {})
        if 'CB' in tags:
            bc = tags['CB']
        elif 'CR' in tags:
            bc = tags['CR']
        else:
            raise AssertionError('CB or CR tag not found!')
        if bc not in wl_id:
            continue
        bcid = wl_id[bc]
        if 'UB' in tags:
            key = (bcid, tags['UB'])
        elif 'UR' in tags:
            key = (bcid, tags['UR'])
        else:
            raise AssertionError('UB or UR tag not found!')
        chrom = read_chrom_key(r['rname'])
        if '_' in chrom or 'alt' in chrom:
            continue
        cls = '%s:%s' % (chrom, ('-' if r['flag'] & 0x10 else '+') if strand else 'NA')
        s = '%s:%d:%d' % (cls, r['pos'], r['end'])
        if key in cur:
            if cur[key][0].startswith(cls):
                continue  # already seen (te_count.py:452-454)
            if s not in cur[key]:
                cur[key].append(s)
        else:
            cur[key] = [s]
        part1_counts[bcid] = part1_counts.get(bcid, 0) + 1
    if cur:
        bundles.append(cur)

    # ---- Part 2 (te_count.py:493-575): best cells + k-way merge with the first-line drop rule
    ranked = sorted(part1_counts.items(), key=lambda kv: kv[1], reverse=True)  # stable
    todo = sorted(set(k for k, _ in ranked[0:maxcells + pad]))
    files = []
    for b in bundles:
        lines = [(k[0], k[1], v) for k, v in sorted(b.items(), key=lambda kv: kv[0])]
        files.append({'lines': lines, 'pos': 1, 'bc': lines[0][0], 'open': True})  # line 0 primes
    merged = []
    kept_lines = 0
    for c in todo:
        seen = OrderedDict()
        for f in files:
            while f['bc'] <= c:
                if not f['open']:
                    break
                if f['pos'] >= len(f['lines']):
                    f['open'] = False
                    continue
                ln = f['lines'][f['pos']]
                f['pos'] += 1
                f['bc'] = ln[0]
                if ln[0] == c:
                    kept_lines += 1
                    if ln[1] not in seen:
                        seen[ln[1]] = ln[2]
        for umi, strings in seen.items():
            merged.append((c, umi, strings))

    # ---- Part 3 (te_count.py:577-691)
    buckets = build_buckets(features)
    result = {n: {} for n in feature_names(features)}
    barcodes = OrderedDict()
    for bcid, umi, strings in merged:
        barcode = wl_sorted[bcid]
        reads = OrderedDict()
        for s in strings:
            t = s.split(':')
            reads[(t[0], t[1])] = (int(t[2]), int(t[3]))
        for (chrom, rstrand), (left, rite) in reads.items():
            if chrom not in buckets:
                continue
            hits = _sc_hits(features, buckets, chrom, left, rite)
            if not hits:
                continue
            barcodes[barcode] = barcodes.get(barcode, 0) + 1
            types = set(features[i]['type'] for i in hits)
            pairs = set((features[i]['ensg'], features[i]['strand']) for i in hits)  # KeyError w/o strand
            if types & set(GENE_TYPES):
                for e, s in pairs:
                    if strand and rstrand != s:
                        continue
                    result[e][barcode] = result[e].get(barcode, 0) + 1
            elif 'TE' in types or 'enhancer' in types:
                for e, s in pairs:
                    result[e][barcode] = result[e].get(barcode, 0) + 1
    if kept_lines == 0:
        raise ZeroDivisionError('division by zero')  # te_count.py:573 / :703
    return result, barcodes, idx


def format_sc_tsv(result, barcodes, maxcells):
    """measureTE.sc_save_result, te_count.py:724-754.  Returns (tsv text, barcode_freq text)."""
    order = [k for k, _ in sorted(barcodes.items(), key=lambda kv: kv[1], reverse=True)]
    if len(barcodes) > maxcells:
        order = order[0:maxcells]
    freq = ''.join('{0}\t{1}\n'.format(b, barcodes[b]) for b in order)
    rows = ['{}\t{}\n'.format('name', '\t'.join(result.keys()))]
    for b in order:
        rows.append('{}\n'.format('\t'.join([b] + [str(result[f].get(b, 0)) for f in result])))
    return ''.join(rows), freq
