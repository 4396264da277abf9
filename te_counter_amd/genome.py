"""Index builders (the reference's `te_genome`): GTF -> `.tecidx`.

`make_custom` follows te_count/genome/make.py:173-226 over the reference's GTF reader
(miniglbase/delayedlist.py:121-172, base_genelist.py:206-265): tab-separated columns without quote
handling, `#` comment lines and a fixed list of header words skipped, attribute column split on
"; " then shell-lexed (the LAST attribute keeps its trailing ';'), values coerced
float > int > str, exons only, chromosomes X, Y, 1..29 after the index-side normalisation.
The prepackaged modes (genes_tes, enhancers, snrnps) need network downloads
(make.py:41-43) and are not built here; `make_genes_tes` accepts local copies of the two files.
"""

import csv
import gzip
import shlex

from .index import FlatIndex, index_chrom_key

# delayedlist.py:30-36
TYPICAL_HEADERS = frozenset(["chipseq_loc", "loc", "chr", "#", "Gene Name", "", "GenBank", "RefSeq", "Systematic",
                             "mm8.refGene.chrom", "mm8", "loc", 'chromosome', "mm9.refGene.chrom", "mm9",
                             "=" * 70, "=" * 69, "=" * 71])

CHR_SET_CUSTOM = frozenset(['X', 'Y'] + ['%s' % i for i in range(1, 30)])          # make.py:181
CHR_SET_GENES_TES = frozenset(['X', 'Y', 'M'] + ['%s' % i for i in range(1, 30)])  # make.py:50


def _guess(value):
    """base_genelist.py:206-238: float if it has a '.' or an 'e' and parses, else int, else str."""
    try:
        if "." in value or 'e' in value:
            return float(value)
        raise ValueError
    except ValueError:
        try:
            return int(value)
        except ValueError:
            return str(value).strip()


def _unquote(v):
    # delayedlist.py:159-165: csv quoting is off, so enclosing quotes are stripped by hand
    if isinstance(v, str) and v:
        if v[0] == "'" and v[-1] == "'":
            v = v.strip("'")
        if v and v[0] == '"' and v[-1] == '"':
            v = v.strip('"')
    return v


def iter_gtf(path):
    """Yields one dict per GTF row as the reference's delayedlist + gtf_format would
    (make.py:8-10): feature_type, feature, strand, loc=(chr key, left, right), + attributes."""
    opener = gzip.open if '.gz' in path else open
    with opener(path, 'rt') as fh:
        for column in csv.reader(fh, dialect=csv.excel_tab, quoting=csv.QUOTE_NONE):
            if not column:
                continue
            if column[0][0:1] == '#' or column[0] in TYPICAL_HEADERS:
                continue
            # key order of gtf_format (make.py:8-10): attributes land between `feature` and `loc`,
            # so a `loc` / `strand` attribute is overwritten by the column, a `feature` one wins
            d = {'feature_type': _guess(column[1]), 'feature': _guess(column[2])}
            for item in column[8].strip().split("; "):
                if item:
                    ss = shlex.split(item.strip())
                    d[ss[0]] = _guess(ss[1].strip('"'))
            d['loc'] = {'chr': index_chrom_key(column[0]), 'left': int(column[3]), 'right': int(column[4])}
            d['strand'] = _guess(column[6])
            for k in d:
                if k != 'loc':
                    d[k] = _unquote(d[k])
            yield d


def custom_features(gtffilename, log=None):
    """make_custom's feature rows (make.py:184-218)."""
    if log:
        log.info('Checking GTF format')
    for idx, item in enumerate(iter_gtf(gtffilename)):
        assert 'ensg' in item, 'the custom GTF must have an "ensg" key that is unique and can be used to score expression'
        assert 'gene_type' in item, 'the custom GTF must have a "gene_type" key, which should be one of "protein_coding/lncRNA"'
        assert 'gene_id' in item, 'the custom GTF must have a "gene_id" key, which should contain a unqie identifier for the gene'
        if idx > 10:
            break
    if log:
        log.info('Processing custom gtf {0}'.format(gtffilename))
    feats = []
    for item in iter_gtf(gtffilename):
        if item['feature'] != 'exon':
            continue
        if item['loc']['chr'] not in CHR_SET_CUSTOM:
            continue
        feats.append({'chr': item['loc']['chr'], 'left': item['loc']['left'], 'right': item['loc']['right'],
                      'strand': item['strand'], 'name': item['gene_name'], 'type': item['gene_type'],
                      'ensg': item['gene_id']})
    return feats


def make_custom(gtffilename, filename_for_index, log=None):
    """te_count/genome/make.py:173-226 -> `.tecidx`."""
    feats = custom_features(gtffilename, log)
    print('\nAdded {:,} features'.format(len(feats)))
    out = filename_for_index if str(filename_for_index).endswith('.tecidx') else str(filename_for_index) + '.tecidx'
    FlatIndex.from_features(feats).save(out)
    return out


KEEP_TE_CLASSES = frozenset(['LINE', 'LTR', 'SINE', 'DNA', 'Retroposon', 'tRNA'])  # make.py:55
GENE_BIOTYPES = ('protein_coding', 'lncRNA', 'lincRNA')                              # make.py:93-97


def iter_rmsk(path):
    """UCSC rmsk.txt(.gz) rows as the reference reads them (make.py:12-13 `rmsk_track_form`):
    tab-separated, columns 5/6/7 = chrom/start/end, 9 = strand, 10/11/12 = repName/repClass/
    repFamily.  The format has no `skiplines`, so the reference's reader drops the FIRST line of
    the file as a header (delayedlist.py:203-204) -- mirrored."""
    opener = gzip.open if '.gz' in path else open
    with opener(path, 'rt') as fh:
        reader = csv.reader(fh, dialect=csv.excel_tab, quoting=csv.QUOTE_NONE)
        next(reader, None)
        for column in reader:
            if not column or column[0] in TYPICAL_HEADERS:
                continue
            yield {'loc': {'chr': index_chrom_key(column[5]), 'left': int(column[6]), 'right': int(column[7])},
                   'strand': _unquote(_guess(column[9])), 'repName': _unquote(_guess(column[10])),
                   'repClass': _unquote(_guess(column[11])), 'repFamily': _unquote(_guess(column[12]))}


def genes_tes_features(genome, gencode_path, rmsk_path):
    """make_genes_tes (make.py:15-127) over local copies of the GENCODE GTF and the UCSC rmsk table
    (the reference downloads them).  macFas5 uses Ensembl's *_biotype attribute names."""
    gene_label, tx_label = ('gene_biotype', 'transcript_biotype') if genome == 'macFas5' else ('gene_type', 'transcript_type')
    feats = []
    for item in iter_rmsk(rmsk_path):
        if item['repClass'] not in KEEP_TE_CLASSES:
            continue
        if str(item['loc']['chr']) not in CHR_SET_GENES_TES:
            continue
        te_name = f"{item['repClass']}:{item['repFamily']}:{item['repName']}"
        feats.append({'chr': item['loc']['chr'], 'left': item['loc']['left'], 'right': item['loc']['right'],
                      'strand': item['strand'], 'name': te_name, 'type': 'TE', 'ensg': te_name})
    for item in iter_gtf(gencode_path):
        if item['feature'] != 'exon':
            continue
        if item[gene_label] not in GENE_BIOTYPES or item[tx_label] not in GENE_BIOTYPES:
            continue
        if item['loc']['chr'] not in CHR_SET_GENES_TES:
            continue
        gene_name = item['gene_name'] if 'gene_name' in item else item['gene_id']
        feats.append({'chr': item['loc']['chr'], 'left': item['loc']['left'], 'right': item['loc']['right'],
                      'strand': item['strand'], 'name': gene_name, 'type': item[gene_label],
                      'ensg': item['gene_id'].split('.')[0]})
    return feats


def make_genes_tes(genome, gencode_path, rmsk_path, filename_for_index):
    feats = genes_tes_features(genome, gencode_path, rmsk_path)
    out = filename_for_index if str(filename_for_index).endswith('.tecidx') else str(filename_for_index) + '.tecidx'
    FlatIndex.from_features(feats).save(out)
    return out


def snrnps_features(gencode_path):
    """make_snrnps (make.py:228-288) over a local GENCODE GTF.  The reference tests
    `item['gene_type'] not in ('snRNA')` -- a SUBSTRING test against the string 'snRNA' (the
    parentheses make no tuple), so e.g. a gene_type of 'RNA' passes too; mirrored."""
    feats = []
    for item in iter_gtf(gencode_path):
        if item['feature'] != 'exon':
            continue
        if item['gene_type'] not in 'snRNA' or item['transcript_type'] not in 'snRNA':
            continue
        if item['loc']['chr'] not in CHR_SET_CUSTOM:
            continue
        feats.append({'chr': item['loc']['chr'], 'left': item['loc']['left'], 'right': item['loc']['right'],
                      'strand': item['strand'], 'name': 'snRNA-' + item['gene_name'], 'type': item['gene_type'],
                      'ensg': item['gene_id'].split('.')[0]})
    return feats


def make_snrnps(gencode_path, filename_for_index):
    feats = snrnps_features(gencode_path)
    out = filename_for_index if str(filename_for_index).endswith('.tecidx') else str(filename_for_index) + '.tecidx'
    FlatIndex.from_features(feats).save(out)
    return out


def enhancers_features(genome, bed_path):
    """make_enh (make.py:129-171) over a local copy of the FANTOM5 enhancer BED the reference
    downloads (F5.<genome>.enhancers.bed.gz): every line is a row (no header is skipped), columns
    0-2 are chr / left / right through `location()` (chr normalised, location.py:20-32); name and
    ensg are 'F5enh_<genome>_<chr>_<left>_<right>', type 'enhancer', no strand key; chromosomes
    outside {X, Y, 1..29} are dropped."""
    import gzip
    feats = []
    opener = gzip.open if str(bed_path).endswith('.gz') else open
    with opener(bed_path, 'rt') as fh:
        for line in fh:
            col = line.rstrip('\r\n').split('\t')
            if len(col) < 3:
                continue
            chrom = index_chrom_key(col[0])
            if chrom not in CHR_SET_CUSTOM:
                continue
            left, right = int(col[1]), int(col[2])
            name = 'F5enh_{0}_{1}_{2}_{3}'.format(genome, chrom, left, right)
            feats.append({'chr': chrom, 'left': left, 'right': right, 'name': name, 'type': 'enhancer', 'ensg': name})
    return feats


def make_enhancers(genome, bed_path, filename_for_index):
    feats = enhancers_features(genome, bed_path)
    print('\nAdded {:,} features'.format(len(feats)))
    out = filename_for_index if str(filename_for_index).endswith('.tecidx') else str(filename_for_index) + '.tecidx'
    FlatIndex.from_features(feats).save(out)
    return out
