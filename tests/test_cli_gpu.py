"""End to end on the GPU box: BAM/SAM file -> bin/te_count (the reference's flags) -> TSV bytes
identical to what the reference wrote for the same records (golden fixtures)."""

import os
import subprocess
import sys

import pytest

from conftest import ROOT, load_golden
from helpers import flat_from_golden
from te_counter_amd import bam, cli
from te_counter_amd.reads import ListSource

pytestmark = pytest.mark.gpu


def _prepare(g, tmp_path, sam=False):
    idx = str(tmp_path / 'custom.tecidx')
    flat_from_golden(g).save(idx)
    aln = str(tmp_path / ('in.sam' if sam else 'in.bam'))
    refs = ListSource(g['records']).reference_names
    (bam.write_sam if sam else bam.write_bam)(aln, refs, g['records'])
    return idx, aln


@pytest.mark.parametrize('name', ['bulk_se_rand11', 'bulk_pe_rand11', 'bulk_se_q0', 'bulk_pe_rand12'])
@pytest.mark.parametrize('sam', [False, True])
def test_bulk_cli(name, sam, tmp_path):
    g = load_golden(name)
    idx, aln = _prepare(g, tmp_path, sam)
    out = str(tmp_path / 'out.tsv')
    argv = ['-i', aln, '-o', out, '-g', idx, '-m', 'custom', '-q', str(g['params']['q'])]
    if not g['params']['paired']:
        argv.append('--se')
    cli.main(argv)
    assert open(out).read() == g['expected']['tsv']


@pytest.mark.parametrize('name', ['sc_rand21_nostrand', 'sc_rand21_strand', 'sc_rand22_strand'])
def test_sc_cli(name, tmp_path):
    g = load_golden(name)
    idx, aln = _prepare(g, tmp_path)
    wl = tmp_path / 'wl.txt'
    wl.write_text(''.join(b + '\n' for b in g['whitelist']))
    out = str(tmp_path / 'cells_out.tsv')
    argv = ['-i', aln, '-o', out, '-g', idx, '-m', 'custom', '--sc', '--se', '-w', str(wl), '--maxcells',
            str(g['params']['maxcells'])]
    if g['params']['strand']:
        argv.append('--strand')
    cli.main(argv)
    assert open(out).read() == g['expected']['tsv']
    assert open(out.replace('.tsv', '.barcode_freq.tsv')).read() == g['expected']['barcode_freq']


def test_script_and_refusals(tmp_path):
    g = load_golden('bulk_se_rand12')
    idx, aln = _prepare(g, tmp_path)
    out = str(tmp_path / 'o.tsv')
    exe = os.path.join(ROOT, 'bin', 'te_count')
    r = subprocess.run([sys.executable, exe, '-i', aln, '-o', out, '-g', idx, '-m', 'custom', '--se'],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert open(out).read() == g['expected']['tsv']
    with pytest.raises(NotImplementedError):  # bulk --strand: te_count.py:58-59
        cli.main(['-i', aln, '-o', out, '-g', idx, '-m', 'custom', '--se', '--strand'])
    # ... unless the extension is asked for by name: then the stranded table holds no more than the unstranded one
    cli.main(['-i', aln, '-o', out, '-g', idx, '-m', 'custom', '--se', '--strand', '--bulk-strand-extension'])
    plain = {ln.split('\t')[0]: int(ln.split('\t')[1]) for ln in g['expected']['tsv'].splitlines()}
    stranded = {ln.split('\t')[0]: int(ln.split('\t')[1]) for ln in open(out).read().splitlines()}
    assert stranded.keys() <= plain.keys() and all(stranded[k] <= plain[k] for k in stranded)
    assert 0 < sum(stranded.values()) < sum(plain.values())
    with pytest.raises(AssertionError):       # --sc without --se: bin/te_count:109-110
        cli.main(['-i', aln, '-o', out, '-g', idx, '-m', 'custom', '--sc', '-w', 'x'])
    with pytest.raises(SystemExit):            # genome not built: bin/te_count:89-92
        cli.main(['-i', aln, '-o', out, '-g', 'hg38', '-m', 'genes_tes'])


def _tests_sh_fixture(tmp_path, types, with_strand=True, sc=False, seed=61):
    """Index + BAM for one line of the reference's test/tests.sh (modes genes_tes / enhancers / snrnps)."""
    import datagen as dg
    from te_counter_amd.index import FlatIndex
    feats = dg.make_features(seed, 300, types=types, with_strand=with_strand)
    idx = str(tmp_path / 'idx.tecidx')
    FlatIndex.from_features(feats).save(idx)
    if sc:
        wl = dg.make_whitelist(seed, 60)
        recs = dg.make_sc_records(seed + 1, 1500, wl, feats, n_cells=8)
        wlf = tmp_path / 'barcodes.txt'
        wlf.write_text(''.join(b + '\n' for b in wl))
    else:
        wl, wlf = None, None
        recs = [r for r in dg.make_records(seed + 1, 1500, feats, paired=True) if r['end'] > r['pos']]
    aln = str(tmp_path / 'in.bam')
    bam.write_bam(aln, ListSource(recs).reference_names, recs)
    return feats, recs, wl, idx, aln, (str(wlf) if wlf else None)


@pytest.mark.parametrize('se', [False, True])
def test_tests_sh_snrnps_bulk(se, tmp_path):
    """tests.sh: `te_count [-se] -i bulk.bam -g hg38 -m snrnps` -- snRNA features are counted in bulk."""
    from oracle import te_oracle as orc
    feats, recs, _, idx, aln, _ = _tests_sh_fixture(tmp_path, ['snRNA', 'snRNA', 'miRNA'])
    out = str(tmp_path / 'o.tsv')
    cli.main(['-i', aln, '-o', out, '-g', idx, '-m', 'custom'] + (['--se'] if se else []))
    res, total, _ = orc.bulk_count(feats, recs, not se)
    assert open(out).read() == orc.format_bulk_tsv(res, total) and sum(res.values()) > 0


def test_tests_sh_enhancers_modes_crash_like_the_reference(tmp_path):
    """tests.sh runs the `enhancers` index through bulk and --sc; the reference dies in both
    (NameError in bulk, te_count.py:145-147; KeyError 'strand' in sc, te_count.py:661)."""
    feats, recs, _, idx, aln, _ = _tests_sh_fixture(tmp_path, ['enhancer'], with_strand=False)
    with pytest.raises(NameError):
        cli.main(['-i', aln, '-o', str(tmp_path / 'o.tsv'), '-g', idx, '-m', 'custom'])
    feats, recs, wl, idx, aln, wlf = _tests_sh_fixture(tmp_path, ['enhancer'], with_strand=False, sc=True)
    with pytest.raises(KeyError):
        cli.main(['-i', aln, '-o', str(tmp_path / 'c.tsv'), '-g', idx, '-m', 'custom', '--sc', '--se', '-w', wlf, '--maxcells', '3'])


def test_tests_sh_sc_maxcells_3(tmp_path):
    """tests.sh: `--se --sc ... --maxcells 3` with and without --strand; snRNA is not counted in sc."""
    from oracle import te_oracle as orc
    types = ['protein_coding', 'lncRNA', 'TE', 'TE', 'snRNA']
    feats, recs, wl, idx, aln, wlf = _tests_sh_fixture(tmp_path, types, sc=True)
    for strand in (False, True):
        out = str(tmp_path / ('single_cell%s.tsv' % ('_strand' if strand else '')))
        cli.main(['-i', aln, '-o', out, '-g', idx, '-m', 'custom', '--sc', '--se', '-w', wlf, '--maxcells', '3'] +
                 (['--strand'] if strand else []))
        res, bcs, _ = orc.sc_count(feats, recs, wl, 3, strand=strand)
        tsv, freq = orc.format_sc_tsv(res, bcs, 3)
        assert open(out).read() == tsv and tsv.count('\n') == 4
        assert open(out.replace('.tsv', '.barcode_freq.tsv')).read() == freq


def test_se_record_without_cigar_raises_like_pysam_would(tmp_path):
    """A mapped SE record without CIGAR has reference_end None in pysam; the reference dies in
    `loc2+1` (te_count.py:223) with TypeError.  Filtered records of that kind are harmless."""
    g = load_golden('bulk_se_rand11')
    idx = str(tmp_path / 'i.tecidx')
    flat_from_golden(g).save(idx)
    ok = {'flag': 0, 'mapq': 255, 'qname': 'a', 'rname': 'chr1', 'pos': 100, 'end': 150}
    nocigar_filtered = {'flag': 0x4, 'mapq': 255, 'qname': 'b', 'rname': 'chr1', 'pos': 100, 'end': 100}
    nocigar_live = {'flag': 0, 'mapq': 255, 'qname': 'c', 'rname': 'chr1', 'pos': 100, 'end': 100}
    aln = str(tmp_path / 'a.bam')
    bam.write_bam(aln, ['chr1'], [ok, nocigar_filtered, ok])
    cli.main(['-i', aln, '-o', str(tmp_path / 'o.tsv'), '-g', idx, '-m', 'custom', '--se'])
    bam.write_bam(aln, ['chr1'], [ok, nocigar_live, ok])
    with pytest.raises(TypeError):
        cli.main(['-i', aln, '-o', str(tmp_path / 'o2.tsv'), '-g', idx, '-m', 'custom', '--se'])


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def _launch(argv, nproc, tmp_path):
    """bin/te_count under torch.distributed.run; the ranks share GPU 0 (gloo carries the exchanges)."""
    exe = os.path.join(ROOT, 'bin', 'te_count')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), exe] + argv + ['--device', '0']
    env = dict(os.environ, TEC_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.parametrize('nproc', [2, 3])
def test_multi_rank_cli_bulk_and_single_cell(nproc, tmp_path):
    """One process per rank as the driver launches them: every rank decodes its byte range of the BAM on the GPU,
    PE mates are paired across the range ends, single-cell cells are dealt over the ranks -- the files rank 0 writes
    are byte-identical to a one-process run (and, for the golden inputs, to the reference's)."""
    from te_counter_amd import synth
    ann = synth.make_annotation('hg38', scale=0.02)
    idx = str(tmp_path / 'ann.tecidx')
    ann.flat.save(idx)
    refs = ann.chrom_names + ['chrUn_1']
    u = synth.make_units(ann, 150_001, seed=31, paired=True)
    tid, pos, fl, mq = bam.units_to_records(u, len(refs) - 1, True)
    pe = str(tmp_path / 'pe.bam')
    bam.write_bam_arrays(pe, refs, tid[:-1], pos[:-1], fl[:-1], mq[:-1], pair_names=True, payload='illumina')  # odd record count
    one, many = str(tmp_path / 'one.tsv'), str(tmp_path / 'many.tsv')
    cli.main(['-i', pe, '-o', one, '-g', idx, '-m', 'custom'])
    _launch(['-i', pe, '-o', many, '-g', idx, '-m', 'custom'], nproc, tmp_path)
    assert open(many).read() == open(one).read()
    # single cell on a golden input: the reference's own bytes
    g = load_golden('sc_rand22_strand' if nproc == 2 else 'sc_topcells_binding')
    gidx, aln = _prepare(g, tmp_path)
    wl = tmp_path / 'wl.txt'
    wl.write_text(''.join(b + '\n' for b in g['whitelist']))
    out = str(tmp_path / 'cells.tsv')
    argv = ['-i', aln, '-o', out, '-g', gidx, '-m', 'custom', '--sc', '--se', '-w', str(wl), '--maxcells', str(g['params']['maxcells'])]
    if g['params']['strand']:
        argv.append('--strand')
    _launch(argv, nproc, tmp_path)
    assert open(out).read() == g['expected']['tsv']
    assert open(out.replace('.tsv', '.barcode_freq.tsv')).read() == g['expected']['barcode_freq']


def test_host_decoder_switch_and_index_threads(tmp_path, monkeypatch):
    """TEC_DECODER=cpu sends a BAM file through the host decoder, TEC_INDEX_THREADS sets the threads of the index
    build: the TSV is the reference's either way."""
    g = load_golden('bulk_pe_rand12')
    idx, aln = _prepare(g, tmp_path)
    outs = []
    for env in ({}, {'TEC_DECODER': 'cpu'}, {'TEC_INDEX_THREADS': '1'}, {'TEC_DECODER': 'cpu', 'TEC_INDEX_THREADS': '3'}):
        for k in ('TEC_DECODER', 'TEC_INDEX_THREADS'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out = str(tmp_path / ('o%d.tsv' % len(outs)))
        cli.main(['-i', aln, '-o', out, '-g', idx, '-m', 'custom', '-q', str(g['params']['q'])])
        outs.append(open(out).read())
    assert all(o == g['expected']['tsv'] for o in outs)


def test_bench_rehearsal_two_ranks_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2` from a bare interpreter starts its two ranks itself; TEC_BENCH_BACKEND=gloo lets them
    share the one GPU of this box.  One JSON line, whole-job value over both ranks."""
    import json
    env = dict(os.environ, TEC_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--records', '4000000',
                        '--scale', '0.05', '--cpu-sample', '0'], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['steps'] == 3 and line['scaling'] == 'weak' and line['value'] > 0
