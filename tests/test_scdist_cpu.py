"""The single-cell multi-GPU orchestration (te_counter_amd/scdist.py) on CPU: per-rank compute by the
numpy stand-in engine (tests/sc_np_engine.py), collectives over gloo, results against the golden vectors
captured from the reference (every sc_* fixture, incl. multi-bundle runs, binding barcode selection and
the degenerate-boundary case)."""

import os
import socket
import subprocess
import sys

import pytest

from conftest import golden_names, load_golden
from oracle import te_oracle as orc
from te_counter_amd import scdist
from te_counter_amd.dist import World

HERE = os.path.dirname(os.path.abspath(__file__))
SC = [n for n in golden_names('sc_') if '_err_' not in n]


def run_rank(g, world, cuts=None):
    """One rank's part of a golden case: its contiguous slice of the records through the stand-in engine."""
    from sc_np_engine import NpScEngine, result_views
    from te_counter_amd.index import read_chrom_key
    p = g['params']
    wl = sorted(set(b.strip() for b in g['whitelist']))
    recs = g['records']
    names = sorted(set(read_chrom_key(r['rname']) for r in recs if r.get('rname')))
    eng = NpScEngine(g['features'], wl, names, p['strand'], p['q'], p['bundle_threshold'])
    n = len(recs)
    cuts = cuts or [n * r // world.size for r in range(world.size + 1)]
    mine = recs[cuts[world.rank]:cuts[world.rank + 1]]
    eng.ingest(mine)
    out = scdist.run_parts(eng, world, len(mine), p['bundle_threshold'], p['maxcells'], 1000)
    result, barcodes = result_views(out, eng.names, wl)
    return out, result, barcodes


def check(g, out, result, barcodes):
    e = g['expected']
    assert out['total_records'] + 1 == e['total_reads']
    assert barcodes == e['barcodes']
    assert result == e['result']
    bc = orc.OrderedDict((b, h) for b, h in barcodes)
    assert orc.format_sc_tsv(result, bc, g['params']['maxcells']) == (e['tsv'], e['barcode_freq'])


@pytest.mark.parametrize('name', SC)
def test_one_rank_equals_the_reference(name):
    g = load_golden(name)
    check(g, *run_rank(g, World()))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('world', [2, 3])
@pytest.mark.parametrize('name', SC)
def test_ranks_merge_to_the_reference_result(name, world):
    """Every sc fixture with the records cut into 2 and 3 contiguous slices (uneven cuts): the merged
    result, the barcode order, the TSV and the barcode_freq file equal the single-process goldens."""
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), TEC_NO_TORCH='')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'scdist_worker.py'), name], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert ('OK %d' % world) in outs[0][0], outs[0]
