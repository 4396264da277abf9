"""Multi-process (world_size 2 and 3) CPU tests of the sharding + reduce plumbing over gloo."""

import os
import socket
import subprocess
import sys

import pytest

import datagen as dg
from te_counter_amd import bam
from te_counter_amd.reads import ListSource

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, args):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), TEC_NO_TORCH='')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dist_worker.py')] + [str(a) for a in args], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert ('OK %d ' % world) in outs[0][0], outs[0]
    return outs[0][0]


@pytest.mark.parametrize('world,mode,batch', [(2, 'se', 1000), (2, 'pe', 998), (3, 'pe', 4096), (2, 'pe', 7)])
def test_sharded_counts_reduce_to_the_single_process_result(world, mode, batch):
    _run(world, [mode, batch])


@pytest.mark.parametrize('world', [2, 3])
@pytest.mark.parametrize('mode,n_records', [('bam_se', 9001), ('bam_pe', 9000), ('bam_pe', 9001), ('bam_pe', 8999)])
def test_byte_range_decoding_equals_the_single_process_result(world, mode, n_records, tmp_path):
    """Every rank decodes its own byte range of one BAM file (small BGZF blocks: records and PE pairs
    straddle the range cuts; odd record totals; odd numbers of records before a cut) -- the reduced
    counts, log counters and the record total equal the single-reader result."""
    feats = dg.make_features(51, 800, span=300000)
    recs = dg.make_records(52, n_records, feats, span=300000, paired=mode == 'bam_pe')
    recs = [r for r in recs if r['end'] > r['pos']] if mode == 'bam_se' else recs
    path = str(tmp_path / 'in.bam')
    bam.write_bam(path, ListSource(recs).reference_names, recs, block_bytes=3011)
    out = _run(world, [mode, path, len(recs)])
    if mode == 'bam_pe' and world == 3:
        assert 'boundar' in out
