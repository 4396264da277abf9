"""Multi-process (world_size 2 and 3) CPU tests of the sharding + reduce plumbing over gloo."""

import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('world,mode,batch', [(2, 'se', 1000), (2, 'pe', 998), (3, 'pe', 4096), (2, 'pe', 7)])
def test_sharded_counts_reduce_to_the_single_process_result(world, mode, batch):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), TEC_NO_TORCH='')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dist_worker.py'), mode, str(batch)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert ('OK %d ' % world) in outs[0][0], outs[0]
