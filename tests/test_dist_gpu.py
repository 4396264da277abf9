"""The RCCL leg of the multi-GPU bulk path on the one-GPU box: a one-rank `nccl` process group is still RCCL, so the
in-place all-reduce of libtecount's HBM count vector (dist.World.sum_device_counts -- what measure._bulk and bench.py do
at N > 1) runs for real: wrapping the foreign device pointer, the collective, the synchronisation, the read-back."""

import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_SCRIPT = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
import numpy as np
import torch, torch.distributed as dist
import datagen as dg
from helpers import units_from_records
from te_counter_amd import Engine, FlatIndex
from te_counter_amd.dist import World

torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
w = World(0, 1, 0)
w.backend = dist.get_backend()
assert w.backend == 'nccl'
feats = dg.make_features(5, 3000, span=600000)
flat = FlatIndex.from_features(feats)
eng = Engine(0)
eng.upload_index(flat)
units, _ = units_from_records(flat, dg.make_records(6, 50000, feats, span=600000, paired=False), False)
eng.bulk_begin(20); eng.bulk_submit(*units); plain, pst = eng.bulk_finish()
eng.bulk_begin(20); eng.bulk_submit(*units); eng.bulk_submit(*units)
eng.bulk_sync()
w.sum_device_counts(eng.bulk_device_buffers()[0], eng.n_names)      # one rank: the sum is the vector itself
got, gst = eng.bulk_finish()
assert np.array_equal(got, 2 * plain), 'count vector damaged by the in-place all-reduce'
total, sv = w.sum_counts(None, [gst['assigned'], 7], got)
assert total is got and list(sv) == [gst['assigned'], 7]
eng.close()
dist.destroy_process_group()
print('RCCL_IN_PLACE_OK', int(plain.sum()))
'''


def test_in_place_rccl_all_reduce_of_the_count_vector(tmp_path):
    script = tmp_path / 'rccl_one_rank.py'
    script.write_text(_SCRIPT)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29561', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and 'RCCL_IN_PLACE_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
