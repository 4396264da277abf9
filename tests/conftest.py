import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    with gzip.open(os.path.join(GOLDEN_DIR, name + '.json.gz'), 'rb') as fh:
        return json.loads(fh.read().decode())


def golden_names(prefix):
    return sorted(f[:-len('.json.gz')] for f in os.listdir(GOLDEN_DIR)
                  if f.startswith(prefix) and f.endswith('.json.gz'))


@pytest.fixture(scope='session')
def gpu_available():
    import torch
    return torch.cuda.is_available()
