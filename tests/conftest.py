import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'backpacks-flash-attn_amd')
for p in (ROOT, PKG, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# the CPU oracle on a many-core host: more than ~16 intra-op threads only adds synchronisation cost
# (scripts/cpu_threads_probe.py: 256 threads ran the fp32 forward 50x slower than 16)
torch.set_num_threads(min(16, os.cpu_count() or 1))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def from_bits16(arr):
    """uint16 bf16 bit patterns (tests/golden/make_golden.py:bits16) -> fp32 tensor."""
    return torch.from_numpy(np.ascontiguousarray(arr)).view(torch.bfloat16).float()


@pytest.fixture(scope='session')
def golden():
    return load_golden
