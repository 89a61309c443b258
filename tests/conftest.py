import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are skipped on a box without a GPU (a plain `pytest tests` then passes on CPU).  On a box WITH a
    GPU nothing is skipped: a missing libpysfm_ba.so must fail there, loudly."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        # every problem of every GPU test starts from NaNs in all LDS and workspace buffers (a read of something nobody
        # wrote must show, not depend on what ran before)
        from pysfm_amd.backend import HipBackend
        HipBackend.poison_after_set_problem = True
        # a kernel that waits for something that never comes must fail a test, not hang the box (pytest-timeout)
        for item in items:
            if 'gpu' in item.keywords:
                item.add_marker(pytest.mark.timeout(600))
        return
    skip = pytest.mark.skip(reason='needs an MI355X (no GPU visible)')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz')))


@pytest.fixture
def golden():
    return load_golden
