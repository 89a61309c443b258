import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


GPU_TEST_TIMEOUT_S = 600


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # hang protection of the GPU tests: pytest-timeout (tests/requirements.txt) - or, where that plugin is not installed,
    # a SIGALRM of our own around every gpu test, so that the protection does not silently go away
    config._ba_own_alarm = not config.pluginmanager.hasplugin('timeout')
    if config._ba_own_alarm:
        config.addinivalue_line('markers', 'timeout(seconds): per-test time limit (fallback implementation in tests/conftest.py)')


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    import signal
    own = getattr(item.config, '_ba_own_alarm', False) and 'gpu' in item.keywords and hasattr(signal, 'SIGALRM')
    if own:
        def on_alarm(signum, frame):
            raise TimeoutError('gpu test exceeded %d s (a kernel waiting for something that never comes?)' % GPU_TEST_TIMEOUT_S)
        old = signal.signal(signal.SIGALRM, on_alarm)
        signal.alarm(GPU_TEST_TIMEOUT_S)
    try:
        yield
    finally:
        if own:
            signal.alarm(0)
            signal.signal(signal.SIGALRM, old)


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are skipped on a box without a GPU (a plain `pytest tests` then passes on CPU).  On a box WITH a
    GPU nothing is skipped: a missing libpysfm_ba.so must fail there, loudly."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    # `-x` must reach every golden-fixture test (the ones that pin the HIP path to the reference's own numbers: fast) before any
    # full-size, statistics-flavoured one: parity -> resident -> fuzz -> configs; everything that needs no GPU first
    order = {'test_gpu_parity.py': 1, 'test_gpu_resident.py': 2, 'test_gpu_fuzz.py': 3, 'test_gpu_configs.py': 4}
    items.sort(key=lambda it: order.get(os.path.basename(str(it.fspath)), 0))          # (stable: the order inside a file stays)
    if have_gpu:
        # every problem of every GPU test starts from NaNs in all LDS and workspace buffers (a read of something nobody
        # wrote must show, not depend on what ran before)
        from pysfm_amd.backend import HipBackend
        HipBackend.poison_after_set_problem = True
        # a kernel that waits for something that never comes must fail a test, not hang the box (pytest-timeout)
        for item in items:
            if 'gpu' in item.keywords:
                item.add_marker(pytest.mark.timeout(GPU_TEST_TIMEOUT_S))
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


class GemanMcClure(object):
    """A robustifier the library does not know: cost = |e|^2 / (|e|^2 + s^2), residual e / sqrt(|e|^2 + s^2) - a plain
    object with the reference's four-method protocol (sensor_model.py:19-32)."""

    def __init__(self, s):
        self.s = s

    def cost_from_error(self, e):
        e = np.asarray(e, float)
        return float(e.dot(e) / (e.dot(e) + self.s ** 2))

    def residual_from_error(self, e):
        e = np.asarray(e, float)
        return e / np.sqrt(e.dot(e) + self.s ** 2)

    def Jresidual_from_error(self, e):
        e = np.asarray(e, float)
        d = e.dot(e) + self.s ** 2
        return np.eye(2) / np.sqrt(d) - np.outer(e, e) / d ** 1.5

    def clone(self):
        return GemanMcClure(self.s)
