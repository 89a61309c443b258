"""The C-ABI shared library loads and exports every symbol include/pysfm_ba.h declares,
and the product path fails loudly without a GPU / without the library.  No compute calls."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT
from pysfm_amd import _capi as capi

HEADER = os.path.join(ROOT, 'include', 'pysfm_ba.h')


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(capi.LIB_PATH):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'pysfm_amd', 'csrc')])
    return capi.load()


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ba_[a-z_]+)\s*\(', src)))


def test_header_declares_what_python_binds(lib):
    names = declared_functions()
    assert len(names) >= 25
    assert sorted(capi.PROTOTYPES) == names
    for n in names:
        assert hasattr(lib, n), 'libpysfm_ba.so does not export %s' % n


def test_exports_are_plain_c_symbols(lib):
    out = subprocess.check_output(['nm', '-D', '--defined-only', capi.LIB_PATH]).decode()
    exported = set(re.findall(r' T (ba_[a-z_]+)$', out, flags=re.M))
    assert set(declared_functions()) <= exported


def test_version_and_kernel_names(lib):
    assert b'gfx950' in lib.ba_version()
    names = [lib.ba_kernel_name(i).decode() for i in range(capi.K_COUNT)]
    assert names[0] == 'k_cost' and 'k_schur_pairs' in names and lib.ba_kernel_name(99) == b'?'


def test_library_embeds_gfx950_code_object():
    blob = open(capi.LIB_PATH, 'rb').read()
    assert b'gfx950' in blob and b'k_schur_pairs' in blob


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_have_gpu(), reason='a GPU is visible')
def test_no_gpu_is_a_loud_error(lib):
    h = ctypes.c_void_p()
    rc = lib.ba_create(0, ctypes.byref(h))
    assert rc == capi.BA_ERR_NO_DEVICE and not h.value
    assert b'no HIP device' in lib.ba_last_error(None)
    from pysfm_amd.backend import HipBackend
    with pytest.raises(capi.HipDeviceError):
        HipBackend(0)
    import numpy as np
    from pysfm_amd import Bundle, BundleAdjuster
    b = Bundle.FromObservations(np.eye(3), np.stack([np.eye(3)] * 2), np.zeros((2, 3)), np.ones((1, 3)),
                                [0, 1], [0, 0], np.zeros((2, 2)))
    with pytest.raises(capi.HipDeviceError):          # no silent CPU fallback anywhere
        BundleAdjuster(b, verbose=False)


def test_missing_library_is_a_loud_error(monkeypatch):
    monkeypatch.setattr(capi, '_lib', None)
    monkeypatch.setattr(capi, 'LIB_PATH', '/nonexistent/libpysfm_ba.so')
    with pytest.raises(capi.HipLibraryMissing):
        capi.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'pysfm_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.h', '.hip', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), f
                assert 'ba_oracle' not in text, f
