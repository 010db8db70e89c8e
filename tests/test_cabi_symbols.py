"""The C-ABI shared library loads and exports every symbol include/nnviterbi.h
declares; argument validation paths that need no GPU.  CPU only: no kernel is
launched here."""
import ctypes
import os
import re

import pytest

import nnviterbi_amd
from nnviterbi_amd import _cabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nnviterbi.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nnv_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_cabi.library_path()):
        _cabi.build_library()
    return _cabi.load_library()


def test_header_and_binding_list_the_same_entry_points():
    assert declared_functions() == sorted(_cabi.EXPORTED_SYMBOLS)


def test_every_declared_symbol_is_exported(lib):
    for name in declared_functions():
        assert getattr(lib, name) is not None, name


def test_header_cites_the_reference_for_each_compute_entry():
    text = open(HEADER).read()
    for needle in ("src/fnn_viterbi.py:494-510", "606-610", "src/fnn_viterbi.py:648-651",
                   "src/fnn_viterbi.py:122-154", "src/fnn_viterbi.py:157-217",
                   "src/fnn_viterbi_mixed_snr.py"):
        assert needle in text, needle


def header_abi_version():
    return int(re.search(r"^#define\s+NNV_ABI_VERSION\s+(\d+)", open(HEADER).read(), re.M).group(1))


def test_version_and_status_strings(lib):
    assert lib.nnv_abi_version() == _cabi.NNV_ABI_VERSION == header_abi_version()
    assert lib.nnv_status_string(0) == b"ok"
    assert b"unsupported" in lib.nnv_status_string(2)
    assert lib.nnv_device_count() >= 0


def test_argument_validation_without_a_device(lib):
    h = ctypes.c_void_p()
    assert lib.nnv_mlp_prepack(None, ctypes.byref(h)) == 1            # INVALID_ARGUMENT
    assert lib.nnv_mlp_decode_f32(None, None, 10, 8, None, None, None, None, None, None) == 1
    assert lib.nnv_mlp_forward_f32(None, None, 10, None, None) == 1
    assert lib.nnv_mlp_free(None) == 0
    sig = (ctypes.c_float * 1)(0.5)
    # (133,171) tiles shorter than 6 bits / (7,5) shorter than 2: an error, as in the reference
    assert lib.nnv_awgn_generate_f32(1, 5, 10, 0, 1, sig, 1, 0, None, None, None) == 1
    assert lib.nnv_awgn_generate_f32(0, 1, 10, 0, 1, sig, 1, 0, None, None, None) == 1
    assert lib.nnv_awgn_generate_f32(1, 8, 0, 0, 1, sig, 1, 0, None, None, None) == 0   # empty: no-op


def test_unsupported_shape_is_reported_before_touching_the_device(lib):
    import numpy as np
    d = _cabi.MlpDesc()
    d.depth = 2
    dims = [16, 4096, 256]                     # hidden width with no kernel instantiation
    Ws = [np.zeros((dims[i], dims[i + 1]), np.float32) for i in range(2)]
    bs = [np.zeros(dims[i + 1], np.float32) for i in range(2)]
    for i, v in enumerate(dims):
        d.dims[i] = v
    for i in range(2):
        d.W_host[i] = Ws[i].ctypes.data
        d.b_host[i] = bs[i].ctypes.data
    h = ctypes.c_void_p()
    assert lib.nnv_mlp_prepack(ctypes.byref(d), ctypes.byref(h)) == 2  # UNSUPPORTED_SHAPE
    d.depth = 7
    assert lib.nnv_mlp_prepack(ctypes.byref(d), ctypes.byref(h)) == 2


def test_device_rule_status_codes(lib):
    """nnv_check_device is the rule every launch entry applies to its handle's device first.
    Without a HIP device it reports NO_DEVICE; with one, the current device is OK and any other
    index is INVALID_ARGUMENT (so a handle packed for another GPU is refused, not launched)."""
    if lib.nnv_device_count() == 0:
        assert lib.nnv_check_device(0) == 3                  # NNV_ERR_NO_DEVICE
    else:
        import torch
        cur = torch.cuda.current_device()
        assert lib.nnv_check_device(cur) == 0
        assert lib.nnv_check_device(cur + 1) == 1            # NNV_ERR_INVALID_ARGUMENT
    assert b"another device" in lib.nnv_status_string(1)


def test_graft_entry_build_runs_end_to_end():
    """The driver's "does it build" check: __graft_entry__.build() compiles the HIP library and the
    oracle's C restatement, loads the library and compares its ABI with the header and the binding.
    Run as the driver runs it (a fresh interpreter), so a stale literal in the entry point fails here
    and not at round end."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "__graft_entry__.py")], cwd=ROOT,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    import __graft_entry__ as entry
    assert entry.header_abi_version() == header_abi_version() == _cabi.NNV_ABI_VERSION
