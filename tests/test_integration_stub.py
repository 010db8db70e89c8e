"""INTEGRATION.md's ctypes stub, executed verbatim: the binding a maintainer of the reference would add
(no torch, no package import) decodes like the oracle."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(# nnviterbi_ffi\.py.*?)```", text, re.S)
    assert m, "INTEGRATION.md lost its stub"
    return m.group(1)


def test_stub_is_valid_python():
    compile(_stub_source(), "nnviterbi_ffi.py", "exec")


@pytest.mark.gpu
def test_stub_decodes_like_the_oracle(cuda_device):
    import nnviterbi_amd
    from nnviterbi_amd import fnn_viterbi as fv
    from oracle import fnn_oracle as O
    pkg = os.path.dirname(nnviterbi_amd.library_path())
    src = _stub_source().replace('"libnnviterbi_hip.so"', repr(os.path.join(pkg, "libnnviterbi_hip.so")))
    ns = {}
    exec(compile(src, "nnviterbi_ffi.py", "exec"), ns)
    cfg = fv.DecoderConfig(block_length=7)
    w, b = fv.initialize_weights(cfg, rng=np.random.RandomState(2))
    Ws, bs = fv.weights_as_lists(w, b)
    x = (np.random.RandomState(3).randn(5001, 14) * 1.2).astype(np.float32)
    pred = ns["decode"](ns["prepack"](Ws, bs), x, 7)
    assert pred.dtype == np.int64 and pred.shape == (5001,)
    ref = O.decode_c(x, Ws, bs, 7, want_bits=False)["cls"]
    assert (pred != ref).sum() <= 1
    labels = np.random.RandomState(4).randint(0, 128, 5001)
    pred2, counters = ns["decode"](ns["prepack"](Ws, bs), x, 7, labels=labels)
    assert np.array_equal(pred2, pred)
    assert np.array_equal(counters.astype(np.int64), O.error_counters(pred, labels, 7))
