import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np
    arrays = np.load(os.path.join(GOLDEN, "reference_numpy_half.npz"), allow_pickle=False)
    with open(os.path.join(GOLDEN, "reference_numpy_half.json")) as f:
        meta = json.load(f)
    return arrays, meta


@pytest.fixture(scope="session")
def forward_golden():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "forward_fp64.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def cuda_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    torch.cuda.set_device(0)
    return torch.device("cuda", 0)
