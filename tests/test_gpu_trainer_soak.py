"""150 random shapes of the training kernels K7 (csrc/nnv_train.hip) and K8 (csrc/nnv_dense_train.hip) against their
float64 restatements (tests/soak_trainers.py; a 3,000-case run is recorded in profiles/r05/soak_trainers.txt)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_random_training_shapes_stay_within_the_bounds():
    import soak_trainers
    worst, flagged = soak_trainers.run(150, seed=11, verbose=False)
    assert not flagged, flagged
    assert worst["k8_grad"] < 2e-6 and worst["k7_grad"] < 2e-6 and worst["k8_cost"] < 5e-6
