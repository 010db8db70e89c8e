#!/usr/bin/env python3
"""Forward-pass fixtures.  The reference cannot run its forward pass here
(TensorFlow is absent) and ships no vectors for it, so these are
produced by the oracle's float64 numpy restatement of
src/fnn_viterbi.py:494-510 on seeded weights and inputs ("parity unpinned":
they pin THIS project's implementations to each other, not to TensorFlow).

    python tests/golden/make_forward_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import nnviterbi_amd  # noqa
from nnviterbi_amd import fnn_viterbi as fv  # noqa
from oracle import fnn_oracle as O  # noqa

CASES = {
    # name: (block_length, n_layers, rows, bias scale)
    "b8_l3": (8, 3, 768, 0.1),      # BASELINE config 3 row shape 16-128-64-32-256
    "b7_l3": (7, 3, 256, 0.0),      # reference default 14-128-64-32-128, zero biases as initialised
    "b10_l2": (10, 2, 96, 0.1),     # config 2 row shape 20-128-64-1024
    "b3_l3": (3, 3, 64, 0.1),       # 8 classes: exercises class-count masking
}


def main():
    out = {}
    for name, (b, nl, rows, bscale) in CASES.items():
        cfg = fv.DecoderConfig(block_length=b)
        cfg.n_layers = nl
        rng = np.random.RandomState(1000 + b * 10 + nl)
        weights, biases = fv.initialize_weights(cfg, rng=rng)
        Ws, bs = fv.weights_as_lists(weights, biases)
        bs = [(rng.randn(*v.shape) * bscale).astype(np.float32) for v in bs]
        cb = fv.generate_class_bits(b)
        labels = rng.randint(0, 2 ** b, size=rows).astype(np.int32)
        enc = fv.encode_133171_batch(cb) if b >= 6 else fv.encode_57_batch(cb)
        x = (fv._bpsk(enc[labels]) + rng.randn(rows, 2 * b) * cfg.sigma).astype(np.float32)
        cls64, bits, o64 = O.decode(x, Ws, bs, b, np.float64)
        for i, (W, v) in enumerate(zip(Ws, bs)):
            out[f"{name}_W{i}"] = W
            out[f"{name}_b{i}"] = v
        out[f"{name}_x"] = x
        out[f"{name}_labels"] = labels
        out[f"{name}_cls"] = cls64
        out[f"{name}_margin"] = O.top2_margin(o64)
        out[f"{name}_out_head"] = o64[:16]
        out[f"{name}_counters"] = O.error_counters(cls64, labels, b)
    np.savez_compressed(os.path.join(HERE, "forward_fp64.npz"), **out)
    print("wrote", os.path.getsize(os.path.join(HERE, "forward_fp64.npz")), "bytes")


if __name__ == "__main__":
    main()
