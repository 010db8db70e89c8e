"""Third arbiter for the (unpinned) forward pass: the same chain on PyTorch's CPU kernels.

TEST INFRASTRUCTURE ONLY (same rule as ``fnn_oracle.py``): imported by ``tests/`` and by
``bench.py``'s ``cpu_baseline`` leg, never by the product package.

Why it exists.  The reference's forward arithmetic lives in TensorFlow 1.x CPU kernels
(``tf.matmul`` -> ``tf.add`` -> ``tf.nn.tanh`` x 4 -> ``tf.argmax``; call sites
src/fnn_viterbi.py:495-508,606-610), which cannot be run here: **parity unpinned**.
numpy (``fnn_oracle.forward``) and C (``fnn_oracle.c``) are two restatements by the same author;
PyTorch's CPU GEMM/tanh/argmax kernels (MKL/oneDNN sgemm, Sleef tanh, first-max argmax) are the only
*independent* fp32 implementation of those four ops importable in this image, so every
implementation (numpy fp32, C fp32, torch-CPU fp32, HIP) is compared against the float64 chain
under the same near-tie rule, and their pairwise class-flip rates are reported (docs/HISTORY.md section 2; summary in DESIGN.md section 2).
This triangulates; it does not pin: TensorFlow's Eigen kernels are a fifth implementation nobody
here can run.

BASELINE.md section 3 item 1 also names this chain (one call over the whole test set, mirroring the
reference's single full-set ``Session.run``, src/fnn_viterbi.py:610) as the CPU stand-in to time.
"""
from __future__ import annotations

import numpy as np
import torch


def forward(x, weights, biases, dtype=torch.float32, activation="tanh") -> torch.Tensor:
    """Op order of src/fnn_viterbi.py:495-508 per layer: matmul, then add, then tanh — three
    separate ops, each materialising ``[N, n]`` (no fused addmm), the output layer tanh'd too."""
    h = torch.as_tensor(np.asarray(x)).to(dtype)
    for W, b in zip(weights, biases):
        z = torch.matmul(h, torch.as_tensor(np.asarray(W)).to(dtype))
        z = torch.add(z, torch.as_tensor(np.asarray(b)).to(dtype))
        h = torch.tanh(z) if activation == "tanh" else torch.relu(z)
    return h


def decode(x, weights, biases, dtype=torch.float32, activation="tanh"):
    """-> (cls int64 [N] numpy, out [N, C] numpy).  ``torch.argmax`` on CPU returns the first
    maximum (the np.argmax rule the reference uses on labels, src/fnn_viterbi.py:635)."""
    with torch.no_grad():
        out = forward(x, weights, biases, dtype, activation)
        cls = torch.argmax(out, dim=1)
    return cls.numpy().astype(np.int64), out.numpy()


def n_threads() -> int:
    return int(torch.get_num_threads())
