"""TensorFlow-1.x's CPU tanh, restated: Eigen's ``generic_fast_tanh_float``.

TEST INFRASTRUCTURE ONLY (same rule as ``fnn_oracle.py``).

Why.  The reference's ``tf.nn.tanh`` (src/fnn_viterbi.py:496,500,504,508) runs, on CPU, Eigen's
``scalar_tanh_op<float>``, which the Eigen snapshot TensorFlow 1.x vendors (``third_party/eigen3``)
implements as ``internal::generic_fast_tanh_float`` (``Eigen/src/Core/MathFunctionsImpl.h``): clamp
the argument to [-9, 9], then a 13th-degree odd over 6th-degree even rational function.  It is NOT
libm's tanh: it is not monotone at the ulp level, it returns values up to 1 + 2^-22, and its steps
just below 1 sit elsewhere (f(8.5) = 1.0 but f(8.8) = 1 - 2^-24).  So which of two classes within
~4e-7 of each other TensorFlow's ``argmax`` picks depends on this function's rounding noise: the
reason the parity contract is a near-tie RULE, and further evidence that the
forward pass cannot be pinned bit for bit without TensorFlow itself (**parity unpinned**).

The restatement is self-checked (``tests/test_oracle_golden.py``): over [-10, 10] it stays within
4e-7 of float64 tanh, which a mistyped coefficient would break by orders of magnitude.  TensorFlow's
pip builds of that era were compiled without FMA, so ``pmadd`` is evaluated as multiply-then-add in
float32, as here.  The GEMM feeding it is Eigen's blocked gebp kernel; for depths inside one k-block
(all of this path's layers) it keeps one accumulator per output and walks k in order, which
``matmul_seq_k`` restates (product rounded, then added: no FMA); ``forward_tf_like(seq_k=True)`` is the
closest statement of TensorFlow-1.x's CPU forward pass available without TensorFlow, numpy's BLAS matmul
(``seq_k=False``) the quick stand-in.  Neither is a pin: thread partitioning, packet width and
compiler flags of the actual build are not known.

Version caveat (the clamp).  This file fixes the clamp at +-9, the constant of the Eigen snapshots of the
TensorFlow 1.0 - 1.13 era.  Later Eigen revisions (vendored by TF 1.14/1.15 and 2.x) clamp at
+-7.90531110763549805 instead ("the smallest |x| for which the rational function already rounds to +-1 in
float32") and keep the same coefficients.  Which of the two the reference's user ran depends on the
TensorFlow 1.x wheel they had (requirements.txt:2 pins a range that does not even contain a working one), so
the constant is a parameter of the arbiter, not a known fact about the reference: ``tanh(a, clamp=...)``
takes either, the self-check runs for both, and the class-flip tables of docs/HISTORY.md section 2 quote +-9.  The
two differ only for |x| in (7.905, 9], where the late one returns exactly +-1 and the early one values within
4.5e-7 of it (1 - 2^-24 ... 1 + 2^-22) -- inside the flat
top that the decode kernel's kFlatBand path (>= 7.9) already treats as ties."""
from __future__ import annotations

import numpy as np

_F = np.float32
_ALPHA = [_F(v) for v in (4.89352455891786e-03, 6.37261928875436e-04, 1.48572235717979e-05,
                          5.12229709037114e-08, -8.60467152213735e-11, 2.00018790482477e-13,
                          -2.76076847742355e-16)]          # numerator, x^1, x^3, ..., x^13
_BETA = [_F(v) for v in (4.89352518554385e-03, 2.26843463243900e-03, 1.18534705686654e-04,
                         1.19825839466702e-06)]            # denominator, x^0, x^2, x^4, x^6


CLAMP_TF_EARLY = 9.0                      # Eigen snapshots vendored by TF 1.0 - 1.13
CLAMP_TF_LATE = 7.90531110763549805       # Eigen from 2019 on (TF 1.14+)


def tanh(a, clamp=CLAMP_TF_EARLY):
    """float32 in, float32 out; every intermediate rounded to float32 (mul, then add)."""
    x = np.clip(np.asarray(a, dtype=_F), _F(-clamp), _F(clamp))
    x2 = x * x
    p = x2 * _ALPHA[6] + _ALPHA[5]
    for k in (4, 3, 2, 1, 0):
        p = x2 * p + _ALPHA[k]
    p = x * p
    q = x2 * _BETA[3] + _BETA[2]
    q = x2 * q + _BETA[1]
    q = x2 * q + _BETA[0]
    return (p / q).astype(_F)


def matmul_seq_k(h, W):
    """C = h W in float32 with ONE accumulator per output, k ascending, every product rounded to
    float32 and then added (no FMA): the order of Eigen's gebp micro-kernel (TensorFlow's CPU MatMul)
    when the whole depth fits one k-block (Eigen blocks k at 256 or more; the networks here have
    K <= 128) in a build without FMA, as TensorFlow 1.x's pip wheels were.  O(K) numpy passes."""
    h = np.asarray(h, dtype=_F)
    W = np.asarray(W, dtype=_F)
    acc = np.zeros((h.shape[0], W.shape[1]), dtype=_F)
    for k in range(W.shape[0]):
        acc += h[:, k:k + 1] * W[k:k + 1, :]              # float32 product, float32 add
    return acc


def forward_tf_like(x, weights, biases, seq_k=False):
    """matmul -> add -> Eigen tanh per layer, all float32 (op order of src/fnn_viterbi.py:495-508).
    ``seq_k``: the matmul as ``matmul_seq_k`` (closest restatement of TF-1.x's CPU arithmetic we can
    state) instead of numpy's BLAS call."""
    h = np.asarray(x, dtype=_F)
    for W, b in zip(weights, biases):
        z = matmul_seq_k(h, W) if seq_k else np.matmul(h, np.asarray(W, dtype=_F))
        h = tanh(z + np.asarray(b, dtype=_F))
    return h


def decode(x, weights, biases, seq_k=False):
    out = forward_tf_like(x, weights, biases, seq_k=seq_k)
    return np.argmax(out, axis=1).astype(np.int64), out
