"""CPU oracle for the feed-forward neural Viterbi decode path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module, and only as the
checker / the reported CPU baseline.  The product package never imports it.

Parity status
-------------
* Forward pass (matmul -> +bias -> tanh x4 -> argmax -> bits): **parity
  unpinned**.  The arithmetic lives in TensorFlow 1.x, which is not vendored
  in the reference and not installed here; the reference's
  own tests pin nothing at that boundary.  This file restates
  ``src/fnn_viterbi.py:494-510`` (forward), ``:606-610`` (argmax axis 1),
  ``:648-651`` (MSB-first bits) in numpy, in fp32 (``forward``) and in fp64
  (the arbiter for near-ties), and ``fnn_oracle.c`` restates it again in C
  with an explicit k-ordered fmaf chain.  Further arbiters: ``torch_cpu.py``
  (PyTorch's CPU kernels, the independent implementation) and ``eigen_tanh.py``
  (the rational function TensorFlow 1.x's CPU tanh evaluates, restated).
* numpy half (encoders, channel, codebook, metrics): **pinned** against
  golden vectors generated from the reference itself
  (``tests/golden/reference_numpy_half.*``, ``tests/test_oracle_golden.py``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# --------------------------------------------------------------------------
# forward / decode  (src/fnn_viterbi.py:494-510, :606-610, :648-651)
# --------------------------------------------------------------------------


def forward(x, weights, biases, dtype=np.float32, activation="tanh"):
    """tanh(...tanh(tanh(x W1 + b1) W2 + b2)... Wout + bout), all in ``dtype``.

    ``weights[l]`` is ``[n_l, n_{l+1}]`` (right-multiply, reference layout
    src/fnn_viterbi.py:432-456), ``biases[l]`` is ``[n_{l+1}]``.  Op order per
    layer is matmul, add, tanh (src/fnn_viterbi.py:495-496); the output layer
    is tanh'd as well (:507-508).
    """
    h = np.asarray(x, dtype=dtype)
    for W, b in zip(weights, biases):
        z = np.matmul(h, np.asarray(W, dtype=dtype))
        z = z + np.asarray(b, dtype=dtype)
        # tanh: src/fnn_viterbi.py:496; relu: src/fnn_viterbi_bitwise.py:388,397
        h = np.tanh(z) if activation == "tanh" else np.maximum(z, np.asarray(0, dtype=dtype))
    return h


def argmax_first(out):
    """tf.argmax(pred, 1) (src/fnn_viterbi.py:607) with np.argmax's
    first-maximum rule (the convention the reference uses on labels, :635)."""
    return np.argmax(out, axis=1).astype(np.int64)


def class_to_bits(cls, b):
    """np.binary_repr(pred, width=b) per row: MSB first
    (src/fnn_viterbi.py:648-651)."""
    cls = np.asarray(cls).astype(np.int64)
    shifts = np.arange(b - 1, -1, -1, dtype=np.int64)
    return ((cls[:, None] >> shifts[None, :]) & 1).astype(np.uint8)


def decode(x, weights, biases, b, dtype=np.float32, activation="tanh"):
    out = forward(x, weights, biases, dtype, activation)
    cls = argmax_first(out)
    return cls, class_to_bits(cls, b), out


def error_counters(cls, labels, b):
    """{bit_errors, frame_errors, total_bits, total_frames}.

    FER numerator: rows with pred != label (src/fnn_viterbi.py:639);
    BER numerator: popcount(pred XOR label) summed (src/fnn_viterbi.py:654-667
    compares MSB-first bit vectors, src/fnn_viterbi_mixed_snr.py:326-328 does
    the same through an integer XOR)."""
    cls = np.asarray(cls).astype(np.int64)
    labels = np.asarray(labels).astype(np.int64)
    diff = cls ^ labels
    pop = np.zeros(diff.shape, dtype=np.int64)
    d = diff.copy()
    for _ in range(max(int(b), 1) + 1):
        pop += d & 1
        d >>= 1
    return np.array([int(pop.sum()), int((diff != 0).sum()),
                     int(len(cls)) * int(b), int(len(cls))], dtype=np.int64)


def top2_margin(out64):
    """fp64 gap between the largest and second-largest output of each row."""
    part = np.partition(out64, -2, axis=1)
    return part[:, -1] - part[:, -2]


# --------------------------------------------------------------------------
# numpy half: encoders, channel, codebook (pinned by golden vectors)
# --------------------------------------------------------------------------

# tap offsets d such that output = sum_d u[i-d] mod 2, u[<0] = 0
_TAPS_133 = (0, 2, 3, 5, 6)   # 133 octal, src/fnn_viterbi.py:207-210
_TAPS_171 = (0, 1, 2, 3, 6)   # 171 octal, src/fnn_viterbi.py:212-215


def encode_133171(bits):
    """src/fnn_viterbi.py:157-217.  Zero start state, no tail; output pair i is
    (g133, g171).  The reference hard-codes pairs 0..5 (they equal the
    zero-state convolution) and therefore indexes bits[5] unconditionally:
    fewer than 6 bits raises IndexError."""
    bits = np.asarray(bits)
    n = len(bits)
    if n < 6:
        raise IndexError("encode_133171 needs at least 6 bits")
    out = np.zeros(2 * n, dtype=np.int32)
    for i in range(n):
        a = sum(int(bits[i - d]) for d in _TAPS_133 if i - d >= 0)
        c = sum(int(bits[i - d]) for d in _TAPS_171 if i - d >= 0)
        out[2 * i] = a % 2
        out[2 * i + 1] = c % 2
    return out


def encode_57(bits):
    """src/fnn_viterbi.py:122-154.  Emits (5,7) order: even = u[i]+u[i-2],
    odd = u[i]+u[i-1]+u[i-2]; output index 1 is u[0]+u[1] (uses the NEXT bit,
    :143), and index 0 is u[0] without the mod-2 (:142).  Fewer than 2 bits
    raises IndexError (:143)."""
    bits = np.asarray(bits)
    n = len(bits)
    if n < 2:
        raise IndexError("encode_57 needs at least 2 bits")
    out = np.zeros(2 * n, dtype=np.int32)
    for i in range(n):
        u0 = int(bits[i])
        u1 = int(bits[i - 1]) if i >= 1 else 0
        u2 = int(bits[i - 2]) if i >= 2 else 0
        out[2 * i] = (u0 + u2) % 2
        out[2 * i + 1] = (u0 + u1 + u2) % 2
    out[0] = int(bits[0])
    out[1] = (int(bits[0]) + int(bits[1])) % 2
    return out


def generate_class_bits(b):
    """src/fnn_viterbi.py:260-287: row i = MSB-first binary of i."""
    return class_to_bits(np.arange(2 ** b), b)


def bpsk(encoded):
    """0 -> -1, everything else unchanged (src/fnn_viterbi.py:244-247)."""
    m = np.asarray(encoded).astype(np.float32)
    m[m == 0] = -1
    return m


def modulate_awgn(encoded, sigma):
    """src/fnn_viterbi.py:224-253: float32 BPSK symbols + float64
    np.random.normal(0, sigma, n) -> float64 result; one RNG call per row."""
    m = bpsk(encoded)
    return m + np.random.normal(0, sigma, len(m))


def sigma_from_snr_db(snr_db):
    """src/fnn_viterbi.py:89-92: Es/N0 per coded symbol, no rate term."""
    return float(np.sqrt(0.5 / (10.0 ** (snr_db / 10.0))))


# --------------------------------------------------------------------------
# C restatement (fnn_oracle.c) through ctypes
# --------------------------------------------------------------------------

_LIB = None


def build_c(force=False):
    so = os.path.join(_HERE, "libfnn_oracle.so")
    src = os.path.join(_HERE, "fnn_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s", "-B", "libfnn_oracle.so"],
                       check=True)
    return so


def load_c():
    global _LIB
    if _LIB is None:
        so = build_c()
        try:
            lib = ctypes.CDLL(so)
        except OSError:
            lib = ctypes.CDLL(build_c(force=True))
        lib.fnn_oracle_decode_f32.restype = ctypes.c_int
        lib.fnn_oracle_decode_f32.argtypes = [
            ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
        lib.fnn_oracle_max_threads.restype = ctypes.c_int
        _LIB = lib
    return _LIB


def _ptr_array(arrs):
    if arrs is None:
        return None, None
    PA = ctypes.c_void_p * len(arrs)
    return PA(*[None if a is None else a.ctypes.data for a in arrs]), arrs


def decode_c(x, weights, biases, b, labels=None, korder=None, want_act=False,
             want_bits=True, n_threads=0, activation="tanh"):
    """Run fnn_oracle.c.  Returns dict(cls, bits, act, counters)."""
    lib = load_c()
    x = np.ascontiguousarray(x, dtype=np.float32)
    Ws = [np.ascontiguousarray(W, dtype=np.float32) for W in weights]
    bs = [np.ascontiguousarray(v, dtype=np.float32) for v in biases]
    dims = np.array([Ws[0].shape[0]] + [W.shape[1] for W in Ws], dtype=np.int32)
    n = x.shape[0]
    assert x.ndim == 2 and x.shape[1] == dims[0]
    cls = np.empty(n, dtype=np.int64)
    bits = np.empty((n, b), dtype=np.uint8) if want_bits else None
    act = np.empty((n, int(dims[-1])), dtype=np.float32) if want_act else None
    lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.int32)
    counters = np.zeros(4, dtype=np.int64) if lab is not None else None
    ko = None
    if korder is not None:
        ko = [None if k is None else np.ascontiguousarray(k, dtype=np.int32)
              for k in korder]
    Wp, _w = _ptr_array(Ws)
    bp, _b = _ptr_array(bs)
    kp, _k = _ptr_array(ko)
    rc = lib.fnn_oracle_decode_f32(
        x.ctypes.data, n, len(Ws), dims.ctypes.data, Wp, bp, kp, int(b),
        None if lab is None else lab.ctypes.data, cls.ctypes.data,
        None if bits is None else bits.ctypes.data,
        None if act is None else act.ctypes.data,
        None if counters is None else counters.ctypes.data, int(n_threads),
        1 if activation == "relu" else 0)
    if rc != 0:
        raise RuntimeError(f"fnn_oracle_decode_f32 failed: {rc}")
    return {"cls": cls, "bits": bits, "act": act, "counters": counters}


def c_max_threads(cap=16):
    """Threads for the C oracle: the CPUs this process may use (affinity and
    cgroup quota), at most ``cap`` (a one-GPU box's CPU share is 16)."""
    n = int(load_c().fnn_oracle_max_threads())
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.999)))
    except (OSError, ValueError):
        pass
    env = os.environ.get("NNV_CPU_THREADS")
    if env:
        return max(1, int(env))
    return max(1, min(n, cap) if cap else n)
