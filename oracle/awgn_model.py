"""numpy model of the on-device synthetic block generator (K3).

TEST INFRASTRUCTURE ONLY (see fnn_oracle.py for the rules).  K3 is this
project's own harness (the reference draws noise from numpy's MT19937, which
is not reproduced on the device), so this model pins the generator's CONTRACT:

  row r (global index):  Philox4x32-10, key = (seed_lo, seed_hi)
     counter (r_lo, r_hi, 0, 0)  -> word 0, low b bits = class (MSB-first info bits)
     counter (r_lo, r_hi, t, 1)  -> 4 words -> two Box-Muller pairs -> the noise of
                                    coded symbols 4t .. 4t+3
  Box-Muller on 24-bit uniforms: u1 = ((a>>8)+1) 2^-24, u2 = (b>>8) 2^-24,
     n0 = sqrt(-2 ln u1) cos(2 pi u2), n1 = sqrt(-2 ln u1) sin(2 pi u2)
  x = BPSK(encode(bits)) + sigma[block % n_sigma] * n,  block = r // rows_per_block
  continuous mode: encode() runs over the concatenated info bits of the whole block (zero state at the
     block's first row only); classes and noise are unchanged

The integer part (Philox words, class, code bits) is bit-exact; the Gaussian
part is evaluated here in float64 and compared with a tolerance.
Philox known answers (Random123 kat_vectors) are checked in
tests/test_generator_model.py.
"""
from __future__ import annotations

import numpy as np

from . import fnn_oracle as O

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint64(0x9E3779B9)
_W1 = np.uint64(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10; all arguments broadcastable uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(v, dtype=np.uint64) & _MASK for v in (c0, c1, c2, c3)]
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint64(k0) & _MASK
    k1 = np.uint64(k1) & _MASK
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & _MASK
        n1 = p1 & _MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & _MASK
        n3 = p0 & _MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + _W0) & _MASK
        k1 = (k1 + _W1) & _MASK
    return [v.astype(np.uint32) for v in (c0, c1, c2, c3)]


def _box_muller(a, b):
    u1 = ((a >> np.uint32(8)).astype(np.float64) + 1.0) * 2.0 ** -24
    u2 = (b >> np.uint32(8)).astype(np.float64) * 2.0 ** -24
    rad = np.sqrt(-2.0 * np.log(u1))
    return rad * np.cos(2.0 * np.pi * u2), rad * np.sin(2.0 * np.pi * u2)


def encode_rows(code: str, class_bits: np.ndarray) -> np.ndarray:
    enc = O.encode_133171 if code == "133171" else O.encode_57
    return np.stack([enc(r) for r in class_bits]).astype(np.int32)


def generate(code: str, b: int, n_rows: int, sigma, seed: int, first_row: int = 0,
             rows_per_block: int = 1, continuous: bool = False):
    """Returns (x float64 [n_rows, 2b], labels int32 [n_rows]).

    ``continuous``: the info bits of a block (rows_per_block consecutive global rows) are ONE stream,
    encoded from zero state by the restated reference encoder and chopped into rows
    (src/fnn_viterbi_mixed_snr.py:464-475, the "random test set"); labels stay each row's own bits."""
    sigma = np.atleast_1d(np.asarray(sigma, dtype=np.float32)).astype(np.float64)
    r = np.arange(first_row, first_row + n_rows, dtype=np.uint64)
    r_lo = (r & _MASK).astype(np.uint32)
    r_hi = (r >> np.uint64(32)).astype(np.uint32)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    w = philox4x32_10(r_lo, r_hi, 0, 0, k0, k1)
    cls = (w[0] & np.uint32((1 << b) - 1)).astype(np.int32)
    bits = O.class_to_bits(cls, b)
    # encode each distinct class once
    if continuous:
        enc = O.encode_133171 if code == "133171" else O.encode_57
        coded = np.zeros((n_rows, 2 * b), dtype=np.int32)
        first_blk, last_blk = first_row // rows_per_block, (first_row + n_rows - 1) // rows_per_block
        for blk_i in range(first_blk, last_blk + 1):          # whole blocks, whatever part of them was asked for
            rr = np.arange(blk_i * rows_per_block, (blk_i + 1) * rows_per_block, dtype=np.uint64)
            wb = philox4x32_10((rr & _MASK).astype(np.uint32), (rr >> np.uint64(32)).astype(np.uint32), 0, 0, k0, k1)
            stream = O.class_to_bits((wb[0] & np.uint32((1 << b) - 1)).astype(np.int32), b).reshape(-1)
            rows = np.asarray(enc(stream)).reshape(rows_per_block, 2 * b)
            lo, hi = max(first_row, int(rr[0])), min(first_row + n_rows, int(rr[-1]) + 1)
            coded[lo - first_row:hi - first_row] = rows[lo - int(rr[0]):hi - int(rr[0])]
    else:
        table = encode_rows(code, O.generate_class_bits(b))
        coded = table[cls]
    noise = np.zeros((n_rows, 2 * b), dtype=np.float64)
    for t in range((b + 1) // 2):
        g = philox4x32_10(r_lo, r_hi, t, 1, k0, k1)
        n0, n1 = _box_muller(g[0], g[1])
        n2, n3 = _box_muller(g[2], g[3])
        for d, nz in enumerate((n0, n1, n2, n3)):
            col = 4 * t + d
            if col < 2 * b:
                noise[:, col] = nz
    blk = (r // np.uint64(rows_per_block)).astype(np.int64)
    sg = sigma[blk % len(sigma)]
    x = np.where(coded == 1, 1.0, -1.0) + sg[:, None] * noise
    assert bits.shape == (n_rows, b)
    return x, cls
