"""numpy model of the on-device generator: Philox4x32-10 known answers
(Random123 kat_vectors) and the model's own invariants.  CPU only."""
import numpy as np

from oracle import awgn_model as A
from oracle import fnn_oracle as O


def test_philox_known_answers():
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff, 0xffffffff),
         (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        got = A.philox4x32_10(*[np.array([c], dtype=np.uint32) for c in ctr], *key)
        assert tuple(int(v[0]) for v in got) == want


def test_rows_do_not_depend_on_sharding():
    x, lab = A.generate("133171", 8, 64, [0.5, 0.7], seed=5, rows_per_block=8)
    xa, la = A.generate("133171", 8, 24, [0.5, 0.7], seed=5, first_row=0, rows_per_block=8)
    xb, lb = A.generate("133171", 8, 40, [0.5, 0.7], seed=5, first_row=24, rows_per_block=8)
    assert np.array_equal(np.vstack([xa, xb]), x) and np.array_equal(np.concatenate([la, lb]), lab)


def test_signal_part_is_the_reference_codeword_and_noise_is_gaussian():
    b, n = 8, 20000
    x, lab = A.generate("133171", b, n, 0.5616748812614791, seed=11, rows_per_block=8)
    x0, lab0 = A.generate("133171", b, n, 0.0, seed=11, rows_per_block=8)
    assert np.array_equal(lab, lab0)
    cb = O.generate_class_bits(b)
    want = np.stack([O.encode_133171(r) for r in cb])[lab0]
    assert np.array_equal(x0, np.where(want == 1, 1.0, -1.0))
    z = (x - x0) / 0.5616748812614791
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01
    assert abs(np.mean(z ** 4) - 3.0) < 0.1
    counts = np.bincount(lab, minlength=256)
    assert counts.min() > 30                      # classes roughly uniform
    x57, l57 = A.generate("57", 5, 64, 0.0, seed=3)
    want57 = np.stack([O.encode_57(r) for r in O.generate_class_bits(5)])[l57]
    assert np.array_equal(x57, np.where(want57 == 1, 1.0, -1.0))


def test_continuous_mode_is_the_reference_random_test_set_shape():
    """continuous=True: a block's info bits are ONE stream through the restated reference encoder, chopped into rows
    (src/fnn_viterbi_mixed_snr.py:464-475); classes, labels and noise are those of the per-row mode; a block of one row
    IS the per-row mode; rows do not depend on how the call range cuts the blocks."""
    b, n, rpb = 8, 96, 12
    x, lab = A.generate("133171", b, n, 0.0, seed=21, rows_per_block=rpb, continuous=True)
    x_rows, lab_rows = A.generate("133171", b, n, 0.0, seed=21, rows_per_block=rpb)
    assert np.array_equal(lab, lab_rows)
    bits = O.class_to_bits(lab, b)
    for blk in range(n // rpb):
        stream = bits[blk * rpb:(blk + 1) * rpb].reshape(-1)
        want = np.where(O.encode_133171(stream) == 1, 1.0, -1.0).reshape(rpb, 2 * b)
        assert np.array_equal(x[blk * rpb:(blk + 1) * rpb], want)
        assert np.array_equal(x[blk * rpb], x_rows[blk * rpb])            # a block's first row starts from zero state
    assert not np.array_equal(x, x_rows)                                   # ... the others do not
    one, _ = A.generate("57", 6, 40, 0.3, seed=4, rows_per_block=1, continuous=True)
    one_rows, _ = A.generate("57", 6, 40, 0.3, seed=4, rows_per_block=1)
    assert np.array_equal(one, one_rows)
    xa, la = A.generate("57", 6, 17, 0.3, seed=4, first_row=5, rows_per_block=7, continuous=True)
    xb, lb = A.generate("57", 6, 30, 0.3, seed=4, first_row=0, rows_per_block=7, continuous=True)
    assert np.array_equal(xa, xb[5:22]) and np.array_equal(la, lb[5:22])
