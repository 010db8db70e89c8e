#!/usr/bin/env python3
"""Generate golden vectors from the reference's importable numpy half.

Run ONLY in the build container (the reference tree does not exist on the GPU
box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_reference_golden.py

What it does
------------
The reference's forward pass lives in TensorFlow, which is not installed here
(SURVEY.md F4), so nothing of the network itself can be executed.  Everything
*around* the forward pass (encoders, AWGN channel, codebook, dataset assembly,
BER/FER metrics) is plain numpy and is imported from /root/reference/src with
an empty module object registered as ``tensorflow`` (the two attributes it
carries are used by the reference only inside def-time type annotations,
src/fnn_viterbi.py:472-475).  No reference source text is copied: only inputs
and the outputs the reference functions returned are stored.

Output: tests/golden/reference_numpy_half.npz + reference_numpy_half.json
"""
import contextlib
import io
import json
import os
import random
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF_SRC = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    tf = types.ModuleType("tensorflow")
    tf.Tensor = object
    tf.Variable = object
    sys.modules["tensorflow"] = tf
    sys.path.insert(0, REF_SRC)
    import fnn_viterbi as ref_fnn  # noqa
    import fnn_viterbi_mixed_snr as ref_mix  # noqa
    import fnn_viterbi_seq_predict as ref_seq  # noqa
    import fnn_viterbi_bitwise as ref_bit  # noqa
    sys.path.pop(0)
    return ref_fnn, ref_mix, ref_seq, ref_bit


def _capture(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


def main():
    ref_fnn, ref_mix, ref_seq, ref_bit = _import_reference()
    arrays = {}
    meta = {}

    # ---- 1. encoder known answers -------------------------------------
    rng = np.random.RandomState(1234)
    enc_inputs = [
        np.array([1, 0, 0, 0, 0], dtype=np.int32),
        np.array([1, 1, 0, 1, 0], dtype=np.int32),
        np.array([1, 0, 0, 0, 0, 0, 0, 0, 0, 0], dtype=np.int32),
        np.array([1, 0, 1, 1, 0, 1, 0, 1, 1, 0], dtype=np.int32),
        np.ones(10, dtype=np.int32),
        np.zeros(10, dtype=np.int32),
        rng.randint(0, 2, size=6).astype(np.int32),
        rng.randint(0, 2, size=7).astype(np.int32),
        rng.randint(0, 2, size=64).astype(np.int32),
        rng.randint(0, 2, size=257).astype(np.int32),
    ]
    for i, bits in enumerate(enc_inputs):
        arrays[f"enc_in_{i}"] = bits
        arrays[f"enc57_out_{i}"] = ref_fnn.encode_57(bits)
        if len(bits) >= 6:
            arrays[f"enc133171_out_{i}"] = ref_fnn.encode_133171(bits)
        # the two modules define the same encoders; pin that too
        assert np.array_equal(ref_mix.encode_57(bits), arrays[f"enc57_out_{i}"])
    meta["n_enc_inputs"] = len(enc_inputs)
    # short-input error behaviour
    errs = {}
    for name, fn, n in (("encode_57", ref_fnn.encode_57, 1),
                        ("encode_133171", ref_fnn.encode_133171, 5)):
        try:
            fn(np.zeros(n, dtype=np.int32))
            errs[name] = None
        except Exception as e:  # noqa
            errs[name] = type(e).__name__
    meta["short_input_errors"] = errs

    # ---- 2. codebooks --------------------------------------------------
    for b in (3, 7, 8, 10):
        cb = ref_fnn.generate_class_bits(b)
        arrays[f"class_bits_{b}"] = cb
        arrays[f"codebook57_{b}"] = np.stack(
            [ref_fnn.encode_57(r) for r in cb]).astype(np.uint8)
        if b >= 6:
            arrays[f"codebook133171_{b}"] = np.stack(
                [ref_fnn.encode_133171(r) for r in cb]).astype(np.uint8)

    # ---- 3. config-derived numbers --------------------------------------
    sig = {}
    for db in range(0, 7):
        sig[str(db)] = float(ref_fnn.DecoderConfig(snr_db=float(db)).sigma)
    meta["sigma_by_snr_db"] = sig
    c = ref_fnn.DecoderConfig()
    meta["default_config"] = {
        k: (float(v) if isinstance(v, (float, np.floating)) else int(v))
        for k, v in vars(c).items()}
    mc = ref_mix.MixedSNRDecoderConfig()
    meta["mixed_default_config"] = {
        "snr_db_vec": [int(v) for v in mc.snr_db_vec],
        "sigma_vec": [float(v) for v in mc.sigma_vec],
        "samples_per_snr": int(mc.samples_per_snr),
        "random_test_length": int(mc.random_test_length),
        "num_training_samples": int(mc.num_training_samples),
        "num_test_samples": int(mc.num_test_samples),
        "n_input": int(mc.n_input), "n_classes": int(mc.n_classes),
    }

    # ---- 4. channel + dataset, seeded ----------------------------------
    np.random.seed(7)
    arrays["awgn_in"] = np.array([0, 1, 1, 0, 1, 0, 0, 1], dtype=np.int32)
    out = ref_fnn.modulate_awgn(arrays["awgn_in"], 0.5616748812614791)
    arrays["awgn_out_seed7"] = out
    meta["awgn_out_dtype"] = str(out.dtype)

    cfg = ref_fnn.DecoderConfig(block_length=7, samples_per_class_train=2,
                                samples_per_class_test=4)
    np.random.seed(11)
    tr, trl, te, tel, cb = ref_fnn.train_test_dataset(cfg)
    arrays["ds7_train"] = tr
    arrays["ds7_train_labels_idx"] = np.argmax(trl, axis=1).astype(np.int32)
    arrays["ds7_test"] = te
    arrays["ds7_test_labels_idx"] = np.argmax(tel, axis=1).astype(np.int32)
    meta["ds7"] = {"seed": 11, "train_dtype": str(tr.dtype),
                   "labels_dtype": str(trl.dtype),
                   "train_shape": list(tr.shape), "test_shape": list(te.shape),
                   "labels_shape": list(tel.shape),
                   "c_contiguous": bool(te.flags["C_CONTIGUOUS"])}

    random.seed(5)
    arrays["info_seq_seed5"] = ref_fnn.generate_info_sequence(16)

    arrays["one_hot_in"] = np.array([0, 2, 1, 3], dtype=np.int32)
    arrays["one_hot_out"] = ref_fnn.create_one_hot_labels(arrays["one_hot_in"], 4)
    arrays["one_hot_out_mixed"] = ref_mix.create_one_hot_labels(
        arrays["one_hot_in"], 4)

    # ---- 5. metrics ------------------------------------------------------
    cfg3 = ref_fnn.DecoderConfig(block_length=3, samples_per_class_test=2)
    cb3 = ref_fnn.generate_class_bits(3)
    labels_idx = np.repeat(np.arange(8), 2)
    preds = labels_idx.copy()
    preds[0] = 7
    preds[5] = 3
    one_hot = ref_fnn.create_one_hot_labels(labels_idx, 8)
    _, text = _capture(ref_fnn.calculate_error, one_hot, preds, cb3, cfg3)
    arrays["err3_preds"] = preds.astype(np.int64)
    arrays["err3_labels_idx"] = labels_idx.astype(np.int32)
    meta["calculate_error_b3_stdout"] = text
    meta["calculate_ber_b3"] = float(ref_mix.calculate_ber(preds, labels_idx, 3))
    meta["calculate_accuracy_b3"] = float(
        ref_mix.calculate_accuracy(preds, labels_idx))

    # a bigger random case
    rng = np.random.RandomState(99)
    cfg5 = ref_fnn.DecoderConfig(block_length=5, samples_per_class_test=6)
    lab5 = np.repeat(np.arange(32), 6)
    pred5 = lab5.copy()
    flip = rng.rand(len(pred5)) < 0.3
    pred5[flip] = rng.randint(0, 32, size=int(flip.sum()))
    _, text5 = _capture(ref_fnn.calculate_error,
                        ref_fnn.create_one_hot_labels(lab5, 32), pred5,
                        ref_fnn.generate_class_bits(5), cfg5)
    arrays["err5_preds"] = pred5.astype(np.int64)
    arrays["err5_labels_idx"] = lab5.astype(np.int32)
    meta["calculate_error_b5_stdout"] = text5
    meta["calculate_ber_b5"] = float(ref_mix.calculate_ber(pred5, lab5, 5))
    meta["calculate_accuracy_b5"] = float(ref_mix.calculate_accuracy(pred5, lab5))

    # ---- 6. mixed-SNR utilities -------------------------------------------
    b2d_in = [[0, 0, 1], [1, 0, 0], [1, 0, 1, 0, 1, 0, 1, 0], [1], [0],
              [1, 1, 1, 1, 1]]
    meta["binary_to_decimal"] = [
        [v, int(ref_mix.binary_to_decimal(np.array(v)))] for v in b2d_in]
    meta["extract_bits_single"] = [
        [d, n, [int(x) for x in ref_mix.extract_bits_single(d, n)]]
        for d, n in ((5, 5), (5, 7), (0, 5), (127, 7), (1, 7), (300, 10))]
    meta["hamming"] = [
        [a, b_, n, int(ref_mix.calculate_hamming_distance_decimal(a, b_, n))]
        for a, b_, n in ((5, 3, 3), (0, 15, 4), (0, 1, 3), (5, 5, 7),
                         (0, 0, 0), (1023, 512, 10))]

    mcfg = ref_mix.MixedSNRDecoderConfig(
        block_length=6, samples_per_class_train=6, samples_per_class_test=3,
        snr_db_range=(0, 3), test_snr_index=1, random_test_set_size=10)
    np.random.seed(21)
    (m_tr, m_te, m_trl, m_rl, m_rs, m_cb) = ref_mix.get_dataset(mcfg)
    arrays["mix_train"] = m_tr
    arrays["mix_test"] = m_te
    arrays["mix_train_labels_idx"] = np.argmax(m_trl, axis=1).astype(np.int32)
    arrays["mix_random_labels_idx"] = np.argmax(m_rl, axis=1).astype(np.int32)
    arrays["mix_random_set"] = m_rs
    arrays["mix_class_bits"] = m_cb
    meta["mix"] = {
        "seed": 21,
        "kwargs": {"block_length": 6, "samples_per_class_train": 6,
                   "samples_per_class_test": 3, "snr_db_range": [0, 3],
                   "test_snr_index": 1, "random_test_set_size": 10},
        "dtypes": [str(a.dtype) for a in (m_tr, m_te, m_trl, m_rl, m_rs, m_cb)],
        "shapes": [list(a.shape) for a in (m_tr, m_te, m_trl, m_rl, m_rs, m_cb)],
    }

    # ---- 7. sliding-window decoder, numpy half (src/fnn_viterbi_seq_predict.py) ----
    sc = ref_seq.SequentialDecoderConfig()
    meta["seq_default_config"] = {
        "total_training_instances": int(sc.total_training_instances),
        "total_test_instances": int(sc.total_test_instances),
        "total_train_length_coded": int(sc.total_train_length_coded),
        "n_input": int(sc.n_input), "n_output": int(sc.n_output),
        "snr_db_vec": [int(v) for v in sc.snr_db_vec],
        "sigma_vec": [float(v) for v in sc.sigma_vec],
        "test_snr_index": int(sc.test_snr_index),
    }
    small = ref_seq.SequentialDecoderConfig(total_train_length_info=70, total_test_length_info=90,
                                            window_length=20, snr_db_range=(0, 4), test_snr_index=2)
    meta["seq_small_config"] = {"total_training_instances": int(small.total_training_instances),
                                "total_test_instances": int(small.total_test_instances)}
    np.random.seed(31)
    tr_info, tr_coded = ref_seq.generate_training_data(70, 1, small.sigma_vec)
    te_info, te_coded = ref_seq.generate_test_data(90, 1, small.sigma_vec, 2)
    arrays["seq_train_info"], arrays["seq_train_coded"] = tr_info, tr_coded
    arrays["seq_test_info"], arrays["seq_test_coded"] = te_info, te_coded
    meta["seq_dtypes"] = [str(a.dtype) for a in (tr_info, tr_coded, te_info, te_coded)]
    inst_x, inst_y = [], []
    for off in (0, 2, 36, 2 * (small.total_test_instances - 1)):
        bx, by = ref_seq.get_single_instance(20, te_coded, te_info, off)
        inst_x.append(bx)
        inst_y.append(by)
    arrays["seq_instance_x"] = np.stack(inst_x)
    arrays["seq_instance_y"] = np.stack(inst_y)
    meta["seq_instance_offsets"] = [0, 2, 36, 2 * (small.total_test_instances - 1)]
    meta["seq_instance_dtypes"] = [str(inst_x[0].dtype), str(inst_y[0].dtype)]
    np.random.seed(32)
    arrays["seq_info_seq_seed32"] = ref_seq.generate_info_sequence(12)

    # ---- 8. bit-wise decoder, numpy half (src/fnn_viterbi_bitwise.py) ----
    bc = ref_bit.BitwiseDecoderConfig()
    meta["bit_default_config"] = {"n_input": int(bc.n_input), "total_length": int(bc.total_length),
                                  "n_output": int(bc.n_output), "n_hidden_1": int(bc.n_hidden_1),
                                  "sigma_vec": [float(v) for v in bc.sigma_vec]}
    meta["extract_state_bits"] = [[s_, n_, [int(v) for v in ref_bit.extract_state_bits(s_, n_)]]
                                  for s_, n_ in ((5, 6), (0, 6), (63, 6), (1, 6), (37, 6))]
    try:
        ref_bit.extract_state_bits(100, 6)
        meta["extract_state_bits_100_error"] = None
    except Exception as e:  # noqa
        meta["extract_state_bits_100_error"] = type(e).__name__
    rng = np.random.RandomState(77)
    for i, st in enumerate((0, 5, 63)):
        bits = rng.randint(0, 2, size=10)
        arrays[f"bit_enc_in_{i}"] = bits
        arrays[f"bit_enc_out_{i}"] = ref_bit.encode_133171_with_state(bits, st)
    meta["bit_enc_states"] = [0, 5, 63]
    sb = ref_bit.BitwiseDecoderConfig(block_length=12, training_size=5, test_size=4, snr_db_range=(0, 3),
                                      test_snr_index=1, starting_state=9)
    np.random.seed(41)
    bi, be = ref_bit.generate_training_data(sb)
    ti, te2 = ref_bit.generate_test_data(sb)
    arrays["bit_train_info"], arrays["bit_train_enc"] = bi, be
    arrays["bit_test_info"], arrays["bit_test_enc"] = ti, te2
    meta["bit_dtypes"] = [str(a.dtype) for a in (bi, be, ti, te2)]
    np.random.seed(42)
    arrays["bit_input_vectors"] = ref_bit.generate_input_vectors(7, 3)

    np.savez_compressed(os.path.join(HERE, "reference_numpy_half.npz"), **arrays)
    with open(os.path.join(HERE, "reference_numpy_half.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", len(arrays), "arrays;",
          os.path.getsize(os.path.join(HERE, "reference_numpy_half.npz")), "bytes")


if __name__ == "__main__":
    main()
