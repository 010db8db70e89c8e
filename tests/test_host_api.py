"""The package's host-side mirror of the reference API (numpy half) against
golden vectors produced by the reference itself, plus the assertions of the
reference's own unit tests that hold for the shipped code (tests/test_encoders.py,
tests/test_utilities.py; those files cannot be collected as shipped).
CPU only: nothing here calls the HIP path."""
import contextlib
import io
import random

import numpy as np
import pytest

import nnviterbi_amd  # noqa: F401
from nnviterbi_amd import fnn_viterbi as fv
from nnviterbi_amd import fnn_viterbi_mixed_snr as mx
from oracle import fnn_oracle as O


def _stdout(fn, *a):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        fn(*a)
    return buf.getvalue()


# ---------------------------------------------------------------- encoders
def test_encoders_match_reference_vectors(golden):
    g, meta = golden
    for i in range(meta["n_enc_inputs"]):
        bits = g[f"enc_in_{i}"]
        out = fv.encode_57(bits)
        assert out.dtype == np.int32 and np.array_equal(out, g[f"enc57_out_{i}"])
        assert np.array_equal(mx.encode_57(bits), out)
        if len(bits) >= 6:
            out = fv.encode_133171(bits)
            assert out.dtype == np.int32 and np.array_equal(out, g[f"enc133171_out_{i}"])


def test_batch_encoders_equal_loop_encoders(golden):
    g, _ = golden
    for b in (3, 7, 8, 10):
        cb = g[f"class_bits_{b}"]
        assert np.array_equal(fv.encode_57_batch(cb), g[f"codebook57_{b}"])
        if b >= 6:
            assert np.array_equal(fv.encode_133171_batch(cb), g[f"codebook133171_{b}"])
    rng = np.random.RandomState(3)
    u = rng.randint(0, 2, size=(5, 40))
    assert np.array_equal(fv.encode_133171_batch(u), np.stack([O.encode_133171(r) for r in u]))
    assert np.array_equal(fv.encode_57_batch(u), np.stack([O.encode_57(r) for r in u]))


def test_encoder_error_behaviour():
    with pytest.raises(IndexError):
        fv.encode_133171(np.array([1]))            # reference tests/test_encoders.py:204-209 expects
    with pytest.raises(IndexError):                # a result; the shipped code raises
        fv.encode_133171(np.zeros(5, dtype=np.int32))
    with pytest.raises(IndexError):
        fv.encode_57(np.array([1]))


def test_reference_encoder_unit_test_assertions():
    # tests/test_encoders.py:44-106 (lengths, zeros, binary, determinism) and :169-182 (linearity)
    assert len(fv.encode_57(np.array([1, 0, 1, 0, 1]))) == 10
    assert np.all(fv.encode_57(np.zeros(5, dtype=np.int32)) == 0)
    assert fv.encode_57(np.array([1, 0, 0, 0, 0]))[0] == 1
    assert len(fv.encode_133171(np.array([1, 0, 1, 0, 1, 1, 0]))) == 14
    assert np.all(fv.encode_133171(np.zeros(10, dtype=np.int32)) == 0)
    e = fv.encode_133171(np.ones(10, dtype=np.int32))
    assert len(e) == 20 and np.all((e == 0) | (e == 1))
    rng = np.random.RandomState(0)
    a, b = rng.randint(0, 2, 20), rng.randint(0, 2, 20)
    assert np.array_equal(fv.encode_133171((a + b) % 2),
                          (fv.encode_133171(a) + fv.encode_133171(b)) % 2)
    s = fv.generate_info_sequence(100)
    assert len(s) == 100 and s.dtype == np.int32 and np.all((s == 0) | (s == 1))


def test_generate_info_sequence_follows_python_random(golden):
    g, _ = golden
    random.seed(5)
    assert np.array_equal(fv.generate_info_sequence(16), g["info_seq_seed5"])


# ---------------------------------------------------------------- channel / dataset
def test_modulate_awgn(golden):
    g, _ = golden
    np.random.seed(7)
    out = fv.modulate_awgn(g["awgn_in"], 0.5616748812614791)
    assert out.dtype == np.float64 and np.array_equal(out, g["awgn_out_seed7"])
    assert g["awgn_in"].tolist() == [0, 1, 1, 0, 1, 0, 0, 1]      # input not modified


def test_config_fields(golden):
    _, meta = golden
    c = fv.DecoderConfig()
    for k, v in meta["default_config"].items():
        assert getattr(c, k) == v, k
    for db, s in meta["sigma_by_snr_db"].items():
        assert fv.DecoderConfig(snr_db=float(db)).sigma == s
    m = mx.MixedSNRDecoderConfig()
    md = meta["mixed_default_config"]
    assert m.snr_db_vec.tolist() == md["snr_db_vec"]
    assert np.array_equal(m.sigma_vec, np.array(md["sigma_vec"]))
    for k in ("samples_per_snr", "random_test_length", "num_training_samples",
              "num_test_samples", "n_input", "n_classes"):
        assert getattr(m, k) == md[k], k


def test_train_test_dataset_is_bit_identical(golden):
    g, meta = golden
    cfg = fv.DecoderConfig(block_length=7, samples_per_class_train=2, samples_per_class_test=4)
    np.random.seed(meta["ds7"]["seed"])
    tr, trl, te, tel, cb = fv.train_test_dataset(cfg)
    assert tr.dtype == np.float32 and te.dtype == np.float32 and trl.dtype == np.float32
    assert list(tr.shape) == meta["ds7"]["train_shape"] and list(tel.shape) == meta["ds7"]["labels_shape"]
    assert te.flags["C_CONTIGUOUS"]
    assert np.array_equal(tr, g["ds7_train"]) and np.array_equal(te, g["ds7_test"])
    assert np.array_equal(np.argmax(trl, 1), g["ds7_train_labels_idx"])
    assert np.array_equal(np.argmax(tel, 1), g["ds7_test_labels_idx"])
    assert np.array_equal(cb, g["class_bits_7"])


def test_get_dataset_is_bit_identical(golden):
    g, meta = golden
    kw = dict(meta["mix"]["kwargs"])
    kw["snr_db_range"] = tuple(kw["snr_db_range"])
    np.random.seed(meta["mix"]["seed"])
    out = mx.get_dataset(mx.MixedSNRDecoderConfig(**kw))
    assert [str(a.dtype) for a in out] == meta["mix"]["dtypes"]
    assert [list(a.shape) for a in out] == meta["mix"]["shapes"]
    assert np.array_equal(out[0], g["mix_train"]) and np.array_equal(out[1], g["mix_test"])
    assert np.array_equal(np.argmax(out[2], 1), g["mix_train_labels_idx"])
    assert np.array_equal(np.argmax(out[3], 1), g["mix_random_labels_idx"])
    assert np.array_equal(out[4], g["mix_random_set"]) and np.array_equal(out[5], g["mix_class_bits"])
    # rows past samples_per_snr * n_snr stay zero in every class (samples_per_snr quirk)
    assert len(mx.get_dataset.last_test_labels) == out[1].shape[0]


def test_one_hot(golden):
    g, _ = golden
    a = fv.create_one_hot_labels(g["one_hot_in"], 4)
    assert a.dtype == np.float32 and np.array_equal(a, g["one_hot_out"])
    h = mx.create_one_hot_labels(g["one_hot_in"], 4)
    assert h.dtype == np.float16 and np.array_equal(h, g["one_hot_out_mixed"])
    # reference tests/test_utilities.py:101-143
    assert np.array_equal(fv.create_one_hot_labels(np.array([0, 1, 2]), 3), np.eye(3))
    assert fv.create_one_hot_labels(np.array([3]), 5).sum() == 1


def test_class_bits(golden):
    g, _ = golden
    for b in (3, 7, 8, 10):
        cb = fv.generate_class_bits(b)
        assert cb.dtype == np.uint8 and np.array_equal(cb, g[f"class_bits_{b}"])
    # reference tests/test_utilities.py:72-87, :303-309, :342-348
    assert fv.generate_class_bits(3).tolist() == [[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1],
                                                  [1, 0, 0], [1, 0, 1], [1, 1, 0], [1, 1, 1]]
    assert fv.generate_class_bits(1).tolist() == [[0], [1]]
    assert fv.generate_class_bits(15).shape == (32768, 15)


# ---------------------------------------------------------------- metrics / utilities
def test_calculate_error_prints_exactly_like_the_reference(golden):
    g, meta = golden
    cfg3 = fv.DecoderConfig(block_length=3, samples_per_class_test=2)
    text = _stdout(fv.calculate_error, fv.create_one_hot_labels(g["err3_labels_idx"], 8),
                   g["err3_preds"], fv.generate_class_bits(3), cfg3)
    assert text == meta["calculate_error_b3_stdout"]
    cfg5 = fv.DecoderConfig(block_length=5, samples_per_class_test=6)
    text = _stdout(fv.calculate_error, fv.create_one_hot_labels(g["err5_labels_idx"], 32),
                   g["err5_preds"], fv.generate_class_bits(5), cfg5)
    assert text == meta["calculate_error_b5_stdout"]


def test_mixed_metrics(golden):
    g, meta = golden
    assert mx.calculate_ber(g["err3_preds"], g["err3_labels_idx"], 3) == meta["calculate_ber_b3"]
    assert mx.calculate_accuracy(g["err3_preds"], g["err3_labels_idx"]) == meta["calculate_accuracy_b3"]
    assert mx.calculate_ber(g["err5_preds"], g["err5_labels_idx"], 5) == meta["calculate_ber_b5"]
    assert mx.calculate_accuracy(g["err5_preds"], g["err5_labels_idx"]) == meta["calculate_accuracy_b5"]


def test_bit_utilities(golden):
    _, meta = golden
    for v, e in meta["binary_to_decimal"]:
        assert mx.binary_to_decimal(np.array(v)) == e          # LSB-first
    for d, n, e in meta["extract_bits_single"]:
        assert mx.extract_bits_single(d, n).tolist() == e
    for a, b, n, e in meta["hamming"]:
        assert mx.calculate_hamming_distance_decimal(a, b, n) == e
    # reference tests/test_utilities.py:193-199, :255-259 (assertions that hold for shipped code)
    assert mx.extract_bits_single(5, 5).tolist() == [0, 0, 1, 0, 1]
    assert mx.calculate_hamming_distance_decimal(5, 3, 3) == 2


# ---------------------------------------------------------------- model helpers
def test_initialize_weights_shapes_and_truncation():
    cfg = fv.DecoderConfig(block_length=8)
    w, b = fv.initialize_weights(cfg, rng=np.random.RandomState(0))
    assert [w[k].shape for k in ("h1", "h2", "h3", "out")] == [(16, 128), (128, 64), (64, 32), (32, 256)]
    assert all(v.dtype == np.float32 for v in w.values())
    assert all(not v.any() for v in b.values())                       # zero biases
    for k, fan_in in (("h1", 16), ("h2", 128), ("h3", 64), ("out", 32)):
        sd = np.sqrt(2.0 / fan_in)
        assert np.abs(w[k]).max() <= 2 * sd * (1 + 1e-6)              # re-drawn beyond 2 sigma
        assert 0.7 * sd < w[k].std() < 1.0 * sd
    Ws, bs = fv.weights_as_lists(w, b)
    assert [W.shape[0] for W in Ws] == [16, 128, 64, 32] and len(bs) == 4
    cfg.n_layers = 2
    assert fv.layer_dims(cfg) == [16, 128, 64, 256]


def test_checkpoint_round_trip(tmp_path):
    cfg = mx.MixedSNRDecoderConfig(block_length=6)
    w, b = mx.initialize_weights(cfg, rng=np.random.RandomState(1))
    p = mx.save_model(str(tmp_path / "m" / "mixed_snr_model"), w, b)
    data = np.load(p)
    assert sorted(data.files) == sorted(["weights_h1", "weights_h2", "weights_h3", "weights_out",
                                         "bias_h1", "bias_h2", "bias_h3", "bias_out"])
    w2, b2 = mx.load_model(str(tmp_path / "m" / "mixed_snr_model"))
    assert all(np.array_equal(w[k], w2[k]) for k in w) and all(np.array_equal(b[k], b2[k]) for k in b)


def test_eval_without_checkpoint_fails_loudly():
    cfg = mx.MixedSNRDecoderConfig(block_length=6, samples_per_class_test=1)
    x = np.zeros((4, 12), np.float32)
    y = mx.create_one_hot_labels(np.zeros(4, int), 64)
    with pytest.raises(RuntimeError):
        mx.create_model_graph(cfg, x, y, x, y, train_mode=False, model_load_path=None)


def test_decode_path_has_no_cpu_fallback():
    """Without a HIP device the product path must raise, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    cfg = fv.DecoderConfig(block_length=6)
    w, b = fv.initialize_weights(cfg, rng=np.random.RandomState(0))
    with pytest.raises(nnviterbi_amd.NNViterbiError):
        fv.predict(np.zeros((2, 12), np.float32), w, b)
    with pytest.raises(nnviterbi_amd.NNViterbiError):
        fv.multilayer_perceptron(np.zeros((2, 12), np.float32), w, b)


def test_trainer_reduces_the_loss_on_cpu():
    """Section 8f rank 1 (trainer): tiny CPU run, loss goes down, shapes kept."""
    from nnviterbi_amd.training import train_network
    cfg = fv.DecoderConfig(block_length=6, samples_per_class_train=8, samples_per_class_test=1,
                           training_epochs=3, batch_size=64, learning_rate=0.01)
    np.random.seed(0)
    tr, trl, _, _, _ = fv.train_test_dataset(cfg)
    w0, b0 = fv.initialize_weights(cfg, rng=np.random.RandomState(0))
    text = _stdout(lambda: train_network(cfg, tr, trl, w0, b0))
    costs = [float(line.split("=")[1]) for line in text.strip().splitlines()]
    assert len(costs) == 3 and costs[-1] < costs[0]
    assert text.startswith("Epoch: 0001 | cost = ")


# ---------------------------------------------------------------- sliding-window decoder, numpy half
def test_seq_predict_config_and_streams_match_reference(golden):
    from nnviterbi_amd import fnn_viterbi_seq_predict as sq
    g, meta = golden
    c = sq.SequentialDecoderConfig()
    d = meta["seq_default_config"]
    assert c.total_training_instances == d["total_training_instances"] == 1000000
    assert c.total_test_instances == d["total_test_instances"]
    assert c.total_train_length_coded == d["total_train_length_coded"]
    assert (c.n_input, c.n_output, c.test_snr_index) == (d["n_input"], d["n_output"], d["test_snr_index"])
    assert c.snr_db_vec.tolist() == d["snr_db_vec"] and np.array_equal(c.sigma_vec, np.array(d["sigma_vec"]))
    small = sq.SequentialDecoderConfig(total_train_length_info=70, total_test_length_info=90,
                                       window_length=20, snr_db_range=(0, 4), test_snr_index=2)
    assert small.total_training_instances == meta["seq_small_config"]["total_training_instances"]
    assert small.total_test_instances == meta["seq_small_config"]["total_test_instances"]
    np.random.seed(31)
    tr_info, tr_coded = sq.generate_training_data(70, 1, small.sigma_vec)
    te_info, te_coded = sq.generate_test_data(90, 1, small.sigma_vec, 2)
    assert [str(a.dtype) for a in (tr_info, tr_coded, te_info, te_coded)] == meta["seq_dtypes"]
    assert np.array_equal(tr_info, g["seq_train_info"]) and np.array_equal(tr_coded, g["seq_train_coded"])
    assert np.array_equal(te_info, g["seq_test_info"]) and np.array_equal(te_coded, g["seq_test_coded"])
    assert te_coded.dtype == np.int32            # soft values truncated toward zero
    for off, wx, wy in zip(meta["seq_instance_offsets"], g["seq_instance_x"], g["seq_instance_y"]):
        bx, by = sq.get_single_instance(20, te_coded, te_info, off)
        assert str(bx.dtype) == meta["seq_instance_dtypes"][0] and by.dtype == np.float32
        assert np.array_equal(bx, wx) and np.array_equal(by, wy)
    win = sq.extract_windows(20, te_coded, 0, small.total_test_instances)
    assert win.shape == (small.total_test_instances, 20) and win.dtype == np.float32
    for i in (0, 1, 17, small.total_test_instances - 1):
        assert np.array_equal(win[i], sq.get_single_instance(20, te_coded, te_info, 2 * i)[0])
    np.random.seed(32)
    assert np.array_equal(sq.generate_info_sequence(12), g["seq_info_seq_seed32"])


def test_seq_predict_weights_and_checkpoint(tmp_path):
    from nnviterbi_amd import fnn_viterbi_seq_predict as sq
    cfg = sq.SequentialDecoderConfig(window_length=20, n_hidden_1=48, n_hidden_2=40, n_hidden_3=36, n_hidden_4=33)
    w, b = sq.initialize_weights(cfg, rng=np.random.RandomState(0))
    assert [w[k].shape for k in ("h1", "h2", "h3", "h4", "out")] == [(20, 48), (48, 40), (40, 36), (36, 33), (33, 2)]
    assert all(not v.any() for v in b.values())
    p = sq.save_model(str(tmp_path / "seq_predict_model"), w, b)
    assert sorted(np.load(p).files) == sorted(["weights_h1", "weights_h2", "weights_h3", "weights_h4",
                                               "weights_out", "bias_h1", "bias_h2", "bias_h3", "bias_h4",
                                               "bias_out"])
    w2, b2 = sq.load_model(str(tmp_path / "seq_predict_model"))
    assert all(np.array_equal(w[k], w2[k]) for k in w) and all(np.array_equal(b[k], b2[k]) for k in b)
    cfg.test_old_model = True
    with pytest.raises(RuntimeError):
        with contextlib.redirect_stdout(io.StringIO()):
            sq.train_and_evaluate(cfg)                # no model_load_path: nothing to evaluate


# ---------------------------------------------------------------- bit-wise decoder, numpy half
def test_bitwise_numpy_half_matches_reference(golden):
    from nnviterbi_amd import fnn_viterbi_bitwise as bw
    g, meta = golden
    c = bw.BitwiseDecoderConfig()
    d = meta["bit_default_config"]
    assert (c.n_input, c.total_length, c.n_output, c.n_hidden_1) == (d["n_input"], d["total_length"], d["n_output"], d["n_hidden_1"])
    assert np.array_equal(c.sigma_vec, np.array(d["sigma_vec"]))
    for s_, n_, want in meta["extract_state_bits"]:
        got = bw.extract_state_bits(s_, n_)
        assert got.dtype == np.int32 and got.tolist() == want
    assert bw.extract_state_bits(5, 6).tolist() == [0, 0, 0, 1, 0, 1]     # reference tests/test_encoders.py:128-134
    assert meta["extract_state_bits_100_error"] == "ValueError"
    with pytest.raises(ValueError):
        bw.extract_state_bits(100, 6)                                      # shipped code raises
    for i, st in enumerate(meta["bit_enc_states"]):
        out = bw.encode_133171_with_state(g[f"bit_enc_in_{i}"], st)
        assert out.dtype == np.int32 and np.array_equal(out, g[f"bit_enc_out_{i}"])
        assert len(out) == 2 * (6 + len(g[f"bit_enc_in_{i}"]))            # tests/test_encoders.py:136-142
    sb = bw.BitwiseDecoderConfig(block_length=12, training_size=5, test_size=4, snr_db_range=(0, 3),
                                 test_snr_index=1, starting_state=9)
    np.random.seed(41)
    bi, be = bw.generate_training_data(sb)
    ti, te = bw.generate_test_data(sb)
    assert [str(a.dtype) for a in (bi, be, ti, te)] == meta["bit_dtypes"]
    assert np.array_equal(bi, g["bit_train_info"]) and np.array_equal(be, g["bit_train_enc"])
    assert np.array_equal(ti, g["bit_test_info"]) and np.array_equal(te, g["bit_test_enc"])
    np.random.seed(42)
    assert np.array_equal(bw.generate_input_vectors(7, 3), g["bit_input_vectors"])
    w, b = bw.initialize_weights(sb, rng=np.random.RandomState(0))
    assert w["out"].shape == (sb.n_hidden_1, 2) and w["h2"].shape == (30, 100) and w["h1"].shape == (36, 30)
    Ws, bs = bw.active_layers(w, b)
    assert [a.shape for a in Ws] == [(36, 30), (30, 2)] and [a.shape for a in bs] == [(30,), (2,)]


def test_missing_library_fails_loudly():
    """No silent fallback when libnnviterbi_hip.so is absent: loading (and hence every compute entry,
    the torch ops' HIP key included) raises NNViterbiError naming the build command."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import nnviterbi_amd\n"
            "try:\n    nnviterbi_amd.load_library()\nexcept nnviterbi_amd.NNViterbiError as e:\n"
            "    print('LOUD:', e)\n" % root)
    env = dict(os.environ, NNV_LIB="/nonexistent/libnnviterbi_hip.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "LOUD:" in r.stdout and "is missing" in r.stdout and "no CPU fallback" in r.stdout, r.stdout + r.stderr
