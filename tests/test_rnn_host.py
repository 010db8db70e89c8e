"""Recurrent decoder (src/rnn_viterbi.py mirror): host API, oracle and trainer graph on CPU.

Parity status of this whole module: UNPINNED.  The reference file cannot be imported
(``Optional`` undefined) so no golden vector can be generated from it, and its
arithmetic is un-vendored TensorFlow 1.x.  These tests pin our pieces to each other and to
hand-computed values: the numpy generators to a literal per-row restatement of
src/rnn_viterbi.py:187-229, the oracle's LSTM cell to a scalar calculation from the published
``LSTMCell.call`` formula, and the torch training graph to the oracle.
Nothing here calls the HIP path."""
import numpy as np
import pytest
import torch

import nnviterbi_amd  # noqa: F401
from nnviterbi_amd import rnn_viterbi as rv
from oracle import rnn_oracle as R


def test_config_fields_follow_the_reference():
    cfg = rv.RNNDecoderConfig()
    assert (cfg.block_length, cfg.encoded_length, cfg.encoder_hidden_units, cfg.decoder_hidden_units) == (100, 200, 20, 20)
    assert cfg.vocab_size == 1 and cfg.PAD == 0 and cfg.EOS == 1 and cfg.display_step == 1
    assert cfg.snr_db_vec.tolist() == list(range(0, 9)) and cfg.test_snr_index == 2
    # sigma = sqrt(0.5 / 10^(dB/10)): 2 dB -> 0.56167488
    assert abs(cfg.sigma_vec[2] - 0.5616748812614791) < 1e-15
    assert rv.RNNDecoderConfig(decoder_hidden_units=7).decoder_hidden_units == 7


def _literal_datasets(cfg):
    """src/rnn_viterbi.py:187-229 line by line (per-row encode and per-row noise)."""
    def noisy(info):
        enc = np.array([rv.encode_133171(seq) for seq in info])
        sigma = cfg.sigma_vec[cfg.test_snr_index]
        for i in range(len(info)):
            enc[i, :] = rv.modulate_awgn(enc[i, :], sigma)
        return enc
    train_info = np.random.randint(2, size=(cfg.training_size, cfg.block_length))
    train_enc = noisy(train_info)
    train_info = train_info.astype(np.float32)
    train_info[np.where(train_info == 0)] = -1
    test_info = np.random.randint(2, size=(cfg.test_size, cfg.block_length))
    test_enc = noisy(test_info)
    test_info = test_info.astype(np.float32)
    test_info[np.where(test_info == 0)] = -1
    return train_info, train_enc, test_info, test_enc


def test_generate_datasets_equals_the_literal_loops():
    cfg = rv.RNNDecoderConfig(block_length=12, training_size=37, test_size=11)
    np.random.seed(5)
    want = _literal_datasets(cfg)
    np.random.seed(5)
    got = rv.generate_datasets(cfg)
    for w, g in zip(want, got):
        assert w.dtype == g.dtype and np.array_equal(w, g)
    assert got[1].dtype.kind == "i" and got[0].dtype == np.float32          # soft values truncated (F12)
    assert set(np.unique(got[0]).tolist()) <= {-1.0, 1.0}
    assert np.abs(got[1]).max() <= 3                                         # +-1 + noise, toward zero


def test_prepare_batch_layout():
    cfg = rv.RNNDecoderConfig(block_length=6)
    info = np.array([[1, -1, 1, 1, -1, -1], [-1, -1, 1, -1, 1, 1]], np.float32)
    enc = np.arange(24, dtype=np.int32).reshape(2, 12)
    f = rv.prepare_batch(info, enc, cfg)
    assert f["encoder_inputs"].shape == (12, 2, 1) and f["decoder_inputs"].shape == (12, 2, 1)
    assert f["decoder_targets"].shape == (12, 2, 1)
    assert np.array_equal(f["encoder_inputs"][:, 1, 0], enc[1])
    assert f["decoder_inputs"][:, 0, 0].tolist() == [1, 1, -1, 1, 1, -1, -1, 0, 0, 0, 0, 0]   # EOS, info, padding
    assert f["decoder_targets"][:, 0, 0].tolist() == [1, -1, 1, 1, -1, -1, 1, 0, 0, 0, 0, 0]  # info, EOS, padding


def test_lstm_cell_known_answer():
    """H = 1, scalar arithmetic of LSTMCell.call: z = [x, m] K + b, (i, j, f, o) = z."""
    K = np.array([[0.5, -0.25, 0.75, 1.0], [0.1, 0.2, -0.3, 0.4]], np.float64)
    b = np.array([0.01, 0.02, 0.03, 0.04], np.float64)
    x, c0, m0 = 0.8, 0.3, -0.2
    z = x * K[0] + m0 * K[1] + b
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    c1 = sig(z[2] + 1.0) * c0 + sig(z[0]) * np.tanh(z[1])
    m1 = sig(z[3]) * np.tanh(c1)
    c, m = R.lstm_cell(np.array([[x]]), np.array([[c0]]), np.array([[m0]]), K, b)
    assert abs(c[0, 0] - c1) < 1e-15 and abs(m[0, 0] - m1) < 1e-15


def _params(H, seed, scale=1.0):
    cfg = rv.RNNDecoderConfig(encoder_hidden_units=H)
    p = rv.initialize_parameters(cfg, rng=np.random.RandomState(seed))
    rng = np.random.RandomState(seed + 1)
    for k in ("enc_bias", "dec_bias", "proj_b"):
        p[k] = (rng.randn(*p[k].shape) * 0.1).astype(np.float32)
    p["enc_kernel"] *= scale
    p["dec_kernel"] *= scale
    return p


def test_oracle_float32_tracks_float64_and_prediction_is_zero():
    p = _params(20, 0)
    rng = np.random.RandomState(1)
    enc = rng.randn(40, 9).astype(np.float32)
    dec = np.sign(rng.randn(40, 9)).astype(np.float32)
    l32, pred = R.seq2seq_forward(enc, dec, p, np.float32)
    l64, _ = R.seq2seq_forward(enc, dec, p, np.float64)
    assert l32.shape == (40, 9) and pred.shape == (40, 9) and pred.dtype == np.int64 and not pred.any()
    assert np.abs(l32 - l64).max() < 2e-5
    bits = R.hard_bits(l64, 20)
    assert bits.shape == (9, 20) and np.array_equal(bits, (l64[:20].T > 0))
    cnt = R.error_counters(bits, 1 - bits)
    assert cnt.tolist() == [180, 9, 180, 9]


def test_initial_parameters_have_tf_default_shapes_and_ranges():
    cfg = rv.RNNDecoderConfig(encoder_hidden_units=20)
    p = rv.initialize_parameters(cfg, rng=np.random.RandomState(0))
    assert p["enc_kernel"].shape == (21, 80) and p["dec_kernel"].shape == (21, 80)
    assert p["proj_w"].shape == (20, 1) and p["proj_b"].shape == (1,)
    assert not p["enc_bias"].any() and not p["dec_bias"].any() and not p["proj_b"].any()
    assert np.abs(p["enc_kernel"]).max() <= np.sqrt(6.0 / 101) and np.abs(p["proj_w"]).max() <= np.sqrt(6.0 / 21)
    with pytest.raises(ValueError):
        rv.initialize_parameters(rv.RNNDecoderConfig(encoder_hidden_units=20, decoder_hidden_units=10))


def test_training_graph_equals_the_oracle():
    p = _params(6, 3)
    rng = np.random.RandomState(4)
    enc = rng.randn(10, 5).astype(np.float32)
    dec = np.sign(rng.randn(10, 5)).astype(np.float32)
    P = {k: torch.tensor(v, dtype=torch.float64) for k, v in p.items()}
    got = rv._torch_forward(P, torch.tensor(enc, dtype=torch.float64), torch.tensor(dec, dtype=torch.float64), 1.0)
    want, _ = R.seq2seq_forward(enc, dec, p, np.float64)
    assert got.shape == (10, 5, 1) and np.abs(got[:, :, 0].numpy() - want).max() < 1e-12


def test_forward_fails_loudly_without_a_device():
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    cfg = rv.RNNDecoderConfig(block_length=6, encoder_hidden_units=4)
    model = rv.build_seq2seq_model(cfg, rng=np.random.RandomState(0))
    feeds = rv.prepare_batch(np.ones((2, 6), np.float32), np.ones((2, 12), np.int32), cfg)
    with pytest.raises(nnviterbi_amd.NNViterbiError):
        rv.run_seq2seq(model, feeds)


def test_oracle_reproduces_its_committed_fixture():
    """tests/golden/rnn_fp64.npz (made by make_rnn_golden.py): the oracle has not drifted."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rnn_fp64.npz"), allow_pickle=False)
    for name in ("h20_l16", "h7_l10"):
        p = {k: g[f"{name}_{k}"] for k in ("enc_kernel", "enc_bias", "dec_kernel", "dec_bias", "proj_w", "proj_b")}
        got, pred = R.seq2seq_forward(g[f"{name}_enc"], g[f"{name}_dec"], p, np.float64)
        assert np.abs(got - g[f"{name}_logits64"]).max() < 1e-13 and not pred.any()
