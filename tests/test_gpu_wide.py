"""Layer-at-a-time path (nnv_wide_*) and the sliding-window decoder on the GPU, against the oracle.
Same stated tolerance as the register-chained kernel: outputs within 1e-5 of the float64
restatement, classes may differ only on float64 near-ties (margin < 1e-5)."""
import contextlib
import io

import numpy as np
import pytest
import torch

import nnviterbi_amd  # noqa: F401
from nnviterbi_amd import fnn_viterbi as fv
from nnviterbi_amd import fnn_viterbi_seq_predict as sq
from nnviterbi_amd.decoder import WideMLP
from oracle import fnn_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _net(dims, seed, bias_scale=0.1):
    rng = np.random.RandomState(seed)
    Ws = [fv.truncated_normal((dims[i], dims[i + 1]), np.sqrt(2.0 / dims[i]), rng) for i in range(len(dims) - 1)]
    bs = [(rng.randn(dims[i + 1]) * bias_scale).astype(np.float32) for i in range(len(dims) - 1)]
    return Ws, bs


def _check(cls, act, x, Ws, bs):
    c64, _, o64 = O.decode(x, Ws, bs, 1, np.float64)
    assert np.abs(act - o64).max() < TOL
    bad = cls != c64
    assert not bad.any() or O.top2_margin(o64)[bad].max() < TOL


def _stream(n_windows, window, seed):
    rng = np.random.RandomState(seed)
    n = 2 * (n_windows - 1) + window
    return np.trunc(rng.choice([-1.0, 1.0], n) + 0.8 * rng.randn(n)).astype(np.float32)   # int-valued, like F12


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("n", [1, 127, 128, 129, 5000])
def test_sliding_window_decoder_matches_oracle(cuda_device, n, precision):
    """Reference shape 200 -> 500 -> 500 -> 500 -> 500 -> 2 (src/fnn_viterbi_seq_predict.py:343-414)."""
    dims = [200, 500, 500, 500, 500, 2]
    Ws, bs = _net(dims, 3)
    net = WideMLP(Ws, bs, chunk_rows=1024, precision=precision)   # several chunks at n = 5000
    s = _stream(n, 200, n)
    r = net.decode_stream(torch.from_numpy(s).cuda(), n, want_act=True, want_cls64=True)
    x = np.ascontiguousarray(np.lib.stride_tricks.sliding_window_view(s, 200)[::2][:n])
    cls, act = r["cls32"].cpu().numpy(), r["act"].cpu().numpy()
    _check(cls, act, x, Ws, bs)
    assert np.array_equal(r["cls64"].cpu().numpy(), cls.astype(np.int64))
    ref = O.decode_c(x, Ws, bs, 1, want_bits=False)
    assert (ref["cls"] != cls).sum() <= max(1, n // 1000)
    r2 = net.decode(torch.from_numpy(x.copy()).cuda(), want_act=True)   # the materialised matrix
    assert torch.equal(r2["cls32"], r["cls32"]) and torch.equal(r2["act"], r["act"])
    net.close()


@pytest.mark.parametrize("dims", [[14, 300, 70, 128], [16, 256, 128, 64, 256], [20, 64, 2], [30, 129, 33]])
def test_wide_path_generic_shapes(cuda_device, dims):
    Ws, bs = _net(dims, len(dims))
    net = WideMLP(Ws, bs)
    x = np.random.RandomState(1).randn(777, dims[0]).astype(np.float32)
    r = net.decode(torch.from_numpy(x).cuda(), want_act=True)
    _check(r["cls32"].cpu().numpy(), r["act"].cpu().numpy(), x, Ws, bs)
    net.close()


def test_wide_path_argument_errors(cuda_device):
    Ws, bs = _net([200, 500, 2], 0)
    net = WideMLP(Ws, bs)
    s = torch.from_numpy(_stream(10, 200, 0)).cuda()
    with pytest.raises(ValueError):
        net.decode_stream(s, 11)                       # more windows than the stream holds
    with pytest.raises(ValueError):
        net.decode_stream(s.cpu(), 5)                  # host tensor
    with pytest.raises(ValueError):
        net.decode(torch.zeros((4, 100), device="cuda"))
    assert net.decode_stream(s, 0)["cls32"].numel() == 0
    with pytest.raises(ValueError):
        WideMLP([np.zeros((8, 2), np.float32)], [np.zeros(2, np.float32)])   # depth 1
    with pytest.raises(nnviterbi_amd.NNViterbiError):
        WideMLP(*_net([16, 256, 128], 0), precision="f16x3")                  # split path needs <= 32 classes
    Wbig, bbig = _net([200, 500, 2], 0)
    Wbig[0][0, 0] = 1e5
    with pytest.raises(nnviterbi_amd.NNViterbiError):
        WideMLP(Wbig, bbig, precision="f16x3")                                # weight outside the f16 range
    net.close()


def test_train_and_evaluate_end_to_end(cuda_device, tmp_path):
    """Tiny configuration: trains (PyTorch), saves, decodes the test stream on the GPU,
    and the reload path reproduces the predictions."""
    path = str(tmp_path / "seq_predict_model")
    cfg = sq.SequentialDecoderConfig(total_train_length_info=6000, total_test_length_info=3000,
                                     window_length=40, n_hidden_1=64, n_hidden_2=64, n_hidden_3=64,
                                     n_hidden_4=64, learning_rate=0.002, training_epochs=12, batch_size=100,
                                     snr_db_range=(4, 8), test_snr_index=2, model_save_path=path)
    np.random.seed(0)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = sq.train_and_evaluate(cfg)
    text = buf.getvalue()
    assert "Epoch: 0001 | cost = " in text and "Optimization Finished!" in text
    assert f"Model saved to {path}" in text and "Test Error Rate: " in text
    costs = [float(l.split("=")[1]) for l in text.splitlines() if l.startswith("Epoch: ")]
    assert len(costs) == 12 and costs[-1] < costs[0]   # the optimiser works (whether this decoder
    assert 0.0 <= out["error_rate"] <= 1.0             # generalises is the reference's own matter)
    Ws, bs = sq.weights_as_lists(out["weights"], out["biases"])
    n = cfg.total_test_instances
    x = sq.extract_windows(40, out["test_coded"], 0, n)
    c64, _, o64 = O.decode(x, Ws, bs, 1, np.float64)
    bad = out["predictions"] != c64
    assert not bad.any() or O.top2_margin(o64)[bad].max() < TOL
    act, w2, b2 = sq.build_network(x[:50], cfg, out["weights"], out["biases"])
    assert np.abs(act - o64[:50]).max() < TOL
    # reload: same test stream (same seed), same predictions
    cfg.test_old_model, cfg.model_load_path = True, path
    np.random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()) as b2:
        again = sq.train_and_evaluate(cfg)
    assert f"Model loaded from {path}" in b2.getvalue()
    assert np.array_equal(again["predictions"], out["predictions"])


def test_fnn_api_falls_back_to_the_wide_path_for_other_hidden_widths(cuda_device):
    """n_hidden_* are free parameters of the reference's DecoderConfig; widths the fused kernel has no
    instantiation for are decoded by the layer-at-a-time HIP path (still on the GPU)."""
    cfg = fv.DecoderConfig(block_length=7, n_hidden_1=200, n_hidden_2=96, n_hidden_3=48,
                           samples_per_class_train=1, samples_per_class_test=3)
    weights, biases = fv.initialize_weights(cfg, rng=np.random.RandomState(2))
    np.random.seed(4)
    _, _, te, tel, _ = fv.train_test_dataset(cfg)
    labels = np.argmax(tel, 1).astype(np.int32)
    preds, counters = fv.predict(te, weights, biases, labels=labels, tile_bits=7)
    Ws, bs = fv.weights_as_lists(weights, biases)
    out = fv.multilayer_perceptron(te, weights, biases)
    _check(preds, out, te, Ws, bs)
    assert np.array_equal(counters, O.error_counters(preds, labels, 7))


# ------------------------------------------------------------------ bit-wise decoder (ReLU)
def test_bitwise_decoder_forward_matches_oracle(cuda_device):
    """2(6+L) -> 30 -> 2 with ReLU on both layers (src/fnn_viterbi_bitwise.py:388-397): outputs tie
    at 0.0 when both pre-activations are negative, and argmax then says class 0."""
    from nnviterbi_amd import fnn_viterbi_bitwise as bw
    cfg = bw.BitwiseDecoderConfig(block_length=512, training_size=4, test_size=3000, test_snr_index=3)
    from nnviterbi_amd.decoder import WideMLP as _W
    np.random.seed(1)
    info, enc = bw.generate_test_data(cfg)
    w, b = bw.initialize_weights(cfg, rng=np.random.RandomState(5))
    b["out"] = np.array([-0.3, 0.1], dtype=np.float32)
    Ws, bs = bw.active_layers(w, b)
    x = enc.astype(np.float32)
    preds = bw.predict(enc, w, b)
    out, _, _ = bw.build_network(enc, cfg, w, b)
    c64, _, o64 = O.decode(x, Ws, bs, 1, np.float64, activation="relu")
    assert np.abs(out - o64).max() < 2e-5 * max(1.0, np.abs(o64).max())
    bad = preds != c64
    assert not bad.any() or O.top2_margin(o64)[bad].max() < 1e-4
    ref = O.decode_c(x, Ws, bs, 1, want_bits=False, activation="relu")
    assert (ref["cls"] != preds).sum() <= 3
    both_zero = (o64 == 0).all(axis=1)
    assert both_zero.any() and (preds[both_zero] == 0).all()            # ReLU'd ties -> first class
    assert preds.dtype == np.int64 and set(np.unique(preds)) <= {0, 1}


def test_bitwise_train_and_evaluate_end_to_end(cuda_device, tmp_path):
    from nnviterbi_amd import fnn_viterbi_bitwise as bw
    cfg = bw.BitwiseDecoderConfig(block_length=24, training_size=4000, test_size=1000, training_epochs=15,
                                  batch_size=200, learning_rate=0.005, snr_db_range=(4, 8), test_snr_index=2)
    np.random.seed(2)
    tr_i, tr_e = bw.generate_training_data(cfg)
    te_i, te_e = bw.generate_test_data(cfg)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        preds, err = bw.train_and_evaluate(cfg, tr_i, tr_e, te_i, te_e, model_save_path=str(tmp_path / "bw"))
    text = buf.getvalue()
    assert "Epoch: 0015 | cost = " in text and "Optimization Finished!" in text and "Test Error Rate: " in text
    assert err == np.mean(preds != te_i[:, 0])      # (ReLU on the logits can die: accuracy is the reference's matter)
    w, b = bw.train_and_evaluate.last_model
    Ws, bs = bw.active_layers(w, b)
    c64, _, o64 = O.decode(te_e.astype(np.float32), Ws, bs, 1, np.float64, activation="relu")
    bad = preds != c64
    assert not bad.any() or O.top2_margin(o64)[bad].max() < 1e-4
    assert sorted(np.load(str(tmp_path / "bw.npz")).files)[0] == "bias_h1"


def test_relu_network_on_the_layer_at_a_time_path(cuda_device):
    """WideMLP(activation="relu"), narrow fp32 layer (32-column tiles) + head kernel: the generic path for
    ReLU networks (bw.predict itself now takes the fused single-launch kernel for n_hidden <= 32)."""
    for dims in ([1036, 30, 2], [300, 48, 2]):
        Ws, bs = _net(dims, 4)
        bs[-1] = np.array([-0.2, 0.05], np.float32)
        x = np.trunc(np.random.RandomState(3).randn(3001, dims[0]) * 1.5).astype(np.float32)
        net = WideMLP(Ws, bs, activation="relu")
        r = net.decode(torch.from_numpy(x).cuda(), want_act=True)
        c64, _, o64 = O.decode(x, Ws, bs, 1, np.float64, activation="relu")
        assert np.abs(r["act"].cpu().numpy() - o64).max() < 2e-5 * max(1.0, np.abs(o64).max())
        bad = r["cls32"].cpu().numpy() != c64
        assert not bad.any() or O.top2_margin(o64)[bad].max() < 1e-4
        net.close()


def test_relu_network_in_f16x3_precision(cuda_device):
    dims = [1036, 30, 2]
    Ws, bs = _net(dims, 9)
    x = np.trunc(np.random.RandomState(2).randn(2000, 1036) * 1.5).astype(np.float32)
    net = WideMLP(Ws, bs, activation="relu", precision="f16x3")
    r = net.decode(torch.from_numpy(x).cuda(), want_act=True)
    c64, _, o64 = O.decode(x, Ws, bs, 1, np.float64, activation="relu")
    assert np.abs(r["act"].cpu().numpy() - o64).max() < 2e-5 * max(1.0, np.abs(o64).max())
    bad = r["cls32"].cpu().numpy() != c64
    assert not bad.any() or O.top2_margin(o64)[bad].max() < 1e-4
    net.close()


def test_default_precision_switch_reaches_the_reference_shaped_modules(cuda_device):
    """decoder.set_default_precision / NNV_PRECISION: objects built without an explicit precision
    (the mirrored modules build theirs internally) follow it; results stay inside the tolerances."""
    from nnviterbi_amd import decoder, fnn_viterbi as fv
    cfg = fv.DecoderConfig(block_length=7)
    w, b = fv.initialize_weights(cfg, rng=np.random.RandomState(6))
    x = (np.random.RandomState(7).randn(4000, 14) * 1.1).astype(np.float32)
    exact, _ = fv.predict(x, w, b)
    decoder.set_default_precision("f16x3")
    try:
        assert decoder.PackedMLP(*fv.weights_as_lists(w, b)).precision == "f16x3"
        fast, _ = fv.predict(x, w, b)
        wide = decoder.WideMLP(*_net([16, 256, 128], 0))          # > 32 classes: the default falls back to f32
        assert wide.precision == "f32"
        wide.close()
    finally:
        decoder.set_default_precision("f32")
    assert (fast != exact).sum() <= 2
    assert decoder.PackedMLP(*fv.weights_as_lists(w, b)).precision == "f32"
    with pytest.raises(ValueError):
        decoder.set_default_precision("bf16")


# ------------------------------------------------------------------ K6: the whole chain in one launch
def _both_paths(net, fn, expect_fused=True):
    """fn() through the handle's default kernel and through the layer-at-a-time path (WideMLP.set_path, C ABI
    nnv_wide_set_path); the handle says which kernel each call ran."""
    net.set_path("auto")
    assert net.path == ("fused" if expect_fused else "layered")
    fused = fn()
    net.set_path("layered")
    assert net.path == "layered"
    layered = fn()
    net.set_path("auto")
    return fused, layered


FUSED_SHAPES = [
    ([200, 500, 500, 500, 500, 2], "tanh"),    # the reference's sliding-window network
    ([14, 300, 70, 5], "tanh"),                # row length not a multiple of 8, widths not multiples of 32, 5 classes
    ([20, 64, 2], "tanh"),                     # one hidden layer, 2 of 16 waves active
    ([30, 129, 33, 32], "tanh"),               # 33 -> two neuron tiles with one live neuron in the second; 32 classes
    ([512, 512, 512, 3], "tanh"),              # the widest eligible row and layers (128 KB of activations)
    ([10, 40, 40, 40, 40, 40, 40, 40, 7], "tanh"),  # 7 hidden layers (the most), odd class count (row strides are even)
    ([66, 30, 2], "relu"),                     # a small bit-wise style network: ReLU on every layer, logits included
    ([200, 500, 500, 2], "relu"),
    # wide class layers (33..512 classes): one more MFMA layer + an argmax over the LDS buffer
    ([14, 200, 96, 48, 128], "tanh"),          # the reference's b = 7 decoder with other hidden widths
    ([16, 256, 128, 64, 256], "tanh"),
    ([20, 64, 500], "tanh"),                   # 500 classes: 12 padding classes in the last tile must never win
    ([16, 512, 256, 128, 512], "tanh"),        # the widest eligible class layer
    ([30, 40, 33], "relu"),                    # 33 classes: one live class in the second tile; ReLU ties at 0 -> first index
]


@pytest.mark.parametrize("dims,activation", FUSED_SHAPES)
@pytest.mark.parametrize("n", [1, 63, 64, 65, 4099])
def test_fused_chain_matches_the_oracle_and_the_layered_path(cuda_device, dims, activation, n):
    """K6 (nnv_wide_fused.hip): the hidden activations never leave the CU.  Same contract as the layer-at-a-time
    path -- outputs within 1e-5 of float64, classes differ only on float64 near-ties -- and the two HIP paths agree
    with each other to the last bits of the class layer's different summation order (checked: 2e-6)."""
    Ws, bs = _net(dims, 11 + len(dims))
    net = WideMLP(Ws, bs, activation=activation)
    x = (np.random.RandomState(n).randn(n, dims[0]) * 0.9).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    fused, layered = _both_paths(net, lambda: net.decode(xd, want_act=True, want_cls64=True))
    act = fused["act"].cpu().numpy()
    cls = fused["cls32"].cpu().numpy()
    c64, _, o64 = O.decode(x, Ws, bs, 1, np.float64, activation=activation)
    # tanh outputs are bounded: absolute 1e-5.  ReLU outputs are not: the layer-at-a-time tests' relative form
    tol = TOL if activation == "tanh" else 2e-5 * max(1.0, np.abs(o64).max())
    assert np.abs(act - o64).max() < tol
    bad = cls != c64
    assert not bad.any() or O.top2_margin(o64)[bad].max() < 10 * tol
    assert np.array_equal(fused["cls64"].cpu().numpy(), cls.astype(np.int64))
    assert np.abs(act - layered["act"].cpu().numpy()).max() < 0.2 * tol
    differ = cls != layered["cls32"].cpu().numpy()
    assert not differ.any() or O.top2_margin(o64)[differ].max() < 10 * tol
    net.close()


def test_fused_chain_on_a_stream_with_every_alignment(cuda_device):
    """Windows of a stream (stride 2: 8-byte aligned rows), the same rows as a matrix (16-byte aligned when the row
    length allows), a stream that does not start on a 16-byte boundary, and a tail block of 1 row."""
    dims = [200, 500, 500, 500, 500, 2]
    Ws, bs = _net(dims, 3)
    net = WideMLP(Ws, bs)
    n = 64 * 5 + 1
    s = _stream(n + 1, 200, 17)
    sd = torch.from_numpy(s).cuda()
    x = np.ascontiguousarray(np.lib.stride_tricks.sliding_window_view(s, 200)[::2])
    for first in (0, 1):                                    # window `first` starts 8 * first bytes into the allocation
        view = sd[2 * first:]
        fused, layered = _both_paths(net, lambda: net.decode_stream(view, n, want_act=True))
        rows = x[first:first + n]
        _check(fused["cls32"].cpu().numpy(), fused["act"].cpu().numpy(), rows, Ws, bs)
        assert np.abs(fused["act"].cpu().numpy() - layered["act"].cpu().numpy()).max() < 2e-6
        mat = net.decode(torch.from_numpy(rows.copy()).cuda(), want_act=True)
        assert torch.equal(mat["cls32"], fused["cls32"]) and torch.equal(mat["act"], fused["act"])
    net.close()


def test_fused_chain_is_deterministic_and_row_independent(cuda_device):
    """One launch twice: identical bits.  A row decoded alone, in the middle of other rows and in another 64-row
    block: identical outputs (rows are columns of the MFMA's B operand; nothing crosses them)."""
    dims = [200, 500, 500, 500, 500, 2]
    Ws, bs = _net(dims, 5)
    net = WideMLP(Ws, bs)
    rng = np.random.RandomState(8)
    x = rng.randn(1000, 200).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    a = net.decode(xd, want_act=True)
    b = net.decode(xd, want_act=True)
    assert torch.equal(a["act"], b["act"]) and torch.equal(a["cls32"], b["cls32"])
    perm = rng.permutation(1000)
    c = net.decode(xd[torch.from_numpy(perm).cuda()].contiguous(), want_act=True)
    assert torch.equal(c["act"], a["act"][torch.from_numpy(perm).cuda()])
    one = net.decode(xd[417:418].contiguous(), want_act=True)
    assert torch.equal(one["act"][0], a["act"][417])
    x2 = x.copy()
    x2[5] = np.nan                                            # a non-finite row stays inside its row
    d = net.decode(torch.from_numpy(x2).cuda(), want_act=True)
    keep = np.arange(1000) != 5
    assert torch.equal(d["act"][torch.from_numpy(keep).cuda()], a["act"][torch.from_numpy(keep).cuda()])
    net.close()


def test_shapes_outside_the_fused_kernel_take_the_layered_path(cuda_device):
    """Hidden width above 512, row length above 512, a class layer wider than 512 and the f16x3 mode are not eligible:
    the layer-at-a-time path decodes them whatever the handle is pinned to, and pinning them to K6 is refused."""
    for dims, kw in (([100, 600, 2], {}), ([1036, 30, 2], {"activation": "relu"}), ([16, 256, 128, 64, 1024], {}),
                     ([200, 500, 500, 2], {"precision": "f16x3"})):
        Ws, bs = _net(dims, 2)
        net = WideMLP(Ws, bs, **kw)
        x = np.random.RandomState(3).randn(300, dims[0]).astype(np.float32)
        xd = torch.from_numpy(x).cuda()
        a, b = _both_paths(net, lambda: net.decode(xd, want_act=True), expect_fused=False)
        assert torch.equal(a["act"], b["act"]) and torch.equal(a["cls32"], b["cls32"])
        with pytest.raises(Exception, match="unsupported"):
            net.set_path("fused")
        assert net.path == "layered"
        c64, _, o64 = O.decode(x, Ws, bs, 1, np.float64, activation=kw.get("activation", "tanh"))
        assert np.abs(a["act"].cpu().numpy() - o64).max() < 2e-5 * max(1.0, np.abs(o64).max())
        net.close()


def test_fused_chain_is_hip_graph_capturable(cuda_device):
    """K6's launch path allocates and synchronises nothing and sets no kernel attribute (prepack did): the whole
    sliding-window decode is ONE graph node, capturable without a warm-up launch and replayable on new data."""
    dims = [200, 500, 500, 500, 500, 2]
    Ws, bs = _net(dims, 9)
    net = WideMLP(Ws, bs)
    n = 3000
    streams = [torch.from_numpy(_stream(n, 200, 40 + k)).cuda() for k in range(2)]
    s = streams[0].clone()
    cls = torch.empty(n, dtype=torch.int32, device="cuda")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = net._lib.nnv_wide_decode_f32(net._handle, s.data_ptr(), n, 2, net._workspace.data_ptr(),
                                          net._workspace.numel(), cls.data_ptr(), None, None,
                                          torch.cuda.current_stream().cuda_stream)
        assert st == 0
    for src in streams:
        s.copy_(src)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(cls, net.decode_stream(src, n)["cls32"])
    net.close()


def test_sliding_window_stream_generated_and_decoded_on_the_device(cuda_device):
    """fnn_viterbi_seq_predict.generate_test_stream_on_device / evaluate_stream_on_device: a long test stream (random bits
    -> (133,171) as ONE stream from zero state -> BPSK + AWGN -> truncation toward zero, src/fnn_viterbi_seq_predict.py:
    255-286 + F12) made by K3's continuous mode, every window decoded in one persistent launch.  The stream equals the
    generator's numpy model run through the restated reference encoder; the error count equals the host pipeline's
    (predict_stream on the downloaded stream)."""
    from oracle import awgn_model
    n_bits, sigma, seed = 4096, 0.5, 77
    info, coded = sq.generate_test_stream_on_device(n_bits, sigma, seed)
    rows = n_bits // 8
    mx_, ml = awgn_model.generate("133171", 8, rows, sigma, seed, rows_per_block=rows, continuous=True)
    assert np.array_equal(info.cpu().numpy(), O.class_to_bits(ml, 8).reshape(-1))
    model_stream = np.trunc(mx_.reshape(-1))
    got = coded.cpu().numpy()
    # truncation is discontinuous at integers: a float32 value within 1e-5 of one may land on the other side
    differ = got != model_stream
    assert differ.mean() < 1e-3 and np.all(np.abs(mx_.reshape(-1)[differ] - np.round(mx_.reshape(-1)[differ])) < 1e-4)
    cfg = sq.SequentialDecoderConfig()
    w, b = sq.initialize_weights(cfg, rng=np.random.RandomState(3))
    r = sq.evaluate_stream_on_device(w, b, n_bits, sigma, seed)
    n_inst = (2 * n_bits - cfg.window_length) // 2 + 1
    assert r["decoded_bits"] == n_inst
    host_pred = sq.predict_stream(got[None, :], n_inst, w, b)
    assert r["bit_errors"] == int((host_pred != info.cpu().numpy()[:n_inst]).sum())
    assert 0.0 <= r["ber"] <= 1.0
    with pytest.raises(ValueError):
        sq.evaluate_stream_on_device(w, b, 64, sigma, seed)          # 128 symbols: shorter than one 200-symbol window


# ------------------------------------------------ block_length 13 and 15: the sizes the reference's own tests touch
@pytest.mark.parametrize("b,n", [(13, 2999), (15, 1031), (16, 257)])
def test_decoder_config_block_lengths_beyond_twelve(cuda_device, b, n):
    """The reference's DecoderConfig is exercised up to block_length 15 in its own tests (tests/test_utilities.py:
    342-348: n_input 30, 32768 classes; src/fnn_viterbi.py:65-73).  Since round 5 the register-chained kernel K1 takes
    these shapes itself (n_in <= 32; the 2^b-class layer streamed from L2 per wave WITH its biases, which no longer
    need LDS), and the wide path's layer-at-a-time kernels take them too (class layer as one fp32-MFMA GEMM launch,
    argmax fused into its epilogue): BOTH are held to the float64 restatement under the 1e-5 rule here, through
    ``fnn_viterbi.predict`` / ``multilayer_perceptron``, the torch ops and an explicit WideMLP, ragged row counts;
    counters exact functions of the classes.  2^16 classes is the header's stated limit for both
    (NNV_MLP_MAX_CLASSES, NNV_WIDE_MAX_CLASSES); beyond it prepack refuses, loudly."""
    from nnviterbi_amd import decoder
    cfg = fv.DecoderConfig(block_length=b)
    assert (cfg.n_input, cfg.n_classes) == (2 * b, 2 ** b)
    w, bd = fv.initialize_weights(cfg, rng=np.random.RandomState(b))
    bd = {k: (np.random.RandomState(b + 1).randn(*v.shape) * 0.1).astype(np.float32) for k, v in bd.items()}
    Ws, bs = fv.weights_as_lists(w, bd)
    rng = np.random.RandomState(100 + b)
    cb = fv.generate_class_bits(b)
    labels = rng.randint(0, 2 ** b, n).astype(np.int32)
    x = (np.where(fv.encode_133171_batch(cb[labels]) == 1, 1.0, -1.0) + cfg.sigma * rng.randn(n, 2 * b)).astype(np.float32)
    c64, _, o64 = O.decode(x, Ws, bs, b, np.float64)
    margin = O.top2_margin(o64)
    decoder.clear_pack_cache()
    # which kernel: K1's stream variant, a handful of class tiles resident, the rest (and their biases) from L2
    k1 = decoder.PackedMLP(Ws, bs)
    info = k1.launch_info(n)
    assert info["out_layer_in_lds"] == 0 and 0 <= info["class_tiles_in_lds"] < 2 ** b // 32 and info["lds_bytes"] <= 160 * 1024
    for precision in ("f32", "f16x3"):
        kp = k1 if precision == "f32" else decoder.PackedMLP(Ws, bs, precision=precision)
        r = kp.decode(torch.from_numpy(x).cuda(), labels=torch.from_numpy(labels).cuda(), tile_bits=b, want_bits=True)
        kc = r["cls32"].cpu().numpy()
        bad = kc != c64
        assert not bad.any() or margin[bad].max() < TOL, precision
        assert np.array_equal(r["counters"].cpu().numpy(), O.error_counters(kc, labels, b))
        assert np.array_equal(r["bits"].cpu().numpy(), O.class_to_bits(kc, b))
        assert np.abs(kp.forward(torch.from_numpy(x[:101]).cuda()).cpu().numpy() - o64[:101]).max() < TOL
        kp.close()
    # the wide path on the same network (what ran these shapes before K1 did)
    wide = WideMLP(Ws, bs)
    assert wide.path == "layered"
    wc = wide.decode(torch.from_numpy(x).cuda())["cls32"].cpu().numpy()
    bad = wc != c64
    assert not bad.any() or margin[bad].max() < TOL
    wide.close()
    # reference-shaped calls
    preds, counters = fv.predict(x, w, bd, labels=labels, tile_bits=b)
    assert preds.dtype == np.int64 and preds.shape == (n,)
    bad = preds != c64
    assert not bad.any() or margin[bad].max() < TOL
    assert np.array_equal(counters, O.error_counters(preds, labels, b))
    out = fv.multilayer_perceptron(x[:201], w, bd)
    assert out.shape == (201, 2 ** b) and np.abs(out - o64[:201]).max() < TOL
    # torch ops: raw weights and prepacked handle
    xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda()
    Wt = [torch.from_numpy(a).cuda() for a in Ws]
    bt = [torch.from_numpy(a).cuda() for a in bs]
    cls, bits, cnt = torch.ops.nnviterbi.mlp_decode(xd, Wt, bt, b, ld)
    assert np.array_equal(cls.cpu().numpy(), preds)
    assert np.array_equal(bits.cpu().numpy(), O.class_to_bits(preds, b))
    assert np.array_equal(cnt.cpu().numpy(), O.error_counters(preds, labels, b))
    h = torch.ops.nnviterbi.mlp_prepack(Wt, bt, "f32")
    cls2, _, cnt2 = torch.ops.nnviterbi.mlp_decode_packed(xd, h, b, ld)
    assert torch.equal(cls2, cls) and torch.equal(cnt2, cnt)
    torch.ops.nnviterbi.mlp_free(h)
    decoder.clear_pack_cache()


def test_class_layers_beyond_the_stated_limit_are_refused(cuda_device):
    from nnviterbi_amd.decoder import PackedMLP
    Ws = [np.zeros((34, 32), np.float32), np.zeros((32, 2 ** 17), np.float32)]
    bs = [np.zeros(32, np.float32), np.zeros(2 ** 17, np.float32)]
    with pytest.raises(Exception, match="unsupported"):
        WideMLP(Ws, bs)
    with pytest.raises(Exception, match="unsupported"):                # K1: 2^17 classes, and 34 inputs
        PackedMLP([np.zeros((32, 128), np.float32), np.zeros((128, 2 ** 17), np.float32)],
                  [np.zeros(128, np.float32), np.zeros(2 ** 17, np.float32)])
    with pytest.raises(Exception, match="unsupported"):
        PackedMLP([np.zeros((34, 128), np.float32), np.zeros((128, 64), np.float32)],
                  [np.zeros(128, np.float32), np.zeros(64, np.float32)])
    Ws = [np.zeros((20000, 32), np.float32), np.zeros((32, 4), np.float32)]
    bs = [np.zeros(32, np.float32), np.zeros(4, np.float32)]
    with pytest.raises(Exception, match="unsupported"):
        WideMLP(Ws, bs)


@pytest.mark.parametrize("activation", ["tanh", "relu"])
@pytest.mark.parametrize("n_classes,n", [(65, 1000), (100, 129), (128, 4099), (129, 777), (500, 2000), (1000, 1), (4097, 513)])
def test_fused_class_argmax_epilogue_equals_the_argmax_over_the_stored_row(cuda_device, n_classes, n, activation):
    """Round 5: a wide class layer on the layer-at-a-time path keeps only (maximum, class) per row and 128-column block
    (dense_tanh_kernel<.., CLS>) and merges them (partial_argmax_kernel); with the debug output requested the same call
    stores the activated row and scans it (row_argmax_kernel).  Same values, same first-maximum rule: the classes must be
    IDENTICAL, padding columns (act(0) = 0 would beat negative outputs) must never win, and both equal np.argmax of the
    stored outputs.  ReLU makes exact ties at 0.0 common: the lowest class index wins in both."""
    dims = [24, 40, n_classes]
    Ws, bs = _net(dims, n_classes, bias_scale=0.3)
    if activation == "tanh":
        bs[-1] = bs[-1] - 2.0                      # most outputs negative: a padding column's tanh(0) = 0 would win if unmasked
    net = WideMLP(Ws, bs, activation=activation, chunk_rows=512)
    net.set_path("layered")
    x = torch.from_numpy(np.random.RandomState(n).randn(n, dims[0]).astype(np.float32)).cuda()
    fused = net.decode(x, want_cls64=True)                   # argmax in the GEMM epilogue
    stored = net.decode(x, want_act=True, want_cls64=True)   # activated rows stored, then scanned
    assert torch.equal(fused["cls32"], stored["cls32"]) and torch.equal(fused["cls64"], stored["cls64"])
    act = stored["act"].cpu().numpy()
    assert np.array_equal(fused["cls32"].cpu().numpy(), act.argmax(axis=1))
    assert int(fused["cls32"].max()) < n_classes
    c64, _, o64 = O.decode(x.cpu().numpy(), Ws, bs, 1, np.float64, activation=activation)
    bad = fused["cls32"].cpu().numpy() != c64
    assert not bad.any() or O.top2_margin(o64)[bad].max() < TOL * max(1.0, np.abs(o64).max())
    net.close()
