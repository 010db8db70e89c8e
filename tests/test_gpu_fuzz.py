"""Seeded shape fuzz of the HIP paths against the oracle (MI355X).

Every case draws a network shape, a row count and inputs from a fixed seed, runs the HIP path in
both precision modes and applies the stated tolerances of the dedicated suites:
  K1 / K4 (feed-forward): tanh'd outputs within 1e-5 of the float64 restatement, a class may
      differ from the float64 argmax only on a float64 near-tie (margin < 1e-5); bits / counters
      exact functions of the class   (tests/test_gpu_parity.py, tests/test_gpu_wide.py)
  K5 (recurrent): logits within 5e-5 of the float64 restatement (tests/test_gpu_rnn.py)
The point is the corners the hand-picked shapes miss: odd input widths, class counts that are not
multiples of 32, ragged row counts, one-unit layers, class layers straddling the LDS limit."""
import numpy as np
import pytest
import torch

import nnviterbi_amd  # noqa: F401
from nnviterbi_amd import fnn_viterbi as fv
from nnviterbi_amd import rnn_viterbi as rv
from nnviterbi_amd.decoder import PackedMLP, Seq2SeqLSTM, WideMLP
from oracle import fnn_oracle as O
from oracle import rnn_oracle as R

pytestmark = pytest.mark.gpu
TOL = 1e-5
# NNV_FUZZ_SCALE=k multiplies the number of seeds per path (one-off soak runs; default 1)
import os  # noqa: E402
SCALE = max(1, int(os.environ.get("NNV_FUZZ_SCALE", "1")))


def _net(dims, rng, bias_scale=0.1):
    Ws = [fv.truncated_normal((dims[i], dims[i + 1]), np.sqrt(2.0 / dims[i]), rng)
          for i in range(len(dims) - 1)]
    bs = [(rng.randn(dims[i + 1]) * bias_scale).astype(np.float32) for i in range(len(dims) - 1)]
    return Ws, bs


def _check_ff(cls, act, x, Ws, bs, activation="tanh"):
    o64 = O.forward(x, Ws, bs, np.float64, activation=activation)
    assert np.abs(act - o64).max() < TOL
    c64 = O.argmax_first(o64)
    bad = cls != c64
    if bad.any():
        assert O.top2_margin(o64)[bad].max() < TOL, "class differs on a row that is not a near-tie"


def _pinned_paths_agree(net, xd, labels_d, b, r):
    """K1 (persistent) and, where the network allows it, K1w (split) pinned explicitly: the same classes, bits and
    counters as the automatic choice, bit for bit."""
    from nnviterbi_amd import _cabi
    for path in ("persistent", "split"):
        try:
            net.set_path(path)
        except _cabi.NNViterbiError:
            continue                      # K1w does not take this network (precision / hidden widths)
        q = net.decode(xd, labels=labels_d, tile_bits=b, want_cls64=True, want_bits=True)
        for k in ("cls32", "cls64", "bits", "counters"):
            assert torch.equal(q[k], r[k]), (path, k)
    net.set_path("auto")


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("seed", range(12 * SCALE))
def test_fused_decoder_random_shapes(cuda_device, seed, precision):
    """K1: hidden widths are the first 1..3 of the reference's (128, 64, 32), everything else random."""
    rng = np.random.RandomState(1000 + seed)
    n_hidden = rng.randint(1, 4)
    n_in = int(rng.randint(2, 25))
    b = int(rng.randint(1, 11))
    n_classes = int(2 ** b if rng.rand() < 0.5 else rng.randint(2, 2 ** b + 1))
    dims = [n_in] + [128, 64, 32][:n_hidden] + [n_classes]
    n = int(rng.choice([1, 2, 31, 33, 63, 64, 65, 500, 2049]))
    Ws, bs = _net(dims, rng)
    x = (rng.randn(n, n_in) * 1.3).astype(np.float32)
    labels = rng.randint(0, n_classes, size=n).astype(np.int32)
    net = PackedMLP(Ws, bs, precision=precision)
    xd = torch.from_numpy(x).cuda()
    r = net.decode(xd, labels=torch.from_numpy(labels).cuda(), tile_bits=b, want_cls64=True, want_bits=True)
    cls = r["cls32"].cpu().numpy()
    _check_ff(cls, net.forward(xd).cpu().numpy(), x, Ws, bs)
    assert np.array_equal(r["cls64"].cpu().numpy(), cls.astype(np.int64))
    assert np.array_equal(r["bits"].cpu().numpy(), O.class_to_bits(cls, b))
    assert np.array_equal(r["counters"].cpu().numpy(), O.error_counters(cls, labels, b))
    _pinned_paths_agree(net, xd, torch.from_numpy(labels).cuda(), b, r)
    # the classes+counters call shape (lean instantiation where the shape allows it) and the host-array
    # entry (pinned ring, random chunking / thread count) must give the very same classes and counters
    lean = net.decode(xd, labels=torch.from_numpy(labels).cuda(), tile_bits=b)
    assert np.array_equal(lean["cls32"].cpu().numpy(), cls) and torch.equal(lean["counters"], r["counters"])
    hc, hcnt = net.decode_host(x, labels, b, chunk_rows=int(rng.choice([1, 33, 100, 4096])), threads=int(rng.randint(0, 4)))
    assert np.array_equal(hc, cls.astype(np.int64)) and np.array_equal(hcnt, r["counters"].cpu().numpy())
    net.close()


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("seed", range(6 * SCALE))
def test_fused_decoder_random_shapes_long_rows(cuda_device, seed, precision):
    """K1 on row lengths 25..32 (round 5: block_length 13..16; odd lengths included) with class layers from a handful
    of classes to 20,000 (resident, banded and streamed with their biases), ragged row counts, every output."""
    rng = np.random.RandomState(7000 + seed)
    n_hidden = rng.randint(1, 4)
    n_in = int(rng.randint(25, 33))
    n_classes = int(rng.choice([3, 64, 257, 1000, 4096, 5000, 20000]))
    b = max(1, int(np.ceil(np.log2(n_classes))))
    dims = [n_in] + [128, 64, 32][:n_hidden] + [n_classes]
    n = int(rng.choice([1, 33, 64, 500, 2049, 40001]))
    Ws, bs = _net(dims, rng)
    x = (rng.randn(n, n_in) * 1.3).astype(np.float32)
    labels = rng.randint(0, n_classes, size=n).astype(np.int32)
    net = PackedMLP(Ws, bs, precision=precision)
    xd = torch.from_numpy(x).cuda()
    r = net.decode(xd, labels=torch.from_numpy(labels).cuda(), tile_bits=b, want_cls64=True, want_bits=True)
    cls = r["cls32"].cpu().numpy()
    m = min(n, 600)                                            # the [rows, n_classes] float64 outputs: a sample
    _check_ff(cls[:m], net.forward(xd[:m]).cpu().numpy(), x[:m], Ws, bs)
    ref = O.decode_c(x, Ws, bs, b, labels=labels, want_bits=False)
    diff = np.flatnonzero(cls != ref["cls"])
    assert len(diff) <= max(2, n // 2000)                       # fp32 vs fp32: near-ties only
    if len(diff):
        o64 = O.forward(x[diff], Ws, bs, np.float64)
        assert O.top2_margin(o64).max() < TOL
    assert np.array_equal(r["cls64"].cpu().numpy(), cls.astype(np.int64))
    assert np.array_equal(r["bits"].cpu().numpy(), O.class_to_bits(cls, b))
    assert np.array_equal(r["counters"].cpu().numpy(), O.error_counters(cls, labels, b))
    _pinned_paths_agree(net, xd, torch.from_numpy(labels).cuda(), b, r)
    lean = net.decode(xd, labels=torch.from_numpy(labels).cuda(), tile_bits=b)
    assert np.array_equal(lean["cls32"].cpu().numpy(), cls) and torch.equal(lean["counters"], r["counters"])
    net.close()


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("seed", range(6 * SCALE))
def test_fused_decoder_random_shapes_many_rows(cuda_device, seed, precision):
    """K1 over SEVERAL rounds of its persistent workgroups (40 k - 150 k rows: every wave of every workgroup decodes
    more than one row tile, ragged tails), class counts up to 4096 so that the resident, lean and stream variants
    all occur.  Every row against the C oracle; rows that differ must be float64 near-ties."""
    rng = np.random.RandomState(5000 + seed)
    n_hidden = rng.randint(1, 4)
    n_in = int(2 * rng.randint(1, 13))
    b = int(rng.randint(3, 13))
    n_classes = int(2 ** b if rng.rand() < 0.6 else rng.randint(2 ** (b - 1) + 1, 2 ** b + 1))
    dims = [n_in] + [128, 64, 32][:n_hidden] + [n_classes]
    n = int(rng.choice([40_003, 70_001, 150_017]))
    Ws, bs = _net(dims, rng)
    x = (rng.randn(n, n_in) * 1.3).astype(np.float32)
    labels = rng.randint(0, n_classes, size=n).astype(np.int32)
    try:
        net = PackedMLP(Ws, bs, precision=precision)
    except nnviterbi_amd.NNViterbiError:
        pytest.skip("no instantiation for this shape (1 hidden layer x thousands of classes)")
    r = net.decode(torch.from_numpy(x).cuda(), labels=torch.from_numpy(labels).cuda(), tile_bits=b)
    cls = r["cls32"].cpu().numpy()
    ref = O.decode_c(x, Ws, bs, b, labels=labels, want_bits=False)
    diff = np.flatnonzero(cls != ref["cls"])
    assert len(diff) <= 10
    if len(diff):
        o64 = O.forward(x[diff], Ws, bs, np.float64)
        bad = cls[diff] != O.argmax_first(o64)
        assert not bad.any() or O.top2_margin(o64)[bad].max() < TOL
    assert np.array_equal(r["counters"].cpu().numpy(), O.error_counters(cls, labels, b))
    net.close()


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("seed", range(8 * SCALE))
def test_bitwise_bank_random_shapes(cuda_device, seed, precision):
    """K4b: random input width (even), hidden width <= 32, 1..40 positions, ragged rows."""
    from nnviterbi_amd.decoder import BitwiseBank
    rng = np.random.RandomState(4000 + seed)
    K, H = int(2 * rng.randint(1, 300)), int(rng.randint(1, 33))
    P, n = int(rng.randint(1, 41)), int(rng.choice([1, 63, 128, 129, 1000, 2500]))
    models = []
    for _ in range(P):
        Ws, bs = _net([K, H, 2], rng, bias_scale=0.3)
        models.append((Ws[0], bs[0], Ws[1], bs[1]))
    x = np.trunc(rng.randn(n, K) * 1.5).astype(np.float32)
    labels = (rng.rand(P, n) > 0.5).astype(np.uint8)
    bank = BitwiseBank(models, precision=precision)
    r = bank.decode(torch.from_numpy(x).cuda(), labels=torch.from_numpy(labels).cuda(), want_act=True)
    bits, act, err = r["bits"].cpu().numpy(), r["act"].cpu().numpy(), r["errors"].cpu().numpy()
    for p_ in range(P):
        W1, b1, Wo, bo = models[p_]
        o64 = O.forward(x, [W1, Wo], [b1, bo], np.float64, activation="relu")
        assert np.abs(act[p_] - o64).max() < 2e-5 * max(1.0, np.abs(o64).max())
        bad = bits[p_] != O.argmax_first(o64)
        assert not bad.any() or O.top2_margin(o64)[bad].max() < 1e-4
        assert err[p_] == int((bits[p_] != labels[p_]).sum())
    bank.close()


@pytest.mark.parametrize("seed", range(10 * SCALE))
def test_bitwise_bank_kernel_choice_random_shapes(cuda_device, seed):
    """The fp32 bank's other two kernels, which the shapes above seldom reach: K4d (1-2 positions, rows a multiple of 4
    floats, up to 1152 inputs; row counts past one tile per workgroup) on even seeds, K4c (>= 33 positions) on odd ones."""
    from nnviterbi_amd.decoder import BitwiseBank
    rng = np.random.RandomState(4500 + seed)
    if seed % 2 == 0:
        K, P = int(4 * rng.randint(1, 289)), int(rng.randint(1, 3))
        n = int(rng.choice([1, 31, 33, 1000, 8193, 20001]))
    else:
        K, P = int(2 * rng.randint(1, 200)), int(rng.randint(33, 70))
        n = int(rng.choice([1, 63, 129, 1000, 9000]))
    H = int(rng.randint(1, 33))
    models = []
    for _ in range(P):
        Ws, bs = _net([K, H, 2], rng, bias_scale=0.3)
        models.append((Ws[0], bs[0], Ws[1], bs[1]))
    x = np.trunc(rng.randn(n, K) * 1.5).astype(np.float32)
    labels = (rng.rand(P, n) > 0.5).astype(np.uint8)
    bank = BitwiseBank(models, precision="f32")
    r = bank.decode(torch.from_numpy(x).cuda(), labels=torch.from_numpy(labels).cuda(), want_act=True)
    bits, act, err = r["bits"].cpu().numpy(), r["act"].cpu().numpy(), r["errors"].cpu().numpy()
    for p_ in rng.permutation(P)[:6]:
        W1, b1, Wo, bo = models[p_]
        o64 = O.forward(x, [W1, Wo], [b1, bo], np.float64, activation="relu")
        assert np.abs(act[p_] - o64).max() < 2e-5 * max(1.0, np.abs(o64).max())
        bad = bits[p_] != O.argmax_first(o64)
        assert not bad.any() or O.top2_margin(o64)[bad].max() < 1e-4
        assert err[p_] == int((bits[p_] != labels[p_]).sum())
    bank.close()


@pytest.mark.parametrize("seed", range(10 * SCALE))
def test_wide_decoder_random_shapes(cuda_device, seed):
    """K4: 2..6 layers of arbitrary widths, matrix rows or a strided stream, tanh or ReLU."""
    rng = np.random.RandomState(2000 + seed)
    depth = int(rng.randint(2, 7))
    dims = [int(2 * rng.randint(1, 120))] + [int(rng.randint(1, 600)) for _ in range(depth - 1)] + [int(rng.randint(2, 300))]
    activation = "relu" if seed % 3 == 2 else "tanh"
    precision = "f16x3" if (seed % 2 == 1 and dims[-1] <= 32) else "f32"
    if precision == "f16x3" or seed % 4 == 0:
        dims[-1] = int(rng.randint(2, 33))                 # narrow head kernel
    n = int(rng.choice([1, 127, 129, 1000, 3001]))
    Ws, bs = _net(dims, rng)
    net = WideMLP(Ws, bs, activation=activation, precision=precision)
    if seed % 2 == 0:                                       # windows of one stream, stride 2
        stream = rng.randn(2 * (n - 1) + dims[0]).astype(np.float32)
        x = np.stack([stream[2 * i: 2 * i + dims[0]] for i in range(n)])
        r = net.decode_stream(torch.from_numpy(stream).cuda(), n, want_act=True)
    else:
        x = rng.randn(n, dims[0]).astype(np.float32)
        r = net.decode(torch.from_numpy(x).cuda(), want_act=True)
    _check_ff(r["cls32"].cpu().numpy(), r["act"].cpu().numpy(), x, Ws, bs, activation)
    net.close()


@pytest.mark.parametrize("seed", range(12 * SCALE))
def test_fused_chain_random_shapes(cuda_device, seed):
    """K6 (one persistent launch for the whole chain): random eligible shapes -- 1..7 hidden layers of width 1..512, an even
    row length <= 512, 2..32 classes (vector-ALU head) or 33..512 (MFMA class layer + LDS argmax), tanh or ReLU, matrix
    rows or windows of a stream -- against float64 AND against the
    layer-at-a-time path (set_path), whose hidden layers are bit-identical by construction."""
    rng = np.random.RandomState(5000 + seed)
    n_hidden = int(rng.randint(1, 8))
    top = int(rng.choice([40, 130, 512]))
    n_classes = int(rng.randint(2, 33)) if seed % 4 < 2 else int(rng.randint(33, 513))
    dims = [int(2 * rng.randint(1, 257))] + [int(rng.randint(1, top + 1)) for _ in range(n_hidden)] + [n_classes]
    activation = "relu" if seed % 3 == 2 else "tanh"
    n = int(rng.choice([1, 31, 33, 64, 95, 1000, 3001]))
    Ws, bs = _net(dims, rng)
    net = WideMLP(Ws, bs, activation=activation)
    if seed % 2 == 0:
        stream = rng.randn(2 * (n - 1) + dims[0]).astype(np.float32)
        x = np.stack([stream[2 * i: 2 * i + dims[0]] for i in range(n)])
        run = lambda: net.decode_stream(torch.from_numpy(stream).cuda(), n, want_act=True)
    else:
        x = rng.randn(n, dims[0]).astype(np.float32)
        xd = torch.from_numpy(x).cuda()
        run = lambda: net.decode(xd, want_act=True)
    assert net.path == "fused"
    fused = run()
    net.set_path("layered")
    assert net.path == "layered"
    layered = run()
    act = fused["act"].cpu().numpy()
    _check_ff(fused["cls32"].cpu().numpy(), act, x, Ws, bs, activation)
    scale = 1.0 if activation == "tanh" else max(1.0, float(np.abs(act).max()))
    assert np.abs(act - layered["act"].cpu().numpy()).max() < 4e-6 * scale
    net.close()


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("seed", range(8 * SCALE))
def test_recurrent_decoder_random_shapes(cuda_device, seed, precision):
    """K5: hidden 1..22, independent encoder/decoder lengths (0 allowed for the encoder), ragged batch."""
    rng = np.random.RandomState(3000 + seed)
    H = int(rng.randint(1, 23))
    t_enc, t_dec, batch = int(rng.randint(0, 50)), int(rng.randint(1, 50)), int(rng.randint(1, 200))
    cfg = rv.RNNDecoderConfig(encoder_hidden_units=H)
    p = rv.initialize_parameters(cfg, rng=rng)
    for k in ("enc_bias", "dec_bias", "proj_b"):
        p[k] = (rng.randn(*p[k].shape) * 0.2).astype(np.float32)
    enc = np.trunc(rng.randn(t_enc, batch) * 1.5).astype(np.float32)
    dec = np.sign(rng.randn(t_dec, batch)).astype(np.float32)
    n_bits = int(rng.randint(1, t_dec + 1))
    info = (rng.rand(n_bits, batch) > 0.5).astype(np.uint8)
    net = Seq2SeqLSTM(p, precision=precision)
    r = net.forward(torch.from_numpy(enc).cuda().reshape(t_enc, batch), torch.from_numpy(dec).cuda(),
                    info_bits=torch.from_numpy(info).cuda().reshape(n_bits, batch), n_bits=n_bits, want_bits=True)
    got = r["logits"].cpu().numpy()
    want, _ = R.seq2seq_forward(enc, dec, p, np.float64)
    assert np.abs(got - want).max() < 5e-5
    bits = r["bits"].cpu().numpy()
    assert np.array_equal(bits, (got[:n_bits] > 0).astype(np.uint8))
    assert np.array_equal(r["counters"].cpu().numpy(), R.error_counters(bits.T, info.T))
    net.close()
