"""The single-product 16-bit mode (precision="f16", NNV_PRECISION_F16): SURVEY section 7 step 4's "fast mode with
measured BER delta".  Every operand is rounded ONCE to f16 (11 significant bits), so this mode is deliberately
OUTSIDE the 1e-5 parity contract of tests/test_gpu_parity.py and is never a default.  Its own stated tolerance
(measured on the MI355X, profiles/r03): tanh'd outputs within F16_TOL = 1e-2 absolute of the float64 restatement
(observed <= 2e-3 on the reference's architectures with initialisation-scale weights), a decoded class differs
from the float64 argmax only where the float64 top-2 margin is below 2 * F16_TOL, and -- BASELINE.json's bar for a
reduced-precision mode -- |BER - BER_f32| <= 1e-4 on the trained network at 2 dB on the same noise realisation.
Bits and counters stay exact integer functions of the decoded class."""
import os

import numpy as np
import pytest
import torch

import nnviterbi_amd
from nnviterbi_amd import fnn_viterbi as fv
from nnviterbi_amd.decoder import PackedMLP, Seq2SeqLSTM, WideMLP, generate_awgn_blocks
from oracle import fnn_oracle as O

pytestmark = pytest.mark.gpu
F16_TOL = 1e-2


def _net(dims, seed):
    rng = np.random.RandomState(seed)
    Ws = [fv.truncated_normal((dims[i], dims[i + 1]), np.sqrt(2.0 / dims[i]), rng) for i in range(len(dims) - 1)]
    bs = [(rng.randn(d) * 0.1).astype(np.float32) for d in dims[1:]]
    return Ws, bs


@pytest.mark.parametrize("dims,b", [([16, 128, 64, 32, 256], 8), ([14, 128, 64, 32, 128], 7), ([20, 128, 64, 1024], 10),
                                    ([20, 128, 64, 32, 1024], 10), ([16, 128, 256], 8), ([6, 128, 64, 32, 8], 3),
                                    ([16, 100, 50, 20, 250], 8)])
@pytest.mark.parametrize("n", [1, 33, 4099])
def test_f16_mode_within_its_stated_tolerance(cuda_device, dims, b, n):
    Ws, bs = _net(dims, 3 * b + len(dims))
    rng = np.random.RandomState(n)
    x = (rng.randn(n, dims[0]) * 1.2).astype(np.float32)
    labels = rng.randint(0, dims[-1], n).astype(np.int32)
    net = PackedMLP(Ws, bs, precision="f16")
    xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda()
    full = net.decode(xd, labels=ld, tile_bits=b, want_cls64=True, want_bits=True)
    act = net.forward(xd).cpu().numpy()
    c64, _, o64 = O.decode(x, Ws, bs, b, np.float64)
    assert np.abs(act - o64).max() < F16_TOL
    cls = full["cls32"].cpu().numpy()
    bad = cls != c64
    if bad.any():
        assert O.top2_margin(o64)[bad].max() < 2 * F16_TOL
    assert np.array_equal(full["bits"].cpu().numpy(), O.class_to_bits(cls, b))
    assert np.array_equal(full["counters"].cpu().numpy(), O.error_counters(cls, labels, b))
    if dims[-1] % 32 == 0 and dims[0] % 8 == 0:            # the lean call shape decodes the same classes
        lean = net.decode(xd, labels=ld, tile_bits=b)
        assert torch.equal(lean["cls32"], full["cls32"]) and torch.equal(lean["counters"], full["counters"])
    net.close()


def test_f16_mode_ber_delta_on_the_trained_network(cuda_device):
    """BASELINE.json: a reduced-precision mode must keep BER within 1e-4 of the reference arithmetic.  Trained
    config-3 network, 8 M rows of 2 dB AWGN (the headline's generator and seed)."""
    from nnviterbi_amd import fnn_viterbi_mixed_snr as mx
    ckpt = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        "neural-network-based-viterbi-decoding_amd", "weights", "fnn_133171_b8_2dB.npz")
    Ws, bs = fv.weights_as_lists(*mx.load_model(ckpt))
    n = 8_000_000
    x, lab = generate_awgn_blocks("133171", 8, n, 0.5616748812614791, 20240607, rows_per_block=8)
    res = {}
    for prec in ("f32", "f16"):
        net = PackedMLP(Ws, bs, precision=prec)
        r = net.decode(x, labels=lab, tile_bits=8)
        res[prec] = (r["cls32"], r["counters"].cpu().numpy())
        net.close()
    ber = {k: v[1][0] / v[1][2] for k, v in res.items()}
    flips = int((res["f32"][0] != res["f16"][0]).sum().item())
    assert abs(ber["f16"] - ber["f32"]) <= 1e-4
    assert flips <= n // 500                                 # measured: ~3e-4 of rows, all low-margin
    assert 4e-3 < ber["f32"] < 6e-3


def test_f16_mode_is_explicit_and_mlp_only(cuda_device):
    from nnviterbi_amd import decoder
    with pytest.raises(ValueError):
        decoder.set_default_precision("f16")               # never a default: outside the parity contract
    Ws, bs = _net([200, 500, 2], 1)
    with pytest.raises(ValueError):
        WideMLP(Ws, bs, precision="f16")
    from nnviterbi_amd import rnn_viterbi as rv
    with pytest.raises(ValueError):
        Seq2SeqLSTM(rv.initialize_parameters(rv.RNNDecoderConfig(encoder_hidden_units=8), rng=np.random.RandomState(0)),
                    precision="f16")
    big = _net([16, 128, 64, 32, 256], 0)
    big[0][1][3, 5] = 70000.0
    with pytest.raises(nnviterbi_amd.NNViterbiError):        # outside the f16 range, as in f16x3
        PackedMLP(*big, precision="f16")
