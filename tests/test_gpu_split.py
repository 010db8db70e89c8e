"""K1w (csrc/nnv_mlp_split.hip, the decode kernel for short inputs) against K1 (csrc/nnv_mlp_decode.hip) on the same
packed network and the same rows: the same classes, bits and counters BIT FOR BIT (one arithmetic, one decision rule),
and both against the float64 oracle under the 1e-5 near-tie rule like every other parity test."""
import numpy as np
import pytest

import nnviterbi_amd  # noqa: F401
from oracle import fnn_oracle as FO


def _net(n_in, n_classes, rng, scale=1.0, nhid=3):
    dims = [n_in] + [128, 64, 32][:nhid] + [n_classes]
    Ws = [(rng.randn(dims[i], dims[i + 1]) * scale * np.sqrt(2.0 / dims[i])).astype(np.float32) for i in range(nhid + 1)]
    bs = [(rng.randn(dims[i + 1]) * 0.1).astype(np.float32) for i in range(nhid + 1)]
    return Ws, bs


def _decode(net, x, labels, path, tile_bits):
    import torch
    net.set_path(path)
    assert net.path_for(x.shape[0]) == path
    r = net.decode(x, tile_bits=tile_bits, labels=labels, want_bits=True, want_cls64=True)
    torch.cuda.synchronize()
    return r


SHAPES = [(14, 128, 7, 3), (16, 256, 8, 3), (20, 1024, 10, 3), (12, 64, 6, 3), (13, 100, 7, 3), (32, 4096, 12, 3),
          (7, 33, 6, 3), (26, 8192, 13, 3),
          (20, 1024, 10, 2), (14, 100, 7, 2), (16, 256, 8, 1), (9, 50, 6, 1), (24, 2048, 11, 1)]   # BASELINE configs[1] is a 2-hidden-layer network


@pytest.mark.gpu
@pytest.mark.parametrize("n_in,n_classes,tile_bits,nhid", SHAPES)
@pytest.mark.parametrize("n_rows", [1, 31, 32, 33, 2000])
def test_split_equals_persistent_bit_for_bit(n_in, n_classes, tile_bits, nhid, n_rows):
    import torch
    from nnviterbi_amd.decoder import PackedMLP
    rng = np.random.RandomState(n_in * 7 + n_rows + nhid)
    Ws, bs = _net(n_in, n_classes, rng, nhid=nhid)
    x = torch.tensor(rng.randn(n_rows, n_in).astype(np.float32), device="cuda")
    labels = torch.tensor(rng.randint(0, n_classes, n_rows).astype(np.int32), device="cuda")
    net = PackedMLP(Ws, bs)
    a = _decode(net, x, labels, "persistent", tile_bits)
    b = _decode(net, x, labels, "split", tile_bits)
    for k in ("cls32", "cls64", "bits", "counters"):
        assert torch.equal(a[k], b[k]), k
    # and against the oracle (float64 forward, 1e-5 near-tie rule)
    out = FO.forward(x.cpu().numpy(), Ws, bs, dtype=np.float64)
    ref = out.argmax(axis=1)
    got = b["cls32"].cpu().numpy()
    bad = np.nonzero(got != ref)[0]
    for i in bad:
        assert abs(out[i, got[i]] - out[i, ref[i]]) < 1e-5
    net.set_path("auto")
    assert net.path_for(n_rows) == "split" and net.path_for(40000) == "persistent"
    net.close()


@pytest.mark.gpu
def test_split_flat_top_rows_follow_the_tanh_argmax():
    """Saturated class outputs (pre-activations beyond the flat top of fp32 tanh): both kernels take the lowest index of
    the tanh'd maximum, as argmax over the reference's tanh outputs does (src/fnn_viterbi.py:606-610)."""
    import torch
    from nnviterbi_amd.decoder import PackedMLP
    rng = np.random.RandomState(5)
    Ws, bs = _net(16, 256, rng)
    Ws[3] = (Ws[3] * 40.0).astype(np.float32)          # class pre-activations far beyond 9
    x = torch.tensor(rng.randn(500, 16).astype(np.float32), device="cuda")
    net = PackedMLP(Ws, bs)
    a = _decode(net, x, None, "persistent", 8)
    b = _decode(net, x, None, "split", 8)
    assert torch.equal(a["cls32"], b["cls32"])
    out32 = FO.forward(x.cpu().numpy(), Ws, bs, dtype=np.float32)
    assert (out32.max(axis=1) == 1.0).mean() > 0.5                # the case is what it claims to be
    # numpy's tanhf and the kernels' exp/rcp tanh do not saturate at exactly the same pre-activation, so a few rows
    # whose several classes sit right at the edge of 1.0f take another of the (near-)tied indices
    assert (b["cls32"].cpu().numpy() == FO.argmax_first(out32)).mean() > 0.95
    net.close()


@pytest.mark.gpu
def test_split_refuses_the_16_bit_precisions():
    from nnviterbi_amd import _cabi
    from nnviterbi_amd.decoder import PackedMLP
    rng = np.random.RandomState(1)
    Ws, bs = _net(14, 128, rng)
    for precision in ("f16x3", "f16"):
        net = PackedMLP(Ws, bs, precision=precision)     # 16-bit operands: K1 only
        with pytest.raises(_cabi.NNViterbiError):
            net.set_path("split")
        assert net.path_for(100) == "persistent"
        net.close()
