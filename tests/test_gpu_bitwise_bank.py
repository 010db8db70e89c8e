"""The bit-wise decoder BANK (C ABI nnv_bitwise_*, decoder.BitwiseBank): P networks
2(6+L) -> 30 (ReLU) -> 2 (ReLU) -> argmax, one per information-bit position
(src/fnn_viterbi_bitwise.py:326-399,503-505), in one fused launch - against the oracle PER POSITION
(numpy float64 arbiter + the C fp32 restatement, activation="relu"), and against the package's own
one-position-per-launch path.  Parity of the forward arithmetic stays unpinned (no TensorFlow)."""
import numpy as np
import pytest
import torch

import nnviterbi_amd  # noqa: F401
from nnviterbi_amd import fnn_viterbi_bitwise as bw
from nnviterbi_amd.decoder import BitwiseBank
from oracle import fnn_oracle as O

pytestmark = pytest.mark.gpu


def _models(cfg, n_pos, seed, bout=None):
    models = []
    for p in range(n_pos):
        w, b = bw.initialize_weights(cfg, rng=np.random.RandomState(seed + p))
        b["b1"] = (np.random.RandomState(1000 + seed + p).randn(cfg.n_hidden_1) * 0.05).astype(np.float32)
        b["out"] = np.array([-0.3, 0.1] if bout is None else bout, dtype=np.float32)
        models.append((w, b))
    return models


def _check_position(bits, act, x, w, b):
    Ws, bs = bw.active_layers(w, b)
    c64, _, o64 = O.decode(x, Ws, bs, 1, np.float64, activation="relu")
    assert np.abs(act - o64).max() < 2e-5 * max(1.0, np.abs(o64).max())
    bad = bits != c64
    assert not bad.any() or O.top2_margin(o64)[bad].max() < 1e-4
    ref = O.decode_c(x, Ws, bs, 1, want_bits=False, activation="relu")
    assert (ref["cls"] != bits).sum() <= max(3, len(bits) // 1000)
    both_zero = (o64 == 0).all(axis=1)
    assert (bits[both_zero] == 0).all()                  # ReLU'd ties -> first class (bit 0)
    return int(bad.sum())


@pytest.mark.parametrize("precision", ["f32", "f16x3"])   # exact fp32 MFMA GEMM | hi/lo f16 planes (round 3), heads fp32
@pytest.mark.parametrize("block_length,n_pos,n_rows", [(512, 9, 1500), (24, 24, 777), (24, 1, 300),
                                                       (24, 4, 128), (25, 6, 129), (40, 3, 1)])
def test_bank_matches_oracle_per_position(cuda_device, block_length, n_pos, n_rows, precision):
    cfg = bw.BitwiseDecoderConfig(block_length=block_length, training_size=4, test_size=n_rows, test_snr_index=3)
    np.random.seed(block_length + n_pos)
    info, enc = bw.generate_test_data(cfg)
    models = _models(cfg, n_pos, seed=7)
    x = enc.astype(np.float32)
    bank = BitwiseBank([(w["h1"], b["b1"], w["out"], b["out"]) for w, b in models], precision=precision)
    labels = torch.from_numpy(np.ascontiguousarray(info[:, :n_pos].T.astype(np.uint8))).cuda()
    r = bank.decode(torch.from_numpy(x).cuda(), labels=labels, want_act=True)
    bits, act, errors = r["bits"].cpu().numpy(), r["act"].cpu().numpy(), r["errors"].cpu().numpy()
    assert bits.shape == (n_pos, n_rows) and act.shape == (n_pos, n_rows, 2)
    for p, (w, b) in enumerate(models):
        _check_position(bits[p].astype(np.int64), act[p], x, w, b)
        assert errors[p] == int((bits[p] != info[:, p]).sum())       # counters: exact integer function of the bits
    # accumulate: a second launch adds to the same counters
    r2 = bank.decode(torch.from_numpy(x).cuda(), labels=labels, errors=r["errors"])
    assert np.array_equal(r2["errors"].cpu().numpy(), 2 * errors)
    bank.close()


def test_bank_equals_the_one_position_per_launch_path(cuda_device):
    """predict_all_positions (one fused launch) == predict() called per trained network."""
    cfg = bw.BitwiseDecoderConfig(block_length=64, training_size=4, test_size=5000, test_snr_index=5)
    np.random.seed(3)
    info, enc = bw.generate_test_data(cfg)
    models = _models(cfg, 10, seed=21, bout=[0.05, 0.0])
    preds, err = bw.predict_all_positions(enc, models, test_info=info)
    assert preds.shape == (5000, 10) and preds.dtype == np.int64
    for p, (w, b) in enumerate(models):
        single = bw.predict(enc, w, b)
        assert (single != preds[:, p]).sum() <= 2                     # K4 head kernel vs fused epilogue: near-ties only
        assert err[p] == pytest.approx(np.mean(preds[:, p] != info[:, p]))
    assert np.array_equal(bw.predict_all_positions(enc, models), preds)


def test_bank_argument_validation(cuda_device):
    cfg = bw.BitwiseDecoderConfig(block_length=24)
    models = _models(cfg, 2, seed=1)
    bank = BitwiseBank([(w["h1"], b["b1"], w["out"], b["out"]) for w, b in models])
    x = torch.zeros((10, cfg.n_input), device="cuda")
    assert bank.decode(x[:0])["bits"].shape == (2, 0)                 # empty input: no-op
    with pytest.raises(ValueError):
        bank.decode(x[:, :-2].contiguous())
    with pytest.raises(ValueError):
        bank.decode(x, errors=torch.zeros(2, dtype=torch.int64, device="cuda"))   # errors need labels
    with pytest.raises(ValueError):
        bank.decode(x, labels=torch.zeros((10, 2), dtype=torch.uint8, device="cuda"))  # [P, N], not [N, P]
    bank.close()
    big = [(w["h1"].copy(), b["b1"], w["out"], b["out"]) for w, b in models]
    big[1][0][3, 2] = 70000.0                                          # outside the f16 range: f16x3 refuses, f32 does not
    with pytest.raises(nnviterbi_amd.NNViterbiError):
        BitwiseBank(big, precision="f16x3")
    BitwiseBank(big, precision="f32").close()
    with pytest.raises(ValueError):
        BitwiseBank(big, precision="f16")                              # the single-product mode is nnv_mlp_* only
    with pytest.raises(Exception):
        BitwiseBank([(np.zeros((60, 40), np.float32), np.zeros(40, np.float32),       # n_hidden > 32
                      np.zeros((40, 2), np.float32), np.zeros(2, np.float32))])


@pytest.mark.parametrize("block_length,n_pos,n_rows", [(512, 40, 1500), (24, 24, 777), (25, 17, 129), (40, 3, 1),
                                                       (100, 33, 4099)])
def test_streamed_bank_equals_the_tiled_bank_bit_for_bit(cuda_device, block_length, n_pos, n_rows):
    """K4c (rows resident in LDS, every position's first layer streamed from L2; the fp32 bank's default from 33
    positions on since round 4) against K4b (LDS-tiled GEMM; BitwiseBank.set_path pins either): the same k order in the first layer and the same
    sequential 32-term head chain, so decided bits, ReLU'd outputs and error counters are IDENTICAL, not just close."""
    cfg = bw.BitwiseDecoderConfig(block_length=block_length, training_size=4, test_size=n_rows, test_snr_index=3)
    np.random.seed(3 * block_length + n_pos)
    info, enc = bw.generate_test_data(cfg)
    models = _models(cfg, n_pos, seed=21)
    x = torch.from_numpy(enc.astype(np.float32)).cuda()
    labels = torch.from_numpy(np.ascontiguousarray(np.resize(info.T, (n_pos, n_rows)).astype(np.uint8))).cuda()
    bank = BitwiseBank([(w["h1"], b["b1"], w["out"], b["out"]) for w, b in models], precision="f32")
    assert bank.path_for(x) == ("stream" if n_pos >= 33 else "tiled")    # the default: K4c from 33 positions on
    bank.set_path("stream")                            # K4c whatever the bank's size
    assert bank.path_for(x) == "stream"
    a = bank.decode(x, labels=labels, want_act=True)
    bank.set_path("tiled")
    assert bank.path_for(x) == "tiled"
    b = bank.decode(x, labels=labels, want_act=True)
    bank.set_path("stream")
    assert torch.equal(a["bits"], b["bits"]) and torch.equal(a["act"], b["act"]) and torch.equal(a["errors"], b["errors"])
    # and a view of x that starts 8 bytes into an allocation (rows 8-byte aligned only)
    if enc.shape[1] % 4 == 2:
        pad = torch.empty(x.numel() + 2, dtype=torch.float32, device="cuda")
        pad[2:] = x.reshape(-1)
        view = pad[2:].view_as(x)
        assert bank.path_for(view) == "stream"
        c = bank.decode(view, labels=labels, want_act=True)
        assert torch.equal(c["bits"], a["bits"]) and torch.equal(c["act"], a["act"])
    bank.close()


def _bank_inputs(block_length, n_pos, n_rows, seed):
    cfg = bw.BitwiseDecoderConfig(block_length=block_length, training_size=4, test_size=n_rows, test_snr_index=3)
    np.random.seed(seed)
    info, enc = bw.generate_test_data(cfg)
    models = _models(cfg, n_pos, seed=seed + 5)
    labels = np.ascontiguousarray(np.resize(info.T, (n_pos, n_rows)).astype(np.uint8))
    return cfg, enc.astype(np.float32), models, labels


# block_length -> n_input = 2 (6 + L): 1036 = 129.5 fragment groups (9 / 8 per wave, the row ends INSIDE the last group),
# 412 = 51.5 (4 / 3), 64 = 8 (1 / 0: half the waves own nothing), 144 = 18 (2 / 1), 1152 = 144 (9 / 9, the largest taken)
@pytest.mark.parametrize("precision", ["f32", "f16x3"])   # f16x3: steps of 16 inputs, hi / lo planes split in registers
@pytest.mark.parametrize("block_length,n_pos,n_rows", [(512, 1, 9001), (512, 2, 1500), (200, 1, 4099), (26, 2, 31),
                                                       (66, 1, 257), (570, 1, 640), (512, 1, 1), (28, 1, 100)])
def test_single_position_kernel_against_the_oracle_and_the_tiled_path(cuda_device, block_length, n_pos, n_rows,
                                                                      precision):
    """K4d (round 4; banks of up to 2 positions, 16-byte aligned rows): rows straight from HBM into the MFMA B operand,
    the contraction split over the 16 waves, partial sums meeting in LDS.  Per position against float64 and the C
    restatement; against K4b (set_path) the outputs agree to rounding -- the summation ORDER differs (16
    slice sums instead of one chain), so not bit for bit -- and the decided bits wherever the float64 margin is not a tie;
    counters are exact functions of the bits; a second launch returns the same bytes."""
    cfg, x_np, models, labels_np = _bank_inputs(block_length, n_pos, n_rows, seed=11 * block_length + n_pos)
    x = torch.from_numpy(x_np).cuda()
    labels = torch.from_numpy(labels_np).cuda()
    bank = BitwiseBank([(w["h1"], b["b1"], w["out"], b["out"]) for w, b in models], precision=precision)
    k4d_expected = n_pos == 1 or precision == "f32"          # f16x3: a single position only
    assert bank.path_for(x) == ("single" if k4d_expected else "tiled")   # the dispatch itself, not inferred from bytes
    a = bank.decode(x, labels=labels, want_act=True)
    a2 = bank.decode(x, labels=labels, want_act=True)
    assert torch.equal(a["bits"], a2["bits"]) and torch.equal(a["act"], a2["act"]) and torch.equal(a["errors"], a2["errors"])
    bank.set_path("tiled")
    assert bank.path_for(x) == "tiled"
    t = bank.decode(x, labels=labels, want_act=True)
    bank.set_path("auto")
    bits, act = a["bits"].cpu().numpy(), a["act"].cpu().numpy()
    t_bits, t_act = t["bits"].cpu().numpy(), t["act"].cpu().numpy()
    assert np.abs(act - t_act).max() < 2e-5 * max(1.0, np.abs(t_act).max())
    for p, (w, b) in enumerate(models):
        _check_position(bits[p].astype(np.int64), act[p], x_np, w, b)
        assert a["errors"][p].item() == int((bits[p] != labels_np[p]).sum())
        differ = bits[p] != t_bits[p]
        if differ.any():                                   # only where the two outputs are a rounding apart
            Ws, bs = bw.active_layers(w, b)
            o64 = O.decode(x_np[differ], Ws, bs, 1, np.float64, activation="relu")[2]
            assert O.top2_margin(o64).max() < 1e-4
    # rows that are only 8-byte aligned (a view two floats into an allocation) are not K4d's: the call must still be
    # served -- by the tiled path, whose bytes are known
    pad = torch.empty(x.numel() + 2, dtype=torch.float32, device="cuda")
    pad[2:] = x.reshape(-1)
    assert bank.path_for(pad[2:].view_as(x)) == "tiled"
    u = bank.decode(pad[2:].view_as(x), labels=labels, want_act=True)
    assert torch.equal(u["bits"], t["bits"]) and torch.equal(u["act"], t["act"]) and torch.equal(u["errors"], t["errors"])
    bank.close()


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_single_position_kernel_keeps_rows_apart(cuda_device, precision):
    """Non-finite soft values stay inside their row -- including across the row END: n_input = 1036 is 129.5 fragment
    groups, so the last group's upper half is the NEXT row's first four values, which K4d must never multiply (not even
    by a zero weight: 0 x NaN).  Clean rows decode to the same bytes with and without poisoned neighbours."""
    cfg, x_np, models, _ = _bank_inputs(512, 1, 700, seed=77)
    bank = BitwiseBank([(w["h1"], b["b1"], w["out"], b["out"]) for w, b in models], precision=precision)
    clean = bank.decode(torch.from_numpy(x_np).cuda(), want_act=True)
    dirty = x_np.copy()
    poisoned = np.array([1, 2, 40, 333, 699])
    dirty[poisoned] = np.nan
    dirty[2, :4] = np.inf                                   # the four values behind row 1's end
    dirty[40, 500] = -np.inf
    r = bank.decode(torch.from_numpy(dirty).cuda(), want_act=True)
    r2 = bank.decode(torch.from_numpy(dirty).cuda(), want_act=True)
    keep = np.ones(700, bool)
    keep[poisoned] = False
    keep_t = torch.from_numpy(keep).cuda()
    assert torch.equal(r["bits"][:, keep_t], clean["bits"][:, keep_t])
    assert torch.equal(r["act"][:, keep_t], clean["act"][:, keep_t])
    assert torch.equal(r["bits"], r2["bits"])               # the poisoned rows: unspecified but repeatable
    bank.close()


@pytest.mark.parametrize("n_pos,precision", [(1, "f32"), (2, "f32"), (9, "f32"), (40, "f32"), (1, "f16x3")])   # K4d, K4d x 2, K4b, K4c, K4d f16x3
def test_bank_decode_is_hip_graph_capturable(cuda_device, n_pos, precision):
    """Whatever kernel the bank's size selects, a decode is pure kernel launches on the caller's stream (every kernel
    attribute was set at prepack): capturable into a HIP graph without a warm-up launch, replayable on new rows."""
    cfg, x_np, models, _ = _bank_inputs(100, n_pos, 2100, seed=900 + n_pos)
    bank = BitwiseBank([(w["h1"], b["b1"], w["out"], b["out"]) for w, b in models], precision=precision)
    xs = [torch.from_numpy(x_np).cuda(), torch.from_numpy(np.ascontiguousarray(x_np[::-1])).cuda()]
    x = xs[0].clone()
    assert bank.path_for(x) == {1: "single", 2: "single", 9: "tiled", 40: "stream"}[n_pos]
    bits = torch.empty((n_pos, 2100), dtype=torch.uint8, device="cuda")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = bank._lib.nnv_bitwise_decode_f32(bank._handle, x.data_ptr(), 2100, bits.data_ptr(), None, None, None,
                                              torch.cuda.current_stream().cuda_stream)
        assert st == 0
    for src in xs:
        x.copy_(src)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(bits, bank.decode(src)["bits"])
    bank.close()
