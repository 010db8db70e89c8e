"""K8 (csrc/nnv_dense_train.hip, training.NativeDenseTrainer): the training step of the reference's bit-wise and
sliding-window decoders (src/fnn_viterbi_bitwise.py:438-452, src/fnn_viterbi_seq_predict.py:475-489) against the float64
restatement oracle/dense_train_oracle.py.  Parity unpinned against TensorFlow itself (not installable here): the
restatement's backprop is checked against central differences of its own loss below."""
import contextlib
import io
import os

import numpy as np
import pytest

import nnviterbi_amd  # noqa: F401
from oracle import dense_train_oracle as DO

TOL = 1e-6


def _net(dims, seed, scale=1.0):
    rng = np.random.RandomState(seed)
    Ws = [(rng.randn(dims[i], dims[i + 1]) * scale * np.sqrt(2.0 / dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [(rng.randn(dims[i + 1]) * 0.1).astype(np.float32) for i in range(len(dims) - 1)]
    return Ws, bs


@pytest.mark.parametrize("kind", ["tanh", "relu"])
def test_oracle_backprop_matches_central_differences(kind):
    dims = [9, 7, 5, 2]
    Ws, bs = _net(dims, 3)
    Ws, bs = [w.astype(np.float64) for w in Ws], [b.astype(np.float64) for b in bs]
    extra = [np.random.RandomState(4).randn(3, 4)]
    rng = np.random.RandomState(5)
    x = rng.randn(11, dims[0])
    y = np.eye(2)[rng.randint(0, 2, 11)]
    cost, dW, db, de = DO.loss_and_gradients(x, y, Ws, bs, kind, 1e-2, (0, 1), extra)
    assert abs(cost - DO.loss(x, y, Ws, bs, kind, 1e-2, (0, 1), extra)) < 1e-14
    eps = 1e-6
    for arrs, grads in ((Ws, dW), (bs, db), (extra, de)):
        for a, g in zip(arrs, grads):
            it = np.nditer(a, flags=["multi_index"])
            for _ in range(min(a.size, 12)):
                idx = it.multi_index
                old = a[idx]
                a[idx] = old + eps
                up = DO.loss(x, y, Ws, bs, kind, 1e-2, (0, 1), extra)
                a[idx] = old - eps
                dn = DO.loss(x, y, Ws, bs, kind, 1e-2, (0, 1), extra)
                a[idx] = old
                assert abs((up - dn) / (2 * eps) - g[idx]) < 1e-7, (idx, (up - dn) / (2 * eps), g[idx])
                it.iternext()


def test_torch_steps_of_the_mirror_modules_follow_the_oracle():
    """The PyTorch fallback's step formula (bit-wise decoder: ReLU, an unused-but-regularised h2) on the CPU in float64
    (TFAdam keeps beta1_power / beta2_power in float32 as TensorFlow does: 1e-7, not 1e-12)."""
    import torch
    from nnviterbi_amd.training import TFAdam
    dims = [12, 6, 2]
    Ws, bs = _net(dims, 1)
    h2 = np.random.RandomState(2).randn(6, 5).astype(np.float32)
    rng = np.random.RandomState(7)
    x = rng.randn(3 * 8, dims[0])
    lab = rng.randint(0, 2, 3 * 8)
    y = np.eye(2)[lab]
    ref = DO.Trainer(Ws, bs, "relu", 1e-3, 1e-2, (0,), [h2])
    W = [torch.tensor(w, dtype=torch.float64, requires_grad=True) for w in Ws]
    B = [torch.tensor(b, dtype=torch.float64, requires_grad=True) for b in bs]
    H2 = torch.tensor(h2, dtype=torch.float64, requires_grad=True)
    opt = TFAdam(W + B + [H2], lr=1e-3)
    for t in range(3):
        xb, yb = torch.tensor(x[8 * t:8 * t + 8]), torch.tensor(y[8 * t:8 * t + 8])
        pred = torch.relu(torch.relu(xb @ W[0] + B[0]) @ W[1] + B[1])
        cost = torch.nn.functional.binary_cross_entropy_with_logits(pred, yb) + 1e-2 * 0.5 * ((W[0] ** 2).sum() + (H2 ** 2).sum())
        opt.zero_grad(set_to_none=False)
        cost.backward()
        opt.step()
        assert abs(float(cost) - ref.step(x[8 * t:8 * t + 8], y[8 * t:8 * t + 8])) < 1e-7
    for a, b in zip(W + B + [H2], ref.Ws + ref.bs + ref.extra):
        assert np.abs(a.detach().numpy() - b).max() < 1e-7


CASES = [
    # dims, activation, batch, reg_layers, extra shape, row_stride (None: a [rows, n_in] matrix)
    ([32, 30, 2], "relu", 100, (0,), (30, 100), None),          # the bit-wise decoder's shape at block_length 10
    ([200, 40, 33, 31, 20, 2], "tanh", 100, (0, 1), None, 2),   # the sliding-window decoder's depth and window, widths that
                                                                # pad to a new tile (31 + the ones column = 32; 33 -> 64)
    ([20, 64, 2], "tanh", 37, (0, 1), None, None),              # ragged batch: 37 rows in two 32-row tiles
    ([7, 3], "relu", 5, (0,), None, 3),                         # one layer, three outputs, strided rows
]


@pytest.mark.gpu
@pytest.mark.parametrize("dims,kind,batch,reg,extra_shape,row_stride", CASES)
def test_native_dense_step_against_float64_restatement(dims, kind, batch, reg, extra_shape, row_stride):
    """Cost, every gradient (the extras' included), three Adam steps in ONE call with per-step costs, their running sum."""
    import torch
    from nnviterbi_amd.training import NativeDenseTrainer
    Ws, bs = _net(dims, 11, scale=0.7)
    extra = [np.random.RandomState(12).randn(*extra_shape).astype(np.float32) * 0.3] if extra_shape else None
    rng = np.random.RandomState(13)
    n_steps, n_in, n_out = 3, dims[0], dims[-1]
    stride = n_in if row_stride is None else row_stride
    flat = rng.randn((n_steps * batch - 1) * stride + n_in + 5).astype(np.float32)
    lab = rng.randint(0, n_out, n_steps * batch).astype(np.int32)
    rows = np.stack([flat[m * stride:m * stride + n_in] for m in range(n_steps * batch)])
    y = np.eye(n_out)[lab]
    beta, lr = 1e-2, 1e-3
    ref = DO.Trainer(Ws, bs, kind, lr, beta, reg, extra)
    nt = NativeDenseTrainer(Ws, bs, kind, lr, beta, batch, reg_layers=reg, extra=extra)
    xd, ld = torch.tensor(flat, device="cuda"), torch.tensor(lab, device="cuda")
    cost, gW, gb, ge = nt.gradients(xd, ld, row_stride=row_stride)
    rcost, dW, db, de = ref.gradients(rows[:batch], y[:batch])
    assert abs(cost - rcost) < 2e-6, (cost, rcost)
    for a, b in zip(gW + gb + ge, dW + db + de):
        assert np.abs(a.astype(np.float64) - b).max() < TOL
    costs = nt.steps(xd, ld, n_steps, row_stride=row_stride, step_stride=batch * stride, want_costs=True)
    rc = [ref.step(rows[t * batch:(t + 1) * batch], y[t * batch:(t + 1) * batch]) for t in range(n_steps)]
    assert np.abs(costs.cpu().numpy().astype(np.float64) - np.array(rc)).max() < 2e-6
    got_w, got_b, got_e, cost_sum = nt.export(want_cost_sum=True)
    assert abs(cost_sum - sum(rc)) < 1e-5
    for a, b in zip(got_w + got_b + got_e, ref.Ws + ref.bs + ref.extra):
        assert np.abs(a.astype(np.float64) - b).max() < 5e-6
    assert nt.export(want_cost_sum=True)[3] == 0.0          # the running sum was reset by the export
    nt.close()


@pytest.mark.gpu
def test_native_dense_trainer_refuses_what_it_cannot_take():
    import torch
    from nnviterbi_amd import _cabi
    from nnviterbi_amd.training import NativeDenseTrainer
    Ws, bs = _net([8, 4, 40], 1)
    with pytest.raises(_cabi.NNViterbiError):               # 40 outputs: more than the loss kernel's one tile
        NativeDenseTrainer(Ws, bs, "tanh", 1e-3, 0.0, 16)
    Ws, bs = _net([8, 4, 2], 1)
    nt = NativeDenseTrainer(Ws, bs, "tanh", 1e-3, 0.0, 16)
    with pytest.raises(ValueError):                          # too few elements for two steps
        nt.steps(torch.zeros(16 * 8 + 3, device="cuda"), torch.zeros(32, dtype=torch.int32, device="cuda"), 2)
    nt.close()


def _run_bitwise(env_trainer):
    from nnviterbi_amd import fnn_viterbi_bitwise as bw
    cfg = bw.BitwiseDecoderConfig(block_length=10, training_size=2000, test_size=500, batch_size=100, training_epochs=3)
    np.random.seed(5)
    tri, tre = bw.generate_training_data(cfg)
    tei, tee = bw.generate_test_data(cfg)
    old = os.environ.get("NNV_TRAINER")
    if env_trainer:
        os.environ["NNV_TRAINER"] = env_trainer
    else:
        os.environ.pop("NNV_TRAINER", None)
    try:
        np.random.seed(6)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            pred, err = bw.train_and_evaluate(cfg, tri, tre, tei, tee)
    finally:
        if old is None:
            os.environ.pop("NNV_TRAINER", None)
        else:
            os.environ["NNV_TRAINER"] = old
    return buf.getvalue(), bw.train_and_evaluate.last_model, pred, err


@pytest.mark.gpu
def test_bitwise_train_and_evaluate_native_and_autograd_agree():
    """src/fnn_viterbi_bitwise.py:402-509 through K8 and through PyTorch autograd: same printed costs (1e-5), same weights
    (2e-5; the unused h2 decays identically), same predictions."""
    n_out, n_model, n_pred, n_err = _run_bitwise(None)
    a_out, a_model, a_pred, a_err = _run_bitwise("autograd")
    costs = lambda s: [float(l.split("=")[1]) for l in s.splitlines() if l.startswith("Epoch:")]
    assert len(costs(n_out)) == 3 and np.abs(np.array(costs(n_out)) - np.array(costs(a_out))).max() < 1e-5
    for k in n_model[0]:
        assert np.abs(n_model[0][k] - a_model[0][k]).max() < 2e-5, k
    for k in n_model[1]:
        assert np.abs(n_model[1][k] - a_model[1][k]).max() < 2e-5, k
    assert (n_pred != a_pred).mean() < 0.01 and abs(n_err - a_err) < 0.01


@pytest.mark.gpu
def test_sliding_window_training_native_and_autograd_agree():
    """src/fnn_viterbi_seq_predict.py:421-577 through K8 (windows addressed in the stream, row stride 2) and through
    PyTorch autograd on the unfolded windows."""
    from nnviterbi_amd import fnn_viterbi_seq_predict as sp
    cfg = sp.SequentialDecoderConfig(total_train_length_info=1100, total_test_length_info=600, window_length=40,
                                     n_hidden_1=24, n_hidden_2=31, n_hidden_3=32, n_hidden_4=9, training_epochs=2,
                                     batch_size=50, beta_regularization=1e-3)
    np.random.seed(3)
    info, coded = sp.generate_training_data(cfg.total_train_length_info, cfg.training_size, cfg.sigma_vec)
    res = {}
    for mode in (None, "autograd"):
        old = os.environ.get("NNV_TRAINER")
        if mode:
            os.environ["NNV_TRAINER"] = mode
        else:
            os.environ.pop("NNV_TRAINER", None)
        try:
            np.random.seed(4)
            w0, b0 = sp.initialize_weights(cfg)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                res[mode] = sp._train(cfg, info, coded, w0, b0) + (buf.getvalue(),)
        finally:
            if old is None:
                os.environ.pop("NNV_TRAINER", None)
            else:
                os.environ["NNV_TRAINER"] = old
    costs = lambda s: [float(l.split("=")[1]) for l in s.splitlines() if l.startswith("Epoch:")]
    assert len(costs(res[None][2])) == 2
    assert np.abs(np.array(costs(res[None][2])) - np.array(costs(res["autograd"][2]))).max() < 1e-5
    for k in res[None][0]:
        assert np.abs(res[None][0][k] - res["autograd"][0][k]).max() < 2e-5, k
    for k in res[None][1]:
        assert np.abs(res[None][1][k] - res["autograd"][1][k]).max() < 2e-5, k


@pytest.mark.gpu
@pytest.mark.parametrize("dims,batch", [([18, 128, 64, 32, 512], 256), ([20, 128, 64, 32, 1024], 100), ([14, 128, 64, 100], 37),
                                        ([9, 128, 33], 64)])
def test_native_dense_softmax_step_against_the_fnn_restatement(dims, batch):
    """K8 with the feed-forward decoder's loss (softmax cross-entropy on the tanh outputs + beta l2 of the hidden layers,
    src/fnn_viterbi.py:549-559) at class counts K7 does not take (512, 1024) and at other depths: cost, all gradients,
    three steps against oracle/train_oracle.py."""
    import torch
    from oracle import train_oracle as TO
    from nnviterbi_amd.training import NativeDenseTrainer
    rng = np.random.RandomState(dims[-1] + batch)
    L = len(dims) - 1
    wk, bk = ["h1", "h2", "h3"][:L - 1] + ["out"], ["b1", "b2", "b3"][:L - 1] + ["out"]
    w = {k: (rng.randn(dims[i], dims[i + 1]) * np.sqrt(2.0 / dims[i])).astype(np.float32) for i, k in enumerate(wk)}
    bd = {k: (rng.randn(dims[i + 1]) * 0.1).astype(np.float32) for i, k in enumerate(bk)}
    x = rng.randn(3 * batch, dims[0]).astype(np.float32)
    lab = rng.randint(0, dims[-1], 3 * batch).astype(np.int32)
    y = np.eye(dims[-1], dtype=np.float32)[lab]
    ref = TO.Trainer(w, bd, 1e-4, 1e-3)
    nt = NativeDenseTrainer([w[k] for k in wk], [bd[k] for k in bk], "tanh", 1e-4, 1e-3, batch,
                            reg_layers=tuple(range(L - 1)), loss="softmax")
    xd, ld = torch.tensor(x, device="cuda"), torch.tensor(lab, device="cuda")
    cost, gW, gb, _ = nt.gradients(xd, ld)
    rcost, dW, db = ref.gradients(x[:batch], y[:batch])
    assert abs(cost - rcost) < 2e-6, (cost, rcost)
    for a, b in zip(gW + gb, dW + db):
        assert np.abs(a.astype(np.float64) - b).max() < TOL
    costs = nt.steps(xd, ld, 3, want_costs=True).cpu().numpy()
    rc = [ref.step(x[i * batch:(i + 1) * batch], y[i * batch:(i + 1) * batch]) for i in range(3)]
    assert np.abs(costs - np.array(rc)).max() < 2e-6
    got_w, got_b, _, cost_sum = nt.export(want_cost_sum=True)
    assert abs(cost_sum - sum(rc)) < 1e-5
    for a, b in zip(got_w + got_b, ref.Ws + ref.bs):
        assert np.abs(a.astype(np.float64) - b).max() < 5e-6
    nt.close()


@pytest.mark.gpu
def test_train_network_beyond_256_classes_runs_the_dense_kernels_and_agrees_with_autograd():
    """fnn_viterbi at block_length 9 (512 classes: K7 does not take it): train_network goes through K8 with the softmax
    loss and prints the same costs as the PyTorch-autograd trainer (1e-5), the weights agree to 2e-5."""
    from nnviterbi_amd import fnn_viterbi as fv, training
    cfg = fv.DecoderConfig(block_length=9, training_epochs=2, batch_size=128)
    rng = np.random.RandomState(3)
    n = 128 * 6
    x = rng.randn(n, cfg.n_input).astype(np.float32)
    y = np.eye(cfg.n_classes, dtype=np.float32)[rng.randint(0, cfg.n_classes, n)]
    cfg.num_training_samples = n
    res = {}
    for mode in (None, "autograd"):
        old = os.environ.get("NNV_TRAINER")
        if mode:
            os.environ["NNV_TRAINER"] = mode
        else:
            os.environ.pop("NNV_TRAINER", None)
        try:
            w0, b0 = fv.initialize_weights(cfg, rng=np.random.RandomState(5))
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                res[mode] = training.train_network(cfg, x, y, w0, b0) + (buf.getvalue(),)
        finally:
            if old is None:
                os.environ.pop("NNV_TRAINER", None)
            else:
                os.environ["NNV_TRAINER"] = old
    costs = lambda s: [float(l.split("=")[1]) for l in s.splitlines() if l.startswith("Epoch:")]
    assert len(costs(res[None][2])) == 2 and np.abs(np.array(costs(res[None][2])) - np.array(costs(res["autograd"][2]))).max() < 1e-5
    for k in res[None][0]:
        assert np.abs(res[None][0][k] - res["autograd"][0][k]).max() < 2e-5, k
