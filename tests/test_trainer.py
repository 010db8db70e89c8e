"""the trainer against a float64 numpy restatement of the reference's lines
(oracle/train_oracle.py: src/fnn_viterbi.py:549-563 -- softmax cross-entropy ON THE TANH OUTPUTS, beta * l2 on h1..h3
only, TF-1 AdamOptimizer with epsilon outside the bias correction; dropout src/fnn_viterbi_mixed_snr.py:596-627).

Loss, all eight gradients and the weights after three steps within 1e-6 of the restatement, with and without
dropout (mask fixed), on the CPU here and on the GPU (`-m gpu`), eager and as the captured HIP graph.
TensorFlow is not installable offline and the reference holds no training fixture: the restatement is pinned to the
reference's text and to central differences of its own loss, not to TensorFlow outputs ("parity unpinned")."""
import numpy as np
import pytest
import torch

import nnviterbi_amd  # noqa: F401
from nnviterbi_amd import fnn_viterbi as fv
from nnviterbi_amd import training
from oracle import train_oracle as TO

TOL = 1e-6


def _problem(b=6, batch=64, seed=3, nonzero_bias=True):
    cfg = fv.DecoderConfig(block_length=b, samples_per_class_train=max(4, -(-3 * batch // 2 ** b)), samples_per_class_test=1,
                           learning_rate=1e-4, beta_regularization=0.001, batch_size=batch)
    rng = np.random.RandomState(seed)
    w, bdict = fv.initialize_weights(cfg, rng=rng)
    if nonzero_bias:
        bdict = {k: (0.1 * rng.randn(*v.shape)).astype(np.float32) for k, v in bdict.items()}
    np.random.seed(seed)
    tr, trl, _, _, _ = fv.train_test_dataset(cfg)
    perm = rng.permutation(len(tr))[:3 * batch]            # three different batches, classes mixed
    dims = fv.layer_dims(cfg)
    masks = [[(rng.rand(batch, d) < 0.5).astype(np.float32) for d in dims[1:-1]] for _ in range(3)]
    return cfg, w, bdict, tr[perm], trl[perm], masks


def test_restatement_gradients_match_central_differences():
    """The oracle's hand-written backpropagation against numerical derivatives of its own loss (float64)."""
    cfg, w, b, x, y, masks = _problem(b=6, batch=16)
    ref = TO.Trainer(w, b, cfg.learning_rate, cfg.beta_regularization, keep_prob=0.5)
    for m in (None, [mm[:16] for mm in masks[0]]):
        xb, yb = x[:16], y[:16]
        cost, dW, db = ref.gradients(xb, yb, m)
        assert abs(cost - TO.loss(xb, yb, ref.Ws, ref.bs, ref.beta, m, 0.5)) < 1e-14
        rng = np.random.RandomState(0)
        for arrs, grads in ((ref.Ws, dW), (ref.bs, db)):
            for a, g in zip(arrs, grads):
                for _ in range(5):
                    idx = tuple(rng.randint(0, s) for s in a.shape)
                    old = a[idx]
                    a[idx] = old + 1e-6
                    up = TO.loss(xb, yb, ref.Ws, ref.bs, ref.beta, m, 0.5)
                    a[idx] = old - 1e-6
                    dn = TO.loss(xb, yb, ref.Ws, ref.bs, ref.beta, m, 0.5)
                    a[idx] = old
                    assert abs((up - dn) / 2e-6 - g[idx]) < 1e-8, (idx, (up - dn) / 2e-6, g[idx])


def test_adam_restatement_first_step_is_sign_times_lr():
    """TF-1 Adam's first step: lr_1 = lr sqrt(1 - b2) / (1 - b1), m = (1 - b1) g, v = (1 - b2) g^2 -> the update is
    lr g / (|g| + eps sqrt(1 - b2)) ... with eps OUTSIDE it is lr_1 m / (sqrt(v) + eps): checked literally."""
    p = np.array([1.0, -2.0, 3.0])
    g = np.array([0.5, -1e-9, 0.0])
    opt = TO.AdamTF1([p], lr=1e-2)
    opt.step([g])
    lr1 = 1e-2 * np.sqrt(1 - 0.999) / (1 - 0.9)
    want = np.array([1.0, -2.0, 3.0]) - lr1 * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
    assert np.allclose(p, want, rtol=0, atol=1e-15)
    # torch.optim.Adam's placement would differ on the tiny gradient: eps / sqrt(1 - b2) = 3.2e-7 vs 1e-8
    torch_like = -2.0 - 1e-2 * (-1e-9) / (1e-9 + 1e-8)
    assert abs(p[1] - torch_like) > 1e-4


def _compare(device, use_masks, graphed=False):
    cfg, w, b, x, y, masks = _problem()
    bs = cfg.batch_size
    ref = TO.Trainer(w, b, cfg.learning_rate, cfg.beta_regularization, keep_prob=0.5)
    tr = training.FNNTrainer(w, b, cfg.learning_rate, cfg.beta_regularization, use_dropout=use_masks, keep_prob=0.5,
                             device=device)
    to_dev = lambda a: torch.as_tensor(a, dtype=torch.float32, device=device)
    # gradients of the first batch, before any update
    m0 = masks[0] if use_masks else None
    cost, grads = tr.gradients(to_dev(x[:bs]), to_dev(y[:bs]), None if m0 is None else [to_dev(m) for m in m0])
    rcost, dW, db = ref.gradients(x[:bs], y[:bs], m0)
    assert abs(float(cost) - rcost) < TOL
    assert len(grads) == 8
    for g, r in zip(grads, dW + db):
        assert np.abs(g.cpu().numpy().astype(np.float64) - r).max() < TOL
    # three steps
    if graphed:
        static_masks = [to_dev(m) for m in masks[0]] if use_masks else None
        step = (lambda xb, yb, *ms: tr.step(xb, yb, list(ms))) if use_masks else (lambda xb, yb: tr.step(xb, yb))
        example = [to_dev(x[:bs]), to_dev(y[:bs])] + (static_masks or [])
        run = training.GraphedStep(step, tr.parameters(), tr.opt, example)
        assert run.captured
    for i in range(3):
        xb, yb = x[i * bs:(i + 1) * bs], y[i * bs:(i + 1) * bs]
        mi = masks[i] if use_masks else None
        rc = ref.step(xb, yb, mi)
        if graphed:
            c = run(to_dev(xb), to_dev(yb), *([to_dev(m) for m in mi] if use_masks else []))
        else:
            c = tr.step(to_dev(xb), to_dev(yb), None if mi is None else [to_dev(m) for m in mi])
        assert abs(float(c) - rc) < TOL, (i, float(c), rc)
    got_w, got_b = tr.export()
    for k, r in zip(ref.w_keys, ref.Ws):
        assert np.abs(got_w[k].astype(np.float64) - r).max() < TOL
        assert np.abs(got_w[k] - w[k]).max() > 1e-4 / 2        # three Adam steps of ~lr each really happened
    for k, r in zip(ref.b_keys, ref.bs):
        assert np.abs(got_b[k].astype(np.float64) - r).max() < TOL


@pytest.mark.parametrize("use_masks", [False, True])
def test_trainer_matches_the_restatement_cpu(use_masks):
    _compare(torch.device("cpu"), use_masks)


@pytest.mark.gpu
@pytest.mark.parametrize("graphed", [False, True])
@pytest.mark.parametrize("use_masks", [False, True])
def test_trainer_matches_the_restatement_gpu(cuda_device, use_masks, graphed):
    _compare(cuda_device, use_masks, graphed)


def test_tfadam_is_not_torch_adam():
    """The two placements of epsilon give different weights once gradients are small against 1e-8 / sqrt(1 - b2^t):
    the trainer follows TF-1's."""
    p0 = torch.tensor([1.0, 1.0], requires_grad=True)
    p1 = torch.tensor([1.0, 1.0], requires_grad=True)
    a, b = training.TFAdam([p0], lr=1e-2), torch.optim.Adam([p1], lr=1e-2)
    for p, o in ((p0, a), (p1, b)):
        p.grad = torch.tensor([1e-9, 0.5])
        o.step()
    assert abs(float((p0[1] - p1[1]).detach())) < 1e-7                 # ordinary gradients: the same step
    assert abs(float((p0[0] - p1[0]).detach())) > 1e-4                 # tiny gradient: eps placement shows


def test_dropout_draws_bernoulli_keep_prob_and_eval_has_none():
    cfg, w, b, x, y, _ = _problem()
    tr = training.FNNTrainer(w, b, 1e-4, 0.001, use_dropout=True, keep_prob=0.5, device="cpu")
    xb = torch.as_tensor(x[:64])
    torch.manual_seed(0)
    a = tr.forward(xb, True)
    torch.manual_seed(1)
    c = tr.forward(xb, True)
    e1, e2 = tr.forward(xb, False), tr.forward(xb, False)
    assert not torch.equal(a, c) and torch.equal(e1, e2)


def test_tfadam_leaves_variables_without_a_gradient_alone():
    a = torch.tensor([1.0], requires_grad=True)
    b = torch.tensor([2.0], requires_grad=True)
    opt = training.TFAdam([a, b], lr=0.1)
    (a * 3.0).sum().backward()
    opt.step()
    assert float(b.detach()) == 2.0 and abs(float(a.detach()) - 0.9) < 1e-6


# ------------------------------------------------------------------ K7: the step as two hand-written kernels
def _native_compare(b, batch, use_masks, n_steps=3):
    cfg, w, bd, x, y, masks = _problem(b=b, batch=batch, seed=10 + b)
    dev = torch.device("cuda", 0)
    assert training.NativeFNNTrainer.eligible(w, bd, batch)
    ref = TO.Trainer(w, bd, cfg.learning_rate, cfg.beta_regularization, keep_prob=0.5)
    nt = training.NativeFNNTrainer(w, bd, cfg.learning_rate, cfg.beta_regularization, batch, use_dropout=use_masks, keep_prob=0.5)
    lab = np.argmax(y, axis=1).astype(np.int32)
    xd = torch.as_tensor(x, dtype=torch.float32, device=dev).contiguous()
    ld = torch.as_tensor(lab, device=dev).contiguous()
    to_dev = lambda ms: [torch.as_tensor(np.concatenate(col), dtype=torch.float32, device=dev).contiguous() for col in zip(*ms)]
    # cost and the eight gradients of the first batch, before any update
    m0 = to_dev(masks[:1]) if use_masks else None
    cost, grads = nt.gradients(xd[:batch], ld[:batch], m0)
    rcost, dW, db = ref.gradients(x[:batch], y[:batch], masks[0] if use_masks else None)
    assert abs(cost - rcost) < 2e-6, (cost, rcost)
    for gname, g, r in zip(["dW1", "dW2", "dW3", "dW4", "db1", "db2", "db3", "db4"], grads, dW + db):
        assert g.shape == r.shape and np.abs(g.astype(np.float64) - r).max() < TOL, gname
    # n_steps consecutive batches in ONE call, per-step costs out
    costs = nt.steps(xd[:n_steps * batch], ld[:n_steps * batch], n_steps, masks=to_dev(masks[:n_steps]) if use_masks else None,
                     want_costs=True)
    rc = [ref.step(x[i * batch:(i + 1) * batch], y[i * batch:(i + 1) * batch], masks[i] if use_masks else None) for i in range(n_steps)]
    assert np.abs(costs.cpu().numpy().astype(np.float64) - np.array(rc)).max() < 2e-6
    got_w, got_b, cost_sum = nt.export(want_cost_sum=True)
    assert abs(cost_sum - sum(rc)) < 1e-5
    for k, r in zip(ref.w_keys, ref.Ws):
        assert np.abs(got_w[k].astype(np.float64) - r).max() < TOL, k
        assert np.abs(got_w[k] - w[k]).max() > 1e-4 / 2
    for k, r in zip(ref.b_keys, ref.bs):
        assert np.abs(got_b[k].astype(np.float64) - r).max() < TOL, k
    nt.close()


@pytest.mark.gpu
@pytest.mark.parametrize("use_masks", [False, True])
@pytest.mark.parametrize("b,batch", [(6, 64), (7, 256), (8, 256), (6, 32)])
def test_native_training_step_matches_the_restatement(cuda_device, b, batch, use_masks):
    """csrc/nnv_train.hip against oracle/train_oracle.py: cost, all eight gradients, three Adam steps (one C call),
    per-step costs and their running sum, with and without fixed dropout masks; class counts 32..256 (1..8 class tiles),
    row lengths 10..16, batches of 1..8 workgroups."""
    _native_compare(b, batch, use_masks)


@pytest.mark.gpu
def test_native_trainer_refuses_what_it_does_not_take(cuda_device):
    cfg = fv.DecoderConfig(block_length=9)
    w, bd = fv.initialize_weights(cfg, rng=np.random.RandomState(0))
    assert not training.NativeFNNTrainer.eligible(w, bd, 256)                 # 512 classes
    w6, b6 = fv.initialize_weights(fv.DecoderConfig(block_length=6), rng=np.random.RandomState(0))
    assert training.NativeFNNTrainer.eligible(w6, b6, 100)                    # a ragged last tile is fine
    with pytest.raises(Exception, match="unsupported"):
        training.NativeFNNTrainer(w, bd, 1e-4, 1e-3, 256)
    nt = training.NativeFNNTrainer(w6, b6, 1e-4, 1e-3, 64, use_dropout=True)
    x = torch.zeros((64, 12), device="cuda")
    with pytest.raises(ValueError, match="masks"):
        nt.steps(x, torch.zeros(64, dtype=torch.int32, device="cuda"), 1)
    nt.close()


@pytest.mark.gpu
def test_train_network_native_and_autograd_agree(cuda_device, monkeypatch, capsys):
    """training.train_network picks the two-kernel step for the reference's architecture; NNV_TRAINER=autograd keeps the
    PyTorch path.  Same recipe, same batches: after 3 epochs the weights agree to 2e-5 and the printed costs to 1e-5."""
    cfg = fv.DecoderConfig(block_length=6, samples_per_class_train=16, samples_per_class_test=1, training_epochs=3, batch_size=64)
    np.random.seed(5)
    tr, trl, _, _, _ = fv.train_test_dataset(cfg)
    w0, b0 = fv.initialize_weights(cfg, rng=np.random.RandomState(9))
    nw, nb = training.train_network(cfg, tr, trl, w0, b0, verbose=True)
    native_out = capsys.readouterr().out
    monkeypatch.setenv("NNV_TRAINER", "autograd")
    aw, ab = training.train_network(cfg, tr, trl, w0, b0, verbose=True)
    auto_out = capsys.readouterr().out
    for k in nw:
        assert np.abs(nw[k] - aw[k]).max() < 2e-5 and np.abs(nw[k] - w0[k]).max() > 1e-3
    for k in nb:
        assert np.abs(nb[k] - ab[k]).max() < 2e-5
    costs = lambda s: [float(l.split("=")[1]) for l in s.splitlines() if l.startswith("Epoch:")]
    assert len(costs(native_out)) == 3 and np.abs(np.array(costs(native_out)) - np.array(costs(auto_out))).max() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("n_in,n_classes,batch", [(11, 50, 96), (32, 256, 32), (7, 2, 64), (20, 200, 160),
                                                  (14, 128, 100), (16, 256, 37), (12, 64, 1)])   # ragged last tiles
def test_native_training_step_on_ragged_shapes(cuda_device, n_in, n_classes, batch):
    """K7 outside the 2^b family: odd row lengths (zero-padded to one 32-wide k-tile), class counts that are not multiples
    of 32 (the padding classes must take no part in the softmax and get no gradient), batches of 1..5 workgroups --
    cost, gradients and three steps against the float64 restatement."""
    rng = np.random.RandomState(n_in * 1000 + n_classes)
    dims = [n_in, 128, 64, 32, n_classes]
    w = {k: (rng.randn(dims[i], dims[i + 1]) * np.sqrt(2.0 / dims[i])).astype(np.float32) for i, k in enumerate(("h1", "h2", "h3", "out"))}
    bd = {k: (rng.randn(dims[i + 1]) * 0.1).astype(np.float32) for i, k in enumerate(("b1", "b2", "b3", "out"))}
    x = rng.randn(3 * batch, n_in).astype(np.float32)
    lab = rng.randint(0, n_classes, 3 * batch).astype(np.int32)
    y = np.eye(n_classes, dtype=np.float32)[lab]
    ref = TO.Trainer(w, bd, 1e-4, 1e-3)
    nt = training.NativeFNNTrainer(w, bd, 1e-4, 1e-3, batch)
    xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(lab).cuda()
    cost, grads = nt.gradients(xd[:batch], ld[:batch])
    rcost, dW, db = ref.gradients(x[:batch], y[:batch])
    assert abs(cost - rcost) < 2e-6
    for g, r in zip(grads, dW + db):
        assert g.shape == r.shape and np.abs(g.astype(np.float64) - r).max() < TOL
    costs = nt.steps(xd, ld, 3, want_costs=True).cpu().numpy()
    rc = [ref.step(x[i * batch:(i + 1) * batch], y[i * batch:(i + 1) * batch]) for i in range(3)]
    assert np.abs(costs - np.array(rc)).max() < 2e-6
    got_w, got_b = nt.export()
    # weights after three Adam steps: Adam divides by sqrt(v) + 1e-8, so where a gradient entry is itself ~1e-8 (random
    # dense data makes some) fp32 rounding of the gradient moves the update by a few percent of lr = 1e-4: 5e-6 here, where
    # the gradients themselves are held to 1e-6 above
    for k, r in zip(ref.w_keys, ref.Ws):
        assert np.abs(got_w[k].astype(np.float64) - r).max() < 5e-6
    for k, r in zip(ref.b_keys, ref.bs):
        assert np.abs(got_b[k].astype(np.float64) - r).max() < 5e-6
    nt.close()
