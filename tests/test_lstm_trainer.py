"""K9 (csrc/nnv_lstm_train.hip, training._LSTMSeqFn): the recurrent decoder's back-propagation through time
(src/rnn_viterbi.py:277-316) against float64 PyTorch-CPU autograd of the unrolled graph (rnn_viterbi._torch_forward,
itself checked against oracle/rnn_oracle.py in tests/test_rnn_host.py).  Parity unpinned against TensorFlow (not
installable here; the reference's module cannot even be imported)."""
import contextlib
import io
import os

import numpy as np
import pytest

import nnviterbi_amd  # noqa: F401


def _params(rng, H=20):
    return {"enc_kernel": (rng.randn(1 + H, 4 * H) * 0.4).astype(np.float32), "enc_bias": (rng.randn(4 * H) * 0.1).astype(np.float32),
            "dec_kernel": (rng.randn(1 + H, 4 * H) * 0.4).astype(np.float32), "dec_bias": (rng.randn(4 * H) * 0.1).astype(np.float32),
            "proj_w": (rng.randn(H, 1) * 0.5).astype(np.float32), "proj_b": (rng.randn(1) * 0.1).astype(np.float32)}


@pytest.mark.gpu
@pytest.mark.parametrize("t_enc,t_dec,batch", [(12, 9, 37), (5, 7, 64), (30, 30, 100)])
def test_lstm_bptt_against_float64_autograd(t_enc, t_dec, batch):
    """Logits, the loss and every parameter gradient of the mean squared error, 1e-5 (fp32 against float64)."""
    import torch
    from nnviterbi_amd import rnn_viterbi as rv
    from nnviterbi_amd.training import lstm_seq2seq_logits_native
    rng = np.random.RandomState(t_enc + batch)
    P = _params(rng)
    enc = rng.randn(t_enc, batch).astype(np.float32)
    dec = rng.randint(0, 2, (t_dec, batch)).astype(np.float32)
    tgt = rng.randn(t_dec, batch, 1).astype(np.float32)
    P64 = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in P.items()}
    ref = rv._torch_forward(P64, torch.tensor(enc, dtype=torch.float64), torch.tensor(dec, dtype=torch.float64), 1.0)
    ref_loss = ((torch.tensor(tgt, dtype=torch.float64) - ref) ** 2).mean()
    ref_loss.backward()
    Pd = {k: torch.tensor(v, device="cuda", requires_grad=True) for k, v in P.items()}
    got = lstm_seq2seq_logits_native(Pd, torch.tensor(enc, device="cuda"), torch.tensor(dec, device="cuda"), 1.0)
    loss = ((torch.tensor(tgt, device="cuda") - got) ** 2).mean()
    loss.backward()
    assert np.abs(got.detach().cpu().numpy() - ref.detach().numpy()).max() < 1e-5
    assert abs(float(loss.detach()) - float(ref_loss.detach())) < 1e-5
    for k in P:
        g, r = Pd[k].grad.cpu().numpy().astype(np.float64), P64[k].grad.numpy()
        assert np.abs(g - r).max() < 1e-5 * max(1.0, np.abs(r).max()), k


@pytest.mark.gpu
def test_lstm_entry_points_refuse_other_hidden_sizes():
    import torch
    from nnviterbi_amd import _cabi
    from nnviterbi_amd.training import _LSTMSeqFn
    zx = torch.zeros((3, 4 * 16, 8), device="cuda")
    with pytest.raises(_cabi.NNViterbiError):
        _LSTMSeqFn.apply(zx, torch.zeros((16, 64), device="cuda"), torch.zeros((16, 8), device="cuda"),
                         torch.zeros((16, 8), device="cuda"), 1.0)


@pytest.mark.gpu
def test_train_seq2seq_native_and_autograd_agree():
    """src/rnn_viterbi.py:399-441 through K9 and through the unrolled autograd graph: the same loss history (1e-5)."""
    from nnviterbi_amd import rnn_viterbi as rv
    cfg = rv.RNNDecoderConfig(block_length=8, training_epochs=2, batch_size=20, training_size=60, test_size=10)
    np.random.seed(1)
    tri, tre, tei, tee = rv.generate_datasets(cfg)
    hist = {}
    for mode in (None, "autograd"):
        old = os.environ.get("NNV_TRAINER")
        if mode:
            os.environ["NNV_TRAINER"] = mode
        else:
            os.environ.pop("NNV_TRAINER", None)
        try:
            model = rv.build_seq2seq_model(cfg, rng=np.random.RandomState(2))
            with contextlib.redirect_stdout(io.StringIO()):
                rv.train_seq2seq(cfg, tri, tre, tei, tee, model=model)
            hist[mode] = (np.array(model["loss_history"]), {k: v.copy() for k, v in model["params"].items()})
        finally:
            if old is None:
                os.environ.pop("NNV_TRAINER", None)
            else:
                os.environ["NNV_TRAINER"] = old
    assert len(hist[None][0]) == 6 and np.abs(hist[None][0] - hist["autograd"][0]).max() < 1e-5
    for k in hist[None][1]:
        assert np.abs(hist[None][1][k] - hist["autograd"][1][k]).max() < 5e-4, k     # Adam at lr 0.01 over 6 steps
