"""TEST INFRASTRUCTURE ONLY: float64 restatement of the training step the
reference's bit-wise and sliding-window decoders share.  Only tests/ may import this.

    pred = act(... act(act(x W1 + b1) W2 + b2) ...)          the activation follows EVERY layer, the output layer included
                                                              (src/fnn_viterbi_bitwise.py:388-397 ReLU,
                                                               src/fnn_viterbi_seq_predict.py:404-412 tanh)
    cost = mean over rows AND outputs of sigmoid_cross_entropy_with_logits(logits=pred, labels=y)
           + beta (l2_loss(h1) + l2_loss(h2)),  l2_loss(W) = sum(W^2) / 2
                                                              (src/fnn_viterbi_bitwise.py:438-448,
                                                               src/fnn_viterbi_seq_predict.py:475-485)
    tf.train.AdamOptimizer(lr).minimize(cost)                 (:449-452 / :486-489)  TF-1 rule: oracle/train_oracle.AdamTF1

In the bit-wise decoder `h2` is a variable the network never uses (build_network stops after h1, :388-397) but the
regulariser names it: its gradient is beta * h2 and Adam moves it every step.  `extra` carries such variables.

parity unpinned against TensorFlow itself (not installable here): the backprop below is checked against central
differences of `loss` in tests/test_dense_trainer.py, the HIP kernels and the PyTorch trainers against this file.
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .train_oracle import AdamTF1


def _act(z, kind: str):
    return np.tanh(z) if kind == "tanh" else np.maximum(z, 0.0)


def _dact_from_output(h, kind: str):
    return 1.0 - h * h if kind == "tanh" else (h > 0).astype(np.float64)


def forward(x, Ws: Sequence[np.ndarray], bs: Sequence[np.ndarray], kind: str) -> List[np.ndarray]:
    """Activations of every layer (after the activation), the last one being `pred`."""
    hs, h = [], np.asarray(x, np.float64)
    for W, b in zip(Ws, bs):
        h = _act(h @ W + b, kind)
        hs.append(h)
    return hs


def _sigmoid_ce(z, y):
    return np.maximum(z, 0.0) - z * y + np.log1p(np.exp(-np.abs(z)))


def loss(x, y, Ws, bs, kind: str, beta: float, reg_layers: Sequence[int], extra: Sequence[np.ndarray] = ()) -> float:
    pred = forward(x, Ws, bs, kind)[-1]
    reg = sum(0.5 * float(np.sum(np.asarray(Ws[l], np.float64) ** 2)) for l in reg_layers)
    reg += sum(0.5 * float(np.sum(np.asarray(e, np.float64) ** 2)) for e in extra)
    return float(np.mean(_sigmoid_ce(pred, np.asarray(y, np.float64)))) + beta * reg


def loss_and_gradients(x, y, Ws, bs, kind: str, beta: float, reg_layers: Sequence[int],
                       extra: Sequence[np.ndarray] = ()) -> Tuple[float, List[np.ndarray], List[np.ndarray], List[np.ndarray]]:
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    hs = forward(x, Ws, bs, kind)
    pred = hs[-1]
    cost = loss(x, y, Ws, bs, kind, beta, reg_layers, extra)
    sig = 1.0 / (1.0 + np.exp(-pred))
    dz = (sig - y) / pred.size * _dact_from_output(pred, kind)
    dWs, dbs = [None] * len(Ws), [None] * len(Ws)
    for l in range(len(Ws) - 1, -1, -1):
        h_in = hs[l - 1] if l > 0 else x
        dWs[l] = h_in.T @ dz + (beta * np.asarray(Ws[l], np.float64) if l in reg_layers else 0.0)
        dbs[l] = dz.sum(axis=0)
        if l > 0:
            dz = (dz @ np.asarray(Ws[l], np.float64).T) * _dact_from_output(hs[l - 1], kind)
    return cost, dWs, dbs, [beta * np.asarray(e, np.float64) for e in extra]


class Trainer:
    def __init__(self, Ws, bs, kind: str, learning_rate: float, beta: float, reg_layers: Sequence[int],
                 extra: Optional[Sequence[np.ndarray]] = None):
        self.Ws = [np.array(w, np.float64) for w in Ws]
        self.bs = [np.array(b, np.float64).reshape(-1) for b in bs]
        self.extra = [np.array(e, np.float64) for e in (extra or [])]
        self.kind, self.beta, self.reg_layers = kind, float(beta), tuple(reg_layers)
        self.opt = AdamTF1(self.Ws + self.bs + self.extra, learning_rate)

    def gradients(self, x, y):
        return loss_and_gradients(x, y, self.Ws, self.bs, self.kind, self.beta, self.reg_layers, self.extra)

    def step(self, x, y) -> float:
        cost, dW, db, de = self.gradients(x, y)
        self.opt.step(dW + db + de)
        return cost
