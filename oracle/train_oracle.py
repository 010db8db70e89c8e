"""TEST INFRASTRUCTURE -- numpy float64 restatement of ONE OPTIMISATION STEP of the reference's
feed-forward trainer.  Only tests/ may import this; the product (training.py) never does.

Follows, line by line:
  forward                src/fnn_viterbi.py:494-510 (matmul, +bias, tanh on all four layers);
                         with dropout src/fnn_viterbi_mixed_snr.py:613-627 (tf.nn.dropout after each
                         hidden tanh: x * mask / keep_prob, mask = floor(keep_prob + U[0,1)))
  loss                   src/fnn_viterbi.py:549-560: reduce_mean(softmax_cross_entropy_with_logits(
                         logits=pred, labels=y)) where `pred` is the TANH'd class layer, plus
                         beta * (l2_loss(h1) + l2_loss(h2) + l2_loss(h3)), tf.nn.l2_loss(w) = sum(w^2) / 2;
                         the class layer's weights and all biases are not regularised
                         (same in src/fnn_viterbi_mixed_snr.py:722-732)
  optimiser              src/fnn_viterbi.py:561-563 tf.train.AdamOptimizer(learning_rate).minimize(cost),
                         TF 1.x defaults beta1 = 0.9, beta2 = 0.999, epsilon = 1e-8 and TF 1.x's update
                             t <- t + 1
                             lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
                             m <- m + (g - m) (1 - beta1);  v <- v + (g^2 - v) (1 - beta2)
                             var <- var - lr_t * m / (sqrt(v) + epsilon)
                         ("epsilon hat": epsilon OUTSIDE the bias correction -- not Kingma & Ba's
                         algorithm 1, and not torch.optim.Adam's placement).

Parity status: UNPINNED against TensorFlow itself (not installable offline, the reference holds no
training fixture); the restatement is derived from the text of the call sites above and from TF 1.x's
documented kernels (softmax_cross_entropy_with_logits: backprop = softmax - labels for labels that sum
to one; ApplyAdam as above).  What the tests pin is the build's trainer against THIS restatement, plus
the restatement's analytic gradients against central differences of its own loss (tests/test_trainer.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

W_KEYS = ("h1", "h2", "h3", "out")
B_KEYS = ("b1", "b2", "b3", "out")


def _lists(weights: Dict, biases: Dict):
    wk = [k for k in W_KEYS[:3] if k in weights] + ["out"]
    bk = [k for k in B_KEYS[:3] if k in biases] + ["out"]
    return wk, bk, [np.asarray(weights[k], np.float64) for k in wk], [np.asarray(biases[k], np.float64) for k in bk]


def forward(x, Ws: Sequence[np.ndarray], bs: Sequence[np.ndarray], masks: Optional[Sequence[np.ndarray]] = None,
            keep_prob: float = 1.0):
    """Returns (pred, cache).  ``masks[l]`` (0/1, [B, width_l]) applies tf.nn.dropout after hidden layer l."""
    h = np.asarray(x, np.float64)
    acts, drops = [], []
    n_hidden = len(Ws) - 1
    for l in range(n_hidden):
        a = np.tanh(h @ Ws[l] + bs[l])
        acts.append(a)
        scale = None
        if masks is not None:
            scale = np.asarray(masks[l], np.float64) / keep_prob
            a = a * scale
        drops.append(scale)
        h = a
    pred = np.tanh(h @ Ws[-1] + bs[-1])
    return pred, (np.asarray(x, np.float64), acts, drops, pred)


def loss(x, y, Ws, bs, beta: float, masks=None, keep_prob: float = 1.0) -> float:
    pred, _ = forward(x, Ws, bs, masks, keep_prob)
    y = np.asarray(y, np.float64)
    z = pred - pred.max(axis=1, keepdims=True)
    logsm = z - np.log(np.exp(z).sum(axis=1, keepdims=True))
    ce = -(y * logsm).sum(axis=1).mean()
    reg = sum(0.5 * (W * W).sum() for W in Ws[:-1])
    return float(ce + beta * reg)


def loss_and_gradients(x, y, Ws, bs, beta: float, masks=None, keep_prob: float = 1.0
                       ) -> Tuple[float, List[np.ndarray], List[np.ndarray]]:
    """(cost, dW per layer, db per layer) by hand-written backpropagation in float64.  Labels must be a
    distribution per row (one-hot in the reference): TF's kernel returns softmax - labels."""
    y = np.asarray(y, np.float64)
    pred, (x64, acts, drops, _) = forward(x, Ws, bs, masks, keep_prob)
    B = x64.shape[0]
    z = pred - pred.max(axis=1, keepdims=True)
    e = np.exp(z)
    sm = e / e.sum(axis=1, keepdims=True)
    logsm = z - np.log(e.sum(axis=1, keepdims=True))
    cost = -(y * logsm).sum(axis=1).mean() + beta * sum(0.5 * (W * W).sum() for W in Ws[:-1])
    n_hidden = len(Ws) - 1
    dW = [None] * len(Ws)
    db = [None] * len(Ws)
    d_pred = (sm - y) / B
    dz = d_pred * (1.0 - pred * pred)                       # through the class layer's tanh
    h_in = acts[-1] if drops[-1] is None else acts[-1] * drops[-1]
    dW[-1] = h_in.T @ dz
    db[-1] = dz.sum(axis=0)
    dh = dz @ Ws[-1].T
    for l in range(n_hidden - 1, -1, -1):
        if drops[l] is not None:
            dh = dh * drops[l]
        dz = dh * (1.0 - acts[l] * acts[l])
        if l == 0:
            h_in = x64
        else:
            h_in = acts[l - 1] if drops[l - 1] is None else acts[l - 1] * drops[l - 1]
        dW[l] = h_in.T @ dz + beta * Ws[l]                 # d/dW of beta * sum(W^2) / 2
        db[l] = dz.sum(axis=0)
        dh = dz @ Ws[l].T
    return float(cost), dW, db


class AdamTF1:
    """tf.train.AdamOptimizer of TF 1.x on a list of float64 arrays (updated in place)."""

    def __init__(self, params: Sequence[np.ndarray], lr: float, beta1: float = 0.9, beta2: float = 0.999,
                 epsilon: float = 1e-8):
        self.params = list(params)
        self.lr, self.beta1, self.beta2, self.epsilon = float(lr), beta1, beta2, epsilon
        self.m = [np.zeros_like(p) for p in self.params]
        self.v = [np.zeros_like(p) for p in self.params]
        self.t = 0

    def step(self, grads: Sequence[np.ndarray]) -> None:
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - self.beta2 ** self.t) / (1.0 - self.beta1 ** self.t)
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            m += (g - m) * (1.0 - self.beta1)
            v += (g * g - v) * (1.0 - self.beta2)
            p -= lr_t * m / (np.sqrt(v) + self.epsilon)


class Trainer:
    """The reference's training step as an object: ``step(x, y, masks)`` returns the cost BEFORE the
    update (what ``sess.run([optimizer, cost])`` returns, src/fnn_viterbi.py:589-592)."""

    def __init__(self, weights: Dict, biases: Dict, learning_rate: float, beta_regularization: float,
                 keep_prob: float = 1.0):
        self.w_keys, self.b_keys, self.Ws, self.bs = _lists(weights, biases)
        self.Ws = [W.copy() for W in self.Ws]
        self.bs = [b.copy() for b in self.bs]
        self.beta = float(beta_regularization)
        self.keep_prob = float(keep_prob)
        self.opt = AdamTF1(self.Ws + self.bs, learning_rate)

    def gradients(self, x, y, masks=None):
        return loss_and_gradients(x, y, self.Ws, self.bs, self.beta, masks, self.keep_prob)

    def step(self, x, y, masks=None) -> float:
        cost, dW, db = self.gradients(x, y, masks)
        self.opt.step(dW + db)
        return cost
