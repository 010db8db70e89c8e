"""Training loop of the feed-forward decoder.

Restates the optimisation the reference builds in TensorFlow
(src/fnn_viterbi.py:547-603; src/fnn_viterbi_mixed_snr.py:722-779):

  * loss = mean softmax cross-entropy taken ON THE TANH OUTPUTS (the network's
    last op is tanh, and that tensor is passed as ``logits``)
    + beta * (l2(h1) + l2(h2) + l2(h3)), with tf.nn.l2_loss(w) = sum(w^2)/2;
    the output layer's weights are not regularised;
  * Adam, learning rate from the config, TF defaults otherwise.  torch.optim.Adam places eps as
    lr_t * m_hat / (sqrt(v_hat) + eps); TF-1 AdamOptimizer uses lr * sqrt(1-b2^t)/(1-b1^t) * m /
    (sqrt(v) + eps), i.e. eps OUTSIDE the bias correction ("epsilon hat"): the two differ by a factor
    sqrt(1-b2^t) on eps (1e-8), far below anything the decoded bits can see, so this is a restatement
    of the recipe, not a bit-identical optimiser (and nothing of the reference's training can be
    pinned anyway: no TensorFlow here);
  * contiguous, unshuffled, class-major mini-batches; the remainder that does
    not fill a batch is dropped (``int(num_training_samples / batch_size)``);
  * optional dropout after each hidden tanh (mixed_snr:596-627), evaluation
    always without dropout (:715-720).

Training is not the hot path of this project: it is plain PyTorch autograd
(on the GPU when there is one).  Decoding with the trained weights goes
through the HIP kernel.
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import numpy as np
import torch


def _device():
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


class GraphedStep:
    """``step(*inputs) -> scalar cost`` (forward, backward, optimizer update) captured once into a
    HIP graph and replayed per batch.  The reference's networks are tiny, so an eager step is bound
    by its kernel launches (the feed-forward trainer: ~25 per step; the recurrent one: thousands).
    Same kernels in the same order as eager mode; on the CPU, or if capture fails, calls run eagerly.

    ``params``: the leaf tensors the optimizer updates; ``opt`` must have been built with
    ``capturable=True`` on CUDA; ``example_inputs``: tensors of the batch shapes (copied)."""

    def __init__(self, step, params, opt, example_inputs, verbose: bool = False):
        self._step = step
        self._graph = None
        self._static = None
        self._cost = None
        dev = params[0].device
        if dev.type != "cuda" or os.environ.get("NNV_NO_TRAIN_GRAPH"):   # the switch is for A/B tests
            return
        saved = [t.detach().clone() for t in params]

        def restore():                                   # undo warm-up steps: weights and optimizer state
            with torch.no_grad():
                for t, v in zip(params, saved):
                    t.copy_(v)
                for st in opt.state.values():
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()

        try:
            static = [t.clone() for t in example_inputs]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):                       # warm-up steps outside the graph
                    step(*static)
            torch.cuda.current_stream().wait_stream(side)
            restore()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):                    # records, does not execute
                cost = step(*static)
            self._graph, self._static, self._cost = g, static, cost
        except Exception as e:  # pragma: no cover - depends on the runtime
            if verbose:
                print(f"(HIP graph capture of the training step unavailable: {e}; eager mode)")
            self._graph = None
            try:                                         # the eager fallback must start from the initial
                torch.cuda.synchronize()                 # state too, not 0-3 Adam steps into the run
                restore()
            except Exception:
                pass

    @property
    def captured(self) -> bool:
        return self._graph is not None

    def __call__(self, *inputs):
        if self._graph is None:
            return self._step(*inputs)
        for dst, src in zip(self._static, inputs):
            dst.copy_(src)
        self._graph.replay()
        return self._cost


def train_network(config, training_set: np.ndarray, training_labels: np.ndarray,
                  weights: Dict[str, np.ndarray], biases: Dict[str, np.ndarray],
                  use_dropout: bool = False, keep_prob: float = 0.5,
                  verbose: bool = True) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
    """Returns trained copies of (weights, biases), keyed like the inputs."""
    dev = _device()
    w_keys = [k for k in ("h1", "h2", "h3") if k in weights] + ["out"]
    b_keys = [k for k in ("b1", "b2", "b3") if k in biases] + ["out"]
    W = [torch.tensor(np.asarray(weights[k]), dtype=torch.float32, device=dev, requires_grad=True)
         for k in w_keys]
    B = [torch.tensor(np.asarray(biases[k]), dtype=torch.float32, device=dev, requires_grad=True)
         for k in b_keys]
    x_all = torch.as_tensor(np.asarray(training_set), dtype=torch.float32, device=dev)
    y_all = torch.as_tensor(np.asarray(training_labels), dtype=torch.float32, device=dev)
    opt = torch.optim.Adam(W + B, lr=config.learning_rate, betas=(0.9, 0.999), eps=1e-8,
                           capturable=(dev.type == "cuda"))
    bs = int(config.batch_size)
    total_batch = int(config.num_training_samples / bs)
    beta = float(config.beta_regularization)
    n_hidden = len(W) - 1

    def model(xb, train: bool):
        h = xb
        for i in range(n_hidden):
            h = torch.tanh(h @ W[i] + B[i])
            if train and use_dropout:
                h = torch.nn.functional.dropout(h, p=1.0 - keep_prob, training=True)
        return torch.tanh(h @ W[-1] + B[-1])

    def step(xb, yb):
        pred = model(xb, True)
        ce = -(yb * torch.log_softmax(pred, dim=1)).sum(dim=1).mean()
        reg = sum(0.5 * (w * w).sum() for w in W[:n_hidden])
        cost = ce + beta * reg
        opt.zero_grad(set_to_none=False)
        cost.backward()
        opt.step()
        return cost.detach()

    run = GraphedStep(step, W + B, opt, [x_all[:bs], y_all[:bs]], verbose) if total_batch > 0 else step

    for epoch in range(int(config.training_epochs)):
        avg_cost = torch.zeros((), device=dev)
        for i in range(total_batch):
            avg_cost += run(x_all[i * bs:(i + 1) * bs], y_all[i * bs:(i + 1) * bs]) / max(total_batch, 1)
        if verbose and epoch % int(getattr(config, "display_step", 1)) == 0:
            print(f"Epoch: {epoch + 1:04d} | cost = {float(avg_cost):.9f}")

    new_w = {k: w.detach().cpu().numpy().copy() for k, w in zip(w_keys, W)}
    new_b = {k: b.detach().cpu().numpy().copy() for k, b in zip(b_keys, B)}
    return new_w, new_b
