"""Training loop of the feed-forward decoder.

Restates the optimisation the reference builds in TensorFlow
(src/fnn_viterbi.py:547-603; src/fnn_viterbi_mixed_snr.py:722-779):

  * loss = mean softmax cross-entropy taken ON THE TANH OUTPUTS (the network's
    last op is tanh, and that tensor is passed as ``logits``)
    + beta * (l2(h1) + l2(h2) + l2(h3)), with tf.nn.l2_loss(w) = sum(w^2)/2;
    the output layer's weights are not regularised;
  * tf.train.AdamOptimizer, learning rate from the config, TF-1 defaults otherwise, with TF-1's
    update rule (:class:`TFAdam`): lr_t = lr sqrt(1 - b2^t) / (1 - b1^t), var -= lr_t m / (sqrt(v) + eps),
    i.e. eps OUTSIDE the bias correction -- not torch.optim.Adam's placement;
  * contiguous, unshuffled, class-major mini-batches; the remainder that does
    not fill a batch is dropped (``int(num_training_samples / batch_size)``);
  * optional dropout after each hidden tanh (mixed_snr:596-627: tf.nn.dropout = x * mask / keep_prob),
    evaluation always without dropout (:715-720).

One step (loss, its eight gradients, the Adam update) is checked against a float64 numpy restatement of the
same lines (oracle/train_oracle.py) in tests/test_trainer.py, on the CPU and on the GPU, eager and as the
captured HIP graph.  TensorFlow itself is not installable here: the restatement is pinned to the reference's
text, not to its outputs.

Training is not the hot path of this project: it is plain PyTorch autograd
(on the GPU when there is one).  Decoding with the trained weights goes
through the HIP kernel.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch


def _device():
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


class GraphedStep:
    """``step(*inputs) -> scalar cost`` (forward, backward, optimizer update) captured once into a
    HIP graph and replayed per batch.  The reference's networks are tiny, so an eager step is bound
    by its kernel launches (the feed-forward trainer: ~25 per step; the recurrent one: thousands).
    Same kernels in the same order as eager mode; on the CPU, or if capture fails, calls run eagerly.

    ``params``: the leaf tensors the optimizer updates; ``opt``: a :class:`TFAdam` (or a torch optimizer built
    with ``capturable=True``); ``example_inputs``: tensors of the batch shapes (copied)."""

    def __init__(self, step, params, opt, example_inputs, verbose: bool = False):
        self._step = step
        self._graph = None
        self._static = None
        self._cost = None
        dev = params[0].device
        if dev.type != "cuda" or os.environ.get("NNV_NO_TRAIN_GRAPH"):   # the switch is for A/B tests
            return
        saved = [t.detach().clone() for t in params]
        opt_state = list(opt.state_tensors()) if hasattr(opt, "state_tensors") else []
        saved_state = [t.detach().clone() for t in opt_state]

        def restore():                                   # undo warm-up steps: weights and optimizer state
            with torch.no_grad():
                for t, v in zip(params, saved):
                    t.copy_(v)
                for t, v in zip(opt_state, saved_state):  # TFAdam: moments AND the running beta powers
                    t.copy_(v)
                if not opt_state:                         # a torch optimizer: its (lazily created) state starts at zero
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


class TFAdam:
    """``tf.train.AdamOptimizer`` of TensorFlow 1.x (the optimiser every script of the reference uses:
    src/fnn_viterbi.py:561-563, mixed_snr:734-738, bitwise / seq_predict / rnn alike) on a list of leaf tensors:

        t <- t + 1;  lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
        m <- m + (g - m) (1 - beta1);   v <- v + (g^2 - v) (1 - beta2)
        var <- var - lr_t * m / (sqrt(v) + epsilon)

    epsilon sits OUTSIDE the bias correction ("epsilon hat"), unlike torch.optim.Adam.  The powers beta^t are
    running float32 products held on the device (as TF keeps them in ``beta1_power`` / ``beta2_power``), so a step is
    pure tensor arithmetic: capturable into a HIP graph, no host value changes between replays."""

    def __init__(self, params: Sequence[torch.Tensor], lr: float, beta1: float = 0.9, beta2: float = 0.999,
                 eps: float = 1e-8):
        self.params = list(params)
        self.lr, self.beta1, self.beta2, self.eps = float(lr), float(beta1), float(beta2), float(eps)
        dev = self.params[0].device
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.beta1_power = torch.ones((), dtype=torch.float32, device=dev)
        self.beta2_power = torch.ones((), dtype=torch.float32, device=dev)

    def state_tensors(self) -> List[torch.Tensor]:
        return self.m + self.v + [self.beta1_power, self.beta2_power]

    def zero_grad(self, set_to_none: bool = False) -> None:
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    @torch.no_grad()
    def step(self) -> None:
        # variables the loss does not reach have no gradient and are left alone (as minimize() leaves them)
        live = [i for i, p in enumerate(self.params) if p.grad is not None]
        params = [self.params[i] for i in live]
        grads = [self.params[i].grad for i in live]
        m, v = [self.m[i] for i in live], [self.v[i] for i in live]
        self.beta1_power.mul_(self.beta1)
        self.beta2_power.mul_(self.beta2)
        lr_t = self.lr * torch.sqrt(1.0 - self.beta2_power) / (1.0 - self.beta1_power)
        torch._foreach_lerp_(m, grads, 1.0 - self.beta1)                      # m += (g - m)(1 - beta1)
        torch._foreach_lerp_(v, torch._foreach_mul(grads, grads), 1.0 - self.beta2)
        denom = torch._foreach_sqrt(v)
        torch._foreach_add_(denom, self.eps)
        upd = torch._foreach_div(m, denom)
        torch._foreach_mul_(upd, lr_t)
        torch._foreach_sub_(params, upd)


class FNNTrainer:
    """The reference's training step for the feed-forward decoder as an object (the lines cited in the module
    docstring).  ``weights`` / ``biases``: dictionaries keyed like the reference's (``h1..h3, out`` / ``b1..b3, out``).

    ``step(xb, yb)`` = what ``sess.run([optimizer, cost], feed_dict)`` does (src/fnn_viterbi.py:589-592): returns the
    cost of the batch BEFORE the update and applies the update.  ``masks`` (one 0/1 tensor [B, width] per hidden layer)
    fixes the dropout pattern for tests; without them a fresh Bernoulli(keep_prob) mask is drawn per call."""

    def __init__(self, weights: Dict[str, np.ndarray], biases: Dict[str, np.ndarray], learning_rate: float,
                 beta_regularization: float, use_dropout: bool = False, keep_prob: float = 0.5, device=None):
        self.device = _device() if device is None else torch.device(device)
        self.w_keys = [k for k in ("h1", "h2", "h3") if k in weights] + ["out"]
        self.b_keys = [k for k in ("b1", "b2", "b3") if k in biases] + ["out"]
        mk = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=self.device, requires_grad=True)
        self.W = [mk(weights[k]) for k in self.w_keys]
        self.B = [mk(biases[k]) for k in self.b_keys]
        self.beta = float(beta_regularization)
        self.use_dropout, self.keep_prob = bool(use_dropout), float(keep_prob)
        self.n_hidden = len(self.W) - 1
        self.opt = TFAdam(self.W + self.B, lr=learning_rate)

    def parameters(self) -> List[torch.Tensor]:
        return self.W + self.B

    def forward(self, xb: torch.Tensor, train: bool, masks: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
        h = xb
        for i in range(self.n_hidden):
            h = torch.tanh(h @ self.W[i] + self.B[i])
            if masks is not None:
                h = h * (masks[i] / self.keep_prob)
            elif train and self.use_dropout:                     # tf.nn.dropout: x * floor(keep_prob + U) / keep_prob
                h = h * ((torch.rand_like(h) < self.keep_prob).to(h.dtype) / self.keep_prob)
        return torch.tanh(h @ self.W[-1] + self.B[-1])

    def cost(self, xb: torch.Tensor, yb: torch.Tensor, masks: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
        pred = self.forward(xb, True, masks)
        ce = -(yb * torch.log_softmax(pred, dim=1)).sum(dim=1).mean()
        reg = sum(0.5 * (w * w).sum() for w in self.W[:self.n_hidden])
        return ce + self.beta * reg

    def gradients(self, xb, yb, masks=None) -> Tuple[torch.Tensor, List[torch.Tensor]]:
        """(cost, [dW..., db...]) without touching the weights (parity tests)."""
        self.opt.zero_grad(set_to_none=False)
        c = self.cost(xb, yb, masks)
        c.backward()
        return c.detach(), [p.grad.detach().clone() for p in self.parameters()]

    def step(self, xb: torch.Tensor, yb: torch.Tensor, masks: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
        c = self.cost(xb, yb, masks)
        self.opt.zero_grad(set_to_none=False)
        c.backward()
        self.opt.step()
        return c.detach()

    def export(self) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
        return ({k: w.detach().cpu().numpy().copy() for k, w in zip(self.w_keys, self.W)},
                {k: b.detach().cpu().numpy().copy() for k, b in zip(self.b_keys, self.B)})


class NativeFNNTrainer:
    """The same training step as :class:`FNNTrainer`, as two hand-written HIP kernels per step (C ABI ``nnv_train_*``,
    csrc/nnv_train.hip): forward + loss + backward in one launch (four waves per 32 rows, every layer's
    tiles dealt over them and handed over through LDS), TF-1 Adam + re-packing of the weights in a second.  ~45 launches of a PyTorch-autograd step become 2; the steps of an epoch are issued by ONE call.

    Eligible: the reference's architecture n_in -> 128 -> 64 -> 32 -> n_classes with n_in <= 32, n_classes <= 256, any
    batch size up to 32,768 (a ragged last 32-row tile is masked), on an MI355X.  Labels are class indices (the position of the 1 in the
    reference's one-hot rows).  Checked against oracle/train_oracle.py like FNNTrainer (tests/test_trainer.py)."""

    @staticmethod
    def eligible(weights: Dict, biases: Dict, batch_size: int) -> bool:
        if not torch.cuda.is_available():
            return False
        if [k for k in ("h1", "h2", "h3") if k in weights] != ["h1", "h2", "h3"]:
            return False
        shp = [tuple(np.asarray(weights[k]).shape) for k in ("h1", "h2", "h3", "out")]
        return (shp[0][1], shp[1], shp[2], shp[3][0]) == (128, (128, 64), (64, 32), 32) and shp[0][0] <= 32 \
            and shp[3][1] <= 256 and 0 < batch_size <= 32 * 1024

    def __init__(self, weights: Dict, biases: Dict, learning_rate: float, beta_regularization: float, batch_size: int,
                 use_dropout: bool = False, keep_prob: float = 0.5, device=None):
        import ctypes
        from . import _cabi
        self._cabi, self._ctypes = _cabi, ctypes
        self._lib = _cabi.load_library()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else torch.device(device).index or 0)
        self.w_keys, self.b_keys = ["h1", "h2", "h3", "out"], ["b1", "b2", "b3", "out"]
        Ws = [np.ascontiguousarray(np.asarray(weights[k]), dtype=np.float32) for k in self.w_keys]
        bs = [np.ascontiguousarray(np.asarray(biases[k]), dtype=np.float32).reshape(-1) for k in self.b_keys]
        self.shapes = [w.shape for w in Ws]
        self.batch_size = int(batch_size)
        self.use_dropout, self.keep_prob = bool(use_dropout), float(keep_prob)
        d = _cabi.TrainDesc()
        d.n_input, d.n_classes, d.batch_size = Ws[0].shape[0], Ws[3].shape[1], self.batch_size
        d.n_hidden[0], d.n_hidden[1], d.n_hidden[2] = 128, 64, 32
        d.learning_rate, d.beta_regularization = float(learning_rate), float(beta_regularization)
        d.keep_prob = self.keep_prob if self.use_dropout else 1.0
        for i in range(4):
            d.W_host[i], d.b_host[i] = Ws[i].ctypes.data, bs[i].ctypes.data
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _cabi.check(self._lib.nnv_train_create(ctypes.byref(d), ctypes.byref(h)), "nnv_train_create")
        self._handle = h

    def close(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            self._lib.nnv_train_free(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _mask_array(self, masks):
        if masks is None:
            return None
        arr = (self._ctypes.c_void_p * 3)(*[int(m.data_ptr()) for m in masks])
        return arr

    def _check(self, x, labels, n_steps, masks):
        n = n_steps * self.batch_size
        if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape) == (n, self.shapes[0][0])):
            raise ValueError(f"x must be a contiguous float32 CUDA tensor [{n}, {self.shapes[0][0]}]")
        if not (labels.is_cuda and labels.dtype == torch.int32 and labels.is_contiguous() and tuple(labels.shape) == (n,)):
            raise ValueError(f"labels must be a contiguous int32 CUDA tensor [{n}] of class indices")
        if self.use_dropout and masks is None:
            raise ValueError("this trainer was built with dropout: pass the three masks")
        if masks is not None:
            for m, wdt in zip(masks, (128, 64, 32)):
                if not (m.is_cuda and m.dtype == torch.float32 and m.is_contiguous() and tuple(m.shape) == (n, wdt)):
                    raise ValueError(f"masks must be contiguous float32 CUDA tensors [{n}, 128 | 64 | 32]")

    def draw_masks(self, n_rows: int):
        """Bernoulli(keep_prob) masks for ``n_rows`` rows, as tf.nn.dropout draws floor(keep_prob + U[0, 1))."""
        return [(torch.rand((n_rows, w), device=self.device) < self.keep_prob).to(torch.float32) for w in (128, 64, 32)]

    def steps(self, x: torch.Tensor, labels: torch.Tensor, n_steps: int, masks=None, want_costs: bool = False):
        """``n_steps`` consecutive batches (rows [t B, (t+1) B) of ``x`` / ``labels``), stream-ordered, one C call."""
        self._check(x, labels, n_steps, masks)
        costs = torch.empty(n_steps, dtype=torch.float32, device=self.device) if want_costs else None
        marr = self._mask_array(masks)
        with torch.cuda.device(self.device):
            st = self._lib.nnv_train_steps(self._handle, x.data_ptr(), labels.data_ptr(), int(n_steps),
                                           None if marr is None else self._ctypes.cast(marr, self._ctypes.c_void_p),
                                           None if costs is None else costs.data_ptr(),
                                           self._ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        self._cabi.check(st, "nnv_train_steps")
        return costs

    def gradients(self, xb: torch.Tensor, labels: torch.Tensor, masks=None):
        """(cost, [dW1..dW4, db1..db4]) of one batch at the current weights; no update (parity tests)."""
        self._check(xb, labels, 1, masks)
        gW = [np.zeros(s, np.float32) for s in self.shapes]
        gb = [np.zeros(s[1], np.float32) for s in self.shapes]
        pw = (self._ctypes.c_void_p * 4)(*[g.ctypes.data for g in gW])
        pb = (self._ctypes.c_void_p * 4)(*[g.ctypes.data for g in gb])
        cost = self._ctypes.c_float(0.0)
        marr = self._mask_array(masks)
        with torch.cuda.device(self.device):
            torch.cuda.current_stream(self.device).synchronize()
            st = self._lib.nnv_train_gradients(self._handle, xb.data_ptr(), labels.data_ptr(),
                                               None if marr is None else self._ctypes.cast(marr, self._ctypes.c_void_p),
                                               self._ctypes.byref(cost), self._ctypes.cast(pw, self._ctypes.c_void_p),
                                               self._ctypes.cast(pb, self._ctypes.c_void_p))
        self._cabi.check(st, "nnv_train_gradients")
        return float(cost.value), gW + gb

    def export(self, want_cost_sum: bool = False):
        Ws = [np.zeros(s, np.float32) for s in self.shapes]
        bs = [np.zeros(s[1], np.float32) for s in self.shapes]
        pw = (self._ctypes.c_void_p * 4)(*[g.ctypes.data for g in Ws])
        pb = (self._ctypes.c_void_p * 4)(*[g.ctypes.data for g in bs])
        csum = self._ctypes.c_double(0.0)
        with torch.cuda.device(self.device):
            st = self._lib.nnv_train_export(self._handle, self._ctypes.cast(pw, self._ctypes.c_void_p),
                                            self._ctypes.cast(pb, self._ctypes.c_void_p),
                                            self._ctypes.byref(csum) if want_cost_sum else None)
        self._cabi.check(st, "nnv_train_export")
        out = (dict(zip(self.w_keys, Ws)), dict(zip(self.b_keys, bs)))
        return out + (float(csum.value),) if want_cost_sum else out


class NativeDenseTrainer:
    """The training step of the reference's bit-wise and sliding-window decoders as 2 L + 1 hand-written HIP launches
    per step (C ABI ``nnv_dense_train_*``, csrc/nnv_dense_train.hip) instead of ~60 PyTorch-autograd kernels: a dense
    network of 1..5 layers with the activation after EVERY layer, mean sigmoid cross-entropy on the activated outputs
    (``loss="softmax"``: the feed-forward decoder's softmax cross-entropy on them, src/fnn_viterbi.py:549-553, for class
    counts K7 does not take), beta * l2_loss of the layers in ``reg_layers``, TF-1 Adam (src/fnn_viterbi_bitwise.py:388-397,438-452;
    src/fnn_viterbi_seq_predict.py:404-412,475-489).  ``extra`` holds variables that only the regulariser names (the
    bit-wise decoder's unused ``h2``).  Labels are class indices.  Rows of a step may be overlapping windows of one
    stream (``row_stride``).  Checked against oracle/dense_train_oracle.py (tests/test_dense_trainer.py)."""

    ACT = {"tanh": 0, "relu": 1}
    LOSS = {"sigmoid": 0, "softmax": 1}     # nnv_loss

    @staticmethod
    def eligible(Ws, n_out: int, loss: str = "sigmoid") -> bool:
        return (torch.cuda.is_available() and 1 <= len(Ws) <= 5 and n_out <= (32 if loss == "sigmoid" else 65536)
                and os.environ.get("NNV_TRAINER") != "autograd")

    def __init__(self, Ws, bs, activation: str, learning_rate: float, beta_regularization: float, batch_size: int,
                 reg_layers=(0, 1), extra=None, device=None, loss: str = "sigmoid"):
        import ctypes
        from . import _cabi
        self._cabi, self._ctypes = _cabi, ctypes
        self._lib = _cabi.load_library()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else torch.device(device).index or 0)
        Ws = [np.ascontiguousarray(np.asarray(w), dtype=np.float32) for w in Ws]
        bs = [np.ascontiguousarray(np.asarray(b), dtype=np.float32).reshape(-1) for b in bs]
        self.shapes = [w.shape for w in Ws]
        self.batch_size, self.n_input, self.n_out = int(batch_size), Ws[0].shape[0], Ws[-1].shape[1]
        self.extra_shapes = [np.asarray(e).shape for e in (extra or [])]
        ex = np.concatenate([np.asarray(e, np.float32).reshape(-1) for e in extra]) if extra else np.zeros(0, np.float32)
        self.n_extra = int(ex.size)
        d = _cabi.DenseTrainDesc()
        d.n_layers = len(Ws)
        for i, w in enumerate(Ws):
            d.dims[i], d.dims[i + 1] = w.shape
            d.W_host[i], d.b_host[i] = w.ctypes.data, bs[i].ctypes.data
        d.activation, d.batch_size = self.ACT[activation], self.batch_size
        d.learning_rate, d.beta_regularization = float(learning_rate), float(beta_regularization)
        d.reg_layer_mask = sum(1 << l for l in reg_layers if l < len(Ws))
        d.loss = self.LOSS[loss]
        d.extra_host, d.n_extra = (ex.ctypes.data if self.n_extra else None), self.n_extra
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _cabi.check(self._lib.nnv_dense_train_create(ctypes.byref(d), ctypes.byref(h)), "nnv_dense_train_create")
        self._handle = h

    def close(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            self._lib.nnv_dense_train_free(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, x, labels, n_steps, row_stride, step_stride):
        need = (n_steps - 1) * step_stride + (self.batch_size - 1) * row_stride + self.n_input
        if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.numel() >= need):
            raise ValueError(f"x must be a contiguous float32 CUDA tensor of at least {need} elements")
        if not (labels.is_cuda and labels.dtype == torch.int32 and labels.is_contiguous()
                and labels.numel() >= n_steps * self.batch_size):
            raise ValueError(f"labels must be a contiguous int32 CUDA tensor of at least {n_steps * self.batch_size} class indices")

    def steps(self, x: torch.Tensor, labels: torch.Tensor, n_steps: int, row_stride: int = None, step_stride: int = None,
              want_costs: bool = False):
        """``n_steps`` consecutive batches, stream-ordered, one C call.  Default strides: ``x`` is [rows, n_input]."""
        row_stride = self.n_input if row_stride is None else int(row_stride)
        step_stride = self.batch_size * row_stride if step_stride is None else int(step_stride)
        self._check(x, labels, n_steps, row_stride, step_stride)
        costs = torch.empty(n_steps, dtype=torch.float32, device=self.device) if want_costs else None
        with torch.cuda.device(self.device):
            st = self._lib.nnv_dense_train_steps(self._handle, x.data_ptr(), row_stride, step_stride, labels.data_ptr(), int(n_steps),
                                                 None if costs is None else costs.data_ptr(),
                                                 self._ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        self._cabi.check(st, "nnv_dense_train_steps")
        return costs

    def _out_arrays(self):
        Ws = [np.zeros(s, np.float32) for s in self.shapes]
        bs = [np.zeros(s[1], np.float32) for s in self.shapes]
        pw = (self._ctypes.c_void_p * 5)(*[g.ctypes.data for g in Ws])
        pb = (self._ctypes.c_void_p * 5)(*[g.ctypes.data for g in bs])
        ex = np.zeros(self.n_extra, np.float32)
        return Ws, bs, ex, pw, pb

    def _split_extra(self, ex):
        out, at = [], 0
        for s in self.extra_shapes:
            n = int(np.prod(s))
            out.append(ex[at:at + n].reshape(s).copy())
            at += n
        return out

    def gradients(self, xb: torch.Tensor, labels: torch.Tensor, row_stride: int = None):
        """(cost, dWs, dbs, dextras) of one batch at the current values; no update (parity tests)."""
        row_stride = self.n_input if row_stride is None else int(row_stride)
        self._check(xb, labels, 1, row_stride, 0)
        gW, gb, ex, pw, pb = self._out_arrays()
        cost = self._ctypes.c_float(0.0)
        with torch.cuda.device(self.device):
            torch.cuda.current_stream(self.device).synchronize()
            st = self._lib.nnv_dense_train_gradients(self._handle, xb.data_ptr(), row_stride, labels.data_ptr(),
                                                     self._ctypes.byref(cost), self._ctypes.cast(pw, self._ctypes.c_void_p),
                                                     self._ctypes.cast(pb, self._ctypes.c_void_p),
                                                     ex.ctypes.data if self.n_extra else None)
        self._cabi.check(st, "nnv_dense_train_gradients")
        return float(cost.value), gW, gb, self._split_extra(ex)

    def export(self, want_cost_sum: bool = False):
        """(Ws, bs, extras[, sum of the step costs since the last export])."""
        Ws, bs, ex, pw, pb = self._out_arrays()
        csum = self._ctypes.c_double(0.0)
        with torch.cuda.device(self.device):
            st = self._lib.nnv_dense_train_export(self._handle, self._ctypes.cast(pw, self._ctypes.c_void_p),
                                                  self._ctypes.cast(pb, self._ctypes.c_void_p),
                                                  ex.ctypes.data if self.n_extra else None,
                                                  self._ctypes.byref(csum) if want_cost_sum else None)
        self._cabi.check(st, "nnv_dense_train_export")
        out = (Ws, bs, self._split_extra(ex))
        return out + (float(csum.value),) if want_cost_sum else out


class _LSTMSeqFn(torch.autograd.Function):
    """One tf.contrib.rnn.LSTMCell unrolled over time (src/rnn_viterbi.py:277-299) as ONE HIP launch forward and one
    backward (K9, csrc/nnv_lstm_train.hip); layouts [T, feature, B].  Inputs: zx [T, 4H, B] = x_t kernel[0] + bias,
    wh [H, 4H] = kernel[1:], c0, m0 [H, B].  Outputs: m_all [T, H, B], cT, mT [H, B]."""

    @staticmethod
    def forward(ctx, zx, wh, c0, m0, forget_bias):
        import ctypes
        from . import _cabi
        lib = _cabi.load_library()
        zx, wh, c0, m0 = zx.contiguous(), wh.contiguous(), c0.contiguous(), m0.contiguous()
        T, H4, B = zx.shape
        H = H4 // 4
        gates = torch.empty_like(zx)
        c_all = torch.empty((T, H, B), dtype=torch.float32, device=zx.device)
        m_all = torch.empty_like(c_all)
        st = lib.nnv_lstm_seq_forward_f32(zx.data_ptr(), wh.data_ptr(), c0.data_ptr(), m0.data_ptr(), T, B, H, float(forget_bias),
                                          gates.data_ptr(), c_all.data_ptr(), m_all.data_ptr(),
                                          ctypes.c_void_p(torch.cuda.current_stream(zx.device).cuda_stream))
        _cabi.check(st, "nnv_lstm_seq_forward_f32")
        ctx.save_for_backward(wh, c0, m0, gates, c_all, m_all)
        return m_all, c_all[T - 1].clone(), m_all[T - 1].clone()

    @staticmethod
    def backward(ctx, dm_all, dcT, dmT):
        import ctypes
        from . import _cabi
        lib = _cabi.load_library()
        wh, c0, m0, gates, c_all, m_all = ctx.saved_tensors
        T, H, B = c_all.shape
        dm_all = torch.zeros_like(c_all) if dm_all is None else dm_all.contiguous()
        dcT = torch.zeros_like(c0) if dcT is None else dcT.contiguous()
        dmT = torch.zeros_like(c0) if dmT is None else dmT.contiguous()
        dz = torch.empty_like(gates)
        dc0, dm0 = torch.empty_like(c0), torch.empty_like(c0)
        st = lib.nnv_lstm_seq_backward_f32(dm_all.data_ptr(), dcT.data_ptr(), dmT.data_ptr(), gates.data_ptr(), c_all.data_ptr(),
                                           c0.data_ptr(), wh.data_ptr(), T, B, H, dz.data_ptr(), dc0.data_ptr(), dm0.data_ptr(),
                                           ctypes.c_void_p(torch.cuda.current_stream(dz.device).cuda_stream))
        _cabi.check(st, "nnv_lstm_seq_backward_f32")
        # dW_h = sum over t of m_{t-1}^T dz_t: ONE matrix product over all (t, row) pairs
        m_prev = torch.cat([m0[None], m_all[:-1]], dim=0)                       # [T, H, B]
        dwh = torch.einsum("thb,tgb->hg", m_prev, dz)
        return dz, dwh, dc0, dm0, None


def lstm_seq2seq_logits_native(P, enc, dec, forget_bias: float):
    """The recurrent decoder's graph (src/rnn_viterbi.py:277-305) with both recurrences in K9: time-major [T, B] inputs,
    logits [T_dec, B, 1] -- what rnn_viterbi._torch_forward computes cell by cell."""
    H = P["enc_bias"].shape[0] // 4
    B = enc.shape[1]
    zeros = enc.new_zeros((H, B))

    def zx_of(x, kernel, bias):                      # [T, B] -> [T, 4H, B]
        return x[:, None, :] * kernel[0][None, :, None] + bias[None, :, None]

    _, cT, mT = _LSTMSeqFn.apply(zx_of(enc, P["enc_kernel"], P["enc_bias"]), P["enc_kernel"][1:], zeros, zeros, forget_bias)
    m_dec, _, _ = _LSTMSeqFn.apply(zx_of(dec, P["dec_kernel"], P["dec_bias"]), P["dec_kernel"][1:], cT, mT, forget_bias)
    logits = torch.einsum("thb,h->tb", m_dec, P["proj_w"].reshape(H)) + P["proj_b"]
    return logits[:, :, None]


def lstm_native_eligible(P) -> bool:
    return (torch.cuda.is_available() and os.environ.get("NNV_TRAINER") != "autograd"
            and P["enc_bias"].shape[0] == 80 and P["dec_bias"].shape[0] == 80)


def train_network(config, training_set: np.ndarray, training_labels: np.ndarray,
                  weights: Dict[str, np.ndarray], biases: Dict[str, np.ndarray],
                  use_dropout: bool = False, keep_prob: float = 0.5,
                  verbose: bool = True) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
    """Returns trained copies of (weights, biases), keyed like the inputs.  On an MI355X the reference's architecture
    trains through the two-kernel step (:class:`NativeFNNTrainer`, <= 256 classes), other class counts and depths
    without dropout through the dense kernels with the softmax loss (:class:`NativeDenseTrainer`); the rest, and ``NNV_TRAINER=autograd``, through
    PyTorch autograd (:class:`FNNTrainer`), one HIP-graph replay per step."""
    bs = int(config.batch_size)
    total_batch = int(config.num_training_samples / bs)
    if os.environ.get("NNV_TRAINER") != "autograd" and NativeFNNTrainer.eligible(weights, biases, bs) and total_batch > 0:
        nt = NativeFNNTrainer(weights, biases, config.learning_rate, config.beta_regularization, bs,
                              use_dropout=use_dropout, keep_prob=keep_prob)
        dev = nt.device
        n = total_batch * bs                         # the remainder that does not fill a batch is dropped, as in the reference
        x_all = torch.as_tensor(np.asarray(training_set)[:n], dtype=torch.float32, device=dev).contiguous()
        y = np.asarray(training_labels)[:n]
        lab = torch.as_tensor(np.argmax(y, axis=1) if y.ndim == 2 else y, device=dev).to(torch.int32).contiguous()
        for epoch in range(int(config.training_epochs)):
            nt.steps(x_all, lab, total_batch, masks=nt.draw_masks(n) if use_dropout else None)
            if verbose and epoch % int(getattr(config, "display_step", 1)) == 0:
                cost_sum = nt.export(want_cost_sum=True)[2]
                print(f"Epoch: {epoch + 1:04d} | cost = {cost_sum / total_batch:.9f}")
        out = nt.export()
        nt.close()
        return out
    w_keys = [k for k in ("h1", "h2", "h3", "out") if k in weights]
    n_out = int(np.asarray(weights["out"]).shape[1])
    if (not use_dropout and total_batch > 0 and NativeDenseTrainer.eligible(w_keys, n_out, "softmax")):
        # K8 with the softmax loss (csrc/nnv_dense_train.hip): feed-forward decoders K7 does not take (more than 256 classes,
        # other hidden widths), 2 L + 2 launches per step
        b_keys = [{"h1": "b1", "h2": "b2", "h3": "b3", "out": "out"}[k] for k in w_keys]
        nd = NativeDenseTrainer([weights[k] for k in w_keys], [biases[k] for k in b_keys], "tanh", config.learning_rate,
                                config.beta_regularization, bs, reg_layers=tuple(range(len(w_keys) - 1)), loss="softmax")
        n = total_batch * bs
        x_all = torch.as_tensor(np.asarray(training_set)[:n], dtype=torch.float32, device=nd.device).contiguous()
        y = np.asarray(training_labels)[:n]
        lab = torch.as_tensor(np.argmax(y, axis=1) if y.ndim == 2 else y, device=nd.device).to(torch.int32).contiguous()
        for epoch in range(int(config.training_epochs)):
            nd.steps(x_all, lab, total_batch)
            if verbose and epoch % int(getattr(config, "display_step", 1)) == 0:
                print(f"Epoch: {epoch + 1:04d} | cost = {nd.export(want_cost_sum=True)[3] / total_batch:.9f}")
        Ws, Bs, _ = nd.export()
        nd.close()
        return dict(zip(w_keys, Ws)), dict(zip(b_keys, Bs))
    tr = FNNTrainer(weights, biases, config.learning_rate, config.beta_regularization,
                    use_dropout=use_dropout, keep_prob=keep_prob)
    dev = tr.device
    x_all = torch.as_tensor(np.asarray(training_set), dtype=torch.float32, device=dev)
    y_all = torch.as_tensor(np.asarray(training_labels), dtype=torch.float32, device=dev)
    step = lambda xb, yb: tr.step(xb, yb)
    run = GraphedStep(step, tr.parameters(), tr.opt, [x_all[:bs], y_all[:bs]], verbose) if total_batch > 0 else step

    for epoch in range(int(config.training_epochs)):
        avg_cost = torch.zeros((), device=dev)
        for i in range(total_batch):
            avg_cost += run(x_all[i * bs:(i + 1) * bs], y_all[i * bs:(i + 1) * bs]) / max(total_batch, 1)
        if verbose and epoch % int(getattr(config, "display_step", 1)) == 0:
            print(f"Epoch: {epoch + 1:04d} | cost = {float(avg_cost):.9f}")

    return tr.export()
