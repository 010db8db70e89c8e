#!/usr/bin/env python3
"""SURVEY section 8f ranks 2 and 4, the training side: the bit-wise decoder's step (1036 -> 30 -> 2 ReLU, batch 1000) and
the sliding-window decoder's (200 -> 500 x 4 -> 2 tanh, batch 100, windows of one stream) through K8
(csrc/nnv_dense_train.hip) and through PyTorch autograd replayed from a HIP graph, microseconds per step.  One JSON line.
Usage: python tools/bench_dense_trainer.py"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd  # noqa: F401
from nnviterbi_amd import training

def net(dims, seed):
    rng = np.random.RandomState(seed)
    return ([(rng.randn(dims[i], dims[i + 1]) * np.sqrt(2.0 / dims[i])).astype(np.float32) for i in range(len(dims) - 1)],
            [np.zeros(dims[i + 1], np.float32) for i in range(len(dims) - 1)])

out = {}
for name, dims, act, bs, steps, stride in (("bitwise", [1036, 30, 2], "relu", 1000, 300, None),
                                           ("sliding_window", [200, 500, 500, 500, 500, 2], "tanh", 100, 2000, 2)):
    Ws, Bs = net(dims, 0)
    n_in = dims[0]
    rs = n_in if stride is None else stride
    flat = torch.randn(((steps * bs - 1) * rs + n_in,), device="cuda")
    lab = torch.randint(0, 2, (steps * bs,), device="cuda", dtype=torch.int32)
    res = {"dims": dims, "batch": bs, "steps": steps, "launches_per_step": 2 * (len(dims) - 1) + 1}
    nt = training.NativeDenseTrainer(Ws, Bs, act, 2e-4, 1e-4, bs, reg_layers=(0, 1) if len(dims) > 3 else (0,))
    nt.steps(flat, lab, steps, row_stride=rs, step_stride=bs * rs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        nt.steps(flat, lab, steps, row_stride=rs, step_stride=bs * rs)
    torch.cuda.synchronize()
    res["native_us_per_step"] = (time.perf_counter() - t0) / (3 * steps) * 1e6
    macs = sum(int(np.prod(w.shape)) for w in Ws)
    res["native_tflops"] = 6 * macs * bs / (res["native_us_per_step"] * 1e-6) / 1e12
    nt.close()
    # the PyTorch fallback of the mirror modules: autograd + TFAdam, the step captured once into a HIP graph
    W = [torch.tensor(w, device="cuda", requires_grad=True) for w in Ws]
    B = [torch.tensor(b, device="cuda", requires_grad=True) for b in Bs]
    opt = training.TFAdam(W + B, lr=2e-4)
    rows = flat.unfold(0, n_in, rs)[:steps * bs]
    y = torch.nn.functional.one_hot(lab.long(), 2).to(torch.float32)
    f = torch.relu if act == "relu" else torch.tanh
    def step(xb, yb):
        h = xb
        for w, b in zip(W, B):
            h = f(h @ w + b)
        cost = torch.nn.functional.binary_cross_entropy_with_logits(h, yb) + 1e-4 * 0.5 * ((W[0] ** 2).sum() + (W[1] ** 2).sum())
        opt.zero_grad(set_to_none=False)
        cost.backward()
        opt.step()
        return cost.detach()
    run = training.GraphedStep(step, W + B, opt, [rows[:bs], y[:bs]])
    n_a = min(steps, 300)
    for i in range(10):
        run(rows[i * bs:(i + 1) * bs], y[i * bs:(i + 1) * bs])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_a):
        run(rows[i * bs:(i + 1) * bs], y[i * bs:(i + 1) * bs])
    torch.cuda.synchronize()
    res["autograd_hip_graph_us_per_step"] = (time.perf_counter() - t0) / n_a * 1e6
    res["graph_captured"] = run.captured
    out[name] = res
print(json.dumps(out))
