#!/usr/bin/env python3
"""SURVEY section 8f rank 1, measured: one epoch of the reference's default training set (DecoderConfig(): b = 7, 64,000
rows, batch 256 = 250 steps; and BASELINE configs[2]'s b = 8 network) through the two-kernel step (K7,
csrc/nnv_train.hip) and through PyTorch autograd replayed from a HIP graph (training.FNNTrainer + GraphedStep).
One JSON line.  Usage: python tools/bench_trainer.py"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd  # noqa: F401
from nnviterbi_amd import fnn_viterbi as fv, training

out = {}
for b in (7, 8):
    cfg = fv.DecoderConfig(block_length=b)
    rng = np.random.RandomState(0)
    w, bd = fv.initialize_weights(cfg, rng=rng)
    n, bs = 64000, 256
    steps = n // bs
    x = torch.randn((steps * bs, 2 * b), device="cuda")
    lab = torch.randint(0, 2 ** b, (steps * bs,), device="cuda", dtype=torch.int32)
    y = torch.nn.functional.one_hot(lab.long(), 2 ** b).to(torch.float32)
    res = {"steps_per_epoch": steps, "batch": bs, "dims": fv.layer_dims(cfg)}
    nt = training.NativeFNNTrainer(w, bd, cfg.learning_rate, cfg.beta_regularization, bs)
    nt.steps(x, lab, steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        nt.steps(x, lab, steps)
    torch.cuda.synchronize()
    res["native_us_per_step"] = (time.perf_counter() - t0) / (5 * steps) * 1e6
    # FLOP of a step: forward 2 MAC per weight and row, backward twice that (dH and dW)
    macs = sum(int(np.prod(v.shape)) for v in w.values())
    res["native_tflops"] = 6 * macs * bs / (res["native_us_per_step"] * 1e-6) / 1e12
    nt.close()
    at = training.FNNTrainer(w, bd, cfg.learning_rate, cfg.beta_regularization)
    run = training.GraphedStep(lambda xb, yb: at.step(xb, yb), at.parameters(), at.opt, [x[:bs], y[:bs]])
    for i in range(20):
        run(x[i * bs:(i + 1) * bs], y[i * bs:(i + 1) * bs])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        run(x[i * bs:(i + 1) * bs], y[i * bs:(i + 1) * bs])
    torch.cuda.synchronize()
    res["autograd_hip_graph_us_per_step"] = (time.perf_counter() - t0) / steps * 1e6
    res["graph_captured"] = run.captured
    # the same step on the host's cores (PyTorch CPU kernels; the reference's TensorFlow-CPU path is not installable offline)
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))     # the box's CPU share, not its 128 hardware threads
    ct = training.FNNTrainer(w, bd, cfg.learning_rate, cfg.beta_regularization, device="cpu")
    xc, yc = x[:50 * bs].cpu(), y[:50 * bs].cpu()
    for i in range(5):
        ct.step(xc[i * bs:(i + 1) * bs], yc[i * bs:(i + 1) * bs])
    t0 = time.perf_counter()
    for i in range(50):
        ct.step(xc[i * bs:(i + 1) * bs], yc[i * bs:(i + 1) * bs])
    res["torch_cpu_us_per_step"] = (time.perf_counter() - t0) / 50 * 1e6
    res["torch_cpu_threads"] = torch.get_num_threads()
    out[f"b{b}"] = res
print(json.dumps(out))
