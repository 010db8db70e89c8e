#!/usr/bin/env python3
"""Scratch GPU check of the wide (layer-at-a-time) path vs the oracle + a rough timing."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd
from nnviterbi_amd.decoder import WideMLP
from oracle import fnn_oracle as O

def net(dims, seed):
    rng = np.random.RandomState(seed)
    Ws = [(rng.randn(dims[i], dims[i+1]) * np.sqrt(2.0/dims[i])).clip(-2*np.sqrt(2.0/dims[i]), 2*np.sqrt(2.0/dims[i])).astype(np.float32) for i in range(len(dims)-1)]
    bs = [(rng.randn(dims[i+1]) * 0.1).astype(np.float32) for i in range(len(dims)-1)]
    return Ws, bs

dims = [200, 500, 500, 500, 500, 2]
Ws, bs = net(dims, 0)
w = WideMLP(Ws, bs)
rng = np.random.RandomState(1)
n = 5000
stream = np.trunc(rng.randn(2 * n + 200) * 0.8 + rng.choice([-1.0, 1.0], 2 * n + 200)).astype(np.float32)
sd = torch.from_numpy(stream).cuda()
r = w.decode_stream(sd, n, want_act=True)
x = np.lib.stride_tricks.sliding_window_view(stream, 200)[::2][:n]
ref = O.decode_c(np.ascontiguousarray(x), Ws, bs, 1, want_act=True, want_bits=False)
c64, _, o64 = O.decode(x, Ws, bs, 1, np.float64)
cls = r["cls32"].cpu().numpy(); act = r["act"].cpu().numpy()
print("stream: mismatch vs C", (cls != ref["cls"]).sum(), "vs fp64", (cls != c64).sum(), "max|act-fp64|", np.abs(act - o64).max(), "C-oracle", np.abs(ref["act"] - o64).max())
r2 = w.decode(torch.from_numpy(np.ascontiguousarray(x)).cuda(), want_act=True)
print("matrix == stream:", torch.equal(r2["cls32"], r["cls32"]), float((r2["act"] - r["act"]).abs().max()))
# wide class layer + odd sizes
dims2 = [14, 300, 70, 128]
W2, b2 = net(dims2, 2)
w2 = WideMLP(W2, b2)
x2 = rng.randn(777, 14).astype(np.float32)
r3 = w2.decode(torch.from_numpy(x2).cuda(), want_act=True)
c64, _, o64 = O.decode(x2, W2, b2, 7, np.float64)
print("generic: mismatch vs fp64", (r3["cls32"].cpu().numpy() != c64).sum(), "max|act-fp64|", np.abs(r3["act"].cpu().numpy() - o64).max())
# f16x3 precision
wf = WideMLP(Ws, bs, precision="f16x3")
rf = wf.decode_stream(sd, n, want_act=True)
c64, _, o64 = O.decode(np.ascontiguousarray(x), Ws, bs, 1, np.float64)
print("f16x3 stream: mismatch vs fp64", (rf["cls32"].cpu().numpy() != c64).sum(), "max|act-fp64|", np.abs(rf["act"].cpu().numpy() - o64).max())
# timing: 1M windows
nbig = 1_000_000
big = torch.from_numpy(np.trunc(rng.randn(2 * nbig + 200)).astype(np.float32)).cuda()
w.decode_stream(big, nbig); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
wf.decode_stream(big, nbig)
for rep, net_ in enumerate((w, wf)):
    t0 = time.perf_counter(); e0.record()
    for _ in range(5): net_.decode_stream(big, nbig)
    e1.record(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    flop = 2 * sum(dims[i] * dims[i+1] for i in range(5))
    print(("f32  " if rep == 0 else "f16x3") + f" 1M windows: wall {dt*1e3:.2f} ms, events {e0.elapsed_time(e1)/5:.2f} ms  {nbig/dt/1e6:.1f} M bits/s  {flop*nbig/dt/1e12:.1f} TFLOP/s algorithmic")
