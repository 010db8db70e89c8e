#!/usr/bin/env python3
"""Scratch GPU sanity run (not a test): HIP decode vs the oracle on several
network shapes, plus a rough timing.  Usage: python tools/gpu_check.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd  # noqa
from nnviterbi_amd.decoder import PackedMLP, generate_awgn_blocks  # noqa
from oracle import fnn_oracle as O  # noqa


def make_net(dims, seed):
    rng = np.random.RandomState(seed)
    Ws = [(rng.randn(dims[i], dims[i + 1]) * np.sqrt(2.0 / dims[i])).astype(np.float32)
          for i in range(len(dims) - 1)]
    bs = [(rng.randn(dims[i + 1]) * 0.1).astype(np.float32) for i in range(len(dims) - 1)]
    return Ws, bs


def check(dims, b, n, seed=0, precision='f32'):
    Ws, bs = make_net(dims, seed)
    rng = np.random.RandomState(seed + 1)
    x = rng.randn(n, dims[0]).astype(np.float32)
    labels = rng.randint(0, dims[-1], size=n).astype(np.int32)
    net = PackedMLP(Ws, bs, precision=precision)
    xd = torch.from_numpy(x).cuda()
    ld = torch.from_numpy(labels).cuda()
    r = net.decode(xd, labels=ld, tile_bits=b, want_cls64=True, want_bits=True)
    torch.cuda.synchronize()
    cls = r["cls32"].cpu().numpy()
    act = net.forward(xd).cpu().numpy()
    ref = O.decode_c(x, Ws, bs, b, labels=labels, want_act=True)
    c64, _, o64 = O.decode(x, Ws, bs, b, np.float64)
    mism = cls != ref["cls"]
    mism64 = cls != c64
    margin = O.top2_margin(o64)
    print(f"[{precision}] dims={dims} b={b} n={n} info={net.launch_info(n)}")
    print(f"   mismatch vs C-oracle {mism.sum()}  vs fp64 {mism64.sum()}"
          f"  max fp64 margin on mismatches {margin[mism64].max() if mism64.any() else 0:.3e}")
    print(f"   max|act - fp64| {np.abs(act - o64).max():.3e}   C-oracle {np.abs(ref['act'] - o64).max():.3e}")
    print(f"   counters hip {r['counters'].cpu().numpy()}  oracle {ref['counters']}")
    assert np.array_equal(r["cls64"].cpu().numpy(), cls.astype(np.int64))
    assert np.array_equal(r["bits"].cpu().numpy(), O.class_to_bits(cls, b))
    assert np.array_equal(O.error_counters(cls, labels, b), r["counters"].cpu().numpy())
    net.close()


def bench(dims, b, n, iters=5, precision='f32'):
    Ws, bs = make_net(dims, 3)
    net = PackedMLP(Ws, bs, precision=precision)
    x, labels = generate_awgn_blocks("133171" if b >= 6 else "57", b, n, 0.56, seed=1)
    if x.shape[1] != dims[0]:
        x = torch.randn(n, dims[0], device="cuda")
    cls = torch.empty(n, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    net.decode_timed(x, labels, cls, cnt, 1, tile_bits=b)
    ms = net.decode_timed(x, labels, cls, cnt, iters, tile_bits=b) / iters
    flop = 2 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
    print(f"bench [{precision}] dims={dims} n={n}: {ms:.3f} ms/launch  {n / ms / 1e3:.1f} M rows/s  "
          f"{flop * n / ms / 1e9:.1f} TFLOP/s  info={net.launch_info(n)}")
    net.close()


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    for prec in ("f32", "f16x3"):
        check([16, 128, 64, 32, 256], 8, 100007, precision=prec)
        check([14, 128, 64, 32, 128], 7, 50001, precision=prec)
        check([20, 128, 64, 32, 1024], 10, 20011, precision=prec)
        check([20, 128, 64, 1024], 10, 20000, precision=prec)
        check([6, 128, 64, 32, 8], 3, 1000, precision=prec)
        check([16, 128, 256], 8, 3333, precision=prec)
        check([16, 100, 50, 20, 256], 8, 5000, precision=prec)
        check([16, 128, 64, 32, 256], 8, 5, precision=prec)
    for prec in ("f32", "f16x3"):
        bench([16, 128, 64, 32, 256], 8, 80 * 1000 * 1000, precision=prec)
        bench([20, 128, 64, 32, 1024], 10, 8 * 1000 * 1000, precision=prec)
        bench([20, 128, 64, 1024], 10, 8 * 1000 * 1000, precision=prec)
        bench([14, 128, 64, 32, 128], 7, 16 * 1000 * 1000, precision=prec)
