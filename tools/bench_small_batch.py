#!/usr/bin/env python3
"""Short inputs: K1 (persistent, network staged into LDS) against K1w (a workgroup of four waves per 32-row tile,
weights from L2) over the number of rows, on BASELINE configs[2]'s network (b = 8, 256 classes) and configs[0]/[1]'s
(b = 10, 1024 classes).  Microseconds per call, pipelined (many calls, one synchronisation) -- what the kernel costs --
and synchronised per call -- what a Python caller waits.  One JSON line.  Usage: python tools/bench_small_batch.py"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd  # noqa: F401
from nnviterbi_amd.decoder import PackedMLP

out = {}
for b, n_cls in ((8, 256), (10, 1024)):
    rng = np.random.RandomState(b)
    dims = [2 * b, 128, 64, 32, n_cls]
    Ws = [(rng.randn(dims[i], dims[i + 1]) * np.sqrt(2.0 / dims[i])).astype(np.float32) for i in range(4)]
    bs = [np.zeros(dims[i + 1], np.float32) for i in range(4)]
    net = PackedMLP(Ws, bs)
    res = {}
    for n_rows in (32, 256, 2000, 8192, 16384, 32768, 65536, 131072, 262144):
        x = torch.randn((n_rows, 2 * b), device="cuda")
        row = {}
        for path in ("persistent", "split"):
            net.set_path(path)
            for _ in range(20):
                net.decode(x, tile_bits=b)
            torch.cuda.synchronize()
            n = 300
            t0 = time.perf_counter()
            for _ in range(n):
                net.decode(x, tile_bits=b)
            torch.cuda.synchronize()
            row[path + "_us_pipelined"] = (time.perf_counter() - t0) / n * 1e6
            t0 = time.perf_counter()
            for _ in range(100):
                net.decode(x, tile_bits=b)
                torch.cuda.synchronize()
            row[path + "_us_synced"] = (time.perf_counter() - t0) / 100 * 1e6
            call = net.bound_decode(x, tile_bits=b)          # the host work of a call done once (serving)
            for _ in range(20):
                call()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                call()
            torch.cuda.synchronize()
            row[path + "_us_bound_pipelined"] = (time.perf_counter() - t0) / n * 1e6
            t0 = time.perf_counter()
            for _ in range(100):
                call()
                torch.cuda.synchronize()
            row[path + "_us_bound_synced"] = (time.perf_counter() - t0) / 100 * 1e6
        res[str(n_rows)] = row
    out[f"b{b}"] = res
    net.close()
print(json.dumps(out))
