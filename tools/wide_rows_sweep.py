#!/usr/bin/env python3
"""K6 (whole chain in one persistent launch) against K4 (one launch per layer) over the NUMBER OF ROWS: the
sliding-window decoder's network 200 -> 500 x 4 -> 2 and two narrower ones.  The handle is pinned with WideMLP.set_path.
Usage: python tools/wide_rows_sweep.py"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd  # noqa: F401
from nnviterbi_amd.decoder import WideMLP


def timed(fn, steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


out = []
for dims in ([200, 500, 500, 500, 500, 2], [16, 256, 128, 64, 256], [20, 300, 300, 2]):
    rng = np.random.RandomState(1)
    Ws = [(rng.randn(dims[i], dims[i + 1]) * np.sqrt(2.0 / dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [(rng.randn(d) * 0.1).astype(np.float32) for d in dims[1:]]
    net = WideMLP(Ws, bs)
    for n in (64, 512, 2048, 8192, 16384, 32768, 65536, 262144, 1048576):
        x = torch.randn((n, dims[0]), device="cuda")
        steps = 200 if n <= 65536 else 20
        net.set_path("auto")
        a = timed(lambda: net.decode(x), steps)
        net.set_path("layered")
        b = timed(lambda: net.decode(x), steps)
        net.set_path("auto")
        out.append({"dims": dims, "rows": n, "K6_us": a * 1e3, "K4_us": b * 1e3})
        print("%-28s rows %8d  K6 %9.1f us  K4 %9.1f us  K4/K6 %.2f" % ("-".join(map(str, dims)), n, a * 1e3, b * 1e3, b / a), flush=True)
    net.close()
print(json.dumps(out))
