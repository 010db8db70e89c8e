#!/usr/bin/env python3
"""Small bit-wise banks (the reference's own call shape is ONE position per run, src/fnn_viterbi_bitwise.py:503-505):
K4d (one HBM-streaming launch per position) against K4b (LDS-tiled GEMM), 1036 inputs x --rows rows.
One JSON line.  Usage: python tools/bench_bank_small.py [--rows 300000] [--positions 1 2 3 4]"""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd  # noqa: F401
from nnviterbi_amd import fnn_viterbi_bitwise as bw
from nnviterbi_amd.decoder import BitwiseBank

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=300_000)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--positions", type=int, nargs="+", default=[1, 2, 3, 4])
ap.add_argument("--precision", default="f32")
args = ap.parse_args()
cfg = bw.BitwiseDecoderConfig()
n = args.rows
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.trunc(torch.sign(torch.randn((n, cfg.n_input), device="cuda", generator=g))
                + 0.8 * torch.randn((n, cfg.n_input), device="cuda", generator=g)).contiguous()


def timed(fn, steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, r


out = {"precision": args.precision, "rows": n, "n_input": cfg.n_input, "hbm_bytes_per_position_pass": 4 * cfg.n_input * n, "legs": []}
for P in args.positions:
    models = []
    for p in range(P):
        wp, bp = bw.initialize_weights(cfg, rng=np.random.RandomState(100 + p))
        models.append((wp["h1"], bp["b1"], wp["out"], np.array([0.01, 0.0], np.float32)))
    bank = BitwiseBank(models, precision=args.precision)
    ms_d, r_d = timed(lambda: bank.decode(x), args.steps)
    bank.set_path("tiled")
    ms_t, r_t = timed(lambda: bank.decode(x), args.steps)
    bank.set_path("auto")
    differ = int((r_d["bits"] != r_t["bits"]).sum().item())
    out["legs"].append({"positions": P, "default_ms": ms_d, "tiled_K4b_ms": ms_t,
                        "default_hbm_frac_per_position_pass": 4 * cfg.n_input * n * P / (ms_d * 1e-3) / 8e12,
                        "default_fp32_mfma_frac": 2 * cfg.n_input * cfg.n_hidden_1 * P * n / (ms_d * 1e-3) / 157.3e12,
                        "decided_bits_differing": differ, "decided_bits": n * P})
    bank.close()
print(json.dumps(out))
