#!/usr/bin/env python3
"""Secondary benchmark (NOT the headline): the bit-wise decoder's forward at the reference's default
configuration (src/fnn_viterbi_bitwise.py: block_length 512 -> 2(6+512) = 1036 int-truncated soft values
-> 30 ReLU -> 2 ReLU logits, 300,000 test rows, ONE bit position per model) through the layer-at-a-time
path.  1036 x 30 is an HBM-shaped layer (4,144 B in for 62 kFLOP per row): the roofline is memory.
Prints one JSON line.  Usage: python tools/bench_bitwise.py [--rows 300000]"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd
from nnviterbi_amd import fnn_viterbi_bitwise as bw
from nnviterbi_amd.decoder import WideMLP
from oracle import fnn_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=300_000)
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
cfg = bw.BitwiseDecoderConfig()
w, b = bw.initialize_weights(cfg, rng=np.random.RandomState(3))
Ws, bs = bw.active_layers(w, b)
n = args.rows
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.trunc(torch.sign(torch.randn((n, cfg.n_input), device="cuda", generator=g))
                + 0.8 * torch.randn((n, cfg.n_input), device="cuda", generator=g)).contiguous()
net = WideMLP(Ws, bs, activation="relu")
for _ in range(3):
    r = net.decode(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.steps):
    r = net.decode(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.steps
m = 20_000
xs = x[:m].cpu().numpy()
t0 = time.perf_counter(); ref = O.decode_c(xs, Ws, bs, 1, want_bits=False, activation="relu"); dt = time.perf_counter() - t0
mism = int((ref["cls"] != r["cls32"][:m].cpu().numpy()).sum())
in_bytes = 4 * cfg.n_input + 4
print(json.dumps({
    "metric": "bit-wise decoder rows/sec (fnn_viterbi_bitwise forward, 1036-30-2 ReLU, one bit position)",
    "value": n / (ms * 1e-3), "unit": "rows/s", "ms_per_step": ms, "rows": n, "dtype": "f32",
    "roofline": {"bound": "hbm", "achieved": in_bytes * n / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                 "frac": in_bytes * n / (ms * 1e-3) / 8.0e12, "algorithmic_bytes_per_row": in_bytes},
    "cpu_baseline": {"value": m / dt, "unit": "rows/s", "cores": O.c_max_threads(), "kind": "port",
                     "sample": f"first {m} rows, oracle/fnn_oracle.c"},
    "parity_sample": {"rows": m, "class_mismatches": mism},
}))
