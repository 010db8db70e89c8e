#!/usr/bin/env python3
"""Secondary measurement: K1 launch time and throughput against the batch size (rows per launch), BASELINE
configs[2] network, both precision modes.  One JSON line.  Usage: python tools/bench_batch_sweep.py"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd
from nnviterbi_amd import fnn_viterbi as fv, fnn_viterbi_mixed_snr as mx
from nnviterbi_amd.decoder import PackedMLP, generate_awgn_blocks
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
w, b = mx.load_model(os.path.join(root, "neural-network-based-viterbi-decoding_amd", "weights", "fnn_133171_b8_2dB.npz"))
Ws, bs = fv.weights_as_lists(w, b)
sizes = [32, 256, 2048, 8192, 32768, 131072, 524288, 2097152, 8388608, 33554432, 80000000]
x, lab = generate_awgn_blocks("133171", 8, sizes[-1], 0.5616748812614791, seed=1, rows_per_block=8)
out = {"metric": "K1 launch time vs rows per launch, 16-128-64-32-256", "unit": "us per launch", "rows": sizes}
for prec in ("f32", "f16x3"):
    net = PackedMLP(Ws, bs, precision=prec)
    cls = torch.empty(sizes[-1], dtype=torch.int32, device="cuda")
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    us = []
    for n in sizes:
        iters = 200 if n <= 1 << 19 else (20 if n <= 1 << 23 else 5)
        net.decode_timed(x[:n], lab[:n], cls[:n], cnt, 3, tile_bits=8)
        us.append(1e3 * net.decode_timed(x[:n], lab[:n], cls[:n], cnt, iters, tile_bits=8) / iters)
    out[prec] = {"us_per_launch": [round(v, 2) for v in us],
                 "M_rows_per_s": [round(n / v, 1) for n, v in zip(sizes, us)]}
    net.close()
print(json.dumps(out))
