#!/usr/bin/env python3
"""Secondary benchmark (NOT the headline; bench.py is): BASELINE configs[0] and configs[1], the (7,5)
K=3 code at block_len 20 (1k blocks, 3-layer FNN) and block_len 100 (1M blocks @ SNR 4 dB, 2-layer FNN).
Tiling rule: b = 10 divides both -> 2 / 10 tiles per block, net 20 -> 128 -> 64 (-> 32) -> 1024; the class
layer (256 / 128 KiB) exceeds the LDS: as many M-tiles as fit stay resident, every wave streams the rest from L2 for its own rows.
Weights: a matched-filter network of the reference's architecture (pass-through hidden layers, class layer =
correlation with every codeword: the ML decision per tile); throughput does not depend on the weights.
Prints one JSON line per config.  Usage: python tools/bench_configs01.py [--steps 20]"""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd
from nnviterbi_amd import fnn_viterbi as fv
from nnviterbi_amd.decoder import PackedMLP, generate_awgn_blocks


def matched_filter_network(b, n_layers):
    cb = fv.generate_class_bits(b)
    sym = np.where(fv.encode_57_batch(cb) == 1, 1.0, -1.0).astype(np.float32)
    n = 2 * b
    dims = [n] + [128, 64, 32][:n_layers] + [2 ** b]
    Ws = [np.zeros((dims[i], dims[i + 1]), np.float32) for i in range(n_layers + 1)]
    Ws[0][np.arange(n), np.arange(n)] = 0.05
    for l in range(1, n_layers):
        Ws[l][np.arange(n), np.arange(n)] = 1.0
    Ws[n_layers][:n, :] = sym.T * 2.0
    return Ws, [np.zeros(d, np.float32) for d in dims[1:]], dims


ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--weights", choices=("matched", "random"), default="matched",
                help="matched: the matched-filter network (meaningful BER, but mostly zero weights); random: seeded "
                     "truncated-normal init of the reference (dense operands: MFMA power draw is data dependent)")
args = ap.parse_args()
for name, block_len, n_blocks, n_layers, snr_db in (("configs[0]", 20, 1_000, 3, 2.0), ("configs[1]", 100, 1_000_000, 2, 4.0)):
    b = 10
    rpb = block_len // b
    n = n_blocks * rpb
    Ws, bs, dims = matched_filter_network(b, n_layers)
    if args.weights == "random":
        rng = np.random.RandomState(5)
        Ws = [fv.truncated_normal((dims[i], dims[i + 1]), np.sqrt(2.0 / dims[i]), rng) for i in range(len(dims) - 1)]
        bs = [(rng.randn(d) * 0.1).astype(np.float32) for d in dims[1:]]
    sigma = float(fv.DecoderConfig(block_length=b, snr_db=snr_db).sigma)
    x, lab = generate_awgn_blocks("57", b, n, sigma, seed=11, rows_per_block=rpb)
    out = {"metric": f"decoded blocks/sec, (7,5) block_len={block_len} @ SNR {snr_db} dB, {n_layers}-layer FNN",
           "config": name, "weights": args.weights, "blocks": n_blocks, "rows_per_block": rpb, "net": dims, "unit": "blocks/s"}
    flop = 2 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
    for prec in ("f32", "f16x3", "f16"):      # "f16": the opt-in single-product mode, outside the 1e-5 contract
        net = PackedMLP(Ws, bs, precision=prec)
        cls = torch.empty(n, dtype=torch.int32, device="cuda")
        cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
        net.decode_timed(x, lab, cls, cnt, 3, tile_bits=b)
        cnt.zero_()
        ms = net.decode_timed(x, lab, cls, cnt, args.steps, tile_bits=b) / args.steps
        c = cnt.cpu().numpy()
        out[prec] = {"value": n_blocks / (ms * 1e-3), "ms_per_launch": ms, "ber": float(c[0] / c[2]),
                     "tflops_algorithmic": flop * n / (ms * 1e-3) / 1e12,
                     "frac_of_peak": flop * n / (ms * 1e-3) / 1e12 / (157.3 if prec == "f32" else 2500.0),
                     "class_tiles_in_lds": net.launch_info(n)["class_tiles_in_lds"], "class_tiles": dims[-1] // 32}
        net.close()
    print(json.dumps(out))
