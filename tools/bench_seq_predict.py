#!/usr/bin/env python3
"""Secondary benchmark (NOT the headline; bench.py is): the reference's sliding-window decoder
forward (src/fnn_viterbi_seq_predict.py: 200 -> 500 x4 -> 2, window i = stream[2i : 2i+200]): the whole chain
in one persistent launch (K6, the default), the layer-at-a-time path beside it (K4, WideMLP.set_path) and its f16x3
mode, with the oracle's C restatement timed on a bounded sample.
Prints one JSON line.  Usage: python tools/bench_seq_predict.py [--windows 1000000] [--steps 10]"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd
from nnviterbi_amd import fnn_viterbi_seq_predict as sq
from nnviterbi_amd.decoder import WideMLP
from oracle import fnn_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--windows", type=int, default=1_000_000)   # the reference's default test stream
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()
cfg = sq.SequentialDecoderConfig()
w, b = sq.initialize_weights(cfg, rng=np.random.RandomState(7))
Ws, bs = sq.weights_as_lists(w, b)
net = WideMLP(Ws, bs)
np.random.seed(3)
info, coded = sq.generate_test_data(args.windows + 100, 1, cfg.sigma_vec, cfg.test_snr_index)
stream = torch.from_numpy(coded[0].astype(np.float32)).cuda()
n = args.windows
for _ in range(2):
    r = net.decode_stream(stream, n)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.steps):
    r = net.decode_stream(stream, n)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.steps
net.set_path("layered")                    # nnv_wide_set_path: the layer-at-a-time path
for _ in range(2):
    rl = net.decode_stream(stream, n)
torch.cuda.synchronize()
e0.record()
for _ in range(args.steps):
    rl = net.decode_stream(stream, n)
e1.record(); torch.cuda.synchronize()
ms_layered = e0.elapsed_time(e1) / args.steps
layered_diff = int((rl["cls32"] != r["cls32"]).sum().item())
net.set_path("auto")
fast = WideMLP(Ws, bs, precision="f16x3")
for _ in range(2):
    rf = fast.decode_stream(stream, n)
torch.cuda.synchronize()
e0.record()
for _ in range(args.steps):
    rf = fast.decode_stream(stream, n)
e1.record(); torch.cuda.synchronize()
ms_fast = e0.elapsed_time(e1) / args.steps
fast_diff = int((rf["cls32"] != r["cls32"]).sum().item())
dims = sq.layer_dims(cfg)
flop = 2 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
m = 20_000
x = sq.extract_windows(cfg.window_length, coded, 0, m)
threads = O.c_max_threads()
t0 = time.perf_counter(); ref = O.decode_c(x, Ws, bs, 1, want_bits=False, n_threads=threads); dt = time.perf_counter() - t0
mism = int((ref["cls"] != r["cls32"][:m].cpu().numpy()).sum())
print(json.dumps({
    "metric": "sliding-window decoded bits/sec (fnn_viterbi_seq_predict forward, window 200, 200-500x4-2)",
    "value": n / (ms * 1e-3), "unit": "bits/s", "ms_per_step": ms, "windows": n, "dtype": "f32",
    "roofline": {"bound": "mfma", "achieved": flop * n / (ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                 "frac": flop * n / (ms * 1e-3) / 1e12 / 157.3, "algorithmic_flop_per_window": flop},
    "cpu_baseline": {"value": m / dt, "unit": "bits/s", "cores": threads, "kind": "port",
                     "sample": f"first {m} windows, oracle/fnn_oracle.c"},
    "parity_sample": {"windows": m, "class_mismatches": mism},
    "kernel": "K6 wide_fused_kernel (whole chain in one launch, activations in LDS)",
    "layer_at_a_time": {"kernel": "K4 dense_tanh_kernel x4 + head", "value": n / (ms_layered * 1e-3), "unit": "bits/s",
                        "ms_per_step": ms_layered, "frac": flop * n / (ms_layered * 1e-3) / 1e12 / 157.3,
                        "classes_differing_from_fused": layered_diff},
    "fast_mode": {"dtype": "f16x3", "value": n / (ms_fast * 1e-3), "unit": "bits/s", "ms_per_step": ms_fast,
                  "classes_differing_from_f32": fast_diff},
}))
