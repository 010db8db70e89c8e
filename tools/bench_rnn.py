#!/usr/bin/env python3
"""Secondary benchmark (NOT the headline; bench.py is): BASELINE configs[4], the recurrent decoder
forward (src/rnn_viterbi.py: LSTM(20) encoder over 2L received symbols -> LSTM(20) decoder over 2L
teacher-forced steps -> linear), (133,171) block_len L = 256, through the register-resident HIP
kernel, with oracle/rnn_oracle.py (numpy fp32) timed beside it on a bounded sample.
Prints one JSON line.  Usage: python tools/bench_rnn.py [--blocks 1000000] [--steps 3]"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd
from nnviterbi_amd import rnn_viterbi as rv
from nnviterbi_amd.decoder import Seq2SeqLSTM
from oracle import rnn_oracle as R

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=1_000_000)   # configs[4] names 10M (--blocks 10000000: ~120 GB of HBM); 1M keeps the default run short
ap.add_argument("--block-len", type=int, default=256)
ap.add_argument("--steps", type=int, default=3)
args = ap.parse_args()
L, B = args.block_len, args.blocks
cfg = rv.RNNDecoderConfig(block_length=L, encoder_hidden_units=20)
params = rv.initialize_parameters(cfg, rng=np.random.RandomState(7))
net = Seq2SeqLSTM(params)
g = torch.Generator(device="cuda").manual_seed(1)
# synthetic feeds of the reference's shape: int-truncated noisy BPSK symbols, [EOS | +-1 info | zeros]
info = (torch.rand((L, B), device="cuda", generator=g) > 0.5)
enc = torch.trunc(torch.sign(torch.randn((2 * L, B), device="cuda", generator=g))
                  + float(cfg.sigma_vec[cfg.test_snr_index]) * torch.randn((2 * L, B), device="cuda", generator=g))
dec = torch.zeros((2 * L, B), device="cuda")
dec[0] = 1.0
dec[1:L + 1] = info.float() * 2 - 1
info_u8 = info.to(torch.uint8).contiguous()
counters = torch.zeros(4, dtype=torch.int64, device="cuda")
del info
torch.cuda.empty_cache()   # the generators' temporaries (3 x 20 GB at 10 M blocks) stay cached otherwise, and every forward's
                           # 20 GB of logits then costs a hipMalloc + hipFree round trip (measured: 686 ms per step instead of 444)
r = net.forward(enc, dec, info_bits=info_u8, n_bits=L, counters=counters)
r = net.forward(enc, dec, info_bits=info_u8, n_bits=L, counters=counters)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.steps):
    r = net.forward(enc, dec, info_bits=info_u8, n_bits=L, counters=counters)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.steps
fast = Seq2SeqLSTM(params, precision="f16x3")
rf = fast.forward(enc, dec)
torch.cuda.synchronize()
e0.record()
for _ in range(args.steps):
    rf = fast.forward(enc, dec)
e1.record(); torch.cuda.synchronize()
ms_fast = e0.elapsed_time(e1) / args.steps
fast_dev = float((rf["logits"] - r["logits"]).abs().max().item())
H = 20
flop = 2 * (2 * L) * 2 * (1 + H) * 4 * H + 2 * (2 * L) * H          # both cells' kernels + the projection
m = 2048
t0 = time.perf_counter()
ref, _ = R.seq2seq_forward(enc[:, :m].cpu().numpy(), dec[:, :m].cpu().numpy(), params, np.float32)
dt = time.perf_counter() - t0
ref64, _ = R.seq2seq_forward(enc[:, :m].cpu().numpy(), dec[:, :m].cpu().numpy(), params, np.float64)
err = float(np.abs(r["logits"][:, :m].cpu().numpy() - ref64).max())
print(json.dumps({
    "metric": f"recurrent decoder blocks/sec (rnn_viterbi LSTM(20) seq2seq forward, (133,171) block_len={L})",
    "value": B / (ms * 1e-3), "unit": "blocks/s", "ms_per_step": ms, "blocks": B, "dtype": "f32",
    "config": {"workload": f"BASELINE configs[4]: recurrent path, (133,171) block_len={L}, {B} blocks, 1 GPU",
               "cell": "LSTM, not GRU: BASELINE.json says GRU, the reference builds tf.contrib.rnn.LSTMCell "
                       "(src/rnn_viterbi.py:289-318); the reference script itself cannot be imported (SURVEY F10)"},
    "weights": "Glorot-uniform init (TF defaults of the reference graph), untrained",
    "roofline": {"bound": "mfma", "achieved": flop * B / (ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                 "frac": flop * B / (ms * 1e-3) / 1e12 / 157.3, "algorithmic_flop_per_block": flop,
                 "note": "33 MFMAs/step issue 96x24 of which 80x21 are live; gate VALU work shares the fp32 pipe"},
    "cpu_baseline": {"value": m / dt, "unit": "blocks/s", "cores": len(os.sched_getaffinity(0)), "kind": "port",
                     "sample": f"first {m} blocks, oracle/rnn_oracle.py numpy fp32 (per-step [B,21]x[21,80] "
                               f"matmuls; BLAS decides how many of the cores it uses)"},
    "parity_sample": {"blocks": m, "max_abs_logit_error_vs_float64": err,
                      "cpu_fp32_error_vs_float64": float(np.abs(ref - ref64).max())},
    "fast_mode": {"dtype": "f16x3", "value": B / (ms_fast * 1e-3), "unit": "blocks/s", "ms_per_step": ms_fast,
                  "max_abs_logit_difference_from_f32": fast_dev},
    "untrained_sign_decision_ber": float(counters[0].item() / max(counters[2].item(), 1)),
}))
