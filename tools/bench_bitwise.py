#!/usr/bin/env python3
"""Secondary benchmark (NOT the headline): the bit-wise decoder's forward at the reference's default
configuration (src/fnn_viterbi_bitwise.py: block_length 512 -> 2(6+512) = 1036 int-truncated soft values
-> 30 ReLU -> 2 ReLU logits, 300,000 test rows).

Two legs, one JSON line:
  * `single`: ONE bit position per launch through the layer-at-a-time path (the reference's own call
    shape: one trained network per run).  1036 x 30 is an HBM-shaped layer (4,144 B in for 62 kFLOP).
  * `bank`: P positions in one fused launch (decoder.BitwiseBank, --positions, default all 512):
    the P first layers as one [1036 -> 32 P] fp32-MFMA GEMM, each position's head in the epilogue.
    Per row that is 2*1036*30*P FLOP against the same 4,144 B, i.e. MFMA-bound from P = 2 on
    (ridge: 157.3 TF / 8 TB/s = 19.7 FLOP/B = 1.3 positions), so this leg is priced against the
    fp32-MFMA peak; the HBM figure (x read ONCE for all positions) is reported beside it.
Usage: python tools/bench_bitwise.py [--rows 300000] [--positions 512]"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd
from nnviterbi_amd import fnn_viterbi_bitwise as bw
from nnviterbi_amd.decoder import BitwiseBank, WideMLP
from oracle import fnn_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=300_000)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--positions", type=int, default=512)
args = ap.parse_args()
cfg = bw.BitwiseDecoderConfig()
n, P = args.rows, args.positions
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.trunc(torch.sign(torch.randn((n, cfg.n_input), device="cuda", generator=g))
                + 0.8 * torch.randn((n, cfg.n_input), device="cuda", generator=g)).contiguous()


def timed(fn, steps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, r


# ---- single position per launch ---------------------------------------------------------------
w, b = bw.initialize_weights(cfg, rng=np.random.RandomState(3))
Ws, bs = bw.active_layers(w, b)
net = WideMLP(Ws, bs, activation="relu")
ms1, r1 = timed(lambda: net.decode(x), args.steps)
m = 20_000
xs = x[:m].cpu().numpy()
t0 = time.perf_counter(); ref = O.decode_c(xs, Ws, bs, 1, want_bits=False, activation="relu"); dt = time.perf_counter() - t0
mism1 = int((ref["cls"] != r1["cls32"][:m].cpu().numpy()).sum())
in_bytes = 4 * cfg.n_input + 4

# ---- the bank: P positions, one launch --------------------------------------------------------
models = []
for p in range(P):
    wp, bp = bw.initialize_weights(cfg, rng=np.random.RandomState(100 + p))
    models.append((wp["h1"], bp["b1"], wp["out"], np.array([0.01, 0.0], np.float32)))
bank = BitwiseBank(models)
msP, rP = timed(lambda: bank.decode(x), max(2, args.steps // 4))        # K4c: rows resident in LDS, first layers streamed
bank.set_path("tiled")                                                   # K4b: the LDS-tiled GEMM
msPt, rPt = timed(lambda: bank.decode(x), max(2, args.steps // 4))
bank.set_path("auto")
tiled_same = bool(torch.equal(rPt["bits"], rP["bits"]))
bitsP = rP["bits"][:, :m].cpu().numpy()
mismP = 0
for p in (0, P // 2, P - 1):
    refp = O.decode_c(xs, [models[p][0], models[p][2]], [models[p][1], models[p][3]], 1, want_bits=False,
                      activation="relu")
    mismP += int((refp["cls"] != bitsP[p]).sum())
bank16 = BitwiseBank(models, precision="f16x3")
msP16, rP16 = timed(lambda: bank16.decode(x), max(2, args.steps // 4))
flips16 = int((rP16["bits"] != rP["bits"]).sum().item())
bank16.close()
# ---- one position through the fused kernel (32-column tile + head in the epilogue) ------------
bank1 = BitwiseBank(models[:1])
ms1f, r1f = timed(lambda: bank1.decode(x), args.steps)
bank1h = BitwiseBank(models[:1], precision="f16x3")
ms1h, r1h = timed(lambda: bank1h.decode(x), args.steps)
flips1h = int((r1h["bits"] != r1f["bits"]).sum().item())
bank1h.close()
ref0 = O.decode_c(xs, [models[0][0], models[0][2]], [models[0][1], models[0][3]], 1, want_bits=False, activation="relu")
mism1f = int((ref0["cls"] != r1f["bits"][0, :m].cpu().numpy()).sum())
flop_row = 2 * cfg.n_input * cfg.n_hidden_1 * P + 2 * cfg.n_hidden_1 * 2 * P
bytes_row = 4 * cfg.n_input + P
print(json.dumps({
    "metric": "bit-wise decoder rows/sec (fnn_viterbi_bitwise forward, 1036-30-2 ReLU)",
    "rows": n, "dtype": "f32",
    "single": {
        "what": "one bit position per launch (layer-at-a-time path)", "value": n / (ms1 * 1e-3), "unit": "rows/s",
        "ms_per_step": ms1,
        "roofline": {"bound": "hbm", "achieved": in_bytes * n / (ms1 * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": in_bytes * n / (ms1 * 1e-3) / 8.0e12, "algorithmic_bytes_per_row": in_bytes},
        "parity_sample": {"rows": m, "class_mismatches": mism1},
        "ms_for_all_positions_this_way": ms1 * P,
    },
    "single_fused": {
        "what": "one bit position, first layer + head in ONE launch (BitwiseBank of 1: K4d, k-split over the waves, rows straight from HBM)", "value": n / (ms1f * 1e-3),
        "unit": "rows/s", "ms_per_step": ms1f,
        "roofline": {"bound": "hbm", "achieved": (4 * cfg.n_input + 1) * n / (ms1f * 1e-3) / 1e9, "peak": 8000.0,
                     "unit": "GB/s", "frac": (4 * cfg.n_input + 1) * n / (ms1f * 1e-3) / 8.0e12},
        "parity_sample": {"rows": m, "class_mismatches": mism1f},
    },
    "single_fused_f16x3": {
        "what": "one bit position, f16x3 (128-column tile, 3 of its 4 position slots empty)", "value": n / (ms1h * 1e-3),
        "unit": "rows/s", "ms_per_step": ms1h, "decided_bits_differing_from_fp32": flips1h,
        "roofline": {"bound": "hbm", "achieved": (4 * cfg.n_input + 1) * n / (ms1h * 1e-3) / 1e9, "peak": 8000.0,
                     "unit": "GB/s", "frac": (4 * cfg.n_input + 1) * n / (ms1h * 1e-3) / 8.0e12},
    },
    "bank": {
        "what": f"{P} positions in one fused launch (decoder.BitwiseBank)", "positions": P,
        "value": n / (msP * 1e-3), "unit": "rows/s (all positions decoded)", "ms_per_step": msP,
        "decided_bits_per_s": n * P / (msP * 1e-3),
        "roofline": {"bound": "mfma", "achieved": flop_row * n / (msP * 1e-3) / 1e12, "peak": 157.3,
                     "unit": "TFLOP/s", "frac": flop_row * n / (msP * 1e-3) / 157.3e12,
                     "algorithmic_flop_per_row": flop_row, "algorithmic_bytes_per_row": bytes_row,
                     "hbm_frac_of_8TBs": bytes_row * n / (msP * 1e-3) / 8.0e12},
        "kernel": "K4c bank_stream_kernel (32 rows resident in LDS, every position's first layer streamed from L2)",
        "tiled_gemm_K4b": {"ms_per_step": msPt, "frac": flop_row * n / (msPt * 1e-3) / 157.3e12,
                           "decided_bits_identical": tiled_same},
        "speedup_vs_one_launch_per_position": ms1 * P / msP,
        "parity_sample": {"rows": m, "positions_checked": 3, "class_mismatches": mismP},
    },
    "bank_f16x3": {
        "what": f"the same {P} positions with the stacked first layers as hi/lo f16 planes (BitwiseBank(precision='f16x3'))",
        "value": n / (msP16 * 1e-3), "unit": "rows/s (all positions decoded)", "ms_per_step": msP16,
        "speedup_vs_fp32_bank": msP / msP16, "decided_bits_differing_from_the_fp32_bank": flips16,
        "decided_bits": n * P,
    },
    "cpu_baseline": {"value": m / dt, "unit": "rows/s (one position)", "cores": O.c_max_threads(), "kind": "port",
                     "sample": f"first {m} rows, oracle/fnn_oracle.c"},
}))
