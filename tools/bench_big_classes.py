#!/usr/bin/env python3
"""Feed-forward decoders at block_length 13-16 (2^b classes: the sizes the reference's DecoderConfig tests touch,
tests/test_utilities.py:342-348): K1 (round 5: the class layer and its biases streamed from L2 per wave), the
layer-at-a-time path with the class layer's argmax fused into its GEMM epilogue (round 5), and the same path storing the
activated [rows, 2^b] matrix and scanning it (`want_act=True`).
One JSON line per shape.  Usage: python tools/bench_big_classes.py [rows]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd  # noqa: F401
from nnviterbi_amd import fnn_viterbi as fv
from nnviterbi_amd.decoder import PackedMLP, WideMLP

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for b in (13, 15, 16):
    cfg = fv.DecoderConfig(block_length=b)
    Ws, bs = fv.weights_as_lists(*fv.initialize_weights(cfg, rng=np.random.RandomState(b)))
    net = WideMLP(Ws, bs)
    x = torch.randn((rows, 2 * b), device="cuda")
    flop = 2 * sum(Ws[i].shape[0] * Ws[i].shape[1] for i in range(len(Ws)))
    out = {"dims": [2 * b] + [w.shape[1] for w in Ws], "rows": rows, "workspace_MB": net._workspace.numel() / 2 ** 20}
    for name, kw in (("argmax_in_epilogue", {}), ("rows_stored_then_scanned", {"want_act": True})):
        n = rows if not kw else min(rows, 32768)          # the stored matrix is rows x 2^b floats
        xs = x[:n]
        r = net.decode(xs, **kw)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            r = net.decode(xs, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        out[name] = {"rows": n, "ms": ms, "rows_per_s": n / (ms * 1e-3), "frac_of_fp32_mfma_peak": flop * n / (ms * 1e-3) / 157.3e12,
                     "cls_checksum": int(r["cls32"][:32768].to(torch.int64).sum().item())}
        del r
        torch.cuda.empty_cache()
    net.close()
    for precision in ("f32", "f16x3"):
        k1 = PackedMLP(Ws, bs, precision=precision)
        r = k1.decode(x)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            r = k1.decode(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        out[f"K1_{precision}"] = {"rows": rows, "ms": ms, "rows_per_s": rows / (ms * 1e-3),
                                  "frac_of_fp32_mfma_peak": flop * rows / (ms * 1e-3) / 157.3e12, "launch": k1.launch_info(rows),
                                  "cls_checksum": int(r["cls32"][:32768].to(torch.int64).sum().item())}
        k1.close()
    print(json.dumps(out), flush=True)
