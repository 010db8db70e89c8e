#!/usr/bin/env python3
"""Triangulation of the (unpinned) forward pass: pairwise class-flip rates between every
implementation of  matmul -> add -> tanh x depth -> argmax  available here, on the four fixture
networks (tests/golden/forward_fp64.npz, random weights) and on the trained config-3 network
(weights/fnn_133171_b8_2dB.npz) fed with 2 dB AWGN blocks.

  fp64        numpy float64 restatement (the arbiter)
  numpy_f32   oracle/fnn_oracle.py forward, float32
  c_f32       oracle/fnn_oracle.c   (explicit fmaf chain, libm tanhf)
  torch_f32   oracle/torch_cpu.py   (PyTorch CPU GEMM / tanh / argmax) - the independent one
  torch_f64   the same in float64
  eigen_tanh_f32  numpy float32 matmul + TensorFlow-1.x's CPU tanh restated (oracle/eigen_tanh.py)
  tf_like_f32     sequential-k, no-FMA float32 matmul (Eigen gebp order) + that tanh, on the first 300 k rows
  hip_f32 / hip_f16x3   the product kernel (only with a GPU)
  hip_f16     the opt-in single-product 16-bit mode: NOT held to the 1e-5 rules, listed to show what it costs

For every implementation: max |out - fp64|, classes differing from fp64, and the largest fp64 top-2
margin among those rows (must be < 1e-5: the near-tie rule).  Then the pairwise flip counts.
PARITY STAYS UNPINNED: TensorFlow's own kernels are not among these.

Usage: python tools/flip_rates.py [--rows N] > profiles/r03/flip_rates.json
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import awgn_model, eigen_tanh as E, fnn_oracle as O, torch_cpu as T  # noqa: E402

import torch  # noqa: E402


def implementations(x, Ws, bs, b, gpu):
    res = {}
    c64, _, o64 = O.decode(x, Ws, bs, b, np.float64)
    res["fp64"] = (c64, o64)
    c, _, o = O.decode(x, Ws, bs, b, np.float32)
    res["numpy_f32"] = (c, o)
    r = O.decode_c(x, Ws, bs, b, want_act=True, want_bits=False)
    res["c_f32"] = (r["cls"], r["act"])
    res["torch_f32"] = T.decode(x, Ws, bs, torch.float32)
    res["torch_f64"] = T.decode(x, Ws, bs, torch.float64)
    res["eigen_tanh_f32"] = E.decode(x, Ws, bs)
    m = min(len(x), TF_LIKE_ROWS)                      # sequential-k matmul: O(K) numpy passes, on a prefix
    tcls, tout = E.decode(x[:m], Ws, bs, seq_k=True)
    res["tf_like_f32"] = (tcls, tout)
    if gpu:
        import nnviterbi_amd  # noqa: F401
        from nnviterbi_amd.decoder import PackedMLP
        xd = torch.from_numpy(x).cuda()
        for prec in ("f32", "f16x3", "f16"):   # "f16": the opt-in single-product mode, OUTSIDE the 1e-5 contract (shown for scale)
            net = PackedMLP(Ws, bs, precision=prec)
            cls = net.decode(xd, tile_bits=b)["cls32"].cpu().numpy().astype(np.int64)
            act = net.forward(xd).cpu().numpy()
            res["hip_" + prec] = (cls, act)
            net.close()
    return res


TF_LIKE_ROWS = 300_000


def table(x, Ws, bs, b, gpu):
    res = implementations(x, Ws, bs, b, gpu)
    c64, o64 = res["fp64"]
    margin = O.top2_margin(o64)
    names = list(res)
    per = {}
    for k in names[1:]:
        cls, out = res[k]
        m = len(cls)
        bad = cls != c64[:m]
        per[k] = {"rows": int(m), "max_abs_err_vs_fp64": float(np.abs(out - o64[:m]).max()),
                  "classes_differing_from_fp64": int(bad.sum()),
                  "max_fp64_margin_on_those_rows": float(margin[:m][bad].max()) if bad.any() else 0.0}
    pair = {}
    for i, a in enumerate(names):
        for c in names[i + 1:]:
            m = min(len(res[a][0]), len(res[c][0]))
            pair[f"{a}|{c}"] = int((res[a][0][:m] != res[c][0][:m]).sum())
    return {"rows": int(x.shape[0]), "per_implementation": per, "pairwise_class_flips": pair}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2_000_000, help="rows for the trained-network case")
    args = ap.parse_args()
    gpu = torch.cuda.is_available()
    out = {"gpu": gpu, "tolerance": "outputs within 1e-5 of fp64; class may differ only where the fp64 "
                                    "top-2 margin < 1e-5", "parity": "unpinned (no TensorFlow here)"}
    fg = np.load(os.path.join(ROOT, "tests", "golden", "forward_fp64.npz"), allow_pickle=False)
    for name, depth, b in (("b8_l3", 4, 8), ("b7_l3", 4, 7), ("b10_l2", 3, 10), ("b3_l3", 4, 3)):
        Ws = [fg[f"{name}_W{i}"] for i in range(depth)]
        bs = [fg[f"{name}_b{i}"] for i in range(depth)]
        out["fixture_" + name] = table(fg[f"{name}_x"], Ws, bs, b, gpu)
    ck = np.load(os.path.join(ROOT, "neural-network-based-viterbi-decoding_amd", "weights",
                              "fnn_133171_b8_2dB.npz"), allow_pickle=False)
    Ws = [ck[k] for k in ("weights_h1", "weights_h2", "weights_h3", "weights_out")]
    bs = [ck[k] for k in ("bias_h1", "bias_h2", "bias_h3", "bias_out")]
    x, _ = awgn_model.generate("133171", 8, args.rows, 0.5616748812614791, 20240607, rows_per_block=8)
    out["trained_fnn_133171_b8_2dB"] = table(x.astype(np.float32), Ws, bs, 8, gpu)
    # random weights at scale (the regime with many near-ties: untrained nets are not confident)
    rng = np.random.RandomState(5)
    dims = [16, 128, 64, 32, 256]
    Wr = [(rng.randn(dims[i], dims[i + 1]) * np.sqrt(2.0 / dims[i])).astype(np.float32) for i in range(4)]
    br = [np.zeros(dims[i + 1], np.float32) for i in range(4)]
    xr = np.random.RandomState(1).randn(args.rows, 16).astype(np.float32)
    out["random_16_128_64_32_256"] = table(xr, Wr, br, 8, gpu)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
