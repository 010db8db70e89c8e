#!/usr/bin/env python3
"""Feed-forward decoders whose hidden widths the register-chained kernel (K1) has no instantiation for
(`DecoderConfig(n_hidden_1=..., ...)` are free parameters of the reference, src/fnn_viterbi.py:35-37): the whole
chain in one persistent launch (K6) against the layer-at-a-time path (K4, WideMLP.set_path).
One JSON line per shape.  Usage: python tools/bench_wide_shapes.py [rows]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd
from nnviterbi_amd import fnn_viterbi as fv
from nnviterbi_amd.decoder import WideMLP

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
for dims in ([14, 100, 100, 100, 128], [14, 200, 96, 48, 128], [16, 256, 128, 64, 256], [16, 512, 256, 128, 256], [18, 512, 512, 512, 512], [200, 500, 500, 500, 500, 2]):
    rng = np.random.RandomState(3)
    Ws = [fv.truncated_normal((dims[i], dims[i + 1]), np.sqrt(2.0 / dims[i]), rng) for i in range(len(dims) - 1)]
    bs = [(rng.randn(d) * 0.1).astype(np.float32) for d in dims[1:]]
    rows = n if dims[0] < 100 else n // 4
    x = torch.randn(rows, dims[0], device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    net = WideMLP(Ws, bs)
    flop = 2 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
    out = {"net": dims, "rows": rows, "flop_per_row": flop}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, path in (("fused_K6", "auto"), ("layered_K4", "layered")):
        net.set_path(path)
        r = net.decode(x)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            r = net.decode(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out[name] = {"ms": ms, "rows_per_s": rows / (ms * 1e-3), "frac_of_fp32_mfma_peak": flop * rows / (ms * 1e-3) / 1e12 / 157.3,
                     "cls_checksum": int(r["cls32"].to(torch.int64).sum().item())}
    net.close()
    print(json.dumps(out))
