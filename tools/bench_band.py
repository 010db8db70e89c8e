#!/usr/bin/env python3
# NOTE (round 5): NNV_BAND_RESIDENT_512 is read only by libraries built with tools/build_variant.sh (-DNNV_AB_BUILD);
# point NNV_LIB at such a build -- the shipped libnnviterbi_hip.so reads no environment variable.
"""The 96-128 KB band of K1: a network that fits the LDS beside 512 threads' winner slots but not
beside 1024 threads' -- e.g. 18->128->64->32->512 (120 KB packed).  Default: class layer streamed at 1024 threads;
NNV_BAND_RESIDENT_512=1 (read once per process): the whole network resident at 512 threads instead.
Run once per setting in ONE gpurun call (box-to-box spread is ~1 %).  One JSON line."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd
from nnviterbi_amd import fnn_viterbi as fv
from nnviterbi_amd.decoder import PackedMLP

dims = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "18,128,64,32,512").split(",")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
steps = 10
rng = np.random.RandomState(9)
Ws = [fv.truncated_normal((dims[i], dims[i + 1]), np.sqrt(2.0 / dims[i]), rng) for i in range(len(dims) - 1)]
bs = [(rng.randn(d) * 0.1).astype(np.float32) for d in dims[1:]]
x = torch.randn(n, dims[0], device="cuda", generator=torch.Generator("cuda").manual_seed(1))
lab = torch.zeros(n, dtype=torch.int32, device="cuda")
flop = 2 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
out = {"net": dims, "rows": n, "NNV_BAND_RESIDENT_512": os.environ.get("NNV_BAND_RESIDENT_512")}
for prec in ("f32", "f16x3"):
    net = PackedMLP(Ws, bs, precision=prec)
    cls = torch.empty(n, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    tb = int(np.ceil(np.log2(dims[-1])))
    net.decode_timed(x, lab, cls, cnt, 3, tile_bits=tb)
    ms = net.decode_timed(x, lab, cls, cnt, steps, tile_bits=tb) / steps
    info = net.launch_info(n)
    out[prec] = {"ms_per_launch": ms, "tflops_algorithmic": flop * n / (ms * 1e-3) / 1e12,
                 "frac_of_fp32_mfma_peak": flop * n / (ms * 1e-3) / 1e12 / 157.3 if prec == "f32" else None,
                 "threads": info["threads"], "lds_bytes": info["lds_bytes"], "out_layer_in_lds": info["out_layer_in_lds"],
                 "cls_checksum": int(cls.to(torch.int64).sum().item())}
    net.close()
print(json.dumps(out))
