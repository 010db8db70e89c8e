#!/usr/bin/env python3
# NOTE (round 5): NNV_THREADS / -DNNV_DIAG is read only by libraries built with tools/build_variant.sh (-DNNV_AB_BUILD);
# point NNV_LIB at such a build -- the shipped libnnviterbi_hip.so reads no environment variable.
"""Diagnostic only: per-phase s_memtime shares of (block 0, wave 0) from a -DNNV_DIAG build.
Usage: NNV_LIB=.../ab_DIAG.so NNV_THREADS=256|512|1024 python tools/diag_stamps.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd
from nnviterbi_amd import fnn_viterbi as fv, _cabi
from nnviterbi_amd.decoder import PackedMLP, generate_awgn_blocks
cfg = fv.DecoderConfig(block_length=8)
w, b = fv.initialize_weights(cfg, rng=np.random.RandomState(0))
net = PackedMLP(*fv.weights_as_lists(w, b), precision=os.environ.get('NNV_PREC', 'f32'))
n = 16_000_000
x, lab = generate_awgn_blocks("133171", 8, n, 0.56, 1)
cls = torch.empty(n, dtype=torch.int32, device="cuda"); cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
net.decode_timed(x, lab, cls, cnt, 1, tile_bits=8)
ms = net.decode_timed(x, lab, cls, cnt, 3, tile_bits=8) / 3
lib = _cabi.load_library()
buf = (ctypes.c_ulonglong * 16)()
assert lib.nnv_debug_read_stamps(buf) == 0
v = [int(t) for t in buf]
tiles = v[8]
names = ["x wait/top", "layer1 (32 MFMA)", "layer2 (128)", "layer3 (32)", "class layer (128)", "epilogue"]
info = net.launch_info(n)
wps = info["threads"] // 256
print(f"threads {info['threads']} ({wps} waves/SIMD)  {ms:.3f} ms for {n} rows  tiles by wave0: {tiles}")
tot = sum(v[:6])
mf = [0, 32, 128, 32, 128, 0] if os.environ.get('NNV_PREC', 'f32') == 'f32' else [0, 6, 24, 6, 24, 0]
for nm, c, m in zip(names, v[:6], mf):
    print(f"  {nm:22s} {c / tiles:9.0f} cyc/tile  ({100 * c / tot:5.1f} %)   ideal MFMA-only x waves: {m * 64 * wps}")
print(f"  total {tot / tiles:.0f} cyc/tile/wave  -> per SIMD per tile {tot / tiles / wps:.0f}  (MFMA floor 20480)")
