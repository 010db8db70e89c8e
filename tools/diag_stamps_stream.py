#!/usr/bin/env python3
"""Diagnostic only: per-phase s_memtime shares of (block 0, wave 0) of the K1 STREAM variant on BASELINE
configs[1] (20-128-64-1024) from a -DNNV_DIAG build.
Usage: NNV_LIB=.../ab_diag.so python tools/diag_stamps_stream.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd
from nnviterbi_amd import _cabi
from nnviterbi_amd.decoder import PackedMLP, generate_awgn_blocks
rng = np.random.RandomState(0)
dims = [20, 128, 64, 1024]
Ws = [(rng.randn(dims[i], dims[i + 1]) * 0.1).astype(np.float32) for i in range(3)]
bs = [np.zeros(d, np.float32) for d in dims[1:]]
net = PackedMLP(Ws, bs, precision=os.environ.get('NNV_PREC', 'f32'))
n = 10_000_000
x, lab = generate_awgn_blocks("57", 10, n, 0.45, 1)
cls = torch.empty(n, dtype=torch.int32, device="cuda"); cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
net.decode_timed(x, lab, cls, cnt, 1, tile_bits=10)
ms = net.decode_timed(x, lab, cls, cnt, 1, tile_bits=10)
lib = _cabi.load_library()
buf = (ctypes.c_ulonglong * 16)()
assert lib.nnv_debug_read_stamps(buf) == 0
v = [int(t) for t in buf]
tiles = v[8]
names = ["x wait/top", "layer1", "layer2 act", "layer3", "class: resident + slot compute", "epilogue", "wait vmcnt(0)", "s_barrier"]
info = net.launch_info(n)
print(f"threads {info['threads']} lds_tiles {info['class_tiles_in_lds']} {ms:.3f} ms for {n} rows; row tiles by wave0: {tiles}")
tot = sum(v[:8])
for nm, c in zip(names, v[:8]):
    print(f"  {nm:32s} {c / max(tiles,1):9.0f} cyc/tile  ({100 * c / tot:5.1f} %)")
print(f"  total {tot / tiles:.0f} cyc/tile/wave")
