#!/usr/bin/env python3
"""K3 (on-device generator) alone: ms per 80 M rows of b = 8 for the block shapes in use.  Round 5 used it to A/B shifts
instead of the three 64-bit divisions per row (block index, position in the block, SNR bin): 1.98 -> 1.90 ms and
1.92 -> 1.91 ms on one box, i.e. nothing (the compiler's 32-bit bypass already covers them) -- not kept.
Usage: python tools/bench_generator.py   (NNV_LIB=... for another build)"""
import sys, torch, time
sys.path.insert(0, "/root/repo")
import nnviterbi_amd
from nnviterbi_amd.decoder import generate_awgn_blocks
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, kw in (("configs[2] rows_per_block 8, one sigma", dict(rows_per_block=8, sigma=0.56)),
                 ("configs[3] rows_per_block 128, 7 sigmas", dict(rows_per_block=128, sigma=[0.7, 0.63, 0.56, 0.5, 0.45, 0.4, 0.35])),
                 ("rows_per_block 10 (not a power of two), 7 sigmas", dict(rows_per_block=10, sigma=[0.7, 0.63, 0.56, 0.5, 0.45, 0.4, 0.35])),
                 ("continuous, rows_per_block 128", dict(rows_per_block=128, sigma=0.56, continuous=True))):
    x, l = generate_awgn_blocks("133171", 8, 80_000_000, seed=1, **kw)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        generate_awgn_blocks("133171", 8, 80_000_000, seed=1, **kw)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 5:.3f} ms per 80 M rows, checksum {float(x[:1000].sum()):.6f} {int(l[:1000].sum())}")
    del x, l
