#!/usr/bin/env python3
"""PCIe-inclusive rate of the reference-shaped call (host test_set in, host predictions out):
PackedMLP.decode_host = the C ABI's nnv_mlp_decode_host_f32 (pinned ring + staging threads).
One JSON line.  Usage: python tools/host_path_bench.py [blocks]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd
from nnviterbi_amd import fnn_viterbi as fv, fnn_viterbi_mixed_snr as mx
from nnviterbi_amd.decoder import PackedMLP, generate_awgn_blocks
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
n = blocks * 8
w, b = mx.load_model(os.path.join(os.path.dirname(nnviterbi_amd.__path__[0]), "neural-network-based-viterbi-decoding_amd", "weights", "fnn_133171_b8_2dB.npz"))
net = PackedMLP(*fv.weights_as_lists(w, b))
xd, ld = generate_awgn_blocks("133171", 8, n, 0.5616748812614791, 1, rows_per_block=8)
x, lab = xd.cpu().numpy(), ld.cpu().numpy()
ref = net.decode(xd, labels=ld, tile_bits=8)
want_cls, want_cnt = ref["cls32"].cpu().numpy().astype(np.int64), ref["counters"].cpu().numpy()
del xd, ld
rows = []
for threads in (0, 4, 16):
    for chunk in (0, 1 << 19, 1 << 20, 1 << 21):
        net.decode_host(x[: 1 << 22], lab[: 1 << 22], 8, chunk_rows=chunk, threads=threads)     # ring allocated, pages touched
        best = None
        for _ in range(2):
            t0 = time.perf_counter(); cls, cnt = net.decode_host(x, lab, 8, chunk_rows=chunk, threads=threads); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        ok = bool(np.array_equal(cls, want_cls) and np.array_equal(cnt, want_cnt))
        rows.append({"threads": threads, "chunk_rows": chunk, "ms": best * 1e3, "M_blocks_per_s": blocks / best / 1e6,
                     "GB_per_s_moved": n * 68 / best / 1e9, "bit_exact": ok})
        print(f"threads {threads:2d} chunk {chunk:>8d}: {best*1e3:8.1f} ms  {blocks/best/1e6:7.1f} M blocks/s  "
              f"{n*68/best/1e9:6.1f} GB/s moved  match={ok}", file=sys.stderr)
# small test sets (the reference's own sizes: 2 k rows = BASELINE configs[0]; 64 k rows): latency of ONE host call
small = {}
for m in (2_000, 65_536):
    net.decode_host(x[:m], lab[:m], 8)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); net.decode_host(x[:m], lab[:m], 8); ts.append(time.perf_counter() - t0)
    small[str(m)] = {"rows": m, "median_us_per_call": float(np.median(ts) * 1e6), "min_us_per_call": float(min(ts) * 1e6)}
    print(f"host call over {m} rows: median {np.median(ts)*1e6:.0f} us, min {min(ts)*1e6:.0f} us", file=sys.stderr)
t0 = time.perf_counter(); t = torch.from_numpy(x).cuda(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
px = torch.from_numpy(x[: 1 << 24]).pin_memory()
t2 = torch.empty_like(px, device="cuda"); t2.copy_(px, non_blocking=True); torch.cuda.synchronize()     # warm: allocation, first touch
t0 = time.perf_counter(); t2.copy_(px, non_blocking=True); torch.cuda.synchronize(); dtp = time.perf_counter() - t0
best = max(rows, key=lambda r: r["M_blocks_per_s"])
print(json.dumps({"metric": "host-array decode, PCIe inclusive (BASELINE configs[2] rows, host in / host out)",
                  "blocks": blocks, "best": best, "all": rows, "small_batch_latency": small,
                  "plain_pageable_h2d_GBps": x.nbytes / dt / 1e9, "pinned_h2d_GBps": px.numel() * 4 / dtp / 1e9,
                  "link_bound_M_blocks_per_s": px.numel() * 4 / dtp / 544 / 1e6}))
