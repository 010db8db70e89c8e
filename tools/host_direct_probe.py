#!/usr/bin/env python3
"""Probe for the host-array path (round 4): instead of staging pageable rows through a pinned ring (46-48 GB/s against
56 GB/s for a bare H2D), hand the caller's PAGEABLE rows to the runtime in LARGE chunks (one plain H2D of the whole 5 GB
matrix runs at link speed on this box: the runtime page-locks big ranges in place) and overlap the decode of chunk i
with the copy of chunk i + 1.  Round 2 tried this with 64 MB chunks (61-71 M blocks/s: pin/unpin per chunk).
Prints one JSON line: total ms and M blocks/s per chunk size, classes checked against the resident decode."""
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
x = xd.cpu().numpy()
want = net.decode(xd, tile_bits=8)["cls32"].cpu().numpy()
del xd, ld
out = {"blocks": blocks, "runs": []}
copy_s, cmp_s, out_s = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
for chunk_mb in (128, 256, 512, 1024, 2560, 5120):
    m = min(n, (chunk_mb << 20) // 64)
    nch = (n + m - 1) // m
    dx = [torch.empty((m, 16), dtype=torch.float32, device="cuda") for _ in range(2)]
    dc = [torch.empty(m, dtype=torch.int32, device="cuda") for _ in range(2)]
    cls_host = torch.empty(n, dtype=torch.int32).pin_memory()
    ev_in = [torch.cuda.Event() for _ in range(2)]
    ev_cmp = [torch.cuda.Event() for _ in range(2)]
    ev_out = [torch.cuda.Event() for _ in range(2)]
    best = None
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nch):
            s = i & 1
            lo, cnt = i * m, min(m, n - i * m)
            if i >= 2:
                ev_out[s].synchronize()          # slot s: its previous classes have left the device
            with torch.cuda.stream(copy_s):
                dx[s][:cnt].copy_(torch.from_numpy(x[lo:lo + cnt]), non_blocking=True)   # pageable source
                ev_in[s].record(copy_s)
            with torch.cuda.stream(cmp_s):
                cmp_s.wait_event(ev_in[s])
                r = net.decode(dx[s][:cnt], tile_bits=8)
                dc[s][:cnt].copy_(r["cls32"])
                ev_cmp[s].record(cmp_s)
            with torch.cuda.stream(out_s):
                out_s.wait_event(ev_cmp[s])
                cls_host[lo:lo + cnt].copy_(dc[s][:cnt], non_blocking=True)
                ev_out[s].record(out_s)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    ok = bool(np.array_equal(cls_host.numpy(), want))
    out["runs"].append({"chunk_MB": chunk_mb, "chunks": nch, "ms": best * 1e3, "M_blocks_per_s": blocks / best / 1e6,
                        "GB_per_s_moved": n * 68 / best / 1e9, "classes_match": ok})
    print(out["runs"][-1], file=sys.stderr)
    del dx, dc, cls_host
t0 = time.perf_counter(); t = torch.from_numpy(x).cuda(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["plain_pageable_h2d_GBps_whole_matrix"] = x.nbytes / dt / 1e9
print(json.dumps(out))
