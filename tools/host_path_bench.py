#!/usr/bin/env python3
"""PCIe-inclusive rate of the reference-shaped call (host test_set in, host predictions out).
Usage: python tools/host_path_bench.py [blocks]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd
from nnviterbi_amd import fnn_viterbi as fv, fnn_viterbi_mixed_snr as mx
from nnviterbi_amd.decoder import PackedMLP, generate_awgn_blocks
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
n = blocks * 8
w, b = mx.load_model(os.path.join(os.path.dirname(nnviterbi_amd.__path__[0]), "neural-network-based-viterbi-decoding_amd", "weights", "fnn_133171_b8_2dB.npz"))
net = PackedMLP(*fv.weights_as_lists(w, b))
xd, ld = generate_awgn_blocks("133171", 8, n, 0.5616748812614791, 1, rows_per_block=8)
x, lab = xd.cpu().numpy(), ld.cpu().numpy()
ref = net.decode(xd, labels=ld, tile_bits=8)
for mode in (None, True, False):
    for chunk in (1 << 21, 1 << 23):
        net.decode_host(x[: 1 << 20], lab[: 1 << 20], 8, chunk_rows=chunk, register_host=mode)
        t0 = time.perf_counter(); cls, cnt = net.decode_host(x, lab, 8, chunk_rows=chunk, register_host=mode); dt = time.perf_counter() - t0
        ok = np.array_equal(cls, ref["cls32"].cpu().numpy().astype(np.int64)) and np.array_equal(cnt, ref["counters"].cpu().numpy())
        name = {None: "pageable", True: "registered", False: "pinned staging"}[mode]
        print(f"{name:15s} chunk {chunk:>8d} rows: {dt*1e3:8.1f} ms  {blocks/dt/1e6:7.1f} M blocks/s  {n*68/dt/1e9:6.1f} GB/s moved  match={ok}")
t0 = time.perf_counter(); t = torch.from_numpy(x).cuda(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"plain pageable H2D of x alone: {dt*1e3:.1f} ms = {x.nbytes/dt/1e9:.1f} GB/s")
