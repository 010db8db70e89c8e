#!/usr/bin/env python3
"""BASELINE configs[3] at reduced size on the visible GPUs (secondary; bench.py is the headline):
(133,171) block_len=1024 = 128 tiles of b=8 per block, block i at SNR (i mod 7) dB, blocks sharded
over ranks, per-SNR-bin counters reduced with one all-reduce.  Full size is 100 M blocks over 8 GPUs
(12.5 M blocks = 1.6 G rows per GPU); default here 400 k blocks per GPU, generated and decoded in
chunks.  Prints one JSON line on rank 0.
  python tools/bench_config4.py [--blocks-per-gpu N] [--precision f32|f16x3]
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_config4.py"""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nnviterbi_amd
from nnviterbi_amd import distributed as nd, fnn_viterbi as fv, fnn_viterbi_mixed_snr as mx
from nnviterbi_amd.decoder import PackedMLP

ap = argparse.ArgumentParser()
ap.add_argument("--blocks-per-gpu", type=int, default=400_000)
ap.add_argument("--precision", default="f32")
ap.add_argument("--chunk-rows", type=int, default=1 << 25)
args = ap.parse_args()
idx = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()   # ONE device index per rank
torch.cuda.set_device(idx)
dev = torch.device("cuda", idx)
rank, local_rank, world = nd.init_from_env(device_index=idx)
w, b = mx.load_model(os.path.join(ROOT, "neural-network-based-viterbi-decoding_amd", "weights", "fnn_133171_b8_2dB.npz"))
net = PackedMLP(*fv.weights_as_lists(w, b), device=dev, precision=args.precision)
sig = [float(fv.DecoderConfig(snr_db=float(d)).sigma) for d in range(7)]
total_blocks = args.blocks_per_gpu * world
nd.decode_blocks_sharded(net, "133171", 8, 7 * 64, 128, sig, 1, rank, world, chunk_rows=1 << 20)   # warm-up
torch.cuda.synchronize(dev)
nd.barrier(dev)
t0 = time.perf_counter()
counters = nd.decode_blocks_sharded(net, "133171", 8, total_blocks, 128, sig, 2024, rank, world,
                                    chunk_rows=args.chunk_rows)
torch.cuda.synchronize(dev)
nd.barrier(dev)
dt = nd.max_over_ranks(time.perf_counter() - t0, device=dev)
nd.all_reduce_counters(counters)
if rank == 0:
    c = counters.cpu().numpy()
    print(json.dumps({
        "metric": "decoded blocks/sec incl. on-device generation, (133,171) block_len=1024, SNR sweep 0-6 dB",
        "value": total_blocks / dt, "unit": "blocks/s", "n_gpus": world, "blocks": total_blocks,
        "rows": total_blocks * 128, "seconds": dt, "dtype": args.precision,
        "weights": "trained at 2 dB (tools/train_config3.py)",
        "ber_by_snr_db": {str(d): float(c[d, 0] / c[d, 2]) for d in range(7)},
        "fer_by_snr_db": {str(d): float(c[d, 1] / c[d, 3]) for d in range(7)},
        "blocks_by_snr_db": {str(d): int(c[d, 3] // 128) for d in range(7)},
    }))
nd.barrier(dev)
if world > 1:
    torch.distributed.destroy_process_group()
