"""Block sharding over GPUs and the one collective of this path.

The reference has no distributed code (single process).
Rows of the decoder input are independent (src/fnn_viterbi.py:494-510 has no
cross-row op), so codeword blocks shard embarrassingly: each rank owns a
contiguous range of GLOBAL block indices, generates/reads exactly those blocks
(the on-device generator is keyed by the global row index, so the data do not
depend on the number of ranks), and the only exchange is one SUM all-reduce of
the error counters ``{bit_errors, frame_errors, total_bits, total_frames}``
(x n_snr bins for a mixed-SNR sweep): tens of bytes over RCCL/xGMI.
Decoded bits stay on the GPU that produced them.
"""
from __future__ import annotations

import datetime
import os
from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def block_shard(n_blocks: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous split, remainder to the low ranks -> (first_block, count)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(int(n_blocks), int(world_size))
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def group_timeout_s() -> float:
    """Time limit of the rendezvous and of every collective (NNV_DIST_TIMEOUT_S, default 120 s): a stuck
    RCCL bootstrap must end as an error with a message, not as a kill at some outer limit (PyTorch's own
    default is 10 minutes)."""
    return float(os.environ.get("NNV_DIST_TIMEOUT_S", "120"))


def validate_group(ranks_seen: int, rank_devices: Sequence[int], world_size: int,
                   require_distinct_devices: bool) -> None:
    """The collective must have spanned exactly ``world_size`` ranks, and under RCCL every rank must sit on
    its own GPU.  Raises RuntimeError (with both lists) otherwise: a benchmark must not print a value
    for N GPUs that its one collective did not span."""
    devs = [int(d) for d in rank_devices]
    if int(ranks_seen) != int(world_size) or len(devs) != int(world_size):
        raise RuntimeError(f"process group spans {ranks_seen} rank(s) (devices {devs}), expected {world_size}")
    if require_distinct_devices and len(set(devs)) != int(world_size):
        raise RuntimeError(f"{world_size} ranks share GPUs: rank_devices {devs} (one GPU per rank under RCCL)")


def init_from_env(backend: Optional[str] = None, device_index: Optional[int] = None,
                  force: bool = False) -> Tuple[int, int, int]:
    """Join the process group described by torchrun's environment.

    Returns (rank, local_rank, world_size); a no-op for a single process.
    backend: "nccl" (= RCCL on ROCm) when CUDA/HIP is available, else "gloo".
    device_index: the ONE HIP device this rank uses (default LOCAL_RANK); the RCCL group is bound
    to exactly that device, so the caller's allocations and the collective cannot disagree.
    force: create the group even for WORLD_SIZE == 1 (a one-rank RCCL group: every collective of
    this module then really goes through RCCL - the only way to execute that branch on a one-GPU box).
    """
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            # NNV_DIST_BACKEND=gloo: rehearse the multi-rank path with several ranks on ONE GPU
            backend = os.environ.get("NNV_DIST_BACKEND") or \
                ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            idx = local_rank if device_index is None else int(device_index)
            if idx >= torch.cuda.device_count():
                raise RuntimeError(f"rank {rank}: device {idx} requested, "
                                   f"{torch.cuda.device_count()} visible (RCCL needs one GPU per rank)")
            torch.cuda.set_device(idx)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", idx),
                                    timeout=datetime.timedelta(seconds=group_timeout_s()))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=group_timeout_s()))
    return rank, local_rank, world


def reset_group() -> None:
    """Leave whatever process group exists (or half exists after a failed initialisation), so that init_from_env can
    build another one -- bench.py's fallback from RCCL to gloo for the counters."""
    try:
        if dist.is_initialized():
            dist.destroy_process_group()
    except Exception:   # noqa: BLE001  (a communicator that never came up may refuse to be torn down; nothing to keep)
        pass


def gather_ints(value: int, device=None, group=None) -> list:
    """[value of rank 0, value of rank 1, ...] on every rank (all_gather of one int64)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [int(value)]
    on_host = device is None or dist.get_backend(group) == "gloo"
    t = torch.tensor([int(value)], dtype=torch.int64, device="cpu" if on_host else device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t, group=group)
    return [int(o.item()) for o in out]


def gather_floats(value: float, device=None, group=None) -> list:
    """[value of rank 0, value of rank 1, ...] on every rank (all_gather of one float64)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    on_host = device is None or dist.get_backend(group) == "gloo"
    t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if on_host else device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t, group=group)
    return [float(o.item()) for o in out]


def all_reduce_counters(counters: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM of an int64 counter tensor over all ranks (any shape)."""
    if counters.dtype != torch.int64:
        raise ValueError("counters must be int64")
    if dist.is_available() and dist.is_initialized():
        if counters.is_cuda and dist.get_backend(group) == "gloo":
            host = counters.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            counters.copy_(host)
        else:
            dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=group)
    return counters


def max_over_ranks(value: float, device=None, group=None) -> float:
    """MAX all-reduce of a python float (timing)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    on_host = device is None or dist.get_backend(group) == "gloo"
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if on_host else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def barrier(device=None):
    if dist.is_available() and dist.is_initialized():
        if device is not None and torch.device(device).type == "cuda" and dist.get_backend() != "gloo":
            dist.barrier(device_ids=[torch.device(device).index])
        else:
            dist.barrier()


def rates(counters) -> Tuple[float, float]:
    """(BER, FER) from {bit_errors, frame_errors, total_bits, total_frames}
    (src/fnn_viterbi.py:640,671)."""
    c = [int(v) for v in counters]
    ber = c[0] / c[2] if c[2] else float("nan")
    fer = c[1] / c[3] if c[3] else float("nan")
    return ber, fer


def decode_blocks_sharded(net, code: str, tile_bits: int, n_blocks_total: int, rows_per_block: int,
                          sigmas, seed: int, rank: int = 0, world_size: int = 1,
                          chunk_rows: int = 1 << 26, device=None, continuous: bool = False) -> torch.Tensor:
    """BASELINE config 4 driver: this rank's contiguous share of ``n_blocks_total`` synthetic
    blocks (block i at SNR bin i % len(sigmas)) is generated on the device and decoded chunk by
    chunk (at most ``chunk_rows`` rows resident at a time: 102 GB of input per GPU at full size
    would fit, but need not be resident).  Returns this rank's counters int64 [n_bins, 4];
    combine ranks with :func:`all_reduce_counters`.  ``continuous``: every block is ONE stream encoded from zero
    state and chopped into rows (the reference's "random test set", src/fnn_viterbi_mixed_snr.py:464-475) instead
    of rows that each start from zero state -- chunk and shard boundaries change nothing (a row's encoder memory is
    regenerated from the previous row's counter)."""
    from .decoder import generate_awgn_blocks
    import numpy as np
    sig = np.atleast_1d(np.asarray(sigmas, dtype=np.float32))
    n_bins = int(sig.size)
    first_block, n_blocks = block_shard(n_blocks_total, rank, world_size)
    first_row = first_block * rows_per_block
    n_rows = n_blocks * rows_per_block
    dev = net.device if device is None else torch.device(device)
    counters = torch.zeros((n_bins, 4), dtype=torch.int64, device=dev)
    done = 0
    while done < n_rows:
        m = min(int(chunk_rows), n_rows - done)
        x, labels = generate_awgn_blocks(code, tile_bits, m, sig, seed, first_row=first_row + done,
                                         rows_per_block=rows_per_block, device=dev, continuous=continuous)
        net.decode_binned(x, labels, n_bins, rows_per_block, first_row=first_row + done,
                          counters=counters, tile_bits=tile_bits)
        done += m
    return counters
