"""The N>1 path on CPU: contiguous block sharding + the one collective (SUM of
error counters) over torch.distributed gloo, world_size 2, 4 and 8, plus
bench.py's own launcher (`python bench.py --gpus N` with no torchrun environment).  The per-rank
decode is done by the oracle here (this is a test of the sharding/reduction
logic; the HIP path itself is covered by the gpu tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import nnviterbi_amd  # noqa: F401
from nnviterbi_amd import distributed as nd

B, ROWS_PER_BLOCK, N_BLOCKS, SEED = 6, 3, 101, 99


def _net():
    from nnviterbi_amd import fnn_viterbi as fv
    cfg = fv.DecoderConfig(block_length=B)
    w, b = fv.initialize_weights(cfg, rng=np.random.RandomState(4))
    return fv.weights_as_lists(w, b)


def _counters_for(first_block, n_blocks):
    from oracle import awgn_model, fnn_oracle as O
    Ws, bs = _net()
    x, labels = awgn_model.generate("133171", B, n_blocks * ROWS_PER_BLOCK, 0.6, SEED,
                                    first_row=first_block * ROWS_PER_BLOCK,
                                    rows_per_block=ROWS_PER_BLOCK)
    cls, _, _ = O.decode(x.astype(np.float32), Ws, bs, B)
    return O.error_counters(cls, labels, B)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, lr, w = nd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    first, count = nd.block_shard(N_BLOCKS, rank, world)
    c = torch.from_numpy(_counters_for(first, count))
    nd.barrier()
    nd.all_reduce_counters(c)
    t = nd.max_over_ranks(1.0 + rank)
    ones = torch.ones(1, dtype=torch.int64)
    nd.all_reduce_counters(ones)                     # bench.py's `ranks_seen`
    seen = nd.gather_ints(100 + rank)                # bench.py's `rank_devices`
    assert seen == [100 + r for r in range(world)]
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.append(c.numpy(), [int(t), int(ones.item())]))
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_block_shard_covers_everything_once():
    for n in (0, 1, 7, 8, 101, 10_000_000):
        for world in (1, 2, 3, 8):
            spans = [nd.block_shard(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        nd.block_shard(10, 2, 2)


def test_single_process_helpers_are_noops():
    c = torch.tensor([1, 2, 3, 4], dtype=torch.int64)
    assert nd.all_reduce_counters(c).tolist() == [1, 2, 3, 4]
    assert nd.max_over_ranks(2.5) == 2.5
    assert nd.rates([5, 2, 100, 10]) == (0.05, 0.2)
    with pytest.raises(ValueError):
        nd.all_reduce_counters(torch.zeros(4, dtype=torch.int32))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_counter_reduction_equals_single_rank(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    full = _counters_for(0, N_BLOCKS)
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got[:4].tolist() == full.tolist()
        assert got[4] == world                      # MAX over ranks of (1 + rank)
        assert got[5] == world                      # SUM of ones: every rank took part
    assert full[3] == N_BLOCKS * ROWS_PER_BLOCK and full[2] == full[3] * B


def test_gather_ints_single_process():
    assert nd.gather_ints(7) == [7]


@pytest.mark.timeout(600)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment must start 2 ranks as children of a
    process that has not touched the GPU.  On this CPU box each rank then reports that it needs an
    MI355X and exits 3, and the launcher's return code is non-zero (no CPU fallback)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if torch.cuda.is_available():
        env["HIP_VISIBLE_DEVICES"] = ""             # on a GPU box: same no-device behaviour, cheaply
        env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0"], env=env, capture_output=True, text=True, timeout=500)
    assert "launching 2 ranks" in r.stderr
    assert "rank 0/2: needs an MI355X" in r.stderr and "rank 1/2: needs an MI355X" in r.stderr
    assert r.returncode != 0
    assert r.stdout.strip() == ""                   # no JSON line from a failed run
