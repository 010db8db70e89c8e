"""The N > 1 path on real hardware, launched the way the driver launches it: the bare command
`python bench.py --gpus N` (bench.py starts its own ranks as children before touching the GPU).

* `test_bench_two_ranks_rccl`: one rank per GPU over RCCL (torch backend "nccl"); needs >= 2 GPUs and
  is skipped on the 1-GPU box (the RCCL branch of distributed.init_from_env / barrier /
  all_reduce_counters runs here and nowhere else).
* `test_bench_two_ranks_one_gpu_rehearsal`: the same 2-rank code path with the counters reduced over
  gloo, both ranks on GPU 0 (NNV_DIST_BACKEND=gloo) - runs on the 1-GPU box.
Both check `ranks_seen`, the per-rank device indices, and that the error counters of the 2-rank
run equal those of ONE rank decoding the same global blocks (the generator is keyed by the global
row, so the data cannot depend on the rank count)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCKS = 60_000            # per GPU; 480 k rows


def _bench(n_gpus, blocks_per_gpu, extra_env=None):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n_gpus),
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--blocks-per-gpu", str(blocks_per_gpu)],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # ONE JSON line, from rank 0
    return json.loads(lines[0])


def _check_pair(two, one):
    assert two["n_gpus"] == 2 and two["ranks_seen"] == 2
    assert two["scaling"] == "weak" and two["config"]["blocks_per_gpu"] == BLOCKS
    # 2 ranks x K steps over blocks [0, 2B) == 1 rank x K steps over the same blocks
    assert two["ber"] == one["ber"] and two["fer"] == one["fer"]
    assert two["block_error_rate"] == one["block_error_rate"]
    assert two["value"] > 0 and two["roofline"]["frac"] > 0


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_bench_two_ranks_one_gpu_rehearsal(cuda_device):
    two = _bench(2, BLOCKS, {"NNV_DIST_BACKEND": "gloo"})
    one = _bench(1, 2 * BLOCKS)
    assert two["dist_backend"] == "gloo" and two["rank_devices"] == [0, 0]
    assert one["ranks_seen"] == 1 and one["rank_devices"] == [0]
    _check_pair(two, one)


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_bench_two_ranks_rccl(cuda_device):
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL path needs >= 2 GPUs (the gpurun box has one)")
    two = _bench(2, BLOCKS)
    one = _bench(1, 2 * BLOCKS)
    assert two["dist_backend"].startswith("rccl") and two["rank_devices"] == [0, 1]
    _check_pair(two, one)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_rccl_branch_with_a_one_rank_group(cuda_device):
    """The RCCL code path of distributed.py (init_process_group('nccl', device_id=...), int64 SUM
    all_reduce of device counters, float64 MAX, all_gather, barrier(device_ids=...)) executed for real
    on the one GPU available: a ONE-rank RCCL group.  It cannot show scaling, but it runs every call
    the N-rank bench makes through RCCL itself (dtypes, ops, device binding).  Fresh interpreter."""
    code = r'''
import os, sys
sys.path.insert(0, %r)
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=%r)
import torch, torch.distributed as dist
import nnviterbi_amd
from nnviterbi_amd import distributed as nd
torch.cuda.set_device(0)
assert nd.init_from_env(backend="nccl", device_index=0, force=True) == (0, 0, 1)
assert dist.is_initialized() and dist.get_backend() == "nccl"
dev = torch.device("cuda", 0)
c = torch.tensor([[5, 2, 100, 10], [1, 1, 8, 1]], dtype=torch.int64, device=dev)
assert nd.all_reduce_counters(c).cpu().tolist() == [[5, 2, 100, 10], [1, 1, 8, 1]]
ones = torch.ones(1, dtype=torch.int64, device=dev); nd.all_reduce_counters(ones); assert int(ones.item()) == 1
assert nd.max_over_ranks(2.5, device=dev) == 2.5
assert nd.gather_ints(0, device=dev) == [0]
nd.barrier(dev)
torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl one-rank ok")
''' % (ROOT, str(29000 + os.getpid() % 2000))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0 and "rccl one-rank ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
