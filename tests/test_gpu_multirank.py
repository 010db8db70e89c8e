"""The N > 1 path on real hardware, launched the way the driver launches it: the bare command
`python bench.py --gpus N` (bench.py starts its own ranks as children before touching the GPU).

* `test_bench_two_ranks_rccl`: one rank per GPU over RCCL (torch backend "nccl"); needs >= 2 GPUs and
  is skipped on the 1-GPU box (the RCCL branch of distributed.init_from_env / barrier /
  all_reduce_counters runs here and nowhere else).
* `test_bench_two_ranks_one_gpu_rehearsal`: the same 2-rank code path with the counters reduced over
  gloo, both ranks on GPU 0 (NNV_DIST_BACKEND=gloo) - runs on the 1-GPU box.
* `test_bench_four_ranks_one_gpu_rehearsal`: 4 ranks over gloo on GPU 0 (uneven shard of the configs[3] leg).
All check `ranks_seen`, the per-rank device indices, and that the error counters of the N-rank
run equal those of ONE rank decoding the same global blocks (the generator is keyed by the global
row, so the data cannot depend on the rank count) -- for the headline workload AND for the
BASELINE configs[3] leg (block_len 1024, mixed SNR, per-SNR-bin counters through the one all-reduce):
the [7, 4] counters of the N-rank run are the single-rank run's integers, bit for bit."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCKS = 60_000            # per GPU; 480 k rows
CONFIG3_BLOCKS = 7 * 301 + 5   # configs[3] leg: 2,112 blocks of 128 rows, not divisible by 2, 4 or 7 (ragged shards and bins)


def _bench(n_gpus, blocks_per_gpu, extra_env=None):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["NNV_BENCH_SMALL_LEGS"] = "1"           # the one-rank run's other legs at reduced size (not what is tested here)
    env["NNV_BENCH_LIVE_TRAFFIC"] = "0"
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n_gpus),
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--blocks-per-gpu", str(blocks_per_gpu), "--config3-blocks", str(CONFIG3_BLOCKS)],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # ONE JSON line, from rank 0
    return json.loads(lines[0])


def _check_config3(multi, one, world):
    """BASELINE configs[3] through `bench.py --gpus N`: the same total work for every N (strong scaling), contiguous
    block shards, per-SNR-bin counters reduced by the one all-reduce == the single-rank run's integers."""
    m, o = multi["other_baseline_configs"]["configs[3]"], one["other_baseline_configs"]["configs[3]"]
    assert "error" not in m and "error" not in o, (m, o)
    assert m["n_gpus"] == world and o["n_gpus"] == 1 and m["scaling"] == "strong"
    assert m["blocks_total"] == o["blocks_total"] == CONFIG3_BLOCKS
    assert sum(m["blocks_per_rank"]) == CONFIG3_BLOCKS and len(m["blocks_per_rank"]) == world
    assert max(m["blocks_per_rank"]) - min(m["blocks_per_rank"]) <= 1
    assert len(m["seconds_per_rank"]) == world and m["value"] > 0
    assert m["counters"] == o["counters"]                       # [7][4] int64, bit for bit
    assert m["ber_by_snr_db"] == o["ber_by_snr_db"] and m["fer_by_snr_db"] == o["fer_by_snr_db"]
    c = o["counters"]
    assert sum(row[3] for row in c) == CONFIG3_BLOCKS * 128 and all(row[2] == 8 * row[3] for row in c)
    assert [row[3] for row in c] == [128 * len(range(d, CONFIG3_BLOCKS, 7)) for d in range(7)]   # block i -> bin i % 7
    ber = [o["ber_by_snr_db"][str(d)] for d in range(7)]
    assert all(ber[d] > ber[d + 1] for d in range(6))           # the trained network: BER falls with the SNR
    # rank 0's own share is a strict part of the whole
    assert sum(r[3] for r in m["counters_rank0_before_all_reduce"]) == m["blocks_per_rank"][0] * 128


def _check_pair(two, one, world=2):
    assert two["n_gpus"] == world and two["ranks_seen"] == world
    assert two["scaling"] == "weak" and two["config"]["blocks_per_gpu"] == BLOCKS
    # 2 ranks x K steps over blocks [0, 2B) == 1 rank x K steps over the same blocks
    assert two["ber"] == one["ber"] and two["fer"] == one["fer"]
    assert two["block_error_rate"] == one["block_error_rate"]
    assert two["value"] > 0 and two["roofline"]["frac"] > 0
    _check_config3(two, one, world)


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
def test_bench_four_ranks_one_gpu_rehearsal(cuda_device):
    four = _bench(4, BLOCKS, {"NNV_DIST_BACKEND": "gloo"})
    one = _bench(1, 4 * BLOCKS)
    assert four["dist_backend"] == "gloo" and four["rank_devices"] == [0, 0, 0, 0]
    _check_pair(four, one, world=4)


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_rccl_failure_falls_back_to_gloo_and_says_so(cuda_device):
    """A first multi-GPU run is where RCCL can fail to come up (IPC between processes on this pool).  The data path has
    no collective -- RCCL carries 28 integers after the timed region -- so bench.py then reduces them over gloo and SAYS
    SO in the record (`dist_backend`, `rccl_error`) instead of losing the scaling measurement.  Executed for real on
    one GPU: two ranks ask RCCL for the SAME device (test hook), which RCCL refuses or times out on; the run must still
    produce one JSON line whose counters equal a single rank's.  NNV_RCCL_FALLBACK=0 makes the same failure exit 7."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs the one-GPU box: with two GPUs RCCL comes up and there is nothing to fall back from")
    hook = {"NNV_RCCL_SHARED_GPU_TEST": "1", "NNV_DIST_TIMEOUT_S": "40"}
    two = _bench(2, BLOCKS, hook)
    one = _bench(1, 2 * BLOCKS)
    assert "FALLBACK" in two["dist_backend"] and two["rccl_error"], two["dist_backend"]
    assert two["rank_devices"] == [0, 0]
    _check_pair(two, one)
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(hook, NNV_RCCL_FALLBACK="0", NNV_BENCH_LIVE_TRAFFIC="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--blocks-per-gpu", "1000", "--config3-blocks", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "process group (nccl) failed" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())      # (RCCL prints its version banner there) no JSON line


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
