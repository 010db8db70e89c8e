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
