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


# ---- BASELINE configs[3]'s shape: block i at SNR bin i % n_bins, per-bin counters [n_bins, 4], one all-reduce ----
N_BINS, BIN_SIGMAS = 7, [0.7, 0.65, 0.6, 0.55, 0.5, 0.45, 0.4]


def _binned_counters_for(first_block, n_blocks):
    """What distributed.decode_blocks_sharded + nnv_mlp_decode_binned_f32 compute for a shard, restated with the oracle:
    row r -> block (first_row + r) // rows_per_block -> bin block % n_bins; the generator picks sigma by the same rule."""
    from oracle import awgn_model, fnn_oracle as O
    Ws, bs = _net()
    first_row = first_block * ROWS_PER_BLOCK
    x, labels = awgn_model.generate("133171", B, n_blocks * ROWS_PER_BLOCK, BIN_SIGMAS, SEED, first_row=first_row,
                                    rows_per_block=ROWS_PER_BLOCK)
    cls, _, _ = O.decode(x.astype(np.float32), Ws, bs, B)
    out = np.zeros((N_BINS, 4), np.int64)
    bins = ((first_row + np.arange(len(cls))) // ROWS_PER_BLOCK) % N_BINS
    for k in range(N_BINS):
        m = bins == k
        if m.any():
            out[k] = O.error_counters(cls[m], labels[m], B)
    return out


def _binned_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    nd.init_from_env(backend="gloo")
    first, count = nd.block_shard(N_BLOCKS, rank, world)
    c = torch.from_numpy(_binned_counters_for(first, count))
    mine = c.clone()
    nd.barrier()
    nd.all_reduce_counters(c)                         # [7, 4] int64: the configs[3] leg's one collective
    per_rank_blocks = nd.gather_ints(count)
    np.save(os.path.join(out_dir, f"b{rank}.npy"), np.concatenate([c.numpy().ravel(), mine.numpy().ravel(), per_rank_blocks]))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_binned_counter_reduction_equals_single_rank(tmp_path, world):
    """bench.py's configs[3] leg on CPU: contiguous shards of 101 blocks (not divisible by 2, 4 or 7), per-SNR-bin
    counters reduced by ONE all-reduce == one rank decoding every block; a rank's own share is a strict part."""
    mp.spawn(_binned_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    full = _binned_counters_for(0, N_BLOCKS)
    assert full[:, 3].tolist() == [ROWS_PER_BLOCK * len(range(k, N_BLOCKS, N_BINS)) for k in range(N_BINS)]
    assert (full[:, 2] == B * full[:, 3]).all() and full[:, 0].sum() > 0
    shares = np.zeros((N_BINS, 4), np.int64)
    for r in range(world):
        got = np.load(tmp_path / f"b{r}.npy")
        assert got[:28].reshape(N_BINS, 4).tolist() == full.tolist()
        shares += got[28:56].reshape(N_BINS, 4)
        assert got[56:].tolist() == [nd.block_shard(N_BLOCKS, q, world)[1] for q in range(world)]
    assert shares.tolist() == full.tolist()


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


# ------------------------------------------------------------------ guard rails
def test_launcher_time_limit_terminates_a_stuck_child(tmp_path):
    """A rendezvous that never completes must end with a diagnosis (which rank got how far), exit code 4
    and no JSON line -- not hang until somebody else's limit.  The fake 'bench script' writes its stage
    breadcrumb like a rank does and then sleeps."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = tmp_path / "stuck_rank.py"
    fake.write_text(
        "import os, time\n"
        "d = os.environ['NNV_BENCH_STAGE_DIR']\n"
        "open(os.path.join(d, 'rank' + os.environ['RANK']), 'w').write('joining the process group')\n"
        "open(os.path.join(%r, 'pid' + os.environ['RANK']), 'w').write(str(os.getpid()))\n"
        "time.sleep(600)\n" % str(tmp_path))
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.launch_ranks(2, 20.0, argv=[], script=%r))" % (root, str(fake)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 4
    assert time.time() - t0 < 120
    assert "did not finish within 20 s" in r.stderr
    assert "0: 'joining the process group'" in r.stderr and "1: 'joining the process group'" in r.stderr
    assert r.stdout.strip() == ""
    time.sleep(1.0)
    for rank in (0, 1):                                  # the stuck ranks are gone (their exact pids)
        pid = int((tmp_path / f"pid{rank}").read_text())
        alive = True
        try:
            os.kill(pid, 0)
            with open(f"/proc/{pid}/stat") as f:
                alive = f.read().split()[2] != "Z"
        except (ProcessLookupError, FileNotFoundError):
            alive = False
        assert not alive


def test_validate_group_rules():
    nd.validate_group(1, [0], 1, require_distinct_devices=False)
    nd.validate_group(4, [0, 1, 2, 3], 4, require_distinct_devices=True)
    nd.validate_group(2, [0, 0], 2, require_distinct_devices=False)         # the one-GPU gloo rehearsal
    with pytest.raises(RuntimeError, match="share GPUs"):
        nd.validate_group(2, [0, 0], 2, require_distinct_devices=True)
    with pytest.raises(RuntimeError, match="spans 1 rank"):
        nd.validate_group(1, [0], 2, require_distinct_devices=True)
    with pytest.raises(RuntimeError):
        nd.validate_group(2, [0], 2, require_distinct_devices=False)


def _dup_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    nd.init_from_env(backend="gloo")
    ones = torch.ones(1, dtype=torch.int64)
    nd.all_reduce_counters(ones)
    devices = nd.gather_ints(0)                       # forced duplicate: every rank claims GPU 0
    per_rank = nd.gather_floats(1.5 + rank)
    verdict = "ok"
    try:
        nd.validate_group(int(ones.item()), devices, world, require_distinct_devices=True)
    except RuntimeError as e:
        verdict = str(e)
    with open(os.path.join(out_dir, f"v{rank}.txt"), "w") as f:
        f.write(f"{verdict}|{per_rank}")
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_duplicate_devices_are_refused_on_every_rank_under_gloo(tmp_path):
    mp.spawn(_dup_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        verdict, per_rank = (tmp_path / f"v{r}.txt").read_text().split("|")
        assert "share GPUs" in verdict and "[0, 0]" in verdict
        assert per_rank == "[1.5, 2.5]"               # bench.py's ms_per_step_per_rank gather


def test_group_timeout_is_short_and_overridable(monkeypatch):
    monkeypatch.delenv("NNV_DIST_TIMEOUT_S", raising=False)
    assert nd.group_timeout_s() == 120.0
    monkeypatch.setenv("NNV_DIST_TIMEOUT_S", "7")
    assert nd.group_timeout_s() == 7.0


def test_bench_labels_say_what_ran(tmp_path, monkeypatch):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    dims = [16, 128, 64, 32, 256]
    full = bench.workload_label(10_000_000, dims)
    assert full.startswith("BASELINE configs[2]") and "10M blocks/GPU" in full and "16-128-64-32-256" in full
    red = bench.workload_label(4_000_000, dims)
    assert red.startswith("REDUCED:") and "4M blocks/GPU" in red
    assert "12,345 blocks/GPU" in bench.workload_label(12_345, dims)
    # replayed traffic: only for the profiled launch size AND the kernel sources the counters were taken on
    import json
    with open(os.path.join(root, "profiles", "pmc_traffic.json")) as f:
        tj = json.load(f)
    t, src = bench.replayed_traffic(tj["rows_per_launch"] + 32)
    assert t is None and "rows per launch" in src
    t, src = bench.replayed_traffic(tj["rows_per_launch"])
    if tj.get("k1_source_sha1") == bench.k1_source_hash():
        assert t == tj["hbm_bytes_per_launch"] and "replayed" in src and "not measured in this run" in src
    else:
        assert t is None and "stale" in src
    monkeypatch.setattr(bench, "k1_source_hash", lambda: "0" * 40)
    t, src = bench.replayed_traffic(tj["rows_per_launch"])
    assert t is None and "stale" in src


def test_bench_live_traffic_reads_the_counters_of_the_headline_kernel(tmp_path, monkeypatch):
    """bench.live_traffic: two `rocprofv3 --pmc` child passes, per-launch means of the fp32 LEAN K1 instantiation only,
    FETCH_SIZE x 2 (gfx950) + WRITE_SIZE.  A stand-in `rocprofv3` on PATH writes the CSV a real pass would."""
    import stat
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    fake = tmp_path / "bin"
    fake.mkdir()
    script = fake / "rocprofv3"
    script.write_text('''#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
ctr, out = a[a.index("--pmc") + 1], a[a.index("-d") + 1]
assert "--" in a and a[a.index("--") + 2].endswith("bench.py") and "--pmc-child" in a
mode = os.environ.get("FAKE_MODE", "ok")
if mode == "fail":
    sys.stderr.write("no counters today"); sys.exit(7)
if mode == "stuck":   # like the real launcher: the profiled program is a CHILD of rocprofv3; both hang
    import subprocess, time
    c = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(300)"])
    open(os.environ["FAKE_PID_FILE"], "w").write(str(c.pid))
    time.sleep(300)
os.makedirs(os.path.join(out, "host", "1"), exist_ok=True)
lean = "void nnv::mlp_decode_kernel<0, 2, 3, 4, 2, 1, 1024, false, 2>(nnv::DecodeParams)"
other = "void nnv::mlp_decode_kernel<0, 2, 3, 4, 2, 1, 1024, false, 0>(nnv::DecodeParams)"
f16x3 = "void nnv::mlp_decode_kernel<1, 2, 3, 4, 2, 1, 1024, false, 2>(nnv::DecodeParams)"
val = {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 300.0}[ctr]
rows = [] if mode == "empty" else [(lean, val - 10), (lean, val + 10), (other, 9e9), (f16x3, 9e9)]
with open(os.path.join(out, "host", "1", "run_counter_collection.csv"), "w") as f:
    f.write("Kernel_Name,Counter_Name,Counter_Value\\n")
    for k, v in rows:
        f.write('"%s",%s,%r\\n' % (k, ctr, v))
''')
    script.chmod(script.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(fake) + os.pathsep + os.environ["PATH"])
    t, src = bench.live_traffic(1000, timeout_s=60)
    assert t == 1000.0 * 1024 * 2 + 300.0 * 1024
    assert "measured in this run" in src and "x2" in src and "over 2 launches" in src
    monkeypatch.setenv("FAKE_MODE", "fail")
    t, src = bench.live_traffic(1000, timeout_s=60)
    assert t is None and "exited 7" in src and "no counters today" in src
    monkeypatch.setenv("FAKE_MODE", "empty")
    t, src = bench.live_traffic(1000, timeout_s=60)
    assert t is None and "no counter rows" in src
    # a stuck profiler: the time limit ends the pass AND the profiled grandchild (its whole process group) -- nothing of
    # it may run beside the timed region
    import time
    pid_file = tmp_path / "grandchild.pid"
    monkeypatch.setenv("FAKE_MODE", "stuck")
    monkeypatch.setenv("FAKE_PID_FILE", str(pid_file))
    t0 = time.time()
    t, src = bench.live_traffic(1000, timeout_s=3)
    assert t is None and "exceeded 3 s" in src and time.time() - t0 < 30
    pid = int(pid_file.read_text())
    for _ in range(50):
        try:
            os.kill(pid, 0)                    # still there?  (0 = probe only)
        except ProcessLookupError:
            break
        time.sleep(0.1)
    else:
        raise AssertionError("the profiled grandchild survived the time-out")
    monkeypatch.setenv("PATH", str(tmp_path / "nothing"))
    t, src = bench.live_traffic(1000)
    assert t is None and "not on PATH" in src
