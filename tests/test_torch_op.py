"""torch.ops.nnviterbi.mlp_decode / mlp_forward: one schema, CPU and CUDA(HIP) dispatch keys.

CPU box: the CPU key (plain torch ops) against the oracle on the fixture networks; the packed-
network cache's keying (a freed-and-reallocated checkpoint must never hit the old entry).
GPU box: CPU key vs HIP key on the same inputs under the near-tie rule; the cache on real tensors."""
import gc

import numpy as np
import pytest
import torch

import nnviterbi_amd  # noqa: F401
from nnviterbi_amd import decoder
from oracle import fnn_oracle as O

CASES = [("b8_l3", 4, 8), ("b7_l3", 4, 7), ("b10_l2", 3, 10), ("b3_l3", 4, 3)]
TOL = 1e-5


def _case(fg, name, depth):
    return ([fg[f"{name}_W{i}"] for i in range(depth)], [fg[f"{name}_b{i}"] for i in range(depth)],
            fg[f"{name}_x"], fg[f"{name}_labels"])


def _t(arrs, device="cpu"):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in arrs]


@pytest.mark.parametrize("name,depth,b", CASES)
def test_cpu_key_matches_the_oracle(forward_golden, name, depth, b):
    Ws, bs, x, labels = _case(forward_golden, name, depth)
    cls, bits, cnt = torch.ops.nnviterbi.mlp_decode(torch.from_numpy(x), _t(Ws), _t(bs), b,
                                                    torch.from_numpy(labels.astype(np.int32)))
    out = torch.ops.nnviterbi.mlp_forward(torch.from_numpy(x), _t(Ws), _t(bs))
    assert cls.dtype == torch.int64 and bits.dtype == torch.uint8 and cnt.dtype == torch.int64
    c64, _, o64 = O.decode(x, Ws, bs, b, np.float64)
    assert np.abs(out.numpy() - o64).max() < TOL
    bad = cls.numpy() != c64
    assert not bad.any() or O.top2_margin(o64)[bad].max() < TOL
    assert np.array_equal(bits.numpy(), O.class_to_bits(cls.numpy(), b))
    assert np.array_equal(cnt.numpy(), O.error_counters(cls.numpy(), labels, b))


def test_cpu_key_edge_cases():
    W = _t([np.zeros((4, 8), np.float32), np.zeros((8, 4), np.float32)])
    b = _t([np.zeros(8, np.float32), np.full(4, 30.0, np.float32)])
    cls, bits, cnt = torch.ops.nnviterbi.mlp_decode(torch.ones(3, 4), W, b, 2)
    assert cls.tolist() == [0, 0, 0]                       # saturated ties -> first maximum
    assert bits.shape == (3, 2) and cnt.tolist() == [0, 0, 0, 0]
    cls, bits, cnt = torch.ops.nnviterbi.mlp_decode(torch.empty(0, 4), W, b, 2, torch.empty(0, dtype=torch.int32))
    assert cls.shape == (0,) and bits.shape == (0, 2) and cnt.tolist() == [0, 0, 0, 0]
    with pytest.raises(Exception):
        torch.ops.nnviterbi.mlp_decode(torch.ones(3, 5), W, b, 2)          # n_in mismatch
    with pytest.raises(Exception):
        torch.ops.nnviterbi.mlp_decode(torch.ones(3, 4), W, b, 1)          # 2**1 < 4 classes
    with pytest.raises(Exception):
        torch.ops.nnviterbi.mlp_decode(torch.ones(3, 4), W, b, 2, torch.zeros(3, dtype=torch.int64))


def test_product_never_imports_the_oracle():
    import os
    import re
    pkg = os.path.dirname(decoder.__file__)
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), fn


class _FakePacked:
    """Stands in for PackedMLP so the cache's keying can be exercised without a GPU."""
    made = []

    def __init__(self, weights, biases, device=None, precision=None):
        self.tag = float(weights[0].flatten()[0])
        self.precision = precision
        self.closed = False
        _FakePacked.made.append(self)

    def close(self):
        self.closed = True


@pytest.fixture
def fake_pack(monkeypatch):
    decoder.clear_pack_cache()
    _FakePacked.made = []
    monkeypatch.setattr(decoder, "PackedMLP", _FakePacked)
    yield _FakePacked
    decoder.clear_pack_cache()
    decoder.set_default_precision("f32")


def _weights(v):
    return [torch.full((4, 8), v), torch.full((8, 4), v)], [torch.zeros(8), torch.zeros(4)]


def test_pack_cache_follows_the_tensors_not_their_addresses(fake_pack):
    Wa, ba = _weights(1.0)
    n1 = decoder._cached_pack(Wa, ba, "cpu")
    assert decoder._cached_pack(Wa, ba, "cpu") is n1 and len(fake_pack.made) == 1      # hit
    # the idioms that make NEW tensor objects over the same storage every call must hit
    assert decoder._cached_pack([w.detach() for w in Wa], [v.detach() for v in ba], "cpu") is n1
    assert decoder._cached_pack([w.data for w in Wa], [v.view(-1) for v in ba], "cpu") is n1
    assert len(fake_pack.made) == 1
    assert decoder._cached_pack([Wa[0].t().contiguous().t(), Wa[1]], ba, "cpu") is not n1      # another storage
    ptrs = [t.data_ptr() for t in Wa + ba]
    del Wa, ba
    gc.collect()
    assert not n1.closed                                     # nothing is freed from the garbage collector ...
    # a new checkpoint of the same shapes, very likely at the same addresses with _version 0 again
    Wb, bb = _weights(2.0)
    n2 = decoder._cached_pack(Wb, bb, "cpu")
    assert n1.closed                                         # ... dead entries go at the next call
    assert all(not any(r.expired() for r in e.refs) for e in decoder._PACK_CACHE.values())
    assert n2 is not n1 and n2.tag == 2.0
    if [t.data_ptr() for t in Wb + bb] == ptrs:
        assert all(t._version == 0 for t in bb)              # the case the old (ptr, version) key got wrong
    # in-place update bumps _version -> repack
    Wb[0].add_(1.0)
    n3 = decoder._cached_pack(Wb, bb, "cpu")
    assert n3 is not n2 and n3.tag == 3.0 and n2.closed
    # the resolved default precision is part of the key
    decoder.set_default_precision("f16x3")
    n4 = decoder._cached_pack(Wb, bb, "cpu")
    assert n4 is not n3 and n4.precision == "f16x3" and not n3.closed
    decoder.set_default_precision("f32")
    assert decoder._cached_pack(Wb, bb, "cpu") is n3


def test_pack_cache_is_bounded(fake_pack):
    keep = [_weights(float(i)) for i in range(decoder._PACK_CACHE_MAX + 4)]
    nets = [decoder._cached_pack(W, b, "cpu") for W, b in keep]
    assert len(decoder._PACK_CACHE) == decoder._PACK_CACHE_MAX
    assert all(n.closed for n in nets[:4]) and not any(n.closed for n in nets[4:])


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name,depth,b", CASES)
def test_cpu_key_and_hip_key_agree(cuda_device, forward_golden, name, depth, b):
    Ws, bs, x, labels = _case(forward_golden, name, depth)
    lab = torch.from_numpy(labels.astype(np.int32))
    c_cls, c_bits, c_cnt = torch.ops.nnviterbi.mlp_decode(torch.from_numpy(x), _t(Ws), _t(bs), b, lab)
    c_out = torch.ops.nnviterbi.mlp_forward(torch.from_numpy(x), _t(Ws), _t(bs))
    g_cls, g_bits, g_cnt = torch.ops.nnviterbi.mlp_decode(torch.from_numpy(x).cuda(), _t(Ws, "cuda"),
                                                          _t(bs, "cuda"), b, lab.cuda())
    g_out = torch.ops.nnviterbi.mlp_forward(torch.from_numpy(x).cuda(), _t(Ws, "cuda"), _t(bs, "cuda"))
    assert g_cls.is_cuda and g_bits.is_cuda and g_cnt.is_cuda and g_out.is_cuda
    assert (g_out.cpu() - c_out).abs().max().item() < TOL
    _, _, o64 = O.decode(x, Ws, bs, b, np.float64)
    bad = (g_cls.cpu() != c_cls).numpy()
    assert not bad.any() or O.top2_margin(o64)[bad].max() < TOL
    if not bad.any():
        assert torch.equal(g_bits.cpu(), c_bits) and torch.equal(g_cnt.cpu(), c_cnt)
    decoder.clear_pack_cache()


@pytest.mark.gpu
def test_reloaded_checkpoint_is_not_decoded_with_the_old_network(cuda_device):
    """Free the weights, load another checkpoint of the same shapes (the caching allocator hands
    back the same addresses, _version 0 again): the op must decode with the NEW network."""
    from nnviterbi_amd import fnn_viterbi as fv
    decoder.clear_pack_cache()
    cfg = fv.DecoderConfig(block_length=8)
    x = torch.randn(4096, 16, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
    results, ptrs = [], []
    for seed in (11, 12):
        w, b = fv.initialize_weights(cfg, rng=np.random.RandomState(seed))
        Ws, bs = fv.weights_as_lists(w, b)
        Wd, bd = _t(Ws, "cuda"), _t(bs, "cuda")
        ptrs.append([t.data_ptr() for t in Wd + bd])
        cls, _, _ = torch.ops.nnviterbi.mlp_decode(x, Wd, bd, 8)
        ref = O.decode_c(x.cpu().numpy(), Ws, bs, 8, want_bits=False)["cls"]
        assert (cls.cpu().numpy() != ref).mean() < 1e-3, f"checkpoint {seed} decoded with another network"
        results.append(cls.cpu())
        del Wd, bd
        gc.collect()
    assert (results[0] != results[1]).float().mean() > 0.5
    assert len(decoder._PACK_CACHE) == 1        # the first checkpoint's entry went when the second was packed;
    decoder._sweep_dead_entries()               # the second one's goes at the next call (never from the GC)
    assert len(decoder._PACK_CACHE) == 0
