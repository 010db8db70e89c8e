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
    with pytest.raises(Exception):
        torch.ops.nnviterbi.mlp_forward(torch.ones(3, 5), W, b)            # n_in mismatch, forward op too
    # up to 8 dense layers (the wide path's depth); 9 are refused
    deep_W = _t([np.eye(4, dtype=np.float32)] * 8)
    deep_b = _t([np.zeros(4, np.float32)] * 7 + [np.array([0., 1., 0., 0.], np.float32)])
    assert torch.ops.nnviterbi.mlp_decode(torch.zeros(2, 4), deep_W, deep_b, 2)[0].tolist() == [1, 1]
    with pytest.raises(Exception):
        torch.ops.nnviterbi.mlp_decode(torch.zeros(2, 4), deep_W + deep_W[:1], deep_b + deep_b[:1], 2)


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


@pytest.mark.parametrize("alias", ["same", "detach", "data", "view"])
@pytest.mark.parametrize("write", ["param", "param_data", "alias"])
def test_pack_cache_sees_every_in_place_write(fake_pack, alias, write):
    """``p.data`` carries a fresh version counter that is always 0 and a write through
    ``p.data`` does not bump ``p``'s own, so no counter sees ``[p.data for p in params]`` after an optimizer
    step.  Whatever alias is passed and whatever alias is written through, the next call must re-pack
    with the new values -- and an untouched network must keep hitting."""
    params = [torch.nn.Parameter(torch.full((4, 8), 1.0)), torch.nn.Parameter(torch.full((8, 4), 1.0))]
    biases = [torch.nn.Parameter(torch.zeros(8)), torch.nn.Parameter(torch.zeros(4))]

    def passed(ts):
        return {"same": lambda t: t, "detach": lambda t: t.detach(), "data": lambda t: t.data,
                "view": lambda t: t.detach().view(t.shape)}[alias]

    def call():
        return decoder._cached_pack([passed(params)(t) for t in params], [passed(biases)(t) for t in biases], "cpu")

    n1 = call()
    assert call() is n1 and len(fake_pack.made) == 1 and n1.tag == 1.0          # untouched: a hit
    with torch.no_grad():
        if write == "param":
            params[0].add_(1.0)                   # what optimizer.step() does
        elif write == "param_data":
            params[0].data.add_(1.0)              # bumps no counter the caller can see
        else:
            passed(params)(params[0]).add_(1.0)   # through an alias of the kind that is passed
    n2 = call()
    assert n2 is not n1 and n2.tag == 2.0 and n1.closed, "decoded with the network packed before the write"
    assert call() is n2 and len(fake_pack.made) == 2
    with torch.no_grad():
        biases[1].data[3] = -0.0                   # a bias, one element, value-equal but another bit pattern
    assert call() is not n2


def test_default_precision_never_resolves_to_f16(monkeypatch):
    """'f16 is explicit opt-in only' must hold for the environment too."""
    assert decoder._resolve_precision("f16", allow_f16=True) == "f16"          # explicit argument: allowed
    with pytest.raises(ValueError):
        decoder._resolve_precision("f16")
    monkeypatch.setenv("NNV_PRECISION", "f16")
    with pytest.warns(RuntimeWarning):
        assert decoder._default_precision_from_env() == "f32"
    monkeypatch.setenv("NNV_PRECISION", "f16x3")
    assert decoder._default_precision_from_env() == "f16x3"
    monkeypatch.setenv("NNV_PRECISION", "bogus")
    with pytest.warns(RuntimeWarning):
        assert decoder._default_precision_from_env() == "f32"
    # even a default forced past the setter cannot reach the single-product mode
    monkeypatch.setattr(decoder, "_DEFAULT_PRECISION", ["f16"])
    with pytest.raises(ValueError):
        decoder._resolve_precision(None, allow_f16=True)
    with pytest.raises(ValueError):
        decoder.set_default_precision("f16")


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


@pytest.mark.gpu
def test_optimizer_step_behind_the_data_idiom_is_decoded_with_the_new_weights(cuda_device):
    """ADVICE r3 (high) on real device tensors: pass ``[p.data for p in params]``, update the parameters in
    place (what ``optimizer.step()`` does), call again -- no version counter moves, the bytes do."""
    from nnviterbi_amd import fnn_viterbi as fv
    decoder.clear_pack_cache()
    cfg = fv.DecoderConfig(block_length=8)
    w, b = fv.initialize_weights(cfg, rng=np.random.RandomState(21))
    Ws, bs = fv.weights_as_lists(w, b)
    params = [torch.nn.Parameter(t) for t in _t(Ws, "cuda") + _t(bs, "cuda")]
    x = torch.randn(8192, 16, device="cuda", generator=torch.Generator("cuda").manual_seed(5))

    def op():
        ts = [p.data for p in params]
        return torch.ops.nnviterbi.mlp_decode(x, ts[:4], ts[4:], 8)[0].cpu()

    def fresh():
        net = decoder.PackedMLP([p.detach() for p in params[:4]], [p.detach() for p in params[4:]])
        try:
            return net.decode(x, want_cls32=False, want_cls64=True)["cls64"].cpu()
        finally:
            net.close()

    c1 = op()
    assert torch.equal(c1, fresh()) and torch.equal(op(), c1) and len(decoder._PACK_CACHE) == 1
    opt = torch.optim.SGD(params, lr=1.0)
    for p in params:
        p.grad = torch.randn(p.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(p.numel()))
    opt.step()
    c2 = op()
    assert torch.equal(c2, fresh()), "the op decoded with the network packed before optimizer.step()"
    assert (c2 != c1).float().mean() > 0.2
    with torch.no_grad():
        params[0].data.mul_(-1.0)                 # a write no counter records at all
    assert torch.equal(op(), fresh())
    decoder.clear_pack_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("dims,b", [([14, 200, 96, 48, 128], 7), ([16, 256, 128, 64, 256], 8), ([20, 300, 1024], 10),
                                    ([40, 100, 100, 100, 100, 2], 1), ([24, 64, 48, 48, 40, 32, 64], 6)])
def test_op_takes_networks_of_other_hidden_widths(cuda_device, dims, b):
    """The reference's n_hidden_* are free parameters: shapes the register-chained kernel has no instantiation for must
    not make the OP fail -- it runs them through the wide path (one persistent launch for the whole chain up to
    512-wide layers, one launch per layer beyond), same schema, same contract, cache included.  The last two are DEEPER
    than the register-chained kernel goes (5 and 6 dense layers: the sliding-window decoder's depth,
    src/fnn_viterbi_seq_predict.py:343-414)."""
    rng = np.random.RandomState(len(dims) + b)
    Ws = [(rng.randn(dims[i], dims[i + 1]) * np.sqrt(2.0 / dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [(rng.randn(d) * 0.1).astype(np.float32) for d in dims[1:]]
    x = rng.randn(3001, dims[0]).astype(np.float32)
    labels = rng.randint(0, dims[-1], 3001).astype(np.int32)
    decoder.clear_pack_cache()
    g_cls, g_bits, g_cnt = torch.ops.nnviterbi.mlp_decode(torch.from_numpy(x).cuda(), _t(Ws, "cuda"), _t(bs, "cuda"), b,
                                                          torch.from_numpy(labels).cuda())
    g_out = torch.ops.nnviterbi.mlp_forward(torch.from_numpy(x).cuda(), _t(Ws, "cuda"), _t(bs, "cuda"))
    assert isinstance(next(iter(decoder._PACK_CACHE.values())).net, decoder.WideMLP)
    c64, _, o64 = O.decode(x, Ws, bs, b, np.float64)
    cls = g_cls.cpu().numpy()
    assert np.abs(g_out.cpu().numpy() - o64).max() < TOL
    bad = cls != c64
    assert not bad.any() or O.top2_margin(o64)[bad].max() < TOL
    assert np.array_equal(g_bits.cpu().numpy(), O.class_to_bits(cls, b))
    assert np.array_equal(g_cnt.cpu().numpy(), O.error_counters(cls, labels, b))
    decoder.clear_pack_cache()


@pytest.mark.gpu
@pytest.mark.filterwarnings("ignore:The CUDA Graph is empty")
def test_op_under_graph_capture_says_what_to_do_instead(cuda_device):
    """The ops verify the weights' bytes on every call (a device wait): inside a capture they must say so, not fail in
    the middle of torch.equal -- and the handle's decode() must be what the message promises, capturable."""
    rng = np.random.RandomState(5)
    dims = [14, 128, 64, 32, 128]
    Ws = _t([(rng.randn(dims[i], dims[i + 1]) * 0.3).astype(np.float32) for i in range(4)], "cuda")
    bs = _t([(rng.randn(d) * 0.1).astype(np.float32) for d in dims[1:]], "cuda")
    x = torch.from_numpy(rng.randn(512, 14).astype(np.float32)).cuda()
    eager = torch.ops.nnviterbi.mlp_decode(x, Ws, bs, 7)[0].clone()
    net = decoder.PackedMLP(Ws, bs, device="cuda")
    net.decode(x, want_cls64=True)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with pytest.raises(RuntimeError, match="cannot be captured"):
            with torch.cuda.graph(g, stream=side):
                torch.ops.nnviterbi.mlp_decode(x, Ws, bs, 7)
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        out = net.decode(x, want_cls64=True)["cls64"]
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    net.close()
    decoder.clear_pack_cache()


# ---------------------------------------------------------------- handle-based ops + the reference-shaped calls' cache
def test_cached_network_is_keyed_by_the_bytes(monkeypatch):
    """decoder.cached_network (what fnn_viterbi.predict / run_graph / predict_stream ... keep their packed network in):
    the same BYTES find the same network whatever array objects carry them, an in-place edit does not, and the
    least-recently-used entry is closed when the cache is full.  No GPU: `build` is a stand-in."""
    monkeypatch.setattr(decoder, "_require_cuda", lambda: None)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    decoder.clear_pack_cache()

    class Net:
        def __init__(self):
            self.closed = False

        def close(self):
            self.closed = True

    made = []

    def build():
        made.append(Net())
        return made[-1]

    w = [np.arange(12, dtype=np.float32).reshape(3, 4), np.ones(4, np.float32)]
    a = decoder.cached_network("ff", w, build)
    assert decoder.cached_network("ff", [w[0].copy(), w[1].astype(np.float64)], build) is a and len(made) == 1
    assert decoder.cached_network("ff", w, build, extra=("relu",)) is not a         # another kind of network
    assert decoder.cached_network("ff", [w[0].reshape(4, 3), w[1]], build) is not a   # same bytes, another shape
    w[0][0, 0] = 99.0                                                               # edited in place
    b = decoder.cached_network("ff", w, build)
    assert b is not a and len(made) == 4
    for i in range(decoder._NET_CACHE_MAX):
        decoder.cached_network("ff", [np.full(3, float(i), np.float32)], build)
    assert a.closed and not made[-1].closed                                         # oldest evicted and closed
    decoder.clear_pack_cache()
    assert all(n.closed for n in made) and not decoder._NET_CACHE


@pytest.mark.gpu
@pytest.mark.parametrize("dims,b,precision", [([16, 128, 64, 32, 256], 8, "f32"), ([16, 128, 64, 32, 256], 8, "f16x3"),
                                              ([20, 128, 64, 1024], 10, "f32"), ([14, 200, 96, 48, 128], 7, "f32"),
                                              ([40, 100, 100, 100, 100, 2], 1, "f32")])
def test_prepacked_handle_ops_equal_the_raw_weight_ops_and_capture_into_a_hip_graph(cuda_device, dims, b, precision):
    """SURVEY 8(b) "stream-ordered, re-entrant, no global state": nnviterbi::mlp_prepack once, then
    nnviterbi::mlp_decode_packed is launches on the current stream and nothing else -- so it captures into a HIP graph
    (a capture forbids host waits and hipMalloc) WITHOUT a warm-up call, and the replays on new inputs equal the eager
    raw-weights op, classes, bits and counters.  K1 shapes (resident and streamed class layer), K6 and a deep net."""
    rng = np.random.RandomState(sum(dims) + b)
    Ws = [(rng.randn(dims[i], dims[i + 1]) * np.sqrt(2.0 / dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [(rng.randn(d) * 0.1).astype(np.float32) for d in dims[1:]]
    n = 5003
    xs = [torch.from_numpy(rng.randn(n, dims[0]).astype(np.float32)).cuda() for _ in range(3)]
    labs = [torch.from_numpy(rng.randint(0, dims[-1], n).astype(np.int32)).cuda() for _ in range(3)]
    decoder.clear_pack_cache()
    decoder.set_default_precision(precision)
    try:
        before = dict(decoder.PACK_STATS)
        h = torch.ops.nnviterbi.mlp_prepack(_t(Ws, "cuda"), _t(bs), precision)      # weights from any device
        assert isinstance(h, int) and sum(decoder.PACK_STATS.values()) == sum(before.values()) + 1
        x_static, l_static = xs[0].clone(), labs[0].clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):                                                   # first call ever on this handle
            cls, bits, cnt = torch.ops.nnviterbi.mlp_decode_packed(x_static, h, b, l_static)
        for x, lab in zip(xs, labs):
            x_static.copy_(x)
            l_static.copy_(lab)
            g.replay()
            torch.cuda.synchronize()
            e_cls, e_bits, e_cnt = torch.ops.nnviterbi.mlp_decode(x, _t(Ws, "cuda"), _t(bs, "cuda"), b, lab)
            assert torch.equal(cls, e_cls) and torch.equal(bits, e_bits) and torch.equal(cnt, e_cnt)
            p_cls, p_bits, p_cnt = torch.ops.nnviterbi.mlp_decode_packed(x, h, b, lab)     # eager, same handle
            assert torch.equal(p_cls, e_cls) and torch.equal(p_bits, e_bits) and torch.equal(p_cnt, e_cnt)
        # 20 more decodes: no further prepack (the raw-weights op packed once for its own cache)
        count = sum(decoder.PACK_STATS.values())
        for _ in range(20):
            torch.ops.nnviterbi.mlp_decode_packed(xs[1], h, b, None)
        assert sum(decoder.PACK_STATS.values()) == count
        out = torch.ops.nnviterbi.mlp_forward_packed(xs[2], h)
        assert torch.equal(out, torch.ops.nnviterbi.mlp_forward(xs[2], _t(Ws, "cuda"), _t(bs, "cuda")))
        o64 = O.decode(xs[2].cpu().numpy(), Ws, bs, b, np.float64)[2]
        assert np.abs(out.cpu().numpy() - o64).max() < TOL
        with pytest.raises(Exception):
            torch.ops.nnviterbi.mlp_decode_packed(xs[0][:, :-1].contiguous(), h, b, None)   # wrong row length
        with pytest.raises(Exception):
            torch.ops.nnviterbi.mlp_decode_packed(xs[0], h, b, labs[0].to(torch.int64))
        torch.ops.nnviterbi.mlp_free(h)
        with pytest.raises(Exception, match="not a live handle"):
            torch.ops.nnviterbi.mlp_decode_packed(xs[0], h, b, None)
        torch.ops.nnviterbi.mlp_free(h)                                              # idempotent
    finally:
        decoder.set_default_precision("f32")
        decoder.clear_pack_cache()


@pytest.mark.gpu
def test_two_handles_on_two_streams_do_not_share_state(cuda_device):
    """Re-entrant: two networks, two streams, interleaved launches; each result equals its own network's."""
    rng = np.random.RandomState(77)
    dims = [16, 128, 64, 32, 256]
    nets = []
    for _ in range(2):
        Ws = [(rng.randn(dims[i], dims[i + 1]) * np.sqrt(2.0 / dims[i])).astype(np.float32) for i in range(4)]
        bs = [(rng.randn(d) * 0.1).astype(np.float32) for d in dims[1:]]
        nets.append((Ws, bs, torch.ops.nnviterbi.mlp_prepack(_t(Ws), _t(bs), "f32")))
    x = torch.from_numpy(rng.randn(200_003, 16).astype(np.float32)).cuda()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    got = [[], []]
    for _ in range(4):
        for i, (_, _, h) in enumerate(nets):
            with torch.cuda.stream(streams[i]):
                got[i].append(torch.ops.nnviterbi.mlp_decode_packed(x, h, 8, None)[0])
    torch.cuda.synchronize()
    for i, (Ws, bs, h) in enumerate(nets):
        want = torch.ops.nnviterbi.mlp_decode(x, _t(Ws, "cuda"), _t(bs, "cuda"), 8)[0]
        assert all(torch.equal(c, want) for c in got[i])
        torch.ops.nnviterbi.mlp_free(h)
    assert not torch.equal(got[0][0], got[1][0])
    decoder.clear_pack_cache()


@pytest.mark.gpu
def test_reference_shaped_calls_pack_once(cuda_device, capsys, tmp_path):
    """fnn_viterbi.predict / multilayer_perceptron / run_graph packed, uploaded and freed the
    network around every call.  Now 50 x run_graph(model) on the same restored model = ONE prepack; new weights = one
    more; and the predictions are the same as a fresh pack's."""
    from nnviterbi_amd import fnn_viterbi as fv
    from nnviterbi_amd import fnn_viterbi_mixed_snr as mx
    cfg = mx.MixedSNRDecoderConfig(block_length=7, samples_per_class_train=2, samples_per_class_test=20)
    np.random.seed(4)
    tr, te, trl, rl, rt, _ = mx.get_dataset(cfg)
    tel = mx.create_one_hot_labels(mx.get_dataset.last_test_labels, cfg.n_classes)
    w, b = mx.initialize_weights(cfg, rng=np.random.RandomState(1))
    path = mx.save_model(str(tmp_path / "m"), w, b)
    decoder.clear_pack_cache()
    model = mx.create_model_graph(cfg, tr, trl, te, tel, train_mode=False, model_load_path=path, run=False)
    base = sum(decoder.PACK_STATS.values())
    first = mx.run_graph(model).copy()
    assert sum(decoder.PACK_STATS.values()) == base + 1
    for _ in range(49):
        assert np.array_equal(mx.run_graph(model), first)
    for _ in range(5):
        fv.predict(te, model.weights, model.biases)
        fv.multilayer_perceptron(te[:100], model.weights, model.biases)
    assert sum(decoder.PACK_STATS.values()) == base + 1          # 50 run_graph + 10 direct calls: one prepack
    net = decoder.PackedMLP(*fv.weights_as_lists(model.weights, model.biases))
    assert np.array_equal(net.decode(torch.from_numpy(te).cuda(), want_cls64=True)["cls64"].cpu().numpy(), first)
    net.close()
    w2 = {k: v + np.float32(0.01) for k, v in model.weights.items()}
    fv.predict(te, w2, model.biases)
    assert sum(decoder.PACK_STATS.values()) == base + 3          # the explicit PackedMLP above + the new weights
    model.weights["h1"][0, 0] += 1.0                             # in-place edit of the model's own arrays: seen
    changed = mx.run_graph(model)
    assert sum(decoder.PACK_STATS.values()) == base + 4 and changed.shape == first.shape
    capsys.readouterr()
    decoder.clear_pack_cache()


@pytest.mark.gpu
def test_c_abi_is_reentrant_from_host_threads(cuda_device):
    """include/nnviterbi.h: "stream-ordered, re-entrant, no global mutable state besides a per-(device, kernel)
    attribute cache".  Four host threads (ctypes releases the GIL inside the calls), each with its OWN handle -- two of
    them packed inside the thread, concurrently -- its own stream and its own rows, 40 decodes each, K1 resident,
    K1 streamed, K6 and the bank: every result equals the same handle's single-threaded result."""
    import threading
    from nnviterbi_amd.decoder import BitwiseBank, PackedMLP, WideMLP
    rng = np.random.RandomState(123)

    def net(dims):
        Ws = [(rng.randn(dims[i], dims[i + 1]) * np.sqrt(2.0 / dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
        bs = [(rng.randn(d) * 0.1).astype(np.float32) for d in dims[1:]]
        return Ws, bs

    specs = [("k1", net([16, 128, 64, 32, 256])), ("k1s", net([20, 128, 64, 1024])), ("k6", net([24, 200, 96, 48, 128])),
             ("bank", [(rng.randn(64, 30).astype(np.float32) * 0.2, rng.randn(30).astype(np.float32) * 0.1,
                        rng.randn(30, 2).astype(np.float32), rng.randn(2).astype(np.float32) * 0.1) for _ in range(5)])]
    n = 30_011
    xs = {k: torch.from_numpy(rng.randn(n, (v[0][0].shape[0] if k != "bank" else 64)).astype(np.float32)).cuda()
          for k, v in specs}

    def build(kind, w):
        if kind == "bank":
            return BitwiseBank(w)
        return WideMLP(*w) if kind == "k6" else PackedMLP(*w)

    def run(kind, obj, x):
        if kind == "bank":
            return obj.decode(x)["bits"]
        return obj.decode(x)["cls32"]

    expected = {}
    for kind, w in specs:                       # single-threaded truth, objects closed again
        o = build(kind, w)
        expected[kind] = run(kind, o, xs[kind]).clone()
        o.close()
    torch.cuda.synchronize()
    errors, results = [], {}

    def worker(kind, w, prebuilt):
        try:
            torch.cuda.set_device(0)
            obj = prebuilt if prebuilt is not None else build(kind, w)     # two threads prepack concurrently
            stream = torch.cuda.Stream()
            out = []
            with torch.cuda.stream(stream):
                for _ in range(40):
                    out.append(run(kind, obj, xs[kind]))
            stream.synchronize()
            results[kind] = all(torch.equal(o, expected[kind]) for o in out)
            obj.close()
        except Exception as e:  # noqa: BLE001
            errors.append(f"{kind}: {type(e).__name__}: {e}")

    prebuilt = {"k1": build("k1", specs[0][1]), "k6": build("k6", specs[2][1])}
    threads = [threading.Thread(target=worker, args=(k, w, prebuilt.get(k))) for k, w in specs]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert results == {k: True for k, _ in specs}, results


@pytest.mark.gpu
def test_bound_decode_is_the_same_decode_with_the_host_work_done_once(cuda_device):
    """PackedMLP.bound_decode: a prepared call over fixed buffers (serving short inputs) returns what decode() returns,
    follows in-place refills of x, adds to its counters, and is capturable into a HIP graph."""
    import torch
    from nnviterbi_amd.decoder import PackedMLP
    rng = np.random.RandomState(3)
    dims = [16, 128, 64, 32, 256]
    Ws = [(rng.randn(dims[i], dims[i + 1]) * np.sqrt(2.0 / dims[i])).astype(np.float32) for i in range(4)]
    bs = [(rng.randn(dims[i + 1]) * 0.1).astype(np.float32) for i in range(4)]
    net = PackedMLP(Ws, bs)
    x = torch.tensor(rng.randn(2000, 16).astype(np.float32), device="cuda")
    labels = torch.tensor(rng.randint(0, 256, 2000).astype(np.int32), device="cuda")
    call = net.bound_decode(x, labels=labels, tile_bits=8, want_bits=True)
    ref = net.decode(x, labels=labels, tile_bits=8, want_bits=True)
    out = call()
    torch.cuda.synchronize()
    for k in ("cls32", "bits", "counters"):
        assert torch.equal(out[k], ref[k]), k
    x.copy_(torch.tensor(rng.randn(2000, 16).astype(np.float32), device="cuda"))      # refill in place
    ref2 = net.decode(x, labels=labels, tile_bits=8, want_bits=True)
    out = call()
    torch.cuda.synchronize()
    assert torch.equal(out["cls32"], ref2["cls32"]) and torch.equal(out["counters"], ref["counters"] + ref2["counters"])
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        gcall = net.bound_decode(x, tile_bits=8)           # bound to the side stream
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            gcall()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(gcall.outputs["cls32"], ref2["cls32"])
    net.close()

