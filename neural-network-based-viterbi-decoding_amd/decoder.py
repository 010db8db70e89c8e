"""Device-side decoder object and the ``nnviterbi::mlp_decode`` torch op.

PyTorch is plumbing here (HBM allocations, streams); the arithmetic runs in
``libnnviterbi_hip.so`` through the C ABI (``include/nnviterbi.h``).

Replaces the reference's ``Session.run`` of
``tf.argmax(multilayer_perceptron(x, weights, biases), 1)``
(src/fnn_viterbi.py:494-510,606-610) plus the class->bits and BER/FER loops
(src/fnn_viterbi.py:639-671).
"""
from __future__ import annotations

import collections
import ctypes
import hashlib
import itertools
import os
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _cabi
from ._cabi import NNViterbiError


def _as_host_f32(a) -> np.ndarray:
    if isinstance(a, torch.Tensor):
        a = a.detach().to("cpu").numpy()
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def _default_precision_from_env() -> str:
    """``NNV_PRECISION`` seeds the default under the same rule as ``set_default_precision``: only the two
    modes inside the 1e-5 parity contract.  "f16" (or anything unknown) is refused with a warning and
    the default stays "f32" -- the single-product mode is never reached without an explicit argument."""
    p = os.environ.get("NNV_PRECISION", "f32")
    if p in ("f32", "f16x3"):
        return p
    import warnings
    warnings.warn(f"NNV_PRECISION={p!r} ignored: the default precision must be 'f32' or 'f16x3' "
                  "('f16' is outside the parity contract and has to be passed explicitly)", RuntimeWarning)
    return "f32"


_DEFAULT_PRECISION = [_default_precision_from_env()]


def set_default_precision(precision: str) -> None:
    """Arithmetic of every decoder object built without an explicit ``precision`` (so also of the
    reference-shaped modules, which build them internally): "f32" (exact fp32 MFMA chains, the
    default and the parity reference) or "f16x3" (hi/lo f16 split operands, ~2-3x faster, same
    stated tolerances).  The environment variable ``NNV_PRECISION`` sets the initial value."""
    if precision not in _cabi.NNV_PRECISION or precision == "f16":   # "f16" is outside the parity contract: explicit only
        raise ValueError("default precision must be 'f32' or 'f16x3'")
    _DEFAULT_PRECISION[0] = precision


def _resolve_precision(precision: Optional[str], allow_f16: bool = False) -> str:
    """``allow_f16`` admits "f16" only as an EXPLICIT argument: a default can never resolve to it."""
    explicit = precision is not None
    p = precision if explicit else _DEFAULT_PRECISION[0]
    if p not in _cabi.NNV_PRECISION or (p == "f16" and not (allow_f16 and explicit)):
        allowed = sorted(k for k in _cabi.NNV_PRECISION if (allow_f16 and explicit) or k != "f16")
        raise ValueError(f"precision must be one of {allowed}")
    return p


def _require_cuda():
    if not torch.cuda.is_available():
        raise NNViterbiError(
            "no HIP device visible to PyTorch-ROCm: the decode path runs only "
            "on MI355X (gfx950); there is no CPU fallback")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream_ptr(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


# Networks packed (prepack + upload) by this process, per kind: the tests of the caches below count them.
PACK_STATS = {"mlp": 0, "wide": 0, "bank": 0, "seq2seq": 0}


class PackedMLP:
    """A network pre-packed into MFMA fragment order, resident in HBM.

    ``weights[l]`` is ``[n_l, n_{l+1}]`` (the reference's right-multiply
    layout, src/fnn_viterbi.py:432-456) and ``biases[l]`` is ``[n_{l+1}]``;
    2..4 dense layers, the last one being the class layer.
    """

    def __init__(self, weights: Sequence, biases: Sequence, device=None, precision: Optional[str] = None):
        """``precision``: "f32" (exact fp32 MFMA chain, the default and the parity
        reference), "f16x3" (hi/lo f16 split operands on the 16-bit matrix pipe,
        fp32 accumulate: ~2^-21 relative error per product, ~3x the throughput), or -- explicit
        opt-in only, OUTSIDE the 1e-5 parity contract -- "f16" (operands rounded once to f16, one
        MFMA per product block: outputs within 1e-2 of float64 (the tested bound; 7.8e-3 observed on the
        trained config-3 network), no reference fixture pins it; bench.py reports its class-flip
        rate and BER delta beside the exact modes)."""
        _require_cuda()
        precision = _resolve_precision(precision, allow_f16=True)
        self.precision = precision
        self._lib = _cabi.load_library()
        self.device = torch.device("cuda", torch.cuda.current_device()
                                   if device is None else torch.device(device).index or 0)
        Ws = [_as_host_f32(w) for w in weights]
        bs = [_as_host_f32(b).reshape(-1) for b in biases]
        if len(Ws) != len(bs) or not (2 <= len(Ws) <= _cabi.NNV_MAX_LAYERS):
            raise ValueError("need 2..4 dense layers with one bias vector each")
        dims = [Ws[0].shape[0]]
        for W, b in zip(Ws, bs):
            if W.ndim != 2 or W.shape[0] != dims[-1] or b.shape[0] != W.shape[1]:
                raise ValueError("inconsistent layer shapes")
            dims.append(W.shape[1])
        self.dims = dims
        self.n_input = dims[0]
        self.n_classes = dims[-1]
        desc = _cabi.MlpDesc()
        desc.depth = len(Ws)
        for i, d in enumerate(dims):
            desc.dims[i] = d
        for i, (W, b) in enumerate(zip(Ws, bs)):
            desc.W_host[i] = W.ctypes.data
            desc.b_host[i] = b.ctypes.data
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _cabi.check(self._lib.nnv_mlp_prepack_ex(ctypes.byref(desc),
                                                     _cabi.NNV_PRECISION[precision],
                                                     ctypes.byref(handle)), "nnv_mlp_prepack_ex")
        self._handle = handle
        PACK_STATS["mlp"] += 1

    # -- lifetime ---------------------------------------------------------
    def close(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            self._lib.nnv_mlp_free(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ------------------------------------------------------------
    def _check_x(self, x: torch.Tensor):
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise TypeError("x must be a CUDA (HIP) tensor")
        if x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != self.n_input:
            raise ValueError(f"x must be float32 [N, {self.n_input}]")
        if not x.is_contiguous():
            raise ValueError("x must be C-contiguous")
        if x.device != self.device:
            raise ValueError(f"x is on {x.device}, the network on {self.device}")

    def tile_bits(self) -> int:
        return max(1, int(np.ceil(np.log2(self.n_classes))))

    # -- the hot path -------------------------------------------------------
    def decode(self, x: torch.Tensor, labels: Optional[torch.Tensor] = None,
               tile_bits: Optional[int] = None, want_cls32: bool = True,
               want_cls64: bool = False, want_bits: bool = False,
               counters: Optional[torch.Tensor] = None) -> Dict[str, Optional[torch.Tensor]]:
        """One fused launch: classes (+bits, +error counters) for every row.

        ``counters`` (uint64-as-int64 [4] on the device) is ADDED to; if
        ``labels`` is given and ``counters`` is None a zeroed one is created.
        """
        self._check_x(x)
        n = x.shape[0]
        b = self.tile_bits() if tile_bits is None else int(tile_bits)
        if labels is not None:
            if labels.dtype != torch.int32 or labels.shape != (n,) or labels.device != x.device:
                raise ValueError("labels must be int32 [N] on x's device")
            labels = labels.contiguous()
            if counters is None:
                counters = torch.zeros(4, dtype=torch.int64, device=x.device)
        if counters is not None and (counters.dtype != torch.int64 or counters.numel() != 4):
            raise ValueError("counters must be int64 [4]")
        cls32 = torch.empty(n, dtype=torch.int32, device=x.device) if want_cls32 else None
        cls64 = torch.empty(n, dtype=torch.int64, device=x.device) if want_cls64 else None
        bits = torch.empty((n, b), dtype=torch.uint8, device=x.device) if want_bits else None
        with torch.cuda.device(self.device):
            st = self._lib.nnv_mlp_decode_f32(
                self._handle, _ptr(x), n, b, _ptr(labels), _ptr(cls32), _ptr(cls64),
                _ptr(bits), _ptr(counters), _stream_ptr(self.device))
        _cabi.check(st, "nnv_mlp_decode_f32")
        return {"cls32": cls32, "cls64": cls64, "bits": bits, "counters": counters}

    def bound_decode(self, x: torch.Tensor, labels: Optional[torch.Tensor] = None, tile_bits: Optional[int] = None,
                     want_cls64: bool = False, want_bits: bool = False, counters: Optional[torch.Tensor] = None,
                     stream: Optional[torch.cuda.Stream] = None):
        """A decode of FIXED buffers prepared once, for callers that decode many short inputs (serving): every check,
        output allocation and argument conversion of :meth:`decode` happens here, the returned ``call()`` is one C-ABI
        call (``nnv_mlp_decode_f32`` on ``stream``, default: the stream current NOW) -- about 2 us of host time
        instead of ~10.  ``call.outputs`` holds the result tensors (rewritten by every call; ``counters`` is added to);
        refill ``x`` (and ``labels``) in place between calls.  The handle must outlive the callable."""
        r = self.decode(x, labels=labels, tile_bits=tile_bits, want_cls64=want_cls64, want_bits=want_bits, counters=counters)
        if r["counters"] is not None and counters is None:
            r["counters"].zero_()                     # the preparing call above counted once
        n = x.shape[0]
        b = self.tile_bits() if tile_bits is None else int(tile_bits)
        s = torch.cuda.current_stream(self.device) if stream is None else stream
        args = (self._handle, _ptr(x), ctypes.c_int64(n), ctypes.c_int32(b), _ptr(labels), _ptr(r["cls32"]), _ptr(r["cls64"]),
                _ptr(r["bits"]), _ptr(r["counters"]), ctypes.c_void_p(s.cuda_stream))
        fn, keep = self._lib.nnv_mlp_decode_f32, (x, labels, r, s)      # `keep`: the buffers stay alive with the callable

        def call():
            st = fn(*args)
            if st:
                _cabi.check(st, "nnv_mlp_decode_f32")
            return keep[2]

        call.outputs = r
        return call

    def decode_binned(self, x: torch.Tensor, labels: torch.Tensor, n_bins: int,
                      rows_per_block: int, first_row: int = 0,
                      counters: Optional[torch.Tensor] = None, tile_bits: Optional[int] = None,
                      want_cls32: bool = False) -> Dict[str, Optional[torch.Tensor]]:
        """Decode with error counters per SNR bin (mixed-SNR sweep, BASELINE config 4):
        row r -> block (first_row + r) // rows_per_block -> bin block % n_bins.
        ``counters`` int64 [n_bins, 4] is ADDED to (created zeroed if None)."""
        self._check_x(x)
        n = x.shape[0]
        b = self.tile_bits() if tile_bits is None else int(tile_bits)
        if labels.dtype != torch.int32 or labels.shape != (n,) or labels.device != x.device:
            raise ValueError("labels must be int32 [N] on x's device")
        if counters is None:
            counters = torch.zeros((n_bins, 4), dtype=torch.int64, device=x.device)
        if counters.dtype != torch.int64 or tuple(counters.shape) != (n_bins, 4):
            raise ValueError("counters must be int64 [n_bins, 4]")
        cls32 = torch.empty(n, dtype=torch.int32, device=x.device) if want_cls32 else None
        with torch.cuda.device(self.device):
            st = self._lib.nnv_mlp_decode_binned_f32(
                self._handle, _ptr(x), n, b, _ptr(labels.contiguous()), _ptr(cls32), _ptr(counters),
                int(first_row), int(rows_per_block), int(n_bins), _stream_ptr(self.device))
        _cabi.check(st, "nnv_mlp_decode_binned_f32")
        return {"cls32": cls32, "counters": counters}

    def decode_host(self, x_host: np.ndarray, labels_host: Optional[np.ndarray] = None,
                    tile_bits: Optional[int] = None, chunk_rows: int = 0, threads: int = 0):
        """The reference's call shape: a HOST ``test_set`` in, host predictions out
        (``predictions_tensor.eval(feed_dict={x: test_set})``, src/fnn_viterbi.py:610), as one
        synchronous C-ABI call (``nnv_mlp_decode_host_f32``).

        The rows are streamed through a ring of page-locked staging slots owned by the packed
        network (allocated on the first call, reused afterwards): pageable -> pinned copies on
        ``threads`` host threads (0 = the CPUs this process may use, at most 8), H2D DMA, the fused
        decode launch and the D2H of the classes overlap on different chunks of ``chunk_rows`` rows
        (0 = 64 MB worth of rows).
        Returns ``(cls int64 [N], counters int64[4] | None)``."""
        x_host = np.ascontiguousarray(x_host, dtype=np.float32)
        if x_host.ndim != 2 or x_host.shape[1] != self.n_input:
            raise ValueError(f"x must be float32 [N, {self.n_input}]")
        n = x_host.shape[0]
        b = self.tile_bits() if tile_bits is None else int(tile_bits)
        lab_host = None
        if labels_host is not None:
            lab_host = np.ascontiguousarray(labels_host, dtype=np.int32)
            if lab_host.shape != (n,):
                raise ValueError("labels must be [N]")
        cls64 = np.empty(n, dtype=np.int64)
        counters = np.zeros(4, dtype=np.uint64) if lab_host is not None else None
        with torch.cuda.device(self.device):
            st = self._lib.nnv_mlp_decode_host_f32(
                self._handle, ctypes.c_void_p(x_host.ctypes.data), n, b,
                None if lab_host is None else ctypes.c_void_p(lab_host.ctypes.data), None,
                ctypes.c_void_p(cls64.ctypes.data),
                None if counters is None else ctypes.c_void_p(counters.ctypes.data),
                int(max(0, chunk_rows)), int(threads))
        _cabi.check(st, "nnv_mlp_decode_host_f32")
        return cls64, (None if counters is None else counters.astype(np.int64))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """The tanh'd output layer [N, n_classes] (debug / parity; slow)."""
        self._check_x(x)
        out = torch.empty((x.shape[0], self.n_classes), dtype=torch.float32, device=x.device)
        with torch.cuda.device(self.device):
            st = self._lib.nnv_mlp_forward_f32(self._handle, _ptr(x), x.shape[0], _ptr(out),
                                               _stream_ptr(self.device))
        _cabi.check(st, "nnv_mlp_forward_f32")
        return out

    def decode_timed(self, x: torch.Tensor, labels: Optional[torch.Tensor], cls32: torch.Tensor,
                     counters: Optional[torch.Tensor], iters: int,
                     tile_bits: Optional[int] = None) -> float:
        """``iters`` back-to-back launches between two HIP events recorded on
        the launch stream; returns elapsed milliseconds."""
        self._check_x(x)
        b = self.tile_bits() if tile_bits is None else int(tile_bits)
        ms = ctypes.c_float(0.0)
        with torch.cuda.device(self.device):
            st = self._lib.nnv_mlp_decode_f32_timed(
                self._handle, _ptr(x), x.shape[0], b, _ptr(labels), _ptr(cls32),
                _ptr(counters), _stream_ptr(self.device), int(iters), ctypes.byref(ms))
        _cabi.check(st, "nnv_mlp_decode_f32_timed")
        return float(ms.value)

    def set_path(self, path: str) -> None:
        """Pin ``decode`` to one kernel: "persistent" (K1: the network staged into LDS once per workgroup, tiles walked
        wave by wave -- the throughput shape), "split" (K1w: a workgroup of four waves per 32-row tile, weights from L2
        -- the latency shape for short inputs; fp32 networks only, else raises) or
        "auto" (K1w up to 1,024 tiles where it applies, the default).  The two return the same classes bit for bit."""
        _cabi.check(self._lib.nnv_mlp_set_path(self._handle, _cabi.NNV_MLP_PATH[path]), "nnv_mlp_set_path")

    def path_for(self, n_rows: int) -> str:
        """The kernel a ``decode`` of ``n_rows`` rows runs: "persistent" or "split"."""
        v = int(self._lib.nnv_mlp_path_for(self._handle, int(n_rows)))
        return {1: "persistent", 2: "split"}[v]

    def launch_info(self, n_rows: int) -> Dict[str, int]:
        info = (ctypes.c_int32 * 10)()
        with torch.cuda.device(self.device):
            _cabi.check(self._lib.nnv_mlp_launch_info(self._handle, int(n_rows), info),
                        "nnv_mlp_launch_info")
        keys = ("threads", "workgroups", "lds_bytes", "vgprs", "out_layer_in_lds",
                "rows_per_wave_tile", "depth", "n_classes", "precision", "class_tiles_in_lds")
        return dict(zip(keys, [int(v) for v in info]))


class WideMLP:
    """Layer-at-a-time decoder for WIDE networks (C ABI ``nnv_wide_*``): one fused
    GEMM+bias+tanh launch per layer, activations in an HBM workspace.

    Built for the reference's sliding-window decoder (200 -> 500 x4 -> 2,
    src/fnn_viterbi_seq_predict.py:343-414): ``decode_stream`` takes the received
    stream itself and decodes window i = stream[2i : 2i + window] without ever
    materialising the [N, window] matrix the reference builds in Python (:507-519)."""

    def __init__(self, weights: Sequence, biases: Sequence, device=None, chunk_rows: int = 1 << 18,
                 activation: str = "tanh", precision: Optional[str] = None):
        """``activation``: "tanh" (sliding-window decoder) or "relu" (bit-wise decoder,
        src/fnn_viterbi_bitwise.py:388-397), applied after every layer, the class layer included.
        ``precision``: "f32" (exact fp32 MFMA) or "f16x3" (hi/lo f16 operand planes, fp32
        accumulate; class layers of at most 32 classes)."""
        _require_cuda()
        if activation not in _cabi.NNV_ACTIVATION:
            raise ValueError("activation must be 'tanh' or 'relu'")
        explicit = precision is not None
        precision = _resolve_precision(precision)
        if precision == "f16x3" and not explicit and np.asarray(biases[-1]).reshape(-1).shape[0] > 32:
            precision = "f32"     # the split path ends in the narrow head kernel: a DEFAULT must not break wide heads
        self.precision = precision
        self._lib = _cabi.load_library()
        self.device = torch.device("cuda", torch.cuda.current_device()
                                   if device is None else torch.device(device).index or 0)
        Ws = [_as_host_f32(w) for w in weights]
        bs = [_as_host_f32(b).reshape(-1) for b in biases]
        if len(Ws) != len(bs) or not (2 <= len(Ws) <= _cabi.NNV_WIDE_MAX_LAYERS):
            raise ValueError("need 2..8 dense layers with one bias vector each")
        dims = [Ws[0].shape[0]]
        for W, b in zip(Ws, bs):
            if W.ndim != 2 or W.shape[0] != dims[-1] or b.shape[0] != W.shape[1]:
                raise ValueError("inconsistent layer shapes")
            dims.append(W.shape[1])
        self.dims, self.n_input, self.n_classes = dims, dims[0], dims[-1]
        desc = _cabi.WideDesc()
        desc.activation = _cabi.NNV_ACTIVATION[activation]
        desc.precision = _cabi.NNV_PRECISION[precision]
        desc.depth = len(Ws)
        for i, d in enumerate(dims):
            desc.dims[i] = d
        for i, (W, b) in enumerate(zip(Ws, bs)):
            desc.W_host[i] = W.ctypes.data
            desc.b_host[i] = b.ctypes.data
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _cabi.check(self._lib.nnv_wide_prepack(ctypes.byref(desc), ctypes.byref(handle)),
                        "nnv_wide_prepack")
        self._handle = handle
        PACK_STATS["wide"] += 1
        # two activation buffers of chunk_rows x the widest padded layer; a very wide CLASS layer (2^13..2^16 classes)
        # would make that tens of GB, so the chunk shrinks until the workspace is at most 2 GiB (>= 128 rows)
        per_128 = int(self._lib.nnv_wide_workspace_bytes(handle, 128))
        rows = max(128, min(int(chunk_rows), (2 << 30) // per_128 * 128))
        nbytes = int(self._lib.nnv_wide_workspace_bytes(handle, rows))
        self._workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)

    def close(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            self._lib.nnv_wide_free(h)
        self._workspace = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_path(self, path: str) -> None:
        """Pin the handle to one kernel: "fused" (K6, the whole chain in one persistent launch), "layered" (one GEMM
        launch per layer) or "auto" (K6 whenever the shape allows it, the default).  For parity tests and A/B
        timing; "fused" on a shape K6 does not take raises."""
        _cabi.check(self._lib.nnv_wide_set_path(self._handle, _cabi.NNV_WIDE_PATH[path]), "nnv_wide_set_path")

    @property
    def path(self) -> str:
        """The kernel ``decode`` runs on this handle: "fused" or "layered"."""
        v = int(self._lib.nnv_wide_get_path(self._handle))
        return {1: "fused", 2: "layered"}[v]

    def _run(self, x: torch.Tensor, n_rows: int, row_stride: int, want_act: bool, want_cls64: bool):
        cls32 = torch.empty(n_rows, dtype=torch.int32, device=self.device)
        cls64 = torch.empty(n_rows, dtype=torch.int64, device=self.device) if want_cls64 else None
        act = torch.empty((n_rows, self.n_classes), dtype=torch.float32, device=self.device) if want_act else None
        with torch.cuda.device(self.device):
            st = self._lib.nnv_wide_decode_f32(
                self._handle, _ptr(x), int(n_rows), int(row_stride), _ptr(self._workspace),
                int(self._workspace.numel()), _ptr(cls32), _ptr(cls64), _ptr(act), _stream_ptr(self.device))
        _cabi.check(st, "nnv_wide_decode_f32")
        return {"cls32": cls32, "cls64": cls64, "act": act}

    def decode(self, x: torch.Tensor, want_act: bool = False, want_cls64: bool = False):
        """Rows of an ordinary [N, n_input] float32 CUDA matrix."""
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
                and x.shape[1] == self.n_input and x.is_contiguous()):
            raise ValueError(f"x must be a contiguous float32 CUDA tensor [N, {self.n_input}]")
        return self._run(x, x.shape[0], self.n_input, want_act, want_cls64)

    def decode_stream(self, stream: torch.Tensor, n_windows: Optional[int] = None, stride: int = 2,
                      want_act: bool = False, want_cls64: bool = False):
        """Sliding window over a 1-D received stream: window i = stream[stride*i : stride*i + n_input]
        (get_single_instance with offset = 2i, src/fnn_viterbi_seq_predict.py:293-336)."""
        if not (isinstance(stream, torch.Tensor) and stream.is_cuda and stream.dtype == torch.float32
                and stream.dim() == 1 and stream.is_contiguous()):
            raise ValueError("stream must be a contiguous 1-D float32 CUDA tensor")
        most = (stream.numel() - self.n_input) // stride + 1 if stream.numel() >= self.n_input else 0
        n = most if n_windows is None else int(n_windows)
        if n < 0 or n > most:
            raise ValueError(f"the stream holds at most {most} windows")
        return self._run(stream, n, stride, want_act, want_cls64)


class BitwiseBank:
    """P bit-wise decoder networks ``n_input -> n_hidden (ReLU) -> 2 (ReLU) -> argmax`` evaluated on
    the same received rows in ONE fused launch (C ABI ``nnv_bitwise_*``).

    The reference trains one such network per ``bit_position`` and evaluates them one at a time
    (src/fnn_viterbi_bitwise.py:326-399,503-505).  ``models`` is a sequence of
    ``(W1 [n_input, n_hidden], b1 [n_hidden], Wout [n_hidden, 2], bout [2])``, one per position."""

    def __init__(self, models: Sequence, device=None, precision: Optional[str] = None):
        """``precision``: "f32" (exact fp32 MFMA GEMM) or "f16x3" (the stacked first layers as hi/lo f16 planes on
        the 16-bit matrix pipe, fp32 accumulate; the heads stay fp32 on the fp32 accumulators)."""
        _require_cuda()
        explicit = precision is not None
        self.precision = _resolve_precision(precision)
        if self.precision == "f16x3" and not explicit and len(models) < 2:
            # one position is HBM-shaped: fp32 0.25 ms (K4d) / 0.26 ms (32-column tile); f16x3 0.24 ms where its K4d variant
            # applies (n_input <= 1280, 16-byte rows) but 0.42 ms on the 128-column f16x3 tile elsewhere
            self.precision = "f32"
        self._lib = _cabi.load_library()
        self.device = torch.device("cuda", torch.cuda.current_device()
                                   if device is None else torch.device(device).index or 0)
        if len(models) < 1:
            raise ValueError("need at least one network")
        W1 = np.ascontiguousarray(np.stack([_as_host_f32(m[0]) for m in models]))
        b1 = np.ascontiguousarray(np.stack([_as_host_f32(m[1]).reshape(-1) for m in models]))
        Wo = np.ascontiguousarray(np.stack([_as_host_f32(m[2]) for m in models]))
        bo = np.ascontiguousarray(np.stack([_as_host_f32(m[3]).reshape(-1) for m in models]))
        P, K, H = W1.shape
        if b1.shape != (P, H) or Wo.shape != (P, H, 2) or bo.shape != (P, 2):
            raise ValueError("every network must be (W1 [K, H], b1 [H], Wout [H, 2], bout [2]) with one K, H")
        self.n_positions, self.n_input, self.n_hidden = P, K, H
        desc = _cabi.BitwiseDesc()
        desc.n_input, desc.n_hidden, desc.n_positions = K, H, P
        desc.W1_host, desc.b1_host = W1.ctypes.data, b1.ctypes.data
        desc.Wout_host, desc.bout_host = Wo.ctypes.data, bo.ctypes.data
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _cabi.check(self._lib.nnv_bitwise_prepack_ex(ctypes.byref(desc), _cabi.NNV_PRECISION[self.precision],
                                                         ctypes.byref(handle)), "nnv_bitwise_prepack_ex")
        self._handle = handle
        PACK_STATS["bank"] += 1

    def close(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            self._lib.nnv_bitwise_free(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_path(self, path: str) -> None:
        """Pin the handle to one kernel: "tiled" (K4b, the LDS-tiled GEMM), "stream" (K4c: rows resident in LDS, first
        layers streamed from L2; fp32 banks of >= 2 positions), "single" (K4d: one HBM-streaming launch per
        position; banks of <= 2 positions) or "auto" (by size, the default).  A path whose packing the handle does
        not hold raises."""
        _cabi.check(self._lib.nnv_bitwise_set_path(self._handle, _cabi.NNV_BITWISE_PATH[path]), "nnv_bitwise_set_path")

    def path_for(self, x: torch.Tensor) -> str:
        """The kernel ``decode(x)`` runs: "tiled", "stream" or "single" (K4d needs 16-byte aligned rows)."""
        v = int(self._lib.nnv_bitwise_path_for(self._handle, _ptr(x)))
        return {1: "tiled", 2: "stream", 3: "single"}[v]

    def decode(self, x: torch.Tensor, labels: Optional[torch.Tensor] = None, want_act: bool = False,
               errors: Optional[torch.Tensor] = None):
        """``x`` float32 CUDA [N, n_input] -> ``bits`` uint8 [P, N] (row p = the prediction vector of
        network p).  ``labels`` uint8 [P, N] (true bits) adds per-position wrong-decision counts to
        ``errors`` int64 [P] (created zeroed if None); ``want_act`` also returns the ReLU'd outputs
        float32 [P, N, 2] (parity)."""
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
                and x.shape[1] == self.n_input and x.is_contiguous() and x.device == self.device):
            raise ValueError(f"x must be a contiguous float32 tensor [N, {self.n_input}] on {self.device}")
        n, P = x.shape[0], self.n_positions
        bits = torch.empty((P, n), dtype=torch.uint8, device=self.device)
        act = torch.empty((P, n, 2), dtype=torch.float32, device=self.device) if want_act else None
        if labels is not None:
            if not (labels.is_cuda and labels.dtype == torch.uint8 and tuple(labels.shape) == (P, n)
                    and labels.is_contiguous()):
                raise ValueError("labels must be a contiguous uint8 CUDA tensor [P, N]")
            if errors is None:
                errors = torch.zeros(P, dtype=torch.int64, device=self.device)
        elif errors is not None:
            raise ValueError("errors need labels")
        if errors is not None and (errors.dtype != torch.int64 or tuple(errors.shape) != (P,)):
            raise ValueError("errors must be int64 [P]")
        with torch.cuda.device(self.device):
            st = self._lib.nnv_bitwise_decode_f32(self._handle, _ptr(x), n, _ptr(bits), _ptr(act),
                                                  _ptr(labels), _ptr(errors), _stream_ptr(self.device))
        _cabi.check(st, "nnv_bitwise_decode_f32")
        return {"bits": bits, "act": act, "errors": errors}


class Seq2SeqLSTM:
    """The recurrent decoder's forward pass on the device (C ABI ``nnv_seq2seq_*``): LSTM encoder
    over the received symbols, LSTM decoder started from the encoder's final state, linear
    projection to one logit per decoder step (build_seq2seq_model, src/rnn_viterbi.py:235-340;
    TF-1.x ``LSTMCell`` arithmetic, see include/nnviterbi.h).

    ``params``: ``enc_kernel`` [1+H, 4H], ``enc_bias`` [4H], ``dec_kernel``, ``dec_bias``,
    ``proj_w`` [H] or [H, 1], ``proj_b`` [1].  All activations are time-major ``[T, B]`` float32
    CUDA tensors, the layout of the reference's placeholders (:263-279)."""

    def __init__(self, params: Dict[str, np.ndarray], forget_bias: float = 1.0, device=None,
                 precision: Optional[str] = None):
        _require_cuda()
        precision = _resolve_precision(precision)
        self.precision = precision
        self._lib = _cabi.load_library()
        self.device = torch.device("cuda", torch.cuda.current_device()
                                   if device is None else torch.device(device).index or 0)
        a = {k: _as_host_f32(params[k]) for k in
             ("enc_kernel", "enc_bias", "dec_kernel", "dec_bias", "proj_w", "proj_b")}
        H = a["enc_kernel"].shape[1] // 4
        for cell in ("enc", "dec"):
            if a[cell + "_kernel"].shape != (1 + H, 4 * H) or a[cell + "_bias"].reshape(-1).shape != (4 * H,):
                raise ValueError("kernels must be [1 + H, 4H] and biases [4H] with one H for both cells")
        a["proj_w"] = np.ascontiguousarray(a["proj_w"].reshape(-1))
        a["proj_b"] = np.ascontiguousarray(a["proj_b"].reshape(-1))
        if a["proj_w"].shape != (H,) or a["proj_b"].shape != (1,):
            raise ValueError("proj_w must hold H weights and proj_b one bias")
        self.hidden = H
        desc = _cabi.Seq2SeqDesc()
        desc.hidden = H
        desc.forget_bias = float(forget_bias)
        desc.precision = _cabi.NNV_PRECISION[precision]
        for k, v in a.items():
            setattr(desc, k, v.ctypes.data)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _cabi.check(self._lib.nnv_seq2seq_prepack(ctypes.byref(desc), ctypes.byref(handle)),
                        "nnv_seq2seq_prepack")
        self._handle = handle
        PACK_STATS["seq2seq"] += 1

    def close(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            self._lib.nnv_seq2seq_free(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, encoder_inputs: torch.Tensor, decoder_inputs: torch.Tensor,
                info_bits: Optional[torch.Tensor] = None, n_bits: int = 0,
                want_logits: bool = True, want_bits: bool = False,
                counters: Optional[torch.Tensor] = None):
        """``encoder_inputs`` [T_enc, B], ``decoder_inputs`` [T_dec, B].  Optional hard decisions
        (``logit > 0`` over the first ``n_bits`` decoder steps, an extension: the reference's
        argmax over one logit is constantly 0) as uint8 [n_bits, B], and error counters against
        ``info_bits`` uint8 [n_bits, B] accumulated into ``counters`` int64[4]."""
        for name, t in (("encoder_inputs", encoder_inputs), ("decoder_inputs", decoder_inputs)):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2
                    and t.is_contiguous()):
                raise ValueError(f"{name} must be a contiguous float32 CUDA tensor [T, B]")
        if encoder_inputs.shape[1] != decoder_inputs.shape[1]:
            raise ValueError("encoder and decoder inputs disagree on the batch size")
        t_enc, batch = encoder_inputs.shape
        t_dec = decoder_inputs.shape[0]
        n_bits = int(n_bits)
        if info_bits is not None:
            if not (info_bits.is_cuda and info_bits.dtype == torch.uint8 and info_bits.is_contiguous()
                    and tuple(info_bits.shape) == (n_bits, batch)):
                raise ValueError("info_bits must be a contiguous uint8 CUDA tensor [n_bits, B]")
            if counters is None:
                counters = torch.zeros(4, dtype=torch.int64, device=self.device)
        elif counters is not None:
            raise ValueError("counters need info_bits")
        logits = torch.empty((t_dec, batch), dtype=torch.float32, device=self.device) if want_logits else None
        bits = torch.empty((n_bits, batch), dtype=torch.uint8, device=self.device) if want_bits else None
        with torch.cuda.device(self.device):
            st = self._lib.nnv_seq2seq_forward_f32(
                self._handle, _ptr(encoder_inputs), _ptr(decoder_inputs), int(t_enc), int(t_dec),
                int(batch), _ptr(logits), _ptr(bits), _ptr(info_bits), n_bits, _ptr(counters),
                _stream_ptr(self.device))
        _cabi.check(st, "nnv_seq2seq_forward_f32")
        return {"logits": logits, "bits": bits, "counters": counters}


def generate_awgn_blocks(code: str, tile_bits: int, n_rows: int, sigma, seed: int,
                         first_row: int = 0, rows_per_block: int = 1, device=None,
                         want_labels: bool = True, continuous: bool = False):
    """On-device synthetic received blocks (C ABI ``nnv_awgn_generate_f32``).

    Returns ``(x [n_rows, 2*tile_bits] float32, labels [n_rows] int32 | None)``.
    ``sigma`` is a float or a sequence (block i uses sigma[i % len]).  ``continuous``: a block's rows are one
    stream encoded from zero state and chopped (the reference's "random test set",
    src/fnn_viterbi_mixed_snr.py:464-475) instead of every row starting from zero state."""
    _require_cuda()
    lib = _cabi.load_library()
    dev = torch.device("cuda", torch.cuda.current_device() if device is None
                       else torch.device(device).index or 0)
    code_id = {"57": _cabi.NNV_CODE_57, "133171": _cabi.NNV_CODE_133171}[str(code)]
    if continuous:
        code_id |= _cabi.NNV_CODE_CONTINUOUS
    sig = np.atleast_1d(np.asarray(sigma, dtype=np.float32))
    x = torch.empty((n_rows, 2 * tile_bits), dtype=torch.float32, device=dev)
    labels = torch.empty(n_rows, dtype=torch.int32, device=dev) if want_labels else None
    with torch.cuda.device(dev):
        st = lib.nnv_awgn_generate_f32(
            code_id, int(tile_bits), int(n_rows), int(first_row), int(rows_per_block),
            sig.ctypes.data, int(sig.size), ctypes.c_uint64(int(seed)), _ptr(x), _ptr(labels),
            _stream_ptr(dev))
    _cabi.check(st, "nnv_awgn_generate_f32")
    return x, labels


# ---------------------------------------------------------------------------
# Packed networks of the REFERENCE-SHAPED calls.  ``fnn_viterbi.predict(test_set, weights, biases)``,
# ``run_graph(model)``, ``fnn_viterbi_seq_predict.predict_stream`` ... take the weights as host dictionaries on every
# call, like the reference's Session.run takes them from its variables.  Packing + upload + a synchronising free
# around a 0.4 ms decode would dominate an SNR loop, so those modules keep the packed network here, keyed by the
# BYTES of the weights (a 128-bit digest: robust against in-place edits, new array objects over the same values, reloads),
# the precision and the device -- one prepack for the life of the call site's weights, least-recently-used eviction.
# ---------------------------------------------------------------------------
_NET_CACHE: "collections.OrderedDict[tuple, object]" = collections.OrderedDict()
_NET_CACHE_MAX = 4      # a host-decoding PackedMLP also owns its pinned staging ring (256 MB): keep few


try:                                   # 128-bit xxh3: ~7 GB/s (a 66 MB bank in 10 ms); blake2b (~0.6 GB/s) if it is absent
    import xxhash as _xxhash
except ImportError:                    # pragma: no cover - depends on the image
    _xxhash = None


def _content_digest(arrays: Sequence) -> bytes:
    """128-bit digest of the arrays' shapes and float32 bytes.  A cache key, not a security boundary."""
    h = _xxhash.xxh3_128() if _xxhash is not None else hashlib.blake2b(digest_size=16)
    for a in arrays:
        a = np.ascontiguousarray(np.asarray(a), dtype=np.float32)
        h.update(repr(a.shape).encode())
        h.update(a.view(np.uint8).reshape(-1).data)
    return h.digest()


def cached_network(kind: str, arrays: Sequence, build: Callable[[], object], extra: tuple = ()):
    """The packed network of ``kind`` for the host ``arrays`` (weights and biases, in a fixed order): built by
    ``build()`` on first use, then shared by every later call with the same bytes, ``extra`` (precision, activation ...)
    and current device.  The caller must NOT close it; ``clear_pack_cache()`` releases everything."""
    _require_cuda()
    key = (kind, torch.cuda.current_device(), _DEFAULT_PRECISION[0], tuple(extra), _content_digest(arrays))
    net = _NET_CACHE.get(key)
    if net is not None:
        _NET_CACHE.move_to_end(key)
        return net
    while len(_NET_CACHE) >= _NET_CACHE_MAX:
        _, old = _NET_CACHE.popitem(last=False)
        old.close()
    net = build()
    _NET_CACHE[key] = net
    return net


# ---------------------------------------------------------------------------
# torch custom ops:  torch.ops.nnviterbi.mlp_decode / mlp_forward
#   one schema, two dispatch keys (SURVEY.md 8b "Operator"):
#     CUDA (= HIP on ROCm): the fused gfx950 kernel through the C ABI
#     CPU: the same chain in plain torch ops (BASELINE configs[0] "plumbing, no GPU";
#          the reference's own call is a CPU call, src/fnn_viterbi.py:606-610)
# ---------------------------------------------------------------------------
class _PackEntry:
    __slots__ = ("net", "refs", "versions", "snapshot")

    def __init__(self, net, refs, versions, snapshot):
        self.net, self.refs, self.versions, self.snapshot = net, refs, versions, snapshot


# key: (device, precision, per tensor: identity of its STORAGE + offset / shape / stride).  Keying on the
# storage rather than on the tensor object makes the common idioms find their entry: ``[w.detach() for w in ws]``,
# ``.data``, a view of a live parameter -- new tensor objects every call over the same storage (round 2 keyed
# on ``id(tensor)``: every such call re-packed, hipMalloc + H2D + a synchronising hipFree, an order of
# magnitude slower than the decode it wrapped).  Finding the entry is not yet a hit.  An entry is used only
#   (1) while every one of those storages is still ALIVE: it holds a weak reference to each StorageImpl (which
#       also keeps that address from being handed to another storage, so a freed-and-reloaded checkpoint at
#       the same data_ptr cannot find a stale entry),
#   (2) if the version counters of the passed tensors are the ones seen at pack time (a cheap early "changed"),
#   (3) and if the BYTES of the weights still equal the snapshot taken at pack time.  (2) alone is not enough
#       (ADVICE r3): ``p.data`` carries a fresh counter that is always 0, and a write through ``p.data`` does not
#       bump ``p``'s own counter, so ``[p.data for p in params]`` after ``optimizer.step()`` looks unchanged to
#       every counter there is.  The snapshot is one flat int32 copy of all weights on their own device (<= 0.6
#       MB) and the check one ``torch.equal`` on the bit patterns: for device-resident weights that is one small
#       kernel and one host wait per op call (the price of taking raw tensors instead of a handle; hold a
#       ``PackedMLP`` to pay nothing per call).
# Nothing is freed from a weakref callback (a hipFree from the garbage collector, at an arbitrary point of
# another op call): dead entries are dropped at the next call of ``_cached_pack``.
_PACK_CACHE: "collections.OrderedDict[tuple, _PackEntry]" = collections.OrderedDict()
_PACK_CACHE_MAX = 16


def _evict(key) -> None:
    e = _PACK_CACHE.pop(key, None)
    if e is not None:
        e.net.close()


def clear_pack_cache() -> None:
    """Drop every network the raw-weights torch ops and the reference-shaped calls have packed (frees their HBM;
    never needed for correctness: an entry is only used while the weights' bytes equal the ones it was packed from).
    Handles made by ``nnviterbi::mlp_prepack`` belong to their caller and are not touched."""
    for key in list(_PACK_CACHE):
        _evict(key)
    while _NET_CACHE:
        _, net = _NET_CACHE.popitem()
        net.close()


def _sweep_dead_entries() -> None:
    for key in [k for k, e in _PACK_CACHE.items() if any(r.expired() for r in e.refs)]:
        _evict(key)


def _content_snapshot(tensors: List[torch.Tensor]) -> torch.Tensor:
    """All weights and biases as ONE flat tensor of fp32 bit patterns (int32, so NaNs compare equal to
    themselves and -0.0 differs from 0.0), on the device the weights live on."""
    flat = [t.detach().reshape(-1).to(torch.float32) for t in tensors]
    devs = {f.device for f in flat}
    if len(devs) > 1:                                   # mixed placement: compare on the host
        flat = [f.cpu() for f in flat]
    return torch.cat(flat).view(torch.int32)


def _pack_any(weights, biases, device, precision: str):
    """PackedMLP when the register-chained kernel has an instantiation for the shape, else the wide path (K6 for layers
    up to 512 wide, one launch per layer beyond): hidden widths are free parameters of the reference's DecoderConfig.
    Both run on the GPU, same tolerances."""
    wide_precision = precision if len(biases[-1]) <= 32 else "f32"
    if len(weights) > _cabi.NNV_MAX_LAYERS:               # deeper than K1 chains layers: the wide path (up to 8)
        return WideMLP(weights, biases, device=device, precision=wide_precision)
    try:
        return PackedMLP(weights, biases, device=device, precision=precision)
    except NNViterbiError as e:
        if "unsupported network shape" not in str(e):
            raise
        return WideMLP(weights, biases, device=device, precision=wide_precision)


def _cached_pack(weights: List[torch.Tensor], biases: List[torch.Tensor], device):
    from torch.multiprocessing.reductions import StorageWeakRef
    tensors = list(weights) + list(biases)
    precision = _resolve_precision(None)
    _sweep_dead_entries()
    storages = [t.untyped_storage() for t in tensors]
    key = (str(device), precision) + tuple(
        (st._cdata, t.storage_offset(), tuple(t.shape), tuple(t.stride()), str(t.dtype)) for st, t in zip(storages, tensors))
    versions = tuple(t._version for t in tensors)
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        # check (3) waits for the device, which a capturing stream forbids -- and a graph would replay the weights of
        # capture time anyway: say so instead of failing inside torch.equal
        raise RuntimeError("nnviterbi torch ops take raw weight tensors and verify their bytes on every call (a device "
                           "wait): they cannot be captured into a HIP graph. Pack once with "
                           "torch.ops.nnviterbi.mlp_prepack and capture torch.ops.nnviterbi.mlp_decode_packed (pure "
                           "kernel launches on the capturing stream), or hold a decoder.PackedMLP / WideMLP.")
    snapshot = _content_snapshot(tensors)
    e = _PACK_CACHE.get(key)
    if e is not None:
        if e.versions == versions and torch.equal(e.snapshot, snapshot):
            _PACK_CACHE.move_to_end(key)
            return e.net
        _evict(key)                                   # same storages, written in place since
    while len(_PACK_CACHE) >= _PACK_CACHE_MAX:
        _evict(next(iter(_PACK_CACHE)))               # least recently used
    net = _pack_any(weights, biases, device, precision)
    _PACK_CACHE[key] = _PackEntry(net, [StorageWeakRef(st) for st in storages], versions, snapshot)
    return net


def _decode_with(net, x: torch.Tensor, tile_bits: int, labels: Optional[torch.Tensor]):
    """(cls64, bits, counters) from either kind of packed network.  PackedMLP's fused kernel emits all three; the wide
    path emits the classes and the rest is integer work on them (MSB-first bits, src/fnn_viterbi.py:648-651; error
    counters :639,654-671)."""
    if isinstance(net, PackedMLP):
        r = net.decode(x, labels=labels, tile_bits=tile_bits, want_cls32=False, want_cls64=True, want_bits=True)
        counters = r["counters"]
        if counters is None:
            counters = torch.zeros(4, dtype=torch.int64, device=x.device)
        return r["cls64"], r["bits"], counters
    if not x.is_contiguous():
        x = x.contiguous()
    cls = net.decode(x, want_cls64=True)["cls64"]
    shifts = torch.arange(tile_bits - 1, -1, -1, dtype=torch.int64, device=x.device)
    bits = ((cls[:, None] >> shifts[None, :]) & 1).to(torch.uint8)
    if labels is None:
        return cls, bits, torch.zeros(4, dtype=torch.int64, device=x.device)
    # device arithmetic only (fill kernels, no host-to-device copies): the op must stay capturable into a HIP graph
    diff = cls ^ labels.to(torch.int64)
    n = x.shape[0]
    counters = torch.stack([((diff[:, None] >> shifts[None, :]) & 1).sum(), (diff != 0).sum(),
                            torch.full((), n * tile_bits, dtype=torch.int64, device=x.device),
                            torch.full((), n, dtype=torch.int64, device=x.device)])
    return cls, bits, counters


def _forward_torch(x: torch.Tensor, weights: List[torch.Tensor], biases: List[torch.Tensor]) -> torch.Tensor:
    """multilayer_perceptron in plain torch ops, op order of src/fnn_viterbi.py:495-508 per
    layer: matmul, add, tanh (the class layer too).  The CPU dispatch key's arithmetic."""
    h = x
    for W, b in zip(weights, biases):
        h = torch.tanh(torch.add(torch.matmul(h, W), b))
    return h


def _check_op_args(x, weights, biases, tile_bits=None, labels=None):
    if x.dtype != torch.float32 or x.dim() != 2:
        raise ValueError("x must be float32 [N, n_in]")
    if len(weights) != len(biases) or not (2 <= len(weights) <= _cabi.NNV_WIDE_MAX_LAYERS):
        raise ValueError("need 2..8 dense layers with one bias vector each")
    n_prev = x.shape[1]
    for W, b in zip(weights, biases):
        if W.dim() != 2 or W.shape[0] != n_prev or b.dim() != 1 or b.shape[0] != W.shape[1] \
                or W.dtype != torch.float32 or b.dtype != torch.float32:
            raise ValueError("inconsistent layer shapes / dtypes")
        n_prev = W.shape[1]
    if tile_bits is not None and (tile_bits < 1 or tile_bits > 30 or (1 << tile_bits) < n_prev):
        raise ValueError("tile_bits must satisfy 2**tile_bits >= n_classes")
    if labels is not None and (labels.dtype != torch.int32 or tuple(labels.shape) != (x.shape[0],)):
        raise ValueError("labels must be int32 [N] on x's device")


@torch.library.custom_op("nnviterbi::mlp_decode", mutates_args=(), device_types="cuda")
def mlp_decode(x: torch.Tensor, weights: List[torch.Tensor], biases: List[torch.Tensor],
               tile_bits: int, labels: Optional[torch.Tensor] = None
               ) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(cls int64 [N], bits uint8 [N, b], counters int64 [4]) - HIP kernel."""
    _check_op_args(x, weights, biases, tile_bits, labels)
    return _decode_with(_cached_pack(list(weights), list(biases), x.device), x, tile_bits, labels)


@mlp_decode.register_kernel("cpu")
def _mlp_decode_cpu(x, weights, biases, tile_bits, labels=None):
    """CPU dispatch key: forward in torch ops, first-max argmax (src/fnn_viterbi.py:606-610),
    MSB-first bits (:648-651), {bit_errors, frame_errors, total_bits, total_frames} (:639,654-671)."""
    _check_op_args(x, weights, biases, tile_bits, labels)
    n = x.shape[0]
    cls = torch.argmax(_forward_torch(x, list(weights), list(biases)), dim=1) if n else \
        torch.empty(0, dtype=torch.int64)
    shifts = torch.arange(tile_bits - 1, -1, -1, dtype=torch.int64)
    bits = ((cls[:, None] >> shifts[None, :]) & 1).to(torch.uint8)
    counters = torch.zeros(4, dtype=torch.int64)
    if labels is not None:
        diff = cls ^ labels.to(torch.int64)
        counters[0] = ((diff[:, None] >> shifts[None, :]) & 1).sum()
        counters[1] = (diff != 0).sum()
        counters[2] = n * tile_bits
        counters[3] = n
    return cls, bits, counters


@mlp_decode.register_fake
def _(x, weights, biases, tile_bits, labels=None):
    n = x.shape[0]
    return (x.new_empty((n,), dtype=torch.int64), x.new_empty((n, tile_bits), dtype=torch.uint8),
            x.new_empty((4,), dtype=torch.int64))


@torch.library.custom_op("nnviterbi::mlp_forward", mutates_args=(), device_types="cuda")
def mlp_forward(x: torch.Tensor, weights: List[torch.Tensor], biases: List[torch.Tensor]) -> torch.Tensor:
    """multilayer_perceptron(x, weights, biases) (src/fnn_viterbi.py:471-510): the tanh'd class layer
    [N, n_classes] float32 - debug / parity op."""
    _check_op_args(x, weights, biases)
    net = _cached_pack(list(weights), list(biases), x.device)
    if isinstance(net, PackedMLP):
        return net.forward(x)
    return net.decode(x if x.is_contiguous() else x.contiguous(), want_act=True)["act"]


@mlp_forward.register_kernel("cpu")
def _mlp_forward_cpu(x, weights, biases):
    _check_op_args(x, weights, biases)
    return _forward_torch(x, list(weights), list(biases))


@mlp_forward.register_fake
def _(x, weights, biases):
    return x.new_empty((x.shape[0], biases[-1].shape[0]), dtype=torch.float32)


# ---------------------------------------------------------------------------
# Handle-based ops (SURVEY.md 8b "Threading": stream-ordered, re-entrant, no global state besides the handle table):
#   h = torch.ops.nnviterbi.mlp_prepack(weights, biases, "f32")        once: pack + upload (synchronous)
#   cls, bits, counters = torch.ops.nnviterbi.mlp_decode_packed(x, h, tile_bits, labels)
#   out = torch.ops.nnviterbi.mlp_forward_packed(x, h)
#   torch.ops.nnviterbi.mlp_free(h)
# A decode is kernel launches on the CURRENT stream and allocations from torch's caching allocator: no host wait, no
# lookup by content, nothing evicted behind the caller's back -- so it can be captured into a HIP graph and replayed on
# new inputs (tests/test_torch_op.py).  A handle is a plain int: the table below only maps it to the packed network
# its caller created and will free.  The raw-weights ops above stay as the convenience form.
# ---------------------------------------------------------------------------
_HANDLES: Dict[int, object] = {}
_HANDLE_IDS = itertools.count(1)


def _from_handle(handle: int):
    net = _HANDLES.get(int(handle))
    if net is None:
        raise ValueError(f"nnviterbi: {handle} is not a live handle of mlp_prepack (freed, or from another process)")
    return net


@torch.library.custom_op("nnviterbi::mlp_prepack", mutates_args=())
def mlp_prepack(weights: List[torch.Tensor], biases: List[torch.Tensor], precision: str = "f32") -> int:
    """Packs the network (``weights[l]`` [n_l, n_{l+1}], ``biases[l]`` [n_{l+1}], any device) for the CURRENT HIP
    device and returns its handle.  Replaces building the TF variables + graph (src/fnn_viterbi.py:538-545) /
    Saver.restore (mixed_snr:809-811).  ``precision``: "f32", "f16x3", or -- explicit opt-in outside the 1e-5 parity
    contract, register-chained shapes only -- "f16".  The weights are COPIED: later edits of the tensors do not reach
    the handle (pack again)."""
    _require_cuda()
    if len(weights) != len(biases) or not (2 <= len(weights) <= _cabi.NNV_WIDE_MAX_LAYERS):
        raise ValueError("need 2..8 dense layers with one bias vector each")
    precision = _resolve_precision(precision, allow_f16=True)
    dev = torch.device("cuda", torch.cuda.current_device())
    if precision == "f16":
        net = PackedMLP(list(weights), list(biases), device=dev, precision="f16")
    else:
        net = _pack_any(list(weights), list(biases), dev, precision)
    handle = next(_HANDLE_IDS)
    _HANDLES[handle] = net
    return handle


@mlp_prepack.register_fake
def _(weights, biases, precision="f32"):
    raise RuntimeError("nnviterbi::mlp_prepack packs and uploads real weights: call it outside tracing / compilation "
                       "and pass the handle in")


@torch.library.custom_op("nnviterbi::mlp_free", mutates_args=())
def mlp_free(handle: int) -> None:
    """Releases the packed network (a synchronising hipFree: not inside a capture)."""
    net = _HANDLES.pop(int(handle), None)
    if net is not None:
        net.close()


@torch.library.custom_op("nnviterbi::mlp_decode_packed", mutates_args=(), device_types="cuda")
def mlp_decode_packed(x: torch.Tensor, handle: int, tile_bits: int, labels: Optional[torch.Tensor] = None
                      ) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(cls int64 [N], bits uint8 [N, b], counters int64 [4]) like ``mlp_decode``, from a prepacked handle: THE HOT PATH
    (src/fnn_viterbi.py:494-510,606-610,648-651) as launches on the current stream and nothing else."""
    net = _from_handle(handle)
    if x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != net.n_input:
        raise ValueError(f"x must be float32 [N, {net.n_input}]")
    if tile_bits < 1 or tile_bits > 30 or (1 << tile_bits) < net.n_classes:
        raise ValueError("tile_bits must satisfy 2**tile_bits >= n_classes")
    if labels is not None and (labels.dtype != torch.int32 or tuple(labels.shape) != (x.shape[0],)):
        raise ValueError("labels must be int32 [N] on x's device")
    return _decode_with(net, x, tile_bits, labels)


@mlp_decode_packed.register_fake
def _(x, handle, tile_bits, labels=None):
    n = x.shape[0]
    return (x.new_empty((n,), dtype=torch.int64), x.new_empty((n, tile_bits), dtype=torch.uint8),
            x.new_empty((4,), dtype=torch.int64))


@torch.library.custom_op("nnviterbi::mlp_forward_packed", mutates_args=(), device_types="cuda")
def mlp_forward_packed(x: torch.Tensor, handle: int) -> torch.Tensor:
    """multilayer_perceptron(x, weights, biases) (src/fnn_viterbi.py:471-510) from a prepacked handle: the tanh'd class
    layer [N, n_classes] float32 (debug / parity op)."""
    net = _from_handle(handle)
    if x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != net.n_input:
        raise ValueError(f"x must be float32 [N, {net.n_input}]")
    if isinstance(net, PackedMLP):
        return net.forward(x)
    return net.decode(x if x.is_contiguous() else x.contiguous(), want_act=True)["act"]


@mlp_forward_packed.register_fake
def _(x, handle):
    return x.new_empty((x.shape[0], _from_handle(handle).n_classes), dtype=torch.float32)
