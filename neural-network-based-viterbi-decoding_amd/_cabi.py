"""ctypes binding of the C ABI declared in include/nnviterbi.h.

This is the only place the package touches native code.  There is no CPU
fallback: if ``libnnviterbi_hip.so`` is missing or no gfx950 device is
present, the compute entry points raise ``NNViterbiError``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# NNV_LIB: A/B a differently-built library (tuning only); default = the in-tree build
_LIB_PATH = os.environ.get("NNV_LIB") or os.path.join(_PKG_DIR, "libnnviterbi_hip.so")
_CSRC = os.path.join(_PKG_DIR, "csrc")

NNV_ABI_VERSION = 11
NNV_MAX_LAYERS = 4
NNV_OK = 0
NNV_CODE_57 = 0
NNV_CODE_133171 = 1
NNV_CODE_CONTINUOUS = 0x100
NNV_PRECISION = {"f32": 0, "f16x3": 1, "f16": 2}   # "f16": PackedMLP only (NNV_PRECISION_F16, outside the parity contract)

EXPORTED_SYMBOLS = (
    "nnv_abi_version", "nnv_status_string", "nnv_device_count", "nnv_check_device",
    "nnv_mlp_prepack", "nnv_mlp_prepack_ex", "nnv_mlp_free", "nnv_mlp_decode_f32", "nnv_mlp_decode_binned_f32",
    "nnv_mlp_decode_host_f32",
    "nnv_mlp_forward_f32", "nnv_awgn_generate_f32",
    "nnv_mlp_decode_f32_timed", "nnv_mlp_launch_info",
    "nnv_wide_prepack", "nnv_wide_free", "nnv_wide_workspace_bytes", "nnv_wide_decode_f32",
    "nnv_wide_set_path", "nnv_wide_get_path", "nnv_bitwise_set_path", "nnv_bitwise_path_for",
    "nnv_train_create", "nnv_train_free", "nnv_train_steps", "nnv_train_gradients", "nnv_train_export",
    "nnv_dense_train_create", "nnv_dense_train_free", "nnv_dense_train_steps", "nnv_dense_train_gradients",
    "nnv_dense_train_export",
    "nnv_lstm_seq_forward_f32", "nnv_lstm_seq_backward_f32",
    "nnv_mlp_set_path", "nnv_mlp_path_for",
    "nnv_seq2seq_prepack", "nnv_seq2seq_free", "nnv_seq2seq_forward_f32",
    "nnv_bitwise_prepack", "nnv_bitwise_prepack_ex", "nnv_bitwise_free", "nnv_bitwise_decode_f32",
)


class NNViterbiError(RuntimeError):
    """A C-ABI call failed, or the HIP library is not available."""


NNV_WIDE_MAX_LAYERS = 8


class WideDesc(ctypes.Structure):
    _fields_ = [
        ("depth", ctypes.c_int32),
        ("dims", ctypes.c_int32 * (NNV_WIDE_MAX_LAYERS + 1)),
        ("W_host", ctypes.c_void_p * NNV_WIDE_MAX_LAYERS),
        ("b_host", ctypes.c_void_p * NNV_WIDE_MAX_LAYERS),
        ("activation", ctypes.c_int32),
        ("precision", ctypes.c_int32),
    ]


NNV_ACTIVATION = {"tanh": 0, "relu": 1}
NNV_MLP_PATH = {"auto": 0, "persistent": 1, "split": 2}
NNV_WIDE_PATH = {"auto": 0, "fused": 1, "layered": 2}                 # nnv_wide_path
NNV_BITWISE_PATH = {"auto": 0, "tiled": 1, "stream": 2, "single": 3}  # nnv_bitwise_path


class Seq2SeqDesc(ctypes.Structure):
    _fields_ = [
        ("hidden", ctypes.c_int32),
        ("forget_bias", ctypes.c_float),
        ("precision", ctypes.c_int32),
        ("enc_kernel", ctypes.c_void_p),
        ("enc_bias", ctypes.c_void_p),
        ("dec_kernel", ctypes.c_void_p),
        ("dec_bias", ctypes.c_void_p),
        ("proj_w", ctypes.c_void_p),
        ("proj_b", ctypes.c_void_p),
    ]


class BitwiseDesc(ctypes.Structure):
    _fields_ = [
        ("n_input", ctypes.c_int32),
        ("n_hidden", ctypes.c_int32),
        ("n_positions", ctypes.c_int32),
        ("W1_host", ctypes.c_void_p),
        ("b1_host", ctypes.c_void_p),
        ("Wout_host", ctypes.c_void_p),
        ("bout_host", ctypes.c_void_p),
    ]


class TrainDesc(ctypes.Structure):
    _fields_ = [
        ("n_input", ctypes.c_int32),
        ("n_classes", ctypes.c_int32),
        ("batch_size", ctypes.c_int32),
        ("n_hidden", ctypes.c_int32 * 3),
        ("learning_rate", ctypes.c_float),
        ("beta_regularization", ctypes.c_float),
        ("keep_prob", ctypes.c_float),
        ("W_host", ctypes.c_void_p * 4),
        ("b_host", ctypes.c_void_p * 4),
    ]


class DenseTrainDesc(ctypes.Structure):
    _fields_ = [
        ("n_layers", ctypes.c_int32),
        ("dims", ctypes.c_int32 * 6),
        ("activation", ctypes.c_int32),
        ("batch_size", ctypes.c_int32),
        ("learning_rate", ctypes.c_float),
        ("beta_regularization", ctypes.c_float),
        ("reg_layer_mask", ctypes.c_uint32),
        ("W_host", ctypes.c_void_p * 5),
        ("b_host", ctypes.c_void_p * 5),
        ("loss", ctypes.c_int32),
        ("extra_host", ctypes.c_void_p),
        ("n_extra", ctypes.c_int64),
    ]


class MlpDesc(ctypes.Structure):
    _fields_ = [
        ("depth", ctypes.c_int32),
        ("dims", ctypes.c_int32 * (NNV_MAX_LAYERS + 1)),
        ("W_host", ctypes.c_void_p * NNV_MAX_LAYERS),
        ("b_host", ctypes.c_void_p * NNV_MAX_LAYERS),
    ]


_lib = None


def library_path() -> str:
    return _LIB_PATH


def build_library(force: bool = False, jobs: int = 4) -> str:
    """Compile csrc/*.hip for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", _CSRC, f"-j{jobs}", "-s"]
    if force:
        args.append("-B")
    subprocess.run(args, check=True)
    return _LIB_PATH


def load_library() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's):
    # if this library were loaded first it would pull in the system copy, torch would then load its own, and
    # the two runtimes do not share a device (measured on the MI355X box: torch.cuda.is_available() turns
    # False, or nnv_mlp_prepack reports "no gfx950 HIP device").  Importing torch first makes the dynamic
    # loader resolve our dependency to the copy that is already mapped.  (A torch-free C or ctypes caller,
    # INTEGRATION.md, uses the system runtime alone and is not affected.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(_LIB_PATH):
        raise NNViterbiError(
            f"{_LIB_PATH} is missing: build it with __graft_entry__.build() "
            "(make -C neural-network-based-viterbi-decoding_amd/csrc). "
            "There is no CPU fallback for the decode path.")
    lib = ctypes.CDLL(_LIB_PATH)
    c = ctypes
    lib.nnv_abi_version.restype = c.c_int32
    lib.nnv_abi_version.argtypes = []
    lib.nnv_status_string.restype = c.c_char_p
    lib.nnv_status_string.argtypes = [c.c_int32]
    lib.nnv_device_count.restype = c.c_int32
    lib.nnv_device_count.argtypes = []
    lib.nnv_check_device.restype = c.c_int32
    lib.nnv_check_device.argtypes = [c.c_int32]
    lib.nnv_mlp_prepack.restype = c.c_int32
    lib.nnv_mlp_prepack.argtypes = [c.POINTER(MlpDesc), c.POINTER(c.c_void_p)]
    lib.nnv_mlp_prepack_ex.restype = c.c_int32
    lib.nnv_mlp_prepack_ex.argtypes = [c.POINTER(MlpDesc), c.c_int32, c.POINTER(c.c_void_p)]
    lib.nnv_mlp_free.restype = c.c_int32
    lib.nnv_mlp_free.argtypes = [c.c_void_p]
    lib.nnv_mlp_decode_f32.restype = c.c_int32
    lib.nnv_mlp_decode_f32.argtypes = [
        c.c_void_p, c.c_void_p, c.c_int64, c.c_int32, c.c_void_p, c.c_void_p,
        c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]
    lib.nnv_mlp_decode_host_f32.restype = c.c_int32
    lib.nnv_mlp_decode_host_f32.argtypes = [
        c.c_void_p, c.c_void_p, c.c_int64, c.c_int32, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p,
        c.c_int64, c.c_int32]
    lib.nnv_mlp_decode_binned_f32.restype = c.c_int32
    lib.nnv_mlp_decode_binned_f32.argtypes = [
        c.c_void_p, c.c_void_p, c.c_int64, c.c_int32, c.c_void_p, c.c_void_p,
        c.c_void_p, c.c_int64, c.c_int64, c.c_int32, c.c_void_p]
    lib.nnv_mlp_forward_f32.restype = c.c_int32
    lib.nnv_mlp_forward_f32.argtypes = [
        c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p]
    lib.nnv_awgn_generate_f32.restype = c.c_int32
    lib.nnv_awgn_generate_f32.argtypes = [
        c.c_int32, c.c_int32, c.c_int64, c.c_int64, c.c_int64, c.c_void_p,
        c.c_int32, c.c_uint64, c.c_void_p, c.c_void_p, c.c_void_p]
    lib.nnv_mlp_decode_f32_timed.restype = c.c_int32
    lib.nnv_mlp_decode_f32_timed.argtypes = [
        c.c_void_p, c.c_void_p, c.c_int64, c.c_int32, c.c_void_p, c.c_void_p,
        c.c_void_p, c.c_void_p, c.c_int32, c.POINTER(c.c_float)]
    lib.nnv_mlp_launch_info.restype = c.c_int32
    lib.nnv_mlp_launch_info.argtypes = [c.c_void_p, c.c_int64, c.POINTER(c.c_int32)]
    lib.nnv_wide_prepack.restype = c.c_int32
    lib.nnv_wide_prepack.argtypes = [c.POINTER(WideDesc), c.POINTER(c.c_void_p)]
    lib.nnv_wide_free.restype = c.c_int32
    lib.nnv_wide_free.argtypes = [c.c_void_p]
    lib.nnv_wide_workspace_bytes.restype = c.c_int64
    lib.nnv_wide_workspace_bytes.argtypes = [c.c_void_p, c.c_int64]
    lib.nnv_wide_decode_f32.restype = c.c_int32
    lib.nnv_wide_decode_f32.argtypes = [
        c.c_void_p, c.c_void_p, c.c_int64, c.c_int64, c.c_void_p, c.c_int64,
        c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]
    lib.nnv_seq2seq_prepack.restype = c.c_int32
    lib.nnv_seq2seq_prepack.argtypes = [c.POINTER(Seq2SeqDesc), c.POINTER(c.c_void_p)]
    lib.nnv_seq2seq_free.restype = c.c_int32
    lib.nnv_seq2seq_free.argtypes = [c.c_void_p]
    lib.nnv_seq2seq_forward_f32.restype = c.c_int32
    lib.nnv_seq2seq_forward_f32.argtypes = [
        c.c_void_p, c.c_void_p, c.c_void_p, c.c_int32, c.c_int32, c.c_int64,
        c.c_void_p, c.c_void_p, c.c_void_p, c.c_int32, c.c_void_p, c.c_void_p]
    lib.nnv_bitwise_prepack.restype = c.c_int32
    lib.nnv_bitwise_prepack.argtypes = [c.POINTER(BitwiseDesc), c.POINTER(c.c_void_p)]
    lib.nnv_bitwise_prepack_ex.restype = c.c_int32
    lib.nnv_bitwise_prepack_ex.argtypes = [c.POINTER(BitwiseDesc), c.c_int32, c.POINTER(c.c_void_p)]
    lib.nnv_bitwise_free.restype = c.c_int32
    lib.nnv_bitwise_free.argtypes = [c.c_void_p]
    lib.nnv_bitwise_decode_f32.restype = c.c_int32
    lib.nnv_bitwise_decode_f32.argtypes = [
        c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]
    lib.nnv_wide_set_path.restype = c.c_int32
    lib.nnv_wide_set_path.argtypes = [c.c_void_p, c.c_int32]
    lib.nnv_wide_get_path.restype = c.c_int32
    lib.nnv_wide_get_path.argtypes = [c.c_void_p]
    lib.nnv_bitwise_set_path.restype = c.c_int32
    lib.nnv_bitwise_set_path.argtypes = [c.c_void_p, c.c_int32]
    lib.nnv_bitwise_path_for.restype = c.c_int32
    lib.nnv_bitwise_path_for.argtypes = [c.c_void_p, c.c_void_p]
    lib.nnv_train_create.restype = c.c_int32
    lib.nnv_train_create.argtypes = [c.POINTER(TrainDesc), c.POINTER(c.c_void_p)]
    lib.nnv_train_free.restype = c.c_int32
    lib.nnv_train_free.argtypes = [c.c_void_p]
    lib.nnv_train_steps.restype = c.c_int32
    lib.nnv_train_steps.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int32, c.c_void_p, c.c_void_p, c.c_void_p]
    lib.nnv_train_gradients.restype = c.c_int32
    lib.nnv_train_gradients.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]
    lib.nnv_train_export.restype = c.c_int32
    lib.nnv_train_export.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]
    lib.nnv_dense_train_create.restype = c.c_int32
    lib.nnv_dense_train_create.argtypes = [c.POINTER(DenseTrainDesc), c.POINTER(c.c_void_p)]
    lib.nnv_dense_train_free.restype = c.c_int32
    lib.nnv_dense_train_free.argtypes = [c.c_void_p]
    lib.nnv_dense_train_steps.restype = c.c_int32
    lib.nnv_dense_train_steps.argtypes = [c.c_void_p, c.c_void_p, c.c_int64, c.c_int64, c.c_void_p, c.c_int32, c.c_void_p, c.c_void_p]
    lib.nnv_dense_train_gradients.restype = c.c_int32
    lib.nnv_dense_train_gradients.argtypes = [c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]
    lib.nnv_dense_train_export.restype = c.c_int32
    lib.nnv_dense_train_export.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]
    lib.nnv_lstm_seq_forward_f32.restype = c.c_int32
    lib.nnv_lstm_seq_forward_f32.argtypes = [c.c_void_p] * 4 + [c.c_int32] * 3 + [c.c_float] + [c.c_void_p] * 4
    lib.nnv_lstm_seq_backward_f32.restype = c.c_int32
    lib.nnv_lstm_seq_backward_f32.argtypes = [c.c_void_p] * 7 + [c.c_int32] * 3 + [c.c_void_p] * 4
    lib.nnv_mlp_set_path.restype = c.c_int32
    lib.nnv_mlp_set_path.argtypes = [c.c_void_p, c.c_int32]
    lib.nnv_mlp_path_for.restype = c.c_int32
    lib.nnv_mlp_path_for.argtypes = [c.c_void_p, c.c_int64]
    if lib.nnv_abi_version() != NNV_ABI_VERSION:
        raise NNViterbiError(
            f"ABI mismatch: library {lib.nnv_abi_version()} != binding {NNV_ABI_VERSION}")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != NNV_OK:
        msg = load_library().nnv_status_string(status).decode()
        raise NNViterbiError(f"{what} failed: {msg} (status {status})")
