"""The C ABI from plain C (examples/decode_c_abi.c): the public header compiles as C11, the program
links against libnnviterbi_hip.so alone (no Python, no torch), and on an MI355X its noiseless round trip
over 1 M blocks is exact; the host-array entry, the device rule and the bit-wise bank are exercised from C too."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "neural-network-based-viterbi-decoding_amd")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _build(tmp_path, name="decode_c_abi"):
    if shutil.which("gcc") is None or not os.path.exists(os.path.join(PKG, "libnnviterbi_hip.so")):
        pytest.skip("gcc or the built library is missing")
    exe = str(tmp_path / name)
    cmd = ["gcc", "-O2", "-std=c11", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__",
           os.path.join(ROOT, "examples", name + ".c"), "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROCM, "include"), "-L" + PKG, "-lnnviterbi_hip", "-L" + os.path.join(ROCM, "lib"),
           "-lamdhip64", "-lm", "-Wl,-rpath," + PKG, "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_example_compiles_as_c11(tmp_path):
    _build(tmp_path)


@pytest.mark.gpu
def test_c_example_round_trip_is_exact(tmp_path, cuda_device):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0].startswith("sigma 0.0000: bit errors 0 / 64000000") and "tile errors 0 / 8000000" in lines[0]
    ber = float(lines[1].split("BER ")[1].split(")")[0])
    assert 1e-3 < ber < 2e-2          # per-tile ML decision of the (133,171) b = 8 tile code at 2 dB
    assert "host entry: 1000007 rows, 0 differ from the resident decode" in r.stdout
    assert "refused: yes" in r.stdout and "bit-wise bank: 3 positions x 5000 rows, 0 wrong" in r.stdout
    assert "bit-wise bank f16x3: 0 wrong" in r.stdout          # nnv_bitwise_prepack_ex from C
    assert "mode f16x3:" in r.stdout and "mode f16:" in r.stdout   # nnv_mlp_prepack_ex, all three arithmetic modes


def test_c_training_example_compiles_as_c11(tmp_path):
    _build(tmp_path, "train_c_abi")


@pytest.mark.gpu
def test_c_training_example_learns(tmp_path, cuda_device):
    """examples/train_c_abi.c: the reference's network trained through nnv_train_* (K7) from plain C on device-generated
    blocks -- the cost falls, the frame errors of the decode fall at least fourfold, K1 and K1w return the same classes
    -- and the bit-wise decoder's recipe through nnv_dense_train_* (K8)."""
    exe = _build(tmp_path, "train_c_abi")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().splitlines()[-1] == "ok" and "K1 == K1w" in r.stdout and "K8:" in r.stdout
