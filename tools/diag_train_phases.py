#!/usr/bin/env python3
"""Where one K7 step's time goes: 100 MHz stamps of block 0 (four waves, every one on the chain) from a `-DNNV_DIAG_TRAIN` side library (tools/build_variant.sh diagtrain -DNNV_DIAG_TRAIN), loaded through
NNV_LIB.  Prints microseconds since the kernel's first stamp.  Usage (on the GPU box):
    NNV_LIB=$PWD/neural-network-based-viterbi-decoding_amd/ab_diagtrain.so python tools/diag_train_phases.py [b]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd  # noqa: F401
from nnviterbi_amd import fnn_viterbi as fv, training, _cabi

b = int(sys.argv[1]) if len(sys.argv) > 1 else 7
cfg = fv.DecoderConfig(block_length=b)
w, bd = fv.initialize_weights(cfg, rng=np.random.RandomState(0))
bs, steps = 256, 250
x = torch.randn((steps * bs, 2 * b), device="cuda")
lab = torch.randint(0, 2 ** b, (steps * bs,), device="cuda", dtype=torch.int32)
nt = training.NativeFNNTrainer(w, bd, cfg.learning_rate, cfg.beta_regularization, bs)
lib = _cabi.load_library()
rd = lib.nnv_train_diag_read
rd.restype, rd.argtypes = ctypes.c_int32, [ctypes.c_void_p]
names = {0: "start", 8: "in front of barrier 1 (layer 1 tile w)", 1: "behind barrier 1",
         9: "in front of barrier 2 (layer 2, waves 0/1)", 2: "behind barrier 2",
         10: "in front of barrier 3 (layer 3, class tiles, exp)", 3: "behind barrier 3",
         11: "in front of barrier 4 (dZ4, share of dH3)", 4: "behind barrier 4",
         12: "in front of barrier 5 (dZ3; dZ2 | dW4)", 5: "behind barrier 5", 6: "end (dZ1, gradients stored)"}
order = [0, 8, 1, 9, 2, 10, 3, 11, 4, 12, 5, 6]
acc = np.zeros(64)
reps = 20
for _ in range(reps):
    nt.steps(x, lab, steps)
    buf = (ctypes.c_uint64 * 64)()
    assert rd(buf) == 0
    v = np.array(buf[:], dtype=np.float64)
    acc += (v - v[0]) / 100.0          # 100 MHz -> us
acc /= reps
print(f"b = {b}: microseconds after wave 0's first stamp, block 0, mean of {reps} last-steps")
print("  " + " " * 52 + "".join(f"wave {w:<5d}" for w in range(4)))
for k in order:
    print(f"  {names[k]:50s}" + "".join(f"{acc[16 * w + k]:9.2f} " for w in range(4)))
