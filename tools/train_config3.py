#!/usr/bin/env python3
"""Train the BASELINE configs[2] network (b=8 tiles of the (133,171) code at 2 dB,
16-128-64-32-256) with the reference's recipe (src/fnn_viterbi.py:537-603:
Adam 1e-4, batch 256, CE on the tanh outputs + 0.001*L2, class-major unshuffled
batches) and save it under the package as an .npz keyed by the reference's
variable names.  Usage (GPU box):  python tools/train_config3.py [epochs]"""
import contextlib, io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd
from nnviterbi_amd import fnn_viterbi as fv, fnn_viterbi_mixed_snr as mx
from nnviterbi_amd.training import train_network

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/fnn_133171_b8_2dB"
cfg = fv.DecoderConfig(block_length=8, snr_db=2.0, training_epochs=epochs,
                       samples_per_class_train=500, samples_per_class_test=2000)
np.random.seed(20240607)
tr, trl, te, tel, cb = fv.train_test_dataset(cfg)
w0, b0 = fv.initialize_weights(cfg)
t0 = time.time()
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    w, b = train_network(cfg, tr, trl, w0, b0)
lines = buf.getvalue().strip().splitlines()
print(lines[0], "...", lines[-1], f"({time.time() - t0:.0f} s)")
labels = np.argmax(tel, axis=1).astype(np.int32)
pred, counters = fv.predict(te, w, b, labels=labels, tile_bits=8)
print("test rows", len(pred), "accuracy", 1 - counters[1] / counters[3], "BER", counters[0] / counters[2])
print("saved", mx.save_model(out, w, b))
