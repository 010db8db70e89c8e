#!/usr/bin/env python3
"""The reference's default script flow (src/fnn_viterbi.py:681-749: DecoderConfig(), 64 k training rows,
200 epochs, 1.28 M test rows) through the mirrored API, timed per stage.  Usage: python tools/e2e_default_main.py"""
import os, sys, time, io, contextlib
sys.path.insert(0, os.getcwd())
import numpy as np
import nnviterbi_amd
from nnviterbi_amd import fnn_viterbi as fv
np.random.seed(0)
cfg = fv.DecoderConfig()
t0 = time.perf_counter()
tr, trl, te, tel, cb = fv.train_test_dataset(cfg)
t1 = time.perf_counter()
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    pred = fv.run_network(cfg, tr, trl, te, tel)
t2 = time.perf_counter()
out = io.StringIO()
with contextlib.redirect_stdout(out):
    fv.calculate_error(tel, pred, cb, cfg)
t3 = time.perf_counter()
# the decode alone, host test_set in, host predictions out (what run_network's tail does)
from nnviterbi_amd.decoder import PackedMLP
print("dataset %.2f s | run_network (train %d epochs + decode + accuracy) %.2f s | calculate_error %.2f s"
      % (t1 - t0, cfg.training_epochs, t2 - t1, t3 - t2))
print(buf.getvalue().strip().splitlines()[-3:])
print(out.getvalue().strip())
