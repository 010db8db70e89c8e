#!/usr/bin/env python3
"""The reference's mixed-SNR script flow (src/fnn_viterbi_mixed_snr.py:852-930: MixedSNRDecoderConfig(), 128 k training
rows over 12 SNRs, 300 epochs, 1.28 M test rows) through the mirrored API, timed per stage, without and with the dropout
network (src/fnn_viterbi_mixed_snr.py:596-627).  Usage: python tools/e2e_mixed_snr_main.py"""
import os, sys, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nnviterbi_amd  # noqa: F401
from nnviterbi_amd import fnn_viterbi_mixed_snr as ms

for drop in (False, True):
    np.random.seed(0)
    cfg = ms.MixedSNRDecoderConfig(use_dropout=drop)
    t0 = time.perf_counter()
    tr, te, trl, rtl, rts, _ = ms.get_dataset(cfg)
    t1 = time.perf_counter()
    onehot = ms.create_one_hot_labels(ms.get_dataset.last_test_labels, cfg.n_classes)
    t2 = time.perf_counter()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ms.create_model_graph(config=cfg, training_set=tr, training_labels=trl, test_set=te, test_labels=onehot,
                              random_test_set=rts, random_test_labels=rtl, train_mode=True, model_save_path=None)
    t3 = time.perf_counter()
    lines = buf.getvalue().strip().splitlines()
    print(f"dropout={drop}: get_dataset {t1 - t0:.2f} s | one-hot test labels {t2 - t1:.2f} s | create_model_graph "
          f"(train {cfg.training_epochs} epochs x {cfg.num_training_samples // cfg.batch_size} steps + evaluate) {t3 - t2:.2f} s")
    print("   ", [l for l in lines if l.startswith("Epoch")][-1], "|", lines[-2], "|", lines[-1])
