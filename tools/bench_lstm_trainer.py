#!/usr/bin/env python3
"""SURVEY section 8f rank 4, the training side: one training step of the recurrent decoder at the reference's default
shape (LSTM(20) encoder over 200 coded symbols, decoder over 200 steps, batch 100: src/rnn_viterbi.py:28-92) with both
recurrences in K9 (csrc/nnv_lstm_train.hip) and through the unrolled PyTorch graph, each replayed from a HIP graph,
milliseconds per step.  One JSON line.  Usage: python tools/bench_lstm_trainer.py"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnviterbi_amd  # noqa: F401
from nnviterbi_amd import rnn_viterbi as rv, training

cfg = rv.RNNDecoderConfig()
model = rv.build_seq2seq_model(cfg, rng=np.random.RandomState(0))
T, B = cfg.encoded_length, cfg.batch_size
g = torch.Generator(device="cuda").manual_seed(0)
enc = torch.randn((T, B), device="cuda", generator=g)
dec = torch.randint(0, 2, (T, B), device="cuda", generator=g).float()
tgt = torch.randn((T, B, 1), device="cuda", generator=g)
out = {"t_enc": T, "t_dec": T, "batch": B, "hidden": cfg.encoder_hidden_units}
for name, fwd, n in (("K9", training.lstm_seq2seq_logits_native, 200), ("unrolled_autograd", rv._torch_forward, 20)):
    P = {k: torch.tensor(v, device="cuda", requires_grad=True) for k, v in model["params"].items()}
    opt = training.TFAdam(list(P.values()), lr=cfg.learning_rate)
    def step(e, d, t):
        loss = ((t - fwd(P, e, d, 1.0)) ** 2).mean()
        opt.zero_grad(set_to_none=False)
        loss.backward()
        opt.step()
        return loss.detach()
    run = training.GraphedStep(step, list(P.values()), opt, (enc, dec, tgt), False)
    for _ in range(3):
        run(enc, dec, tgt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run(enc, dec, tgt)
    torch.cuda.synchronize()
    out[name] = {"ms_per_step": (time.perf_counter() - t0) / n * 1e3, "graph_captured": run.captured}
print(json.dumps(out))
