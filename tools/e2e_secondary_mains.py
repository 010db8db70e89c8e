#!/usr/bin/env python3
"""The reference's other three script flows at their DEFAULT configurations through the mirrored modules, wall time per
flow (src/fnn_viterbi_bitwise.py:512-570, src/fnn_viterbi_seq_predict.py:619-657, src/rnn_viterbi.py:505-557).
Usage: python tools/e2e_secondary_mains.py [bitwise|seq|rnn ...]"""
import os, sys, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nnviterbi_amd  # noqa: F401

which = sys.argv[1:] or ["bitwise", "seq", "rnn"]
for name in which:
    mod = {"bitwise": "fnn_viterbi_bitwise", "seq": "fnn_viterbi_seq_predict", "rnn": "rnn_viterbi"}[name]
    m = __import__("nnviterbi_amd." + mod, fromlist=["main"])
    np.random.seed(0)
    buf = io.StringIO()
    import cProfile, pstats
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(buf):
        pr.enable()
        m.main()
        pr.disable()
    dt = time.perf_counter() - t0
    lines = [l for l in buf.getvalue().splitlines() if l.strip()]
    print(f"{name}: main() {dt:.1f} s, {len(lines)} lines printed; last: {lines[-6:]}", flush=True)
    st = pstats.Stats(pr)
    own = [(v[3], k[2], k[0].split("/")[-1]) for k, v in st.stats.items() if "neural-network-based-viterbi-decoding_amd" in k[0]]
    for cum, fn, fl in sorted(own, reverse=True)[:8]:
        print(f"    {cum:8.2f} s cumulative  {fl}:{fn}", flush=True)
