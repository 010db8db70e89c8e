#!/usr/bin/env python3
"""Probe for docs/HISTORY.md section 9.5's untried host-path route: instead of staging the caller's pageable rows through
pinned slots, page-lock (hipHostRegister) the caller's own pages chunk by chunk on helper threads, one chunk
ahead of the DMA.  That only pays if pages can be locked AND unlocked faster than the link moves them
(56.5 GB/s bare pinned H2D).  Measures, on the 5.12 GB configs[2] input: register / unregister GB/s of 64 MB
chunks on 1, 2, 4, 8 threads, and the H2D rate out of registered memory.  Prints one JSON line."""
import ctypes, json, os, sys, threading, time
import numpy as np, torch

torch.cuda.init()
hip = ctypes.CDLL("libamdhip64.so")          # the runtime torch already mapped
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} -> hipError {rc}")

n_rows, n_in = 80_000_000, 16
x = np.empty((n_rows, n_in), np.float32)
x[:] = 1.0                                    # touch every page (first-touch allocation is not what is measured)
base = x.ctypes.data
total = x.nbytes
chunk = 64 << 20
page = 4096
start = (base + page - 1) // page * page
chunks = [(a, min(chunk, base + total - a) // page * page) for a in range(start, base + total - page, chunk)]
out = {"bytes": total, "chunk_bytes": chunk, "chunks": len(chunks), "thp": open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip()}


def run(threads, fn):
    t0 = time.perf_counter()
    def work(t):
        for a, sz in chunks[t::threads]:
            fn(a, sz)
    ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    [t.start() for t in ths]; [t.join() for t in ths]
    return total / (time.perf_counter() - t0) / 1e9


for threads in (1, 2, 4, 8):
    reg = run(threads, lambda a, sz: check(hip.hipHostRegister(a, sz, 0), 'hipHostRegister'))
    if threads == 1:
        d = torch.empty(chunk // 4, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a, sz in chunks:
            check(hip.hipMemcpyAsync(d.data_ptr(), a, sz, 1, torch.cuda.current_stream().cuda_stream), "hipMemcpyAsync")
        torch.cuda.synchronize()
        out["h2d_from_registered_GBps"] = total / (time.perf_counter() - t0) / 1e9
        t0 = time.perf_counter()
        for a, sz in chunks:
            check(hip.hipMemcpyAsync(d.data_ptr(), a, sz, 1, torch.cuda.current_stream().cuda_stream), "hipMemcpyAsync")
        torch.cuda.synchronize()
        out["h2d_from_registered_second_pass_GBps"] = total / (time.perf_counter() - t0) / 1e9
        ph = torch.empty(chunk // 4, dtype=torch.float32).pin_memory()
        d.copy_(ph, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(len(chunks)):
            d.copy_(ph, non_blocking=True)
        torch.cuda.synchronize()
        out["h2d_from_hipHostMalloc_GBps"] = chunk * len(chunks) / (time.perf_counter() - t0) / 1e9
    unreg = run(threads, lambda a, sz: check(hip.hipHostUnregister(a), 'hipHostUnregister'))
    out[f"threads_{threads}"] = {"register_GBps": reg, "unregister_GBps": unreg,
                                 "both_GBps": 1.0 / (1.0 / reg + 1.0 / unreg)}
print(json.dumps(out))
