#!/usr/bin/env python3
"""Probe (run in a child process: a fault here must not take anything else down): can host threads store straight
into hipMalloc'd device memory through the PCIe BAR, and how fast?  That would take the pinned staging copy (and
two of the three host-DRAM passes) out of the host-array path.  Prints one JSON line."""
import ctypes, json, subprocess, sys, threading, time

CHILD = r'''
import ctypes, json, sys, threading, time
import numpy as np, torch
torch.cuda.init()
n = 1 << 28                                   # 256 MB
d = torch.empty(n // 4, dtype=torch.float32, device="cuda")
h = np.ones(n // 4, np.float32)
torch.cuda.synchronize()
print("ALLOC_OK", flush=True)
ctypes.memmove(d.data_ptr(), h.ctypes.data, 4096)      # faults here if device memory is not host visible
print("FIRST_STORE_OK", flush=True)
res = {}
for threads in (1, 4, 8, 16):
    per = n // threads
    def work(t):
        ctypes.memmove(d.data_ptr() + t * per, h.ctypes.data + t * per, per)
    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    [t.start() for t in ths]; [t.join() for t in ths]
    res[str(threads)] = n / (time.perf_counter() - t0) / 1e9
ok = bool((d[: 1 << 20].cpu().numpy() == 1.0).all())
print("RESULT " + json.dumps({"cpu_store_to_device_GBps_by_threads": res, "data_ok": ok}), flush=True)
'''
p = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=300)
out = {"returncode": p.returncode, "stages": [l for l in p.stdout.splitlines() if l.endswith("_OK")]}
for l in p.stdout.splitlines():
    if l.startswith("RESULT "):
        out.update(json.loads(l[7:]))
if p.returncode != 0:
    out["stderr_tail"] = p.stderr[-300:]
print(json.dumps(out))
