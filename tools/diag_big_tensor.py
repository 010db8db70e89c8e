#!/usr/bin/env python3
"""Diagnostic (scratch): do strided slices / copies of > 2^32-element tensors behave on this torch build?"""
import torch, time
T, B, m = 512, 10_000_000, 2048
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randn((T, B), device="cuda", generator=g)
print("numel", a.numel(), "abs mean on device, whole:", float(a.abs().mean()))
for lo in (0, B // 2, B - m):
    s = a[:, lo:lo + m]
    print(f"cols {lo}: device abs-mean {float(s.abs().mean()):.4f}  cpu abs-mean {float(s.cpu().abs().mean()):.4f}  "
          f"contiguous().cpu() {float(s.contiguous().cpu().abs().mean()):.4f}")
for t in (0, 100, 300, 511):
    print(f"row {t}: device abs-mean {float(a[t].abs().mean()):.4f}  first vals {a[t, :3].tolist()}")
z = torch.trunc(torch.sign(a) + 0.5 * torch.randn((T, B), device="cuda", generator=g))
print("trunc'd: frac zero", float((z == 0).float().mean()), "rows 0/511 frac zero", float((z[0] == 0).float().mean()), float((z[511] == 0).float().mean()))
