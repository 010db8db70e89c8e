#!/usr/bin/env python3
"""Usage: tools/kernel_resources.py <file.hip> [extra hipcc flags]
(nnv_mlp_decode.hip needs -DNNV_TU_PREC=0|1, and -mllvm -amdgpu-sched-strategy=max-ilp for the shipped fp32 build)
One line per kernel: name, VGPRs, AGPRs, spills, scratch, occupancy."""
import os, re, subprocess, sys
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{here}/include", "-c", sys.argv[1],
       "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True)
rows, cur = [], None
for line in (out.stdout + out.stderr).splitlines():
    if "error" in line:
        print(line)
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    n = re.sub(r"^_ZN3nnv\d+", "", r["name"]).replace("EEvNS_12DecodeParamsE", "")
    print("%-58s vgpr=%s agpr=%s spill=%s scratch=%s occ=%s sgpr=%s" % (
        n, r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"),
        r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("SGPRs")))
