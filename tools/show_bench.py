#!/usr/bin/env python3
"""Usage: tools/show_bench.py <bench.json>  -- the few fields of a bench.py line one looks at first."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"value {d['value']:.4g} {d['unit']}  ms/step {d['ms_per_step']:.3f}  frac {r['frac']:.4f}  traffic {r['traffic']}")
print("traffic_source:", r["traffic_source"])
for k in ("fast_mode", "fast_mode_f16"):
    if k in d:
        print(k, f"{d[k]['value']:.4g}")
for k, v in (d.get("other_baseline_configs") or {}).items():
    print(k, {kk: vv for kk, vv in v.items() if kk in ("value", "roofline_frac_of_fp32_mfma_peak", "error")} if isinstance(v, dict) else v)
for k, v in (d.get("next_rows") or {}).items():
    print("next_rows." + k, {kk: ({a: b for a, b in vv.items() if a in ("ms_per_step", "roofline_frac_of_fp32_mfma_peak")} if isinstance(vv, dict) else vv) for kk, vv in v.items() if kk != "workload"} if isinstance(v, dict) else v)
print("cpu_baseline:", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k in ("value", "cores", "kind", "error")})
