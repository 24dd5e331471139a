#!/usr/bin/env python3
"""Per-kernel table of a bench.py JSON line (mean launch ms x launches per step), largest first.  usage: show_bench.py <file>"""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(d["ms_per_step"], "ms/step", d["value"], d["unit"], "peak HBM GiB", d.get("peak_hbm_gib"))
tot = 0.0
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms"] * kv[1]["calls_per_step"]):
    t = v["ms"] * v["calls_per_step"]
    tot += t
    rest = {a: b for a, b in v.items() if a not in ("ms", "calls_per_step")}
    print(f"{k:28s} {v['calls_per_step']:5.1f} x {v['ms']:8.3f} = {t:8.3f}  {rest}")
print("sum of launches", round(tot, 3))
