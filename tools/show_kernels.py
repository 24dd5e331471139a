#!/usr/bin/env python3
"""Print the per-kernel table of a bench.py JSON line (stdin): calls/step, ms/call, ms/step, rate."""
import json
import sys

r = json.loads(sys.stdin.read())
print(r["ms_per_step"], "ms/step", r["value"], r["unit"])
tot = 0.0
for k, v in sorted(r["kernels"].items(), key=lambda kv: -kv[1]["ms"] * kv[1]["calls_per_step"]):
    t = v["ms"] * v["calls_per_step"]
    tot += t
    print("%-24s calls %6.1f  ms/call %8.3f  ms/step %8.2f   %s" % (k, v["calls_per_step"], v["ms"], t, v.get("TFLOP/s", v.get("GB/s"))))
print("sum of timed kernels %.2f ms/step" % tot)
