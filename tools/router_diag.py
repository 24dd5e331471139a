#!/usr/bin/env python3
"""Times csmoe_gate_select alone (results unchecked) -- for the diagnostic twins of `make rfvar` (CSMOE_LIB=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from competesmoe_amd import ops, _lib as L
for T, D, E, K in ((32768, 4096, 64, 2), (32768, 4096, 16, 2)):
    x = torch.randn(T, D, device="cuda").bfloat16()
    wg = (torch.randn(E, D, device="cuda") * D ** -0.5).bfloat16()
    f = lambda: ops.gate_select(x, wg, K, L.SEL_SOFTMAX, True)
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print(f"E={E}: {us:.1f} us = {T * D * 2 / us / 1e6:.2f} TB/s")
