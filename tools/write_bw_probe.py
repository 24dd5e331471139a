#!/usr/bin/env python3
"""HBM write-only / copy bandwidth of this box through plain torch launches (what a store-bound epilogue can hope for)."""
import torch
n = 128 * 4096 * 11008 // 2
x = torch.empty(n, device="cuda", dtype=torch.float32)
y = torch.empty(n, device="cuda", dtype=torch.float32)
def t(f, it=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
ms = t(lambda: x.fill_(1.0)); print(f"fill_ {n*4/1e9:.1f} GB: {ms:.3f} ms = {n*4/ms/1e9:.2f} TB/s written")
ms = t(lambda: x.zero_()); print(f"zero_ : {ms:.3f} ms = {n*4/ms/1e9:.2f} TB/s written")
ms = t(lambda: y.copy_(x)); print(f"copy_ : {ms:.3f} ms = {2*n*4/ms/1e9:.2f} TB/s read+written")
ms = t(lambda: x.sum()); print(f"sum   : {ms:.3f} ms = {n*4/ms/1e9:.2f} TB/s read")
