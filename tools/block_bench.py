#!/usr/bin/env python3
"""Micro-benchmark of the block kernels (LayerNorm+gate forward, LayerNorm backward, residual combine) at the headline shape,
next to the torch ops they replace.  usage: python tools/block_bench.py [--tokens 32768] [--D 4096] [--E 64]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from competesmoe_amd import ops, _lib as L  # noqa: E402


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=32768)
    ap.add_argument("--D", type=int, default=4096)
    ap.add_argument("--E", type=int, default=64)
    a = ap.parse_args()
    T, D, E, dev, dt = a.tokens, a.D, a.E, "cuda", torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(T, D, device=dev, generator=g).to(dt)
    gamma = torch.ones(D, device=dev, dtype=dt)
    beta = torch.zeros(D, device=dev, dtype=dt)
    wg = (torch.randn(E, D, device=dev, generator=g) * 0.02).to(dt)
    d1 = torch.randn(T, D, device=dev, generator=g).to(dt)
    d2 = torch.randn(T, D, device=dev, generator=g).to(dt)
    res = torch.randn(T, D, device=dev, generator=g).to(dt)
    by = T * D * 2
    _, mean, rstd, _ = ops.layernorm_gate(x, gamma, beta, 1e-6, None)

    def rep(name, ms, nbytes):
        print(f"{name:44s} {ms:7.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s", flush=True)

    rep("csmoe layernorm_gate (LN + gate, fused)", timeit(lambda: ops.layernorm_gate(x, gamma, beta, 1e-6, wg)), 2 * by)
    rep("csmoe layernorm (no gate)", timeit(lambda: ops.layernorm_gate(x, gamma, beta, 1e-6, None)), 2 * by)
    rep("torch layer_norm + csmoe gate_logits", timeit(lambda: ops.gate_logits(F.layer_norm(x, (D,), gamma, beta, 1e-6), wg)), 3 * by)
    rep("csmoe layernorm_bwd (2 grads + residual)", timeit(lambda: ops.layernorm_bwd(d1, x, gamma, mean, rstd, add=res, dxn2=d2)), 5 * by)
    rep("csmoe layernorm_bwd (1 grad)", timeit(lambda: ops.layernorm_bwd(d1, x, gamma, mean, rstd)), 3 * by)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)

    def torch_bwd():
        xr.grad = gr.grad = br.grad = None
        y = F.layer_norm(xr, (D,), gr, br, 1e-6)
        y.backward(d1 + d2)
        return xr.grad + res

    rep("torch: add + LN fwd+bwd + add", timeit(torch_bwd), 10 * by)
    rep("torch: LN fwd only", timeit(lambda: F.layer_norm(x, (D,), gamma, beta, 1e-6)), 2 * by)


if __name__ == "__main__":
    main()
