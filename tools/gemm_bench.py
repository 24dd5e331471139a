#!/usr/bin/env python3
"""Micro-benchmark of the grouped expert GEMM kernels alone at the headline shapes (A/B runs, rocprofv3 --pmc runs).
usage: python tools/gemm_bench.py [--which nt1,nt2,nn1,nn2,tn1,tn2] [--iters 5] [--balanced]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from competesmoe_amd import ops, _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="nt1,nt2,nn1,nn2,tn1,tn2")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--tokens", type=int, default=32768)
    ap.add_argument("--balanced", action="store_true", help="exactly T*K/E rows per expert (no ragged tiles)")
    ap.add_argument("--deal", action="store_true", help="weight-gradient launches: experts dealt to XCDs by row count (csmoe_expert_order)")
    ap.add_argument("--skew", type=float, default=0.0, help="add this to the routing scores of experts 0..7 (hot experts)")
    ap.add_argument("--E", type=int, default=64)
    ap.add_argument("--D", type=int, default=4096)
    ap.add_argument("--F", type=int, default=11008)
    ap.add_argument("--f32-out", action="store_true", help="weight-gradient launches write fp32 (the pretrain stack's master-weight gradients)")
    ap.add_argument("--same-weights", action="store_true", help="every expert points at expert 0's matrices (weight panels stay cache-resident): what the loop does without the weight traffic")
    a = ap.parse_args()
    dev = "cuda"
    T, K, E, D, F = a.tokens, 2, a.E, a.D, a.F
    M = T * K
    g = torch.Generator(device=dev).manual_seed(0)
    if a.balanced:
        counts = torch.full((E,), M // E, dtype=torch.int64)
    else:
        sc = torch.rand(T, E, generator=torch.Generator().manual_seed(0))
        sc[:, :8] += a.skew
        idx = sc.topk(K, -1).indices
        counts = torch.bincount(idx.flatten(), minlength=E)
    off = torch.zeros(E + 1, dtype=torch.int32)
    off[1:] = counts.cumsum(0)
    off = off.to(dev)
    bf = torch.bfloat16
    xs = torch.randn(M, D, device=dev, generator=g).to(bf)
    h = torch.randn(M, F, device=dev, generator=g).to(bf)
    W1 = (torch.randn(E, F, D, device=dev, generator=g) * 0.02).to(bf)
    W2 = (torch.randn(E, D, F, device=dev, generator=g) * 0.02).to(bf)
    b1 = torch.zeros(E, F, device=dev, dtype=bf)
    es = 2
    ar = torch.arange(E, device=dev, dtype=torch.int64)
    p1 = W1.data_ptr() + ar * (F * D * es)
    p2 = W2.data_ptr() + ar * (D * F * es)
    pb1 = b1.data_ptr() + ar * (F * es)
    if a.same_weights:
        p1, p2 = p1 * 0 + W1.data_ptr(), p2 * 0 + W2.data_ptr()
    gdt = torch.float32 if a.f32_out else bf
    gW1 = torch.empty(E, F, D, device=dev, dtype=gdt)
    gW2 = torch.empty(E, D, F, device=dev, dtype=gdt)
    pg1 = gW1.data_ptr() + ar * (F * D * gW1.element_size())
    pg2 = gW2.data_ptr() + ar * (D * F * gW2.element_size())
    order = ops.expert_order(off, E) if a.deal else None
    runs = {
        "nt1": (lambda: ops.grouped_gemm(xs, p1, L.B_NK, D, F, off, E, bias_ptrs=pb1, epilogue=L.EPI_BIAS_ACT, act=L.ACT_GELU, want_c2=True), 2.0 * M * D * F),
        "nt2": (lambda: ops.grouped_gemm(h, p2, L.B_NK, F, D, off, E), 2.0 * M * D * F),
        "nn1": (lambda: ops.grouped_gemm(xs, p2, L.B_KN, F, F, off, E, epilogue=L.EPI_ACTGRAD, act=L.ACT_GELU, aux=h), 2.0 * M * D * F),
        "nn2": (lambda: ops.grouped_gemm(h, p1, L.B_KN, D, D, off, E), 2.0 * M * D * F),
        "tn1": (lambda: ops.grouped_wgrad(h, xs, off, E, gW1, pg1, xcd_order=order), 2.0 * M * D * F),
        "tn2": (lambda: ops.grouped_wgrad(xs, h, off, E, gW2, pg2, xcd_order=order), 2.0 * M * D * F),
    }
    for name in a.which.split(","):
        fn, flops = runs[name]
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / a.iters
        print(f"{name}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
