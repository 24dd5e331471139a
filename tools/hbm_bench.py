#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound row kernels at the headline shape (A/B runs).  usage: python tools/hbm_bench.py [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from competesmoe_amd import ops, _lib as L  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--tokens", type=int, default=32768)
    ap.add_argument("--D", type=int, default=4096)
    ap.add_argument("--E", type=int, default=64)
    a = ap.parse_args()
    T, D, E, K = a.tokens, a.D, a.E, 2
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    idx = torch.rand(T, E, generator=g).topk(K, -1).indices.int().to(dev)
    bins = ops.bin_tokens(idx, E)
    x = torch.randn(T, D, device=dev).bfloat16()
    y = torch.randn(T * K, D, device=dev).bfloat16()
    w = torch.rand(T, K, device=dev)
    dout = torch.randn(T, D, device=dev).bfloat16()
    es = 2
    runs = {
        "dispatch_tokens": (lambda: ops.dispatch_tokens(x, bins), (T + T * K) * D * es),
        "combine(seq)": (lambda: ops.combine(y, bins, idx, w, L.COMBINE_SEQ, T), (T + T * K) * D * es),
        "combine(dot)": (lambda: ops.combine(y, bins, idx, w, L.COMBINE_DOT, T), (T + T * K) * D * es),
        "combine_bwd": (lambda: ops.combine_bwd(dout, y, bins, w), (T + 2 * T * K) * D * es),
        "dispatch_rows_bwd": (lambda: ops.dispatch_rows_bwd(y, bins, T), (T + T * K) * D * es),
    }
    for name, (fn, nbytes) in runs.items():
        ms = timeit(fn, a.iters)
        print(f"{name:20s} {ms:7.4f} ms  {nbytes / ms / 1e6:7.1f} GB/s")


if __name__ == "__main__":
    main()
