#!/usr/bin/env python3
"""Where a row-space grouped-GEMM workgroup's time goes, from the -DCSMOE_STAMPS build (competesmoe_amd/lib/libcsmoe_hip_stamps.so,
built by `make stamps` in csrc): per workgroup the CU it ran on and the 100 MHz clock at entry / K-loop start / K-loop end / exit.
Prints, for full 256-row tiles, the medians of: idle gap between consecutive workgroups on one CU, set-up, K-loop, epilogue.
usage (GPU box): CSMOE_LIB=$PWD/competesmoe_amd/lib/libcsmoe_hip_stamps.so python tools/tile_stamps.py [--which nn1,nt1,nn2,nt2]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from competesmoe_amd import ops, _lib as L  # noqa: E402


def analyse(path, name, ms):
    a = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
    a = a[a[:, 1] != 0]
    hw = a[:, 0] & np.uint64(0xffffffff)
    xcc = (a[:, 0] >> np.uint64(32)) & np.uint64(0xf)
    cu = (hw >> np.uint64(8)) & np.uint64(0xf)
    sh = (hw >> np.uint64(12)) & np.uint64(0x1)
    se = (hw >> np.uint64(13)) & np.uint64(0x7)
    key = (xcc * np.uint64(1024) + se * np.uint64(64) + sh * np.uint64(32) + cu).astype(np.int64)
    rows = (a[:, 5] >> np.uint64(32)).astype(np.int64)
    nk = (a[:, 5] & np.uint64(0xffffffff)).astype(np.int64)
    t = a[:, 1:5].astype(np.int64)
    gaps, first_gap = [], []
    t0 = t[:, 0].min()
    for k in np.unique(key):
        m = np.nonzero(key == k)[0]
        m = m[np.argsort(t[m, 0])]
        g = t[m[1:], 0] - t[m[:-1], 3]
        gaps.append(g)
        first_gap.append(t[m[0], 0] - t0)
    gaps = np.concatenate(gaps)
    full = rows == 256
    us = lambda x: 0.01 * float(np.median(x))
    n_cu = len(np.unique(key))
    span = (t[:, 3].max() - t0) * 0.01
    busy = (t[:, 3] - t[:, 0]).sum() * 0.01 / n_cu
    print(f"{name}: launch {ms:.3f} ms; {len(a)} workgroups on {n_cu} CUs; span {span:.0f} us, per-CU resident {busy:.0f} us "
          f"({100 * busy / span:.1f} %)")
    print(f"   full tiles ({int(full.sum())}, nk={int(np.median(nk))}): set-up {us(t[full, 1] - t[full, 0]):.2f} us, "
          f"K-loop (incl. first fetch) {us(t[full, 2] - t[full, 1]):.2f} us, epilogue {us(t[full, 3] - t[full, 2]):.2f} us, "
          f"total {us(t[full, 3] - t[full, 0]):.2f} us")
    thin = ~full
    if thin.any():
        for lo, hi in ((1, 64), (65, 128), (129, 192), (193, 255)):
            m = (rows >= lo) & (rows <= hi)
            if m.any():
                print(f"   ragged tiles of {lo}..{hi} rows ({int(m.sum())}): total {us(t[m, 3] - t[m, 0]):.2f} us, "
                      f"K-loop {us(t[m, 2] - t[m, 1]):.2f} us")
    print(f"   gap exit -> next entry on the same CU: median {us(gaps):.2f} us, mean {0.01 * gaps.mean():.2f} us, "
          f"p90 {0.01 * np.percentile(gaps, 90):.2f} us, negative (overlap) {int((gaps < 0).sum())}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="nn1,nt1,nn2,nt2")
    ap.add_argument("--balanced", action="store_true")
    a = ap.parse_args()
    dev = "cuda"
    T, K, E, D, F = 32768, 2, 64, 4096, 11008
    M = T * K
    g = torch.Generator(device=dev).manual_seed(0)
    if a.balanced:
        counts = torch.full((E,), M // E, dtype=torch.int64)
    else:
        sc = torch.rand(T, E, generator=torch.Generator().manual_seed(0))
        counts = torch.bincount(sc.topk(K, -1).indices.flatten(), minlength=E)
    off = torch.zeros(E + 1, dtype=torch.int32)
    off[1:] = counts.cumsum(0)
    off = off.to(dev)
    bf = torch.bfloat16
    xs = torch.randn(M, D, device=dev, generator=g).to(bf)
    h = torch.randn(M, F, device=dev, generator=g).to(bf)
    W1 = (torch.randn(E, F, D, device=dev, generator=g) * 0.02).to(bf)
    W2 = (torch.randn(E, D, F, device=dev, generator=g) * 0.02).to(bf)
    b1 = torch.zeros(E, F, device=dev, dtype=bf)
    ar = torch.arange(E, device=dev, dtype=torch.int64)
    p1 = W1.data_ptr() + ar * (F * D * 2)
    p2 = W2.data_ptr() + ar * (D * F * 2)
    pb1 = b1.data_ptr() + ar * (F * 2)
    runs = {
        "nt1": lambda: ops.grouped_gemm(xs, p1, L.B_NK, D, F, off, E, bias_ptrs=pb1, epilogue=L.EPI_BIAS_ACT, act=L.ACT_GELU, want_c2=True),
        "nt2": lambda: ops.grouped_gemm(h, p2, L.B_NK, F, D, off, E),
        "nn1": lambda: ops.grouped_gemm(xs, p2, L.B_KN, F, F, off, E, epilogue=L.EPI_ACTGRAD, act=L.ACT_GELU, aux=h),
        "nn2": lambda: ops.grouped_gemm(h, p1, L.B_KN, D, D, off, E),
    }
    path = "/tmp/csmoe_stamps.bin"
    for name in a.which.split(","):
        fn = runs[name]
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        os.environ["CSMOE_STAMP_FILE"] = path
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        del os.environ["CSMOE_STAMP_FILE"]
        analyse(path, name, s.elapsed_time(e))


if __name__ == "__main__":
    main()
