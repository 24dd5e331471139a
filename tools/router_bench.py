#!/usr/bin/env python3
"""One-pass router (csmoe_gate_select) against gate GEMM + router_select + counting pass at the headline token count.
usage (GPU box): python tools/router_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from competesmoe_amd import ops, _lib as L  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    dev = "cuda"
    for T, D, E, K in ((32768, 4096, 64, 2), (32768, 4096, 64, 8), (32768, 1024, 64, 2), (32768, 4096, 16, 2), (12800, 1152, 4, 2)):
        x = torch.randn(T, D, device=dev).to(torch.bfloat16)
        wg = (torch.randn(E, D, device=dev) * D ** -0.5).to(torch.bfloat16)
        t_f = timed(lambda: ops.gate_select(x, wg, K, L.SEL_SOFTMAX, True))
        t_g = timed(lambda: ops.gate_logits(x, wg))
        lg = ops.gate_logits(x, wg)
        t_s = timed(lambda: ops.router_select(lg, K, L.SEL_SOFTMAX, True))
        _, idx, _ = ops.router_select(lg, K, L.SEL_SOFTMAX, True)
        t_b = timed(lambda: ops.bin_tokens(idx.clone(), E))
        _, _, idx2, _ = ops.gate_select(x, wg, K, L.SEL_SOFTMAX, True)
        t_bh = timed(lambda: ops.bin_tokens(idx2, E))
        gb = T * D * 2 / 1e9
        print(f"T={T} D={D} E={E} K={K}: fused {t_f:.1f} us ({gb / t_f * 1e6:.0f} GB/s) + bins-from-hist {t_bh:.1f} us | "
              f"gate {t_g:.1f} us ({gb / t_g * 1e6:.0f} GB/s) + select {t_s:.1f} us + bins (incl. a 256 KB clone) {t_b:.1f} us", flush=True)


if __name__ == "__main__":
    main()
