#!/usr/bin/env python3
"""Where the dense (shared-expert) fp8 FFN and oracle/mxfp8.py part: per-stage comparison of h, mask, dh on the data of
tests/test_fp8_gpu.py::test_ffn_functions_on_the_fp8_pipe_match_the_mx_oracle (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from competesmoe_amd import ops, _lib as L
from oracle import mxfp8 as MX

rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
T, D, Fh, E, K = 384, 256, 384, 8, 2
for seed in (21, 22, 23):
    g = torch.Generator().manual_seed(seed)
    x2 = torch.randn(T, D, generator=g).bfloat16()
    idx = torch.stack([torch.randperm(E, generator=g)[:K] for _ in range(T)])
    w = torch.rand(T, K, generator=g)
    keys, values = torch.randn(E, D, Fh, generator=g) * 0.06, torch.randn(E, Fh, D, generator=g) * 0.05
    bias = torch.randn(E, Fh, generator=g) * 0.1
    dout = torch.randn(T, D, generator=g).bfloat16()
    w1, w2 = torch.randn(D, 2 * Fh, generator=g) * 0.06, torch.randn(2 * Fh, D, generator=g) * 0.04
    xq, xs = ops.quantize_mxfp8(x2.cuda())
    (w1q_b, w1s_b), (w1q, w1s) = ops.quantize_mxfp8_both(w1.cuda())
    hpre, hact = ops.dense_gemm_mxfp8(xq, xs, w1q, w1s, epilogue=L.EPI_BIAS_ACT, act=L.ACT_RELU, want_c2=True, want_c=True)
    h64 = MX._mx(x2) @ MX._mx(w1.t().contiguous()).t()
    h_ref = MX._bf(h64)
    mm = ((hact.cpu() > 0) != (h_ref > 0))
    print("seed", seed, "h", rel(hpre, h_ref), "mask mismatches", int(mm.sum()), "of", h_ref.numel(), "|h64| there", h64[mm].abs().tolist(),
          "hip h there", hpre.cpu()[mm].tolist(), "min |h64|", float(h64.abs().min()))
    (w2q_b, w2s_b), (w2q, w2s) = ops.quantize_mxfp8_both(w2.cuda())
    dyq, dys = ops.quantize_mxfp8(dout.cuda())
    dh = ops.dense_gemm_mxfp8(dyq, dys, w2q_b, w2s_b, epilogue=L.EPI_ACTGRAD, act=L.ACT_RELU, aux=hact)
    dh_ref = MX._bf(MX._bf(MX._mx(dout) @ MX._mx(w2).t()).double() * (h_ref > 0))
    print("   dh", rel(dh, dh_ref))
    gw1 = torch.zeros(D, 2 * Fh, device="cuda")
    from competesmoe_amd.functional import _dense_wgrad
    gw = _dense_wgrad(x2.cuda(), dh, torch.float32)
    print("   gw1 from hip dh vs fp64 product of the same", rel(gw, x2.double().t() @ dh.double().cpu()), "vs oracle", rel(gw, x2.double().t() @ dh_ref.double()))
