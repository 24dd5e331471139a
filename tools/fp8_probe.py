#!/usr/bin/env python3
"""Diagnostics of the MXFP8 GEMM: which part of the operand / scale mapping is wrong?  (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from competesmoe_amd import ops
from oracle import mxfp8 as MX

DEV = "cuda"
M, N, Kd = 40, 64, 128
if len(sys.argv) > 1:
    M, N, Kd = (int(v) for v in sys.argv[1:4])
g = torch.Generator().manual_seed(1)
off = torch.tensor([0, M], dtype=torch.int32)


def run(Aq, As, Bq, Bs, name):
    c = ops.grouped_gemm_mxfp8(Aq.to(DEV), As.to(DEV), Bq.to(DEV), Bs.to(DEV), off.to(DEV)).cpu().float()
    ref = MX.grouped_matmul(Aq, As, Bq, Bs, off).float()
    bad = (c != ref.bfloat16().float())
    print(f"{name:34s} max|err| {float((c - ref).abs().max()):10.3f}  wrong {int(bad.sum())}/{bad.numel()}  rows-with-errors {int(bad.any(1).sum())} cols-with-errors {int(bad.any(0).sum())}")
    return c, ref


Ai = torch.randint(-2, 3, (M, Kd), generator=g).float()
Bi = torch.randint(-1, 3, (1, N, Kd), generator=g).float()
one_a = torch.full((M, Kd // 32), 127, dtype=torch.uint8)
one_b = torch.full((1, N, Kd // 32), 127, dtype=torch.uint8)
run(MX.to_e4m3_bytes(Ai), one_a, MX.to_e4m3_bytes(Bi), one_b, "unit scales, random ints")
# k-map probe: A row m has a single 1 at k = (7 m + 3) % Kd; B[n, k] = k % 5 + n % 3 -> C[m, n] = B[n, k(m)]
A1 = torch.zeros(M, Kd); A1[torch.arange(M), (7 * torch.arange(M) + 3) % Kd] = 1
B1 = (torch.arange(Kd).view(1, 1, Kd) % 5 + torch.arange(N).view(1, N, 1) % 3).float()
c, ref = run(MX.to_e4m3_bytes(A1), one_a, MX.to_e4m3_bytes(B1), one_b, "one-hot A rows (k map)")
if not torch.equal(c, ref):
    print("   got row 0..3:", c[:4, :8].tolist()); print("   ref row 0..3:", ref[:4, :8].tolist())
sa_row = torch.randint(125, 130, (M, 1), generator=g, dtype=torch.uint8).expand(M, Kd // 32).contiguous()
run(MX.to_e4m3_bytes(Ai), sa_row, MX.to_e4m3_bytes(Bi), one_b, "A scale per row only")
sa_blk = torch.randint(125, 130, (1, Kd // 32), generator=g, dtype=torch.uint8).expand(M, Kd // 32).contiguous()
run(MX.to_e4m3_bytes(Ai), sa_blk, MX.to_e4m3_bytes(Bi), one_b, "A scale per k-block only")
sb_row = torch.randint(125, 130, (1, N, 1), generator=g, dtype=torch.uint8).expand(1, N, Kd // 32).contiguous()
run(MX.to_e4m3_bytes(Ai), one_a, MX.to_e4m3_bytes(Bi), sb_row, "B scale per column only")
sb_blk = torch.randint(125, 130, (1, 1, Kd // 32), generator=g, dtype=torch.uint8).expand(1, N, Kd // 32).contiguous()
run(MX.to_e4m3_bytes(Ai), one_a, MX.to_e4m3_bytes(Bi), sb_blk, "B scale per k-block only")
run(MX.to_e4m3_bytes(Ai), torch.randint(125, 130, (M, Kd // 32), generator=g, dtype=torch.uint8), MX.to_e4m3_bytes(Bi),
    torch.randint(126, 129, (1, N, Kd // 32), generator=g, dtype=torch.uint8), "all scales random")
