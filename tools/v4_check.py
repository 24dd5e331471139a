"""Quick check of the row-space kernel chosen by CSMOE_GEMM_KERNEL (v2 / v4) against fp64 references at shapes both take
(K a multiple of 128), every epilogue and both weight layouts.  usage (GPU box): CSMOE_GEMM_KERNEL=v4 python tools/v4_check.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from competesmoe_amd import ops, _lib as L  # noqa: E402
from tests.test_ops_gpu import make_groups, ref_rowspace  # noqa: E402

DEV = "cuda"
bad = 0
for i, (E, M, N, Kd) in enumerate([(1, 2048, 256, 128), (3, 2500, 512, 256), (5, 3000, 776, 512), (8, 6000, 1024, 1280), (2, 2304, 264, 384)]):
    for b_layout in (0, 1):
        for epi, act in ((0, 0), (1, 0), (2, 1), (2, 2), (3, 1), (3, 2)):
            g = torch.Generator().manual_seed(100 + i)
            off = make_groups(E, M, seed=i, empty=E > 2)
            A = torch.randn(M, Kd, generator=g).bfloat16().to(DEV)
            shape = (N, Kd) if b_layout == 0 else (Kd, N)
            Bs = [(torch.randn(*shape, generator=g) / math.sqrt(Kd)).bfloat16().to(DEV) for _ in range(E)]
            bias = [(torch.randn(N, generator=g) * 0.5).bfloat16().to(DEV) for _ in range(E)]
            aux = torch.randn(M, N, generator=g).bfloat16().to(DEV)
            kw = dict(epilogue=epi, act=act)
            if epi in (1, 2):
                kw["bias_ptrs"] = ops.ptr_array(bias, DEV)
            if epi == 2:
                kw["want_c2"] = True
            if epi == 3:
                kw["aux"] = aux
            res = ops.grouped_gemm(A, ops.ptr_array(Bs, DEV), b_layout, Bs[0].stride(0), N, off.to(DEV), E, **kw)
            c, c2 = res if epi == 2 else (res, None)
            rc, rc2 = ref_rowspace(A, Bs, b_layout, off, bias if epi in (1, 2) else None, epi, act if epi >= 2 else 0, aux if epi == 3 else None)
            ok = torch.allclose(c.float(), rc.float(), rtol=2 ** -7, atol=2e-2)
            if epi == 2:
                ok = ok and torch.allclose(c2.float(), rc2.float(), rtol=2 ** -7, atol=2e-2)
            if not ok:
                bad += 1
                print("MISMATCH", (E, M, N, Kd), b_layout, epi, act, float((c.float() - rc.float()).abs().max()))
print("kernel", os.environ.get("CSMOE_GEMM_KERNEL", "default"), "mismatches:", bad)
sys.exit(1 if bad else 0)
