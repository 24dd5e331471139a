import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from competesmoe_amd import ops, _lib as L
from tests.test_ops_gpu import make_groups, ref_rowspace
DEV = "cuda"
E, M, N, Kd = 1, 2048, 256, 128
for b_layout in (1,):
    for epi, act in ((0, 0), (1, 0)):
        g = torch.Generator().manual_seed(100)
        off = make_groups(E, M, seed=0, empty=False)
        A = torch.randn(M, Kd, generator=g).bfloat16().to(DEV)
        shape = (N, Kd) if b_layout == 0 else (Kd, N)
        Bs = [(torch.randn(*shape, generator=g) / math.sqrt(Kd)).bfloat16().to(DEV) for _ in range(E)]
        bias = [(torch.randn(N, generator=g) * 0.5).bfloat16().to(DEV) for _ in range(E)]
        kw = dict(epilogue=epi, act=act)
        if epi == 1:
            kw["bias_ptrs"] = ops.ptr_array(bias, DEV)
        c = ops.grouped_gemm(A, ops.ptr_array(Bs, DEV), b_layout, Bs[0].stride(0), N, off.to(DEV), E, **kw)
        rc, _ = ref_rowspace(A, Bs, b_layout, off, bias if epi == 1 else None, epi, 0, None)
        d = (c.float() - rc.float()).abs()
        badm = d > 0.05
        print("epi", epi, "bad frac", float(badm.float().mean()), "bad rows", badm.any(1).nonzero().flatten()[:20].tolist(), "n bad rows", int(badm.any(1).sum()),
              "bad cols", badm.any(0).nonzero().flatten()[:40].tolist(), "n bad cols", int(badm.any(0).sum()))
        # which K-half is wrong?  compare with products over k < 64 and k >= 64 only
        for lo, hi in ((0, 64), (64, 128)):
            part = (A[:, lo:hi].double() @ Bs[0][lo:hi].double()).float().cpu()
            print("   corr with k in", lo, hi, float(((c.float().cpu() - part) ** 2).mean()))
