import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from competesmoe_amd import ops, _lib as L
from tests.test_ops_gpu import make_groups, ref_rowspace
DEV = "cuda"
def run(E, M, N, Kd, b_layout, seed, empty):
    g = torch.Generator().manual_seed(1000 + seed)
    off = make_groups(E, M, seed=seed, empty=empty)
    A = torch.randn(M, Kd, generator=g).bfloat16().to(DEV)
    shape = (N, Kd) if b_layout == 0 else (Kd, N)
    Bs = [(torch.randn(*shape, generator=g) / math.sqrt(Kd)).bfloat16().to(DEV) for _ in range(E)]
    rc, _ = ref_rowspace(A, Bs, b_layout, off, None, 0, 0, None)
    c = ops.grouped_gemm(A, ops.ptr_array(Bs, DEV), b_layout, Bs[0].stride(0), N, off.to(DEV), E, kernel=4)
    bad = (c.float() - rc.float()).abs() > 0.05
    blocks = sorted(set((int(r) // 16, int(cc) // 16) for r, cc in bad.nonzero().tolist()))
    print(os.environ.get("CSMOE_LIB", "default")[-12:], (E, M, N, Kd, b_layout), "bad frac", float(bad.float().mean()), "bad 16x16 blocks (row blk, col blk):", blocks[:40])
run(4, 2048, 512, 256, 0, 5, False)
run(4, 2048, 512, 256, 1, 5, False)
