"""Run-to-run determinism at the headline shape: every big kernel launched repeatedly on the same inputs must return bit-identical
outputs.  The GEMM loops rely on hand-placed waits and barriers (counted vmcnt, staggered wave groups, LDS slots re-filled while
others are read); a missing wait shows up as run-to-run differences long before it shows up as a wrong mean."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

if torch.cuda.is_available():
    from competesmoe_amd import ops, _lib as L


def _setup(T=32768, K=2, E=64, D=4096, F=11008, skew=0.0):
    M = T * K
    g = torch.Generator(device=DEV).manual_seed(0)
    sc = torch.rand(T, E, generator=torch.Generator().manual_seed(0))
    sc[:, :8] += skew
    counts = torch.bincount(sc.topk(K, -1).indices.flatten(), minlength=E)
    off = torch.zeros(E + 1, dtype=torch.int32)
    off[1:] = counts.cumsum(0)
    off = off.to(DEV)
    bf = torch.bfloat16
    xs = torch.randn(M, D, device=DEV, generator=g).to(bf)
    h = torch.randn(M, F, device=DEV, generator=g).to(bf)
    W1 = (torch.randn(E, F, D, device=DEV, generator=g) * 0.02).to(bf)
    W2 = (torch.randn(E, D, F, device=DEV, generator=g) * 0.02).to(bf)
    b1 = (torch.randn(E, F, device=DEV, generator=g) * 0.1).to(bf)
    ar = torch.arange(E, device=DEV, dtype=torch.int64)
    return dict(M=M, E=E, D=D, F=F, off=off, xs=xs, h=h, W1=W1, W2=W2, b1=b1,
                p1=W1.data_ptr() + ar * (F * D * 2), p2=W2.data_ptr() + ar * (D * F * 2), pb1=b1.data_ptr() + ar * (F * 2), ar=ar)


@pytest.mark.parametrize("skew", [0.0, 0.3])
def test_big_kernels_are_bitwise_reproducible(skew):
    s = _setup(skew=skew)
    E, D, F, off = s["E"], s["D"], s["F"], s["off"]
    order = ops.expert_order(off, E)

    def wgrad(a, b, Na, Nb, od):
        out = torch.empty(E, Na, Nb, device=DEV, dtype=od)
        ops.grouped_wgrad(a, b, off, E, out, out.data_ptr() + s["ar"] * (Na * Nb * out.element_size()), xcd_order=order)
        return (out,)

    runs = {
        "gemm1 nt bias+gelu": lambda: ops.grouped_gemm(s["xs"], s["p1"], L.B_NK, D, F, off, E, bias_ptrs=s["pb1"], epilogue=L.EPI_BIAS_ACT,
                                                        act=L.ACT_GELU, want_c2=True),
        "gemm2 nt": lambda: (ops.grouped_gemm(s["h"], s["p2"], L.B_NK, F, D, off, E),),
        "dh nn actgrad": lambda: (ops.grouped_gemm(s["xs"], s["p2"], L.B_KN, F, F, off, E, epilogue=L.EPI_ACTGRAD, act=L.ACT_GELU, aux=s["h"]),),
        "dxs nn": lambda: (ops.grouped_gemm(s["h"], s["p1"], L.B_KN, D, D, off, E),),
        "dW1 tn bf16": lambda: wgrad(s["h"], s["xs"], F, D, torch.bfloat16),
        "dW2 tn fp32": lambda: wgrad(s["xs"], s["h"], D, F, torch.float32),
    }
    for name, fn in runs.items():
        ref = [t.clone() for t in fn()]
        for it in range(6):
            out = fn()
            for a, b in zip(out, ref):
                assert torch.equal(a, b), (name, it)
        del ref
        torch.cuda.empty_cache()
