"""Seeded random-shape sweep of the big bf16 GEMM kernels (256x256 row-space tiles and the persistent weight-gradient kernel)
against fp64 torch references: ragged / empty experts, N and K that fit no tile, every epilogue, both weight layouts, bf16 and
fp32 gradient outputs, accumulate, dealt and contiguous XCD order.  Shapes are drawn so that the v2 kernels are the ones that run
(N >= 256, K >= 128, M >= 2048 / Na, Nb >= 256, M >= 512)."""
import math
import os
import random

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

if torch.cuda.is_available():
    from competesmoe_amd import ops, _lib as L
    from tests.test_ops_gpu import make_groups, ref_rowspace


def _cases(n, seed):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        E = rng.choice([1, 2, 3, 5, 8, 17])
        M = rng.randrange(2048, 7000)
        N = 8 * rng.randrange(32, 140)            # 256 .. 1112
        Kd = 8 * rng.randrange(16, 190)           # 128 .. 1512
        b_layout, epi, act = rng.choice([0, 1]), rng.choice([0, 1, 2, 3]), rng.choice([1, 2, 3, 4])
        # which 256x256 kernel: 0 = the library's choice, 2 = 8 waves, 4 = one wave per SIMD (takes K = whole pairs of K-tiles: such
        # a K is drawn for it, and for half of the automatic cases, which then land on it too)
        kern = rng.choice([0, 0, 2, 4, 4])
        if kern == 4 or (kern == 0 and i % 2 == 0):
            Kd = 128 * rng.randrange(1, 13)       # 128 .. 1536
        out.append((i, E, M, N, Kd, b_layout, epi, act, kern))
    return out


N_ROW = int(os.environ.get("CSMOE_FUZZ_CASES", "28"))      # a one-off long sweep: CSMOE_FUZZ_CASES=300
N_WGRAD = int(os.environ.get("CSMOE_FUZZ_CASES", "14"))


@pytest.mark.parametrize("i,E,M,N,Kd,b_layout,epi,act,kern", _cases(N_ROW, 1234))
def test_rowspace_fuzz(i, E, M, N, Kd, b_layout, epi, act, kern):
    g = torch.Generator().manual_seed(1000 + i)
    off = make_groups(E, M, seed=i, empty=E > 2)
    A = torch.randn(M, Kd, generator=g).bfloat16().to(DEV)
    shape = (N, Kd) if b_layout == 0 else (Kd, N)
    Bs = [(torch.randn(*shape, generator=g) / math.sqrt(Kd)).bfloat16().to(DEV) for _ in range(E)]
    bias = [(torch.randn(N, generator=g) * 0.5).bfloat16().to(DEV) for _ in range(E)]
    aux = torch.randn(M, N, generator=g).bfloat16().to(DEV)
    kw = dict(epilogue=epi, act=act, kernel=kern)
    if epi in (1, 2):
        kw["bias_ptrs"] = ops.ptr_array(bias, DEV)
    if epi == 2:
        kw["want_c2"] = True
    if epi == 3:
        kw["aux"] = aux
    res = ops.grouped_gemm(A, ops.ptr_array(Bs, DEV), b_layout, Bs[0].stride(0), N, off.to(DEV), E, **kw)
    c, c2 = res if epi == 2 else (res, None)
    if kern == 0 and Kd % 128 == 0:
        # the two 256x256 kernels are independent implementations of the same sums, rounded at the same points: bit-identical
        r2 = ops.grouped_gemm(A, ops.ptr_array(Bs, DEV), b_layout, Bs[0].stride(0), N, off.to(DEV), E, **{**kw, "kernel": 2})
        r4 = ops.grouped_gemm(A, ops.ptr_array(Bs, DEV), b_layout, Bs[0].stride(0), N, off.to(DEV), E, **{**kw, "kernel": 4})
        for a_, b_ in zip(r2 if epi == 2 else (r2,), r4 if epi == 2 else (r4,)):
            assert (a_.float() - b_.float()).abs().max() <= 2 ** -7 * a_.float().abs().max()
    rc, rc2 = ref_rowspace(A, Bs, b_layout, off, bias if epi in (1, 2) else None, epi, act if epi >= 2 else 0, aux if epi == 3 else None)
    tol = dict(rtol=2 ** -7, atol=2e-2)
    assert torch.allclose(c.float(), rc.float(), **tol), (i, (c.float() - rc.float()).abs().max())
    if epi == 2:
        assert torch.allclose(c2.float(), rc2.float(), **tol), i


def _wcases(n, seed):
    rng = random.Random(seed)
    return [(i, rng.choice([1, 2, 4, 7, 16]), rng.randrange(512, 6000), 8 * rng.randrange(32, 120), 8 * rng.randrange(32, 120),
             rng.choice([torch.bfloat16, torch.float32]), rng.choice([False, True]), rng.choice([False, True])) for i in range(n)]


@pytest.mark.parametrize("i,E,M,Na,Nb,out_dtype,accumulate,dealt", _wcases(N_WGRAD, 99))
def test_wgrad_fuzz(i, E, M, Na, Nb, out_dtype, accumulate, dealt):
    g = torch.Generator().manual_seed(2000 + i)
    off = make_groups(E, M, seed=50 + i, empty=E > 2).to(DEV)
    A = torch.randn(M, Na, generator=g).bfloat16().to(DEV)
    B = torch.randn(M, Nb, generator=g).bfloat16().to(DEV)
    init = torch.randn(E, Na, Nb, generator=g).to(out_dtype).to(DEV) if accumulate else torch.full((E, Na, Nb), float("nan"), dtype=out_dtype, device=DEV)
    out = init.clone()
    ptrs = ops.ptr_array([out[e] for e in range(E)], DEV)
    ops.grouped_wgrad(A, B, off, E, out, ptrs, accumulate=accumulate, xcd_order=ops.expert_order(off, E) if dealt else None)
    ref = torch.zeros(E, Na, Nb, dtype=torch.float64, device=DEV)
    for e in range(E):
        r0, r1 = int(off[e]), int(off[e + 1])
        ref[e] = A[r0:r1].double().T @ B[r0:r1].double()
    if accumulate:
        ref = ref + init.double()
    scale = ref.abs().max().item() + 1e-9
    err = (out.double() - ref).abs().max().item() / scale
    assert err <= (2e-5 if out_dtype == torch.float32 else 2 ** -7), (i, err)
