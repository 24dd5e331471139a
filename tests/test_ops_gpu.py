"""GPU parity tests of every C-ABI kernel (through ctypes) against plain torch / the CPU oracle.

bit-exact for integer / index work and for exact-integer GEMM data; fp32 <= 1e-5; bf16 compared with a
fp32 computation of the same bf16 inputs (<= 1 bf16 ulp of rounding, i.e. rel 2^-8)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from competesmoe_amd import ops, _lib as L
    from oracle import moe_oracle as O

DEV = "cuda"


def rand_idx(T, K, E, seed=0, skew=False, empty=()):
    g = torch.Generator().manual_seed(seed)
    sc = torch.rand(T, E, generator=g)
    if skew:
        sc[:, : max(1, E // 8)] += 0.5
    for e in empty:
        sc[:, e] = -1
    return sc.topk(K, dim=-1).indices.int()


# ------------------------------------------------------------------------------------------------ binning
@pytest.mark.parametrize("T,K,E", [(1, 1, 1), (7, 2, 8), (128, 2, 8), (1000, 3, 64), (4096, 2, 64), (3000, 8, 100),
                                   (5000, 2, 300), (32768, 2, 64)])
def test_bin_tokens(T, K, E):
    idx = rand_idx(T, K, E, seed=T + K, skew=True, empty=(E - 1,) if E > 2 else ())
    counts, offsets, perm = O.bin_tokens(idx, E)
    b = ops.bin_tokens(idx.to(DEV), E)
    assert torch.equal(b.counts.cpu().long(), counts)
    assert torch.equal(b.offsets.cpu().long(), offsets)
    assert torch.equal(b.perm.cpu().long(), perm)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel())
    assert torch.equal(b.slot_of.cpu().long(), inv)


def test_bin_tokens_empty_input():
    b = ops.bin_tokens(torch.zeros(0, 2, dtype=torch.int32, device=DEV), 8)
    assert int(b.offsets[-1]) == 0 and int(b.counts.sum()) == 0


# ------------------------------------------------------------------------------------------------ dispatch
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("T,K,E,D", [(5, 2, 4, 8), (300, 2, 8, 64), (1024, 3, 16, 200), (513, 2, 8, 1027), (2048, 2, 64, 4096)])
def test_dispatch_and_bwd(T, K, E, D, dtype):
    idx = rand_idx(T, K, E, seed=D).to(DEV)
    b = ops.bin_tokens(idx, E)
    x = torch.randn(T, D, device=DEV).to(dtype)
    xs = ops.dispatch_rows(x, b)
    assert torch.equal(xs, x[(b.perm // K).long()])
    assert torch.equal(ops.dispatch_tokens(x, b), xs)
    # backward: gather-sum of K rows per token (fp32 sum, one rounding)
    dxs = torch.randn(T * K, D, device=DEV).to(dtype)
    add = torch.randn(T, D, device=DEV).to(dtype)
    dx = ops.dispatch_rows_bwd(dxs, b, T)
    ref = dxs[b.slot_of.long()].view(T, K, D).float().sum(1)
    tol = 1e-6 if dtype == torch.float32 else 2 ** -8
    assert torch.allclose(dx.float(), ref.to(dtype).float(), rtol=tol, atol=1e-6)
    dx2 = ops.dispatch_rows_bwd(dxs, b, T, add=add)
    ref2 = (ref.to(dtype).float() + add.float()).to(dtype)
    assert torch.allclose(dx2.float(), ref2.float(), rtol=tol, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("T,K,E,D", [(7, 2, 4, 8), (300, 3, 8, 64), (513, 4, 16, 200), (1024, 2, 64, 4096)])
def test_dispatch_bwd_sequential_form_is_autograds_chain(T, K, E, D, dtype):
    """csmoe_dispatch_rows_bwd with `idx` (+ `pre`, + `add`): x.dtype accumulation one expert at a time in DESCENDING expert order,
    starting from `pre`, `add` last, a rounding after every add -- the order the reference's autograd adds the gradient streams of x
    in for per-expert modules.  Against that chain written with tensor ops, bit for bit."""
    idx = rand_idx(T, K, E, seed=D + K).to(DEV)
    b = ops.bin_tokens(idx, E)
    dxs = torch.randn(T * K, D, device=DEV).to(dtype)
    pre = torch.randn(T, D, device=DEV).to(dtype)
    add = torch.randn(T, D, device=DEV).to(dtype)
    rows = dxs[b.slot_of.long()].view(T, K, D)                       # [t, k] = the row of expert idx[t, k]
    order = torch.argsort(idx.long(), dim=-1, descending=True, stable=True)
    for use_pre, use_add in ((True, True), (True, False), (False, True), (False, False)):
        acc = pre.clone() if use_pre else torch.zeros(T, D, device=DEV, dtype=dtype)
        for j in range(K):
            acc = acc + rows[torch.arange(T, device=DEV), order[:, j]]          # a dtype add: rounded
        if use_add:
            acc = acc + add
        got = ops.dispatch_rows_bwd(dxs, b, T, add=add if use_add else None, idx=idx, pre=pre if use_pre else None)
        assert torch.equal(got, acc), (use_pre, use_add, float((got.float() - acc.float()).abs().max()))
    # K <= 2 without a first addend: the chain IS the fp32 sum rounded once (what the fast kernel computes)
    if K == 2:
        assert torch.equal(ops.dispatch_rows_bwd(dxs, b, T, idx=idx), ops.dispatch_rows_bwd(dxs, b, T))


# ------------------------------------------------------------------------------------------------ router
def ref_select(scores, K, mode, round_bf16):
    s = scores.float()
    sm = torch.softmax(s, -1)
    dt = scores.dtype
    if mode == 0:
        key = sm
    elif mode == 3:
        key = torch.sigmoid(s).to(dt).float()
    else:
        key = s
    idx = O.topk_lowest_index(key.detach(), K)[1]
    v = torch.gather(key, -1, idx)
    if mode == 0:
        den = v.sum(-1, keepdim=True)
        if round_bf16:
            den = den.bfloat16().float()
        w = v / den
    elif mode == 1:
        den = v.sum(-1, keepdim=True).to(dt).float()
        w = (v / den).to(dt).float()
    elif mode == 2:
        w = torch.softmax(v, -1)
    elif mode == 4:      # top-k of the logits, sigmoid(v / scale) in the logits' dtype, / fp32 sum rounded to x.dtype
        sv = torch.sigmoid((v / SIG_SCALE).to(dt).float()).to(dt).float()
        den = sv.sum(-1, keepdim=True)
        if round_bf16:
            den = den.bfloat16().float()
        w = sv / den
    else:               # sigmoid in the scores' dtype, fp32 sum (+1e-20), fp32 quotient
        den = v.sum(-1, keepdim=True) + 1e-20
        w = v / den
    return sm, idx, w


SIG_SCALE = 2.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("T,E,K", [(1, 4, 1), (77, 8, 2), (512, 64, 2), (300, 63, 7), (256, 128, 8), (64, 300, 4)])
def test_router_select(T, E, K, mode, dtype):
    g = torch.Generator().manual_seed(E * 7 + K)
    scores = (torch.randn(T, E, generator=g) * 2).to(dtype)
    if mode == 1:
        scores = scores.abs() + 0.1   # affinities are positive (mean softplus)
    scores = scores.to(DEV)
    sm, idx, w = ops.router_select(scores, K, mode, round_sum_bf16=(dtype == torch.bfloat16), param=SIG_SCALE)
    rsm, ridx, rw = ref_select(scores.cpu(), K, mode, dtype == torch.bfloat16)
    assert torch.allclose(sm.cpu(), rsm, rtol=2e-6, atol=1e-8)
    # indices bit-exact unless the GPU softmax differs from the CPU one in the last ulp on a near-tie
    if mode in (1, 2, 3, 4) or dtype == torch.bfloat16:
        assert torch.equal(idx.cpu().long(), ridx)
    else:
        mism = (idx.cpu().long() != ridx).any(-1)
        assert mism.float().mean() < 0.01
        if mism.any():
            a = torch.gather(rsm, -1, idx.cpu().long())[mism].sort(-1).values
            bb = torch.gather(rsm, -1, ridx)[mism].sort(-1).values
            assert torch.allclose(a, bb, rtol=1e-6)
    ok = (idx.cpu().long() == ridx).all(-1)
    assert torch.allclose(w.cpu()[ok], rw[ok], rtol=(1e-5 if dtype == torch.float32 or mode in (0, 2, 3, 4) else 2 ** -7), atol=1e-7)


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_router_select_bwd_fp32(mode):
    T, E, K = 200, 16, 3
    g = torch.Generator().manual_seed(5)
    scores = torch.randn(T, E, generator=g)
    if mode == 1:
        scores = scores.abs() + 0.1
    dw = torch.randn(T, K, generator=g)
    dsm = torch.randn(T, E, generator=g)
    sd = scores.to(DEV)
    sm, idx, w = ops.router_select(sd, K, mode, False, param=SIG_SCALE)
    ds = ops.router_select_bwd(sd, K, mode, False, sm, idx, w, dw.to(DEV), dsm.to(DEV), param=SIG_SCALE)
    # autograd reference with the SAME indices
    s = scores.clone().requires_grad_(True)
    smr = torch.softmax(s, -1)
    ii = idx.cpu().long()
    if mode == 0:
        v = torch.gather(smr, -1, ii); wr = v / v.sum(-1, keepdim=True)
    elif mode == 1:
        v = torch.gather(s, -1, ii); wr = v / v.sum(-1, keepdim=True)
    elif mode == 2:
        wr = torch.softmax(torch.gather(s, -1, ii), -1)
    elif mode == 4:
        v = torch.sigmoid(torch.gather(s, -1, ii) / SIG_SCALE); wr = v / v.sum(-1, keepdim=True)
    else:
        v = torch.gather(torch.sigmoid(s), -1, ii); wr = v / (v.sum(-1, keepdim=True) + 1e-20)
    ((wr * dw).sum() + (smr * dsm).sum()).backward()
    assert torch.allclose(ds.cpu(), s.grad, rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------------------------------------ grouped GEMM
def make_groups(E, M, seed, empty=True):
    g = torch.Generator().manual_seed(seed)
    cuts = torch.sort(torch.randint(0, M + 1, (E - 1,), generator=g)).values
    off = torch.cat([torch.zeros(1, dtype=torch.long), cuts, torch.tensor([M])])
    if empty and E > 2:
        off[2] = off[1]          # expert 1 empty
    return off.int()


def ref_rowspace(A, Bs, b_layout, off, bias, epi, act, aux):
    M = A.shape[0]
    N = Bs[0].shape[0] if b_layout == 0 else Bs[0].shape[1]
    dt = A.dtype
    out = torch.zeros(M, N, dtype=torch.float64, device=A.device)
    for e in range(len(Bs)):
        r0, r1 = int(off[e]), int(off[e + 1])
        if r1 > r0:
            Bm = Bs[e].double()
            out[r0:r1] = A[r0:r1].double() @ (Bm.T if b_layout == 0 else Bm)
            if bias is not None and epi in (1, 2):
                out[r0:r1] += bias[e].double()
    actf = {0: lambda h: h, 1: torch.relu, 2: torch.nn.functional.gelu,
            3: lambda h: torch.nn.functional.gelu(h, approximate="tanh"), 4: torch.nn.functional.silu}[act]
    if epi == 3:
        h = aux.double().requires_grad_(True)
        d = torch.autograd.grad(actf(h).sum(), h)[0]
        return (out.to(dt).double() * d).to(dt), None
    c = out.to(dt)
    c2 = actf(c.double()).to(dt) if epi == 2 else None
    return c, c2


GEMM_SHAPES = [  # E, M, N, Kd
    (1, 5, 8, 8), (4, 100, 64, 64), (8, 700, 200, 136), (3, 300, 72, 328), (8, 1500, 256, 192), (64, 5000, 384, 256),
]


@pytest.mark.parametrize("force_generic", [True, False])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("b_layout", [0, 1])
@pytest.mark.parametrize("E,M,N,Kd", GEMM_SHAPES)
def test_grouped_gemm_plain_and_bias_act(E, M, N, Kd, b_layout, dtype, force_generic):
    g = torch.Generator().manual_seed(M + N)
    off = make_groups(E, M, seed=M)
    A = torch.randn(M, Kd, generator=g).to(dtype).to(DEV)
    shape = (N, Kd) if b_layout == 0 else (Kd, N)
    Bs = [(torch.randn(*shape, generator=g) / math.sqrt(Kd)).to(dtype).to(DEV) for _ in range(E)]
    bias = [(torch.randn(N, generator=g) * 0.5).to(dtype).to(DEV) for _ in range(E)]
    bp, biasp = ops.ptr_array(Bs, DEV), ops.ptr_array(bias, DEV)
    offd = off.to(DEV)
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=2 ** -7, atol=2e-2)
    for act in (2, 1, 3):
        c, c2 = ops.grouped_gemm(A, bp, b_layout, Bs[0].stride(0), N, offd, E, bias_ptrs=biasp, epilogue=L.EPI_BIAS_ACT, act=act,
                                 want_c2=True, force_generic=force_generic)
        rc, rc2 = ref_rowspace(A, Bs, b_layout, off, bias, 2, act, None)
        assert torch.allclose(c.float(), rc.float(), **tol), (c.float() - rc.float()).abs().max()
        assert torch.allclose(c2.float(), rc2.float(), **tol)
    c = ops.grouped_gemm(A, bp, b_layout, Bs[0].stride(0), N, offd, E, force_generic=force_generic)
    rc, _ = ref_rowspace(A, Bs, b_layout, off, None, 0, 0, None)
    assert torch.allclose(c.float(), rc.float(), **tol)
    # activation-gradient epilogue
    aux = torch.randn(M, N, generator=g).to(dtype).to(DEV)
    c = ops.grouped_gemm(A, bp, b_layout, Bs[0].stride(0), N, offd, E, epilogue=L.EPI_ACTGRAD, act=2, aux=aux,
                         force_generic=force_generic)
    rc, _ = ref_rowspace(A, Bs, b_layout, off, None, 3, 2, aux)
    assert torch.allclose(c.float(), rc.float(), **tol)


@pytest.mark.parametrize("force_generic", [True, False])
@pytest.mark.parametrize("E,M,N,Kd", [(4, 100, 64, 64), (8, 1500, 256, 192), (8, 3000, 512, 320)])
def test_grouped_gemm_activated_output_only_and_post_rounding_bias(E, M, N, Kd, force_generic):
    """(1) C == null with EPI_BIAS_ACT: only the activated output is written, bit-identical to the two-output launch (what the
    ReLU experts of the pretrain stack use: act'(pre) = (act(pre) > 0), so ACTGRAD with aux = hact equals aux = hpre).
    (2) EPI_ROUND_BIAS32_ACT: act(round(acc) + fp32 bias), the pretrain stack's `cvmm(...) + bias[sel]` under autocast."""
    g = torch.Generator().manual_seed(M + N + 1)
    off = make_groups(E, M, seed=M)
    offd = off.to(DEV)
    A = torch.randn(M, Kd, generator=g).bfloat16().to(DEV)
    Bs = [(torch.randn(Kd, N, generator=g) / math.sqrt(Kd)).bfloat16().to(DEV) for _ in range(E)]
    bias = [(torch.randn(N, generator=g) * 0.5).bfloat16().to(DEV) for _ in range(E)]
    bias32 = [(torch.randn(N, generator=g) * 0.5).to(DEV) for _ in range(E)]
    bp = ops.ptr_array(Bs, DEV)
    c, c2 = ops.grouped_gemm(A, bp, 1, N, N, offd, E, bias_ptrs=ops.ptr_array(bias, DEV), epilogue=L.EPI_BIAS_ACT, act=L.ACT_RELU,
                             want_c2=True, force_generic=force_generic)
    none, only = ops.grouped_gemm(A, bp, 1, N, N, offd, E, bias_ptrs=ops.ptr_array(bias, DEV), epilogue=L.EPI_BIAS_ACT,
                                  act=L.ACT_RELU, want_c2=True, want_c=False, force_generic=force_generic)
    assert none is None and torch.equal(only, c2)
    dy = torch.randn(M, Kd, generator=g).bfloat16().to(DEV)
    Wt = [(torch.randn(N, Kd, generator=g) / math.sqrt(Kd)).bfloat16().to(DEV) for _ in range(E)]     # [N, Kd]: dh = dy @ Wt^T
    d1 = ops.grouped_gemm(dy, ops.ptr_array(Wt, DEV), 0, Kd, N, offd, E, epilogue=L.EPI_ACTGRAD, act=L.ACT_RELU, aux=c,
                          force_generic=force_generic)
    d2 = ops.grouped_gemm(dy, ops.ptr_array(Wt, DEV), 0, Kd, N, offd, E, epilogue=L.EPI_ACTGRAD, act=L.ACT_RELU, aux=c2,
                          force_generic=force_generic)
    assert torch.equal(d1, d2)
    # post-rounding fp32 bias
    pre, act = ops.grouped_gemm(A, bp, 1, N, N, offd, E, bias_ptrs=ops.ptr_array(bias32, DEV), epilogue=L.EPI_ROUND_BIAS32_ACT,
                                act=L.ACT_RELU, want_c2=True, force_generic=force_generic)
    plain = ops.grouped_gemm(A, bp, 1, N, N, offd, E, force_generic=force_generic)            # round(acc)
    u = plain.float()
    for e in range(E):
        u[int(off[e]):int(off[e + 1])] += bias32[e]
    assert torch.equal(pre, u.bfloat16()) and torch.equal(act, torch.relu(u).bfloat16())


@pytest.mark.parametrize("b_layout", [0, 1])
@pytest.mark.parametrize("force_generic", [True, False])
def test_grouped_gemm_exact_integers(b_layout, force_generic):
    """{-1,0,1} data, K<=128: every product and partial sum is exact in bf16 -> results must be bit-identical.
    Asymmetric operands catch a transposed fragment map (guide §3)."""
    E, M, N, Kd = 5, 777, 264, 128
    g = torch.Generator().manual_seed(3)
    off = make_groups(E, M, seed=11)
    A = torch.randint(-1, 2, (M, Kd), generator=g).bfloat16().to(DEV)
    shape = (N, Kd) if b_layout == 0 else (Kd, N)
    Bs = [torch.randint(-1, 2, shape, generator=g).bfloat16().to(DEV) for _ in range(E)]
    c = ops.grouped_gemm(A, ops.ptr_array(Bs, DEV), b_layout, Bs[0].stride(0), N, off.to(DEV), E, force_generic=force_generic)
    rc, _ = ref_rowspace(A, Bs, b_layout, off, None, 0, 0, None)
    assert torch.equal(c, rc)


@pytest.mark.parametrize("force_generic", [True, False])
@pytest.mark.parametrize("dtype,out_dtype", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16),
                                             (torch.bfloat16, torch.float32)])
@pytest.mark.parametrize("E,M,Na,Nb", [(1, 3, 8, 8), (4, 100, 64, 72), (8, 900, 200, 136), (8, 2000, 256, 384), (64, 4000, 128, 64)])
def test_grouped_wgrad(E, M, Na, Nb, dtype, out_dtype, force_generic):
    g = torch.Generator().manual_seed(M)
    off = make_groups(E, M, seed=M + 1)
    A = torch.randn(M, Na, generator=g).to(dtype).to(DEV)
    B = torch.randn(M, Nb, generator=g).to(dtype).to(DEV)
    out = torch.full((E, Na, Nb), float("nan"), dtype=out_dtype, device=DEV)
    ptrs = ops.ptr_array([out[e] for e in range(E)], DEV)
    ops.grouped_wgrad(A, B, off.to(DEV), E, out, ptrs, force_generic=force_generic)
    ref = torch.zeros(E, Na, Nb, dtype=torch.float64, device=DEV)
    for e in range(E):
        r0, r1 = int(off[e]), int(off[e + 1])
        ref[e] = A[r0:r1].double().T @ B[r0:r1].double()
    scale = ref.abs().max().item() + 1e-9
    err = (out.double() - ref).abs().max().item() / scale
    assert err <= (1e-5 if out_dtype == torch.float32 and dtype == torch.float32 else (1e-5 if out_dtype == torch.float32 else 2 ** -8)), err
    # accumulate
    if out_dtype == torch.float32:
        ops.grouped_wgrad(A, B, off.to(DEV), E, out, ptrs, accumulate=True, force_generic=force_generic)
        assert (out.double() - 2 * ref).abs().max().item() / scale <= 2e-5


def test_wgrad_exact_integers():
    E, M, Na, Nb = 3, 100, 136, 264
    g = torch.Generator().manual_seed(9)
    off = torch.tensor([0, 100, 100, 100], dtype=torch.int32)   # <= 128 rows per expert keeps sums exact in bf16
    A = torch.randint(-1, 2, (M, Na), generator=g).bfloat16().to(DEV)
    B = torch.randint(-1, 2, (M, Nb), generator=g).bfloat16().to(DEV)
    for fg in (True, False):
        out = torch.full((E, Na, Nb), float("nan"), dtype=torch.bfloat16, device=DEV)
        ops.grouped_wgrad(A, B, off.to(DEV), E, out, ops.ptr_array([out[e] for e in range(E)], DEV), force_generic=fg)
        ref = torch.zeros(E, Na, Nb, device=DEV)
        ref[0] = A.float().T @ B.float()
        assert torch.equal(out.float(), ref), fg


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dense_gemm_and_gate(dtype):
    T, D, E = 1000, 256, 64
    g = torch.Generator().manual_seed(1)
    x = torch.randn(T, D, generator=g).to(dtype).to(DEV)
    wg = (torch.randn(E, D, generator=g) * 0.02).to(dtype).to(DEV)
    lg = ops.gate_logits(x, wg)
    ref = (x.double() @ wg.double().T).to(dtype)
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=2 ** -7, atol=1e-2)
    assert torch.allclose(lg.float(), ref.float(), **tol)
    W = (torch.randn(D, 320, generator=g) / 16).to(dtype).to(DEV)
    bias = torch.randn(320, generator=g).to(dtype).to(DEV)
    c, c2 = ops.dense_gemm(x, W, L.B_KN, bias=bias, epilogue=L.EPI_BIAS_ACT, act=L.ACT_RELU, want_c2=True)
    r = (x.double() @ W.double() + bias.double()).to(dtype)
    assert torch.allclose(c.float(), r.float(), **tol) and torch.allclose(c2.float(), torch.relu(r).float(), **tol)
    gw = ops.dense_wgrad(x, c)
    rg = (x.double().T @ c.double())
    assert (gw.double() - rg).abs().max() / rg.abs().max() <= (1e-5 if dtype == torch.float32 else 2 ** -8)


# ------------------------------------------------------------------------------------------------ combine
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("T,K,E,D", [(3, 1, 2, 8), (200, 2, 8, 64), (1000, 3, 16, 200), (512, 2, 64, 4096), (100, 8, 64, 33)])
def test_combine_and_bwd(T, K, E, D, mode, dtype):
    idx = rand_idx(T, K, E, seed=D + K).to(DEV)
    b = ops.bin_tokens(idx, E)
    g = torch.Generator().manual_seed(D)
    y = torch.randn(T * K, D, generator=g).to(dtype).to(DEV)
    w = torch.rand(T, K, generator=g).to(DEV)
    if mode == 2:
        w = w.to(dtype).float()
    out = ops.combine(y, b, idx, w, mode, T)
    yk = y[b.slot_of.long()].view(T, K, D)
    if mode == 1:
        ref = (w.unsqueeze(-1) * yk.float()).sum(1).to(dtype)
        tol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=2 ** -7, atol=1e-6)
        assert torch.allclose(out.float(), ref.float(), **tol)
    else:
        order = idx.long().argsort(dim=-1, stable=True)
        ref = torch.zeros(T, D, dtype=dtype, device=DEV)
        for j in range(K):
            k = order[:, j]
            yj = yk[torch.arange(T), k].float()
            wj = w[torch.arange(T), k].unsqueeze(-1)
            prod = wj * yj
            if mode == 2:
                prod = prod.to(dtype).float()
            ref = (ref.float() + prod).to(dtype)
        assert torch.equal(out, ref)
    dout = torch.randn(T, D, generator=g).to(dtype).to(DEV)
    dy, dw = ops.combine_bwd(dout, y, b, w)
    flat_t = (b.perm // K).long()
    wf = w.flatten()[b.perm.long()]
    rdy = (dout[flat_t].float() * wf.unsqueeze(-1)).to(dtype)
    assert torch.equal(dy, rdy)
    rdw = (dout.float().unsqueeze(1) * yk.float()).sum(-1)
    assert torch.allclose(dw, rdw, rtol=1e-4, atol=1e-4 * D ** 0.5)


# ------------------------------------------------------------------------------------------------ small reductions
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_colsum_and_softplus(dtype):
    E, M, N = 8, 1500, 200
    off = make_groups(E, M, seed=4)
    g = torch.Generator().manual_seed(2)
    G = torch.randn(M, N, generator=g).to(dtype).to(DEV)
    out = torch.full((E, N), float("nan"), dtype=dtype, device=DEV)
    ops.grouped_colsum(G, off.to(DEV), E, out, ops.ptr_array([out[e] for e in range(E)], DEV))
    ref = torch.stack([G[int(off[e]):int(off[e + 1])].double().sum(0) for e in range(E)])
    assert (out.double() - ref).abs().max() <= (1e-4 if dtype == torch.float32 else 0.5)
    d = ops.dense_colsum(G)
    assert (d.double() - G.double().sum(0)).abs().max() <= (1e-3 if dtype == torch.float32 else 1.0)
    y = (torch.randn(300, 96, generator=g) * 3).to(dtype).to(DEV)
    aff = ops.softplus_mean(y)
    raff = torch.nn.functional.softplus(y.float()).to(dtype).float().mean(-1).to(dtype)
    assert torch.allclose(aff.float(), raff.float(), rtol=(1e-5 if dtype == torch.float32 else 2 ** -7), atol=1e-6)
    daff = torch.randn(300, generator=g).to(dtype).to(DEV)
    dy = ops.softplus_mean_bwd(y, daff)
    rdy = ((daff.float() / 96).to(dtype).float().unsqueeze(-1) * torch.sigmoid(y.float())).to(dtype)
    assert torch.allclose(dy.float(), rdy.float(), rtol=(1e-5 if dtype == torch.float32 else 2 ** -7), atol=1e-7)
    if dtype == torch.bfloat16:
        # CUDA-autocast semantics (pretrain stack): fp32 softplus / mean of the bf16 rows, fp32 affinities, one rounding of dy
        aff32 = ops.softplus_mean(y, torch.float32)
        assert aff32.dtype == torch.float32
        r32 = torch.nn.functional.softplus(y.float()).mean(-1)
        assert torch.allclose(aff32, r32, rtol=2e-6, atol=1e-7)
        d32 = torch.randn(300, generator=g).to(DEV)
        dy32 = ops.softplus_mean_bwd(y, d32)
        rdy32 = ((d32 / 96).unsqueeze(-1) * torch.sigmoid(y.float())).to(dtype)
        assert torch.allclose(dy32.float(), rdy32.float(), rtol=2 ** -8, atol=1e-9)


def test_argument_errors_raise():
    x = torch.randn(4, 8, device=DEV)
    with pytest.raises(ValueError):
        ops.router_select(x, 9, 0, False)          # K > E
    with pytest.raises(ValueError):
        ops.router_select(x.half(), 2, 0, False)   # unsupported dtype
    with pytest.raises(ValueError):
        ops.bin_tokens(torch.zeros(4, 2, dtype=torch.int64, device=DEV), 4)


# ---------------------------------------------------------------------------------------------------- diversity loss
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("T,K,D", [(50, 2, 64), (129, 3, 200), (33, 4, 4096), (7, 8, 96), (64, 1, 32)])
def test_pair_cosine_matches_torch_autograd(dt, T, K, D):
    """DiversityLoss (csmoe_pair_cosine / _bwd) against the reference formula (normalize -> bmm -> zero diagonal -> mean) and
    its autograd, both evaluated in fp32 on the same x.dtype inputs.  fp32 1e-5, bf16 gradients 2e-3 (rounded to bf16)."""
    from competesmoe_amd.functional import DiversityLoss
    from tests.golden_util import rel_l2
    g = torch.Generator(device=DEV).manual_seed(T * 13 + K)
    y = (torch.randn(2, T, K, D, device=DEV, generator=g) * 1.5 + 0.2).to(dt).requires_grad_(True)
    loss = DiversityLoss.apply(y)
    (loss * 3.0).backward()
    yr = y.detach().clone().requires_grad_(True)
    eo = yr.to(torch.float32)
    nrm = torch.nn.functional.normalize(eo, p=2, dim=-1).view(-1, K, D)
    sim = torch.bmm(nrm, nrm.transpose(1, 2)) * (1 - torch.eye(K, device=DEV))
    ref = sim.mean()
    (ref * 3.0).backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    if K > 1:
        assert rel_l2(y.grad, yr.grad) <= (1e-5 if dt == torch.float32 else 4e-3)
    else:
        assert float(y.grad.abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------------- wgrad load balance
@pytest.mark.parametrize("E", [64, 61, 8, 3, 130])
def test_expert_order_deals_by_row_count(E):
    g = torch.Generator().manual_seed(E)
    counts = torch.randint(0, 500, (E,), generator=g)
    counts[: min(8, E)] += 3000                       # hot experts with consecutive indices (the skewed regime)
    off = torch.zeros(E + 1, dtype=torch.int32)
    off[1:] = counts.cumsum(0)
    order = ops.expert_order(off.to(DEV), E).cpu()
    slots = (E + 7) // 8
    tab = order[: 8 * slots].view(8, slots)
    seen = sorted(int(v) for v in tab.flatten() if v >= 0)
    assert seen == list(range(E))                      # every expert exactly once
    ranks = sorted(range(E), key=lambda e: (-int(counts[e]), e))
    for r, e in enumerate(ranks):
        k, pos = r // 8, r % 8
        x = 7 - pos if k & 1 else pos
        assert int(tab[x, k]) == e
    loads = [sum(int(counts[int(e)]) for e in tab[x] if e >= 0) for x in range(8)]
    if E >= 16:
        assert max(loads) <= 1.6 * (sum(loads) / 8)    # the hot experts end up on different XCDs


def test_wgrad_with_dealt_order_is_bit_identical():
    E, D, F_ = 16, 512, 768
    g = torch.Generator(device=DEV).manual_seed(3)
    counts = torch.tensor([3000, 2500, 40, 0, 700, 1, 256, 257, 900, 33, 512, 64, 2000, 5, 128, 1024])
    off = torch.zeros(E + 1, dtype=torch.int32)
    off[1:] = counts.cumsum(0)
    off = off.to(DEV)
    M = int(counts.sum())
    a = torch.randn(M, F_, device=DEV, generator=g).to(torch.bfloat16)
    b = torch.randn(M, D, device=DEV, generator=g).to(torch.bfloat16)
    outs = []
    for order in (None, ops.expert_order(off, E)):
        out = torch.full((E, F_, D), float("nan"), device=DEV, dtype=torch.bfloat16)
        ptrs = out.data_ptr() + torch.arange(E, device=DEV, dtype=torch.int64) * (F_ * D * 2)
        ops.grouped_wgrad(a, b, off, E, out, ptrs, xcd_order=order)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    ref = torch.stack([a[int(off[e]):int(off[e + 1])].float().t() @ b[int(off[e]):int(off[e + 1])].float() for e in range(E)])
    assert rel_l2_(outs[1].float(), ref) <= 2e-3


def rel_l2_(x, y):
    return float((x.double() - y.double()).norm() / (y.double().norm() + 1e-30))


# ---------------------------------------------------------------------------------------------------- skinny gate (few experts)
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("T,D,E", [(100, 64, 4), (12800, 1152, 4), (333, 4096, 3), (17, 200, 2), (50, 72, 1)])
def test_gate_small_kernels(dt, T, D, E):
    """gate logits / dx / dWg for E <= 4 (row-pass kernels) against fp32 torch products of the same x.dtype operands."""
    if (dt == torch.bfloat16 and D % 8) or (dt == torch.float32 and D % 4):
        pytest.skip("D not a multiple of the 16-byte chunk")
    g = torch.Generator(device=DEV).manual_seed(T + E)
    x = torch.randn(T, D, device=DEV, generator=g).to(dt)
    wg = (torch.randn(E, D, device=DEV, generator=g) * 0.1).to(dt)
    dl = torch.randn(T, E, device=DEV, generator=g).to(dt)
    tol = 1e-5 if dt == torch.float32 else 2e-3
    assert ops.gate_bwd_small_ok(D, E, dt)
    lg = ops.gate_logits(x, wg)
    assert rel_l2_(lg.float(), (x.float() @ wg.float().t()).to(dt).float()) <= tol
    dx = ops.gate_bwd_dx(dl, wg)
    assert rel_l2_(dx.float(), (dl.float() @ wg.float()).to(dt).float()) <= tol
    dw = ops.gate_bwd_dw(dl, x, torch.float32)
    assert rel_l2_(dw, dl.float().t() @ x.float()) <= (1e-5 if dt == torch.float32 else 1e-4)
    assert not ops.gate_bwd_small_ok(D, 64, dt)


# ------------------------------------------------------------------------------------------------ router aux losses
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,E,K", [(1, 1, 2, 1), (2, 96, 8, 2), (3, 700, 64, 2), (16, 2048, 64, 2), (2, 513, 300, 4), (5, 33, 4, 2)])
def test_router_aux_matches_torch_autograd(dt, B, N, E, K):
    """balance (moe.py:90-110) and z-loss (moe.py:71-88) kernels against the torch formulation of the oracle, values and gradients."""
    from competesmoe_amd.functional import RouterAux, RouterSelect
    torch.manual_seed(B * 1000 + N + E)
    logits = (torch.randn(B, N, E, device=DEV) * 2).to(dt)
    with torch.no_grad():
        sm, idx, _ = RouterSelect.apply(logits.reshape(-1, E), K, L.SEL_SOFTMAX, dt == torch.bfloat16)
    idx = idx.view(B, N, K)
    # torch reference on the same softmax values
    lr = logits.clone().requires_grad_(True)
    sr = sm.view(B, N, E).clone().requires_grad_(True)
    dens = torch.nn.functional.one_hot(idx[..., 0].long(), E).float().mean(dim=-2)
    bal_ref = (sr.mean(dim=-2) * dens).mean() * float(E ** 2)
    z_ref = torch.square(torch.logsumexp(lr, dim=-1)).mean()
    (bal_ref * 0.7 + z_ref.float() * 1.3).backward()
    lk = logits.clone().requires_grad_(True)
    sk = sm.view(B, N, E).clone().requires_grad_(True)
    bal, z = RouterAux.apply(lk, sk, idx)
    assert bal.dtype == torch.float32 and z.dtype == dt and bal.dim() == 0 and z.dim() == 0
    (bal * 0.7 + z.float() * 1.3).backward()
    assert abs(float(bal) - float(bal_ref)) <= 1e-5 * abs(float(bal_ref))
    ztol = 1e-5 if dt == torch.float32 else 2.0 ** -7        # bf16: the reference rounds lse, its square and the mean to bf16
    assert abs(float(z) - float(z_ref)) <= ztol * abs(float(z_ref)), (float(z), float(z_ref))
    assert (sk.grad - sr.grad).abs().max() <= 1e-6 * sr.grad.abs().max() + 1e-12
    gtol = 1e-5 if dt == torch.float32 else 2.0 ** -6
    gk, gr = lk.grad.float(), lr.grad.float()
    assert float((gk - gr).norm() / gr.norm()) <= gtol
    # balance alone
    b2, z2 = RouterAux.apply(None, sm.view(B, N, E), idx)
    assert float(b2) == float(bal) and float(z2) == 0.0


# ------------------------------------------------------------------------------------------------ K = 2 fast paths
_K2_SCRIPT = r"""
import sys, torch
from competesmoe_amd import ops, _lib as L
T, K, E, D = 1500, 2, 16, 1024
g = torch.Generator().manual_seed(11)
idx = torch.rand(T, E, generator=g).topk(K, -1).indices.int().cuda()
b = ops.bin_tokens(idx, E)
y = torch.randn(T * K, D, generator=g).bfloat16().cuda()
w = torch.rand(T, K, generator=g).cuda()
res = torch.randn(T, D, generator=g).bfloat16().cuda()
res32 = torch.randn(T, D, generator=g).cuda()
ob = torch.randn(D, generator=g).bfloat16().cuda()
dout = torch.randn(T, D, generator=g).bfloat16().cuda()
dout32 = torch.randn(T, D, generator=g).cuda()
out = {}
for mode in (0, 1, 2):
    out[f"c{mode}"] = ops.combine(y, b, idx, w, mode, T)
    out[f"c{mode}_res"] = ops.combine(y, b, idx, w, mode, T, residual=res)
out["c1_ob"] = ops.combine(y, b, idx, w, 1, T, obias=ob)
out["c1_res32"] = ops.combine(y, b, idx, w, 1, T, residual=res32)
out["gather_sum"] = ops.dispatch_rows_bwd(y, b, T)
out["gather_sum_add"] = ops.dispatch_rows_bwd(y, b, T, add=res)
out["dy"], out["dw"] = ops.combine_bwd(dout, y, b, w)
out["dy_nody"], _ = ops.combine_bwd(dout, None, b, w, want_dw=False)
out["dy32"], out["dw32"] = ops.combine_bwd(dout32, y, b, w, act_dtype=torch.bfloat16)
torch.cuda.synchronize()
torch.save({k: v.cpu() for k, v in out.items()}, sys.argv[1])
"""


def test_k2_fast_kernels_are_bit_identical_to_the_generic_ones(tmp_path):
    """combine_k2_kernel / combine_bwd_k2_kernel (K = 2, D % 512 == 0) against the generic kernels (CSMOE_COMBINE_GENERIC=1 in a
    second process), every rounding rule, bias / residual (bf16 and fp32) variants, bf16 and fp32 upstream gradients."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for tag, extra in (("fast", {}), ("generic", {"CSMOE_COMBINE_GENERIC": "1"})):
        f = str(tmp_path / f"{tag}.pt")
        env = dict(os.environ, **extra)
        env.pop("CSMOE_COMBINE_GENERIC", None) if not extra else None
        r = subprocess.run([sys.executable, "-c", _K2_SCRIPT, f], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(torch.load(f, weights_only=True))
    fast, gen = outs
    assert set(fast) == set(gen)
    for k in fast:
        assert torch.equal(fast[k], gen[k]), k


@pytest.mark.parametrize("E,P,N", [(1, 2, 8), (4, 2, 1000), (3, 5, 4099), (4, 2, 4304 * 64)])
def test_sum_partials_and_chunk_offsets(E, P, N):
    """fp32 partial rows -> per-expert sums in bf16 / fp32 in one launch (fp32-in, bf16-out column sums), and the chunk offsets of
    csmoe_chunk_offsets against their definition."""
    part = torch.randn(E * P, N, device=DEV)
    for od in (torch.bfloat16, torch.float32):
        out = ops.sum_partials(part, E, P, od)
        ref = part.view(E, P, N).double().sum(1)
        assert out.dtype == od and out.shape == (E, N)
        assert float((out.double() - ref).abs().max()) <= (2 ** -8 if od == torch.bfloat16 else 1e-6) * float(ref.abs().max())
    d = ops.dense_colsum(part, out_dtype=torch.bfloat16)
    assert d.dtype == torch.bfloat16 and float((d.double() - part.double().sum(0)).abs().max()) <= 2 ** -8 * float(part.double().sum(0).abs().max())
    counts = torch.randint(0, 5000, (E,))
    counts[0] = 0
    off = torch.zeros(E + 1, dtype=torch.int32)
    off[1:] = counts.cumsum(0)
    for align in (1, 64):
        co = ops.chunk_offsets(off.to(DEV), E, P, align).cpu().long()
        want = []
        for e in range(E):
            lo, hi = int(off[e]), int(off[e + 1])
            for j in range(P):
                st = ((hi - lo) * j) // P
                st = (st + align - 1) // align * align
                want.append(min(hi, lo + st))
        want.append(int(off[E]))
        assert co.tolist() == want


@pytest.mark.parametrize("E,M,N,Kd", [(4, 2100, 256, 128), (8, 5000, 512, 320), (3, 3000, 776, 1032), (64, 9000, 1024, 512)])
def test_grouped_gemm_from_fp32_masters_is_bit_identical_to_cast_then_gemm(E, M, N, Kd):
    """csmoe_grouped_gemm_f32w converts the fp32 weight tiles inside the tile fill (cvmm.py:126-140 does the same per tile): the
    result, every epilogue, and the bf16 copy it writes for the backward must equal "weights.to(bf16), then csmoe_grouped_gemm"
    BIT FOR BIT (the conversion is the same round-to-nearest-even, the K order of the accumulation is the same).  Experts without
    rows keep their slice of the copy untouched."""
    g = torch.Generator().manual_seed(E + N)
    off = make_groups(E, M, seed=N)                     # expert 1 is empty
    offd = off.to(DEV)
    A = torch.randn(M, Kd, generator=g).bfloat16().to(DEV)
    W32 = (torch.randn(E, Kd, N, generator=g) / math.sqrt(Kd)).to(DEV)
    Wb = W32.bfloat16()
    bias = (torch.randn(E, N, generator=g) * 0.5).bfloat16().to(DEV)
    bias32 = (torch.randn(E, N, generator=g) * 0.5).to(DEV)
    bp = ops.ptr_table(Wb, E, Kd * N * 2)
    copy = torch.full((E, Kd, N), 7.0, dtype=torch.bfloat16, device=DEV)
    c = ops.grouped_gemm_f32w(A, W32, offd, copy=copy)
    ref = ops.grouped_gemm(A, bp, L.B_KN, N, N, offd, E)
    assert torch.equal(c, ref)
    for e in range(E):
        if int(off[e + 1]) > int(off[e]):
            assert torch.equal(copy[e], Wb[e]), e
        else:
            assert bool((copy[e] == 7.0).all()), e
    pre, act = ops.grouped_gemm_f32w(A, W32, offd, bias_ptrs=ops.ptr_table(bias, E, N * 2), epilogue=L.EPI_BIAS_ACT, act=L.ACT_GELU,
                                     want_c2=True)
    rpre, ract = ops.grouped_gemm(A, bp, L.B_KN, N, N, offd, E, bias_ptrs=ops.ptr_table(bias, E, N * 2), epilogue=L.EPI_BIAS_ACT,
                                  act=L.ACT_GELU, want_c2=True)
    assert torch.equal(pre, rpre) and torch.equal(act, ract)
    none, only = ops.grouped_gemm_f32w(A, W32, offd, bias_ptrs=ops.ptr_table(bias32, E, N * 4), epilogue=L.EPI_ROUND_BIAS32_ACT,
                                       act=L.ACT_RELU, want_c2=True, want_c=False)
    _, ronly = ops.grouped_gemm(A, bp, L.B_KN, N, N, offd, E, bias_ptrs=ops.ptr_table(bias32, E, N * 4), epilogue=L.EPI_ROUND_BIAS32_ACT,
                                act=L.ACT_RELU, want_c2=True, want_c=False)
    assert none is None and torch.equal(only, ronly)


# ------------------------------------------------------------------------------------------------ one-pass router
@pytest.mark.gpu
@pytest.mark.parametrize("T,D,E,K", [(1, 64, 4, 2), (63, 256, 8, 2), (65, 1152, 4, 2), (300, 264, 40, 4), (1000, 4096, 64, 2),
                                     (257, 520, 16, 1), (128, 1024, 48, 8)])
@pytest.mark.parametrize("mode_name", ["SEL_SOFTMAX", "SEL_SIGMOID", "SEL_TOPK_SOFTMAX", "SEL_RAW"])
def test_gate_select_equals_gate_then_select(T, D, E, K, mode_name):
    """csmoe_gate_select = csmoe_gate_logits + csmoe_router_select + the counting pass of csmoe_bin_tokens.  The logits come from a
    different kernel (another MFMA tiling of the same fp32 sums): equal to 1 bf16 ulp; on ITS OWN logits the selection must give the
    bits router_select gives, and the bins built from its histogram must be the bins bin_tokens builds."""
    dev = "cuda"
    mode = getattr(L, mode_name)
    g = torch.Generator(device=dev).manual_seed(T * 7 + E)
    x = torch.randn(T, D, device=dev, generator=g).to(torch.bfloat16)
    wg = (torch.randn(E, D, device=dev, generator=g) * D ** -0.5).to(torch.bfloat16)
    assert ops.gate_select_ok(x, wg, K)
    logits, sm, idx, w = ops.gate_select(x, wg, K, mode, True)
    ref_logits = ops.gate_logits(x, wg)
    d = (logits.float() - ref_logits.float()).abs()
    assert (d <= 2.0 ** -7 * ref_logits.float().abs().clamp_min(2.0 ** -6)).all(), d.max()
    exact = (x.double() @ wg.double().t())
    assert (logits.double() - exact).abs().max() <= 2.0 ** -7 * exact.abs().max().clamp_min(1.0)
    sm2, idx2, w2 = ops.router_select(logits, K, mode, True)
    assert torch.equal(idx, idx2) and torch.equal(w, w2) and torch.equal(sm, sm2)
    b_hist = ops.bin_tokens(idx, E)                 # picks the histogram of this routing up
    assert ops._route_hist_for(idx, E) is not None
    idx_copy = idx.clone()
    b_plain = ops.bin_tokens(idx_copy, E)
    assert ops._route_hist_for(idx_copy, E) is None
    for a, b in ((b_hist.counts, b_plain.counts), (b_hist.offsets, b_plain.offsets), (b_hist.perm, b_plain.perm),
                 (b_hist.slot_of, b_plain.slot_of)):
        assert torch.equal(a, b)
    idx[0, 0] = (idx[0, 0] + 1) % E                 # an in-place edit retires the histogram
    assert ops._route_hist_for(idx, E) is None


@pytest.mark.gpu
def test_smoe_layer_fused_router_equals_two_launch_router(monkeypatch):
    """The LLaVA `smoe` layer with the one-pass router against the same layer with CSMOE_FUSED_ROUTER=0: same routing, outputs and
    gradients to the tolerance of the logits' different MFMA tiling."""
    from competesmoe_amd.moe.smoe import SMoeLayer
    import torch.nn as nn
    dev = "cuda"
    torch.manual_seed(3)
    D, F, E, K = 256, 512, 8, 2
    expert = nn.Sequential(nn.Linear(D, F), nn.GELU(), nn.Linear(F, D))
    import types
    args = types.SimpleNamespace(balance_loss_coef=0.01, router_z_loss_coef=0.001)
    layer = SMoeLayer(D, D, E, K, expert, args).to(dev).to(torch.bfloat16)
    x = (torch.randn(2, 96, D, device=dev) * 0.5).to(torch.bfloat16)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("CSMOE_FUSED_ROUTER", flag)
        xi = x.clone().requires_grad_(True)
        layer.zero_grad(set_to_none=True)
        out, aux, _, _ = layer(xi)
        (out.float().pow(2).mean() + aux.float()).backward()
        outs.append((out.detach().float(), aux.detach().float(), xi.grad.float(), layer.gate.weight.grad.float(),
                     layer.log_metrics["selected_experts"].clone()))
    a, b = outs
    same_rows = (a[4] == b[4]).all(-1).float().mean().item()
    assert same_rows >= 0.98, same_rows                              # a logit 1 ulp apart can flip a near-tie
    assert (a[0] - b[0]).abs().max() <= 0.05 * b[0].abs().max() or same_rows < 1.0
    if same_rows == 1.0:
        assert torch.allclose(a[0], b[0], rtol=2e-2, atol=2e-3)
        assert torch.allclose(a[1], b[1], rtol=1e-3, atol=1e-5)
        assert torch.allclose(a[2], b[2], rtol=5e-2, atol=5e-3 * b[2].abs().max().item())
        assert torch.allclose(a[3], b[3], rtol=5e-2, atol=5e-3 * b[3].abs().max().item())


# ------------------------------------------------------------------------------------------------ competition pass without its outputs
@pytest.mark.gpu
@pytest.mark.parametrize("T,F,Dout,bias", [(300, 512, 256, True), (2100, 1024, 520, False), (64, 256, 264, True)])
@pytest.mark.parametrize("rounded", [True, False])
def test_affinity_epilogues_match_gemm_then_softplus_mean(T, F, Dout, bias, rounded):
    """SOFTPLUS_ROWSUM / SOFTPLUS_GRAD epilogues of csmoe_dense_gemm against the stored form: y = dense_gemm (+bias), then the
    softplus_mean kernels.  Same rounded y, same softplus values; the row sums are added in another order (column tiles), so the
    affinities agree to fp32 rounding (1 ulp of the affinity dtype) and the gradient to 1 bf16 ulp."""
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(T + F)
    h = torch.randn(T, F, device=dev, generator=g).to(torch.bfloat16)
    w2 = (torch.randn(Dout, F, device=dev, generator=g) * F ** -0.5 * 2).to(torch.bfloat16)
    b2 = (torch.randn(Dout, device=dev, generator=g) * 0.1).to(torch.bfloat16) if bias else None
    y = ops.dense_gemm(h, w2, L.B_NK, bias=b2, epilogue=L.EPI_BIAS if bias else L.EPI_PLAIN)
    adt = torch.bfloat16 if rounded else torch.float32
    ref = ops.softplus_mean(y, adt)
    aff = torch.empty(T, 3, dtype=adt, device=dev)
    ops.dense_gemm_affinity(h, w2, L.B_NK, b2, aff[:, 1], rounded=rounded)
    got = aff[:, 1]
    tol = 2.0 ** -7 if rounded else 2e-6
    assert ((got.float() - ref.float()).abs() <= tol * ref.float().abs()).all()
    if rounded:
        assert (got != ref).float().mean().item() <= 0.01            # a bf16 rounding boundary now and then
    daff = torch.randn(T, device=dev, generator=g).to(adt)
    ref_dy = ops.softplus_mean_bwd(y, daff)
    dy = ops.dense_gemm_affinity_grad(h, w2, L.B_NK, b2, daff.float().contiguous(), rounded)
    assert torch.equal(dy, ref_dy)


@pytest.mark.gpu
def test_competesmoe_lean_competition_equals_stored_competition(monkeypatch):
    """LLaVA `competesmoe` on a competition step with CSMOE_COMPETITION_LEAN=1 (affinities from the GEMM epilogue, no dense outputs
    kept, backward by recomputation, selected outputs from the sparse step) against the stored form: same routing up to affinity
    ties, same losses / outputs / gradients to bf16 rounding."""
    import types
    import torch.nn as nn
    from competesmoe_amd.moe.competesmoe import CompeteSMoE
    dev = "cuda"
    torch.manual_seed(11)
    D, F, E, K = 256, 512, 4, 2
    args = types.SimpleNamespace(rate_flip=1.0, warm_up=0.0, max_compete_in_iter=1, balance_loss_coef=0.01, router_z_loss_coef=0.001,
                                 router_loss_coef=0.1, diversity_loss_coef=0.1, bal_comp_loss_coef=0.05, router_theta=0.5)
    layer = CompeteSMoE(D, D, E, K, nn.Sequential(nn.Linear(D, F), nn.GELU(), nn.Linear(F, D)), args).to(dev).to(torch.bfloat16)
    layer.set_total_steps(4, 0, {})
    layer.prob_flips.fill_(True)             # every step competes
    layer._flips_host = None
    layer.set_current_steps(1)
    x = (torch.randn(2, 160, D, device=dev) * 0.7).to(torch.bfloat16)
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv("CSMOE_COMPETITION_LEAN", flag)
        xi = x.clone().requires_grad_(True)
        layer.zero_grad(set_to_none=True)
        assert layer._competing(xi)
        out, aux, _, info = layer(xi)
        (out.float().pow(2).mean() + aux.float()).backward()
        grads = torch.cat([p.grad.float().flatten() for p in layer.parameters() if p.grad is not None])
        res.append((out.detach().float(), aux.detach().float(), xi.grad.float(), grads, {k: float(v) for k, v in info.items()}))
    a, b = res
    assert set(a[4]) == set(b[4]) and "diversity_loss" in a[4]
    same = ((a[0] - b[0]).abs().amax(-1) <= 0.02 * a[0].abs().amax()).float().mean().item()
    assert same >= 0.97, same                                       # rows routed differently only on affinity ties
    for k in a[4]:
        assert abs(a[4][k] - b[4][k]) <= 2e-2 * max(abs(a[4][k]), 1e-3), (k, a[4][k], b[4][k])
    assert (a[2] - b[2]).abs().mean() <= 0.05 * a[2].abs().mean()
    assert (a[3] - b[3]).abs().mean() <= 0.05 * a[3].abs().mean()


# ------------------------------------------------------------------------------------------------ weighted-cvmm backward order
@pytest.mark.gpu
@pytest.mark.parametrize("T,D,F,E,K", [(96, 64, 128, 4, 2), (3000, 256, 520, 6, 2), (2500, 128, 136, 3, 1)])
def test_actgrad_rowscale_epilogue_and_dot_table(T, D, F, E, K):
    """EPI_ACTGRAD_ROWSCALE: dh = round(round(w * round(g @ W2^T)) * relu'(h)) and the dot table sum_f round(g @ W2^T) * h against the
    same arithmetic in torch (both bf16 kernels: 128-tile at the small shape, 256-tile at the large ones)."""
    dev = "cuda"
    bf = torch.bfloat16
    gen = torch.Generator(device=dev).manual_seed(T + F)
    idx = rand_idx(T, K, E, seed=5).to(dev)
    bins = ops.bin_tokens(idx, E)
    n = bins.n
    g = torch.randn(n, D, device=dev, generator=gen).to(bf)                      # unscaled upstream rows, binned order
    W2 = (torch.randn(E, F, D, device=dev, generator=gen) * D ** -0.5).to(bf)    # K-major [F, D] per expert = cvmm `values`
    h = torch.relu(torch.randn(n, F, device=dev, generator=gen)).to(bf)
    wrow = torch.rand(n, device=dev, generator=gen).to(bf).float()
    cols = ops.rowdot_cols(n, F, D, D, D, F, bf)
    assert cols == ((F + 127) // 128 if n >= 2048 and F >= 256 and D >= 128 else F // 8)
    dot = torch.full((n, cols), float("nan"), device=dev)
    ptrs = ops.ptr_table(W2, E, F * D * 2)
    dh = ops.grouped_gemm(g, ptrs, L.B_NK, D, F, bins.offsets, E, epilogue=L.EPI_ACTGRAD_ROWSCALE, act=L.ACT_RELU, aux=h,
                          row_scale=wrow, row_dot=dot)
    off = bins.offsets.tolist()
    G = torch.empty(n, F, device=dev, dtype=bf)
    for e in range(E):
        G[off[e]:off[e + 1]] = (g[off[e]:off[e + 1]].float() @ W2[e].float().t()).to(bf)
    ref = ((wrow[:, None] * G.float()).to(bf).float() * (h > 0).float()).to(bf)
    # the torch product sums in another order than the MFMA chain: G may sit on the other side of a bf16 rounding boundary now and then
    d = (dh.float() - ref.float()).abs()
    assert (d > 2.0 ** -6 * ref.float().abs() + 1e-30).float().mean().item() < 1e-4
    assert (dh != ref).float().mean().item() < 5e-3
    assert float((dh.float() - ref.float()).norm() / ref.float().norm()) < 1e-3
    assert not torch.isnan(dot).any()
    dref = (G.float() * h.float()).sum(-1)
    got = ops.finish_row_dot(dot)
    assert torch.allclose(got, dref, rtol=2e-2, atol=2e-2 * dref.abs().mean().item())
