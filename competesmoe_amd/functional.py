"""torch.autograd.Function wrappers: every forward/backward below is a sequence of C-ABI HIP kernel launches
(competesmoe_amd.ops).  No torch math on the hot path; torch only owns memory, streams and the autograd graph.

Reference parity notes are on each class (file:line of the reference code whose forward AND autograd it replaces).
"""
from __future__ import annotations

import os
import weakref
from dataclasses import dataclass
from typing import List, Optional

import torch

from . import _lib as L
from . import ops


def _flip(layout: int) -> int:
    return L.B_KN if layout == L.B_NK else L.B_NK


# ======================================================================================================== gate
_CHUNK_OFFSETS = {}


def _chunked_dense_wgrad(a: torch.Tensor, b: torch.Tensor, out_dtype, P: Optional[int] = None) -> torch.Tensor:
    """a^T b for a long, skinny reduction ([T,E]^T [T,D], T >> E): a single output tile row would occupy E*D/(128*128) CUs
    only, so the T rows are cut into chunks that the grouped weight-gradient kernel treats as pseudo-experts (fp32
    partials, deterministic), followed by one small sum."""
    T, Na = a.shape
    Nb = b.shape[1]
    if P is None:
        P = max(1, min(32, T // 1024))
    if P == 1:
        return ops.dense_wgrad(a, b, out_dtype=out_dtype)
    key = (T, P, a.device)
    off = _CHUNK_OFFSETS.get(key)
    if off is None:
        step = ((T + P - 1) // P + 63) // 64 * 64          # whole K-tiles per chunk
        off = _CHUNK_OFFSETS[key] = torch.clamp(torch.arange(P + 1, dtype=torch.int64) * step, max=T).int().to(a.device)
    part = torch.empty(P, Na, Nb, dtype=torch.float32, device=a.device)
    ptrs = ops.ptr_table(part, P, Na * Nb * 4)
    ops.grouped_wgrad(a, b, off, P, part, ptrs, tag="gate_wgrad" if Na <= 64 else "grouped_wgrad_splitk")
    # sum over the P partials and the cast in one launch of the column-sum kernel (torch: a reduce kernel and a cast)
    return ops.dense_colsum(part.view(P, Na * Nb), out_dtype=out_dtype).view(Na, Nb)


def _dense_wgrad(a: torch.Tensor, b: torch.Tensor, out_dtype) -> torch.Tensor:
    """Weight gradient of ONE dense expert, a^T b with [M, Na]^T [M, Nb].  ceil(Na/256) * ceil(Nb/256) output tiles: 85 at the
    reference's SigLIP shape (4304 x 1152) on a 256-CU chip -- then the reduction is split over row chunks (pseudo-experts of
    the grouped kernel, fp32 partials, one small sum) until about two tiles per CU are in flight."""
    M, Na = a.shape
    Nb = b.shape[1]
    tiles = ((Na + 255) // 256) * ((Nb + 255) // 256)
    P = min(16, (512 + tiles - 1) // tiles, M // 1024)
    if tiles >= 384 or P < 2:
        return ops.dense_wgrad(a, b, out_dtype=out_dtype)
    return _chunked_dense_wgrad(a, b, out_dtype, P=P)


def _chunked_dense_colsum(g: torch.Tensor, out_dtype) -> torch.Tensor:
    """Column sums of a tall dense [M, N] matrix (bias gradient of an always-on / dense-competition expert).  One launch over all
    M rows has only ceil(N / 512) workgroups (22 at N = 11008 on a 256-CU chip: 0.4 TB/s); the rows are cut into chunks that the
    grouped column-sum kernel treats as pseudo-experts (fp32 partials, deterministic), followed by one small sum."""
    M, N = g.shape
    P = max(1, min(64, M // 512))
    if P == 1:
        return ops.dense_colsum(g, out_dtype=out_dtype)
    key = (M, P, g.device)
    off = _CHUNK_OFFSETS.get(key)
    if off is None:
        step = (M + P - 1) // P
        off = _CHUNK_OFFSETS[key] = torch.clamp(torch.arange(P + 1, dtype=torch.int64) * step, max=M).int().to(g.device)
    part = torch.empty(P, N, dtype=torch.float32, device=g.device)
    ptrs = ops.ptr_table(part, P, N * 4)
    ops.grouped_colsum(g, off, P, part, ptrs)
    return ops.dense_colsum(part, out_dtype=out_dtype)


def _grouped_colsum(g: torch.Tensor, offsets: torch.Tensor, E: int, pd) -> torch.Tensor:
    """Per-expert column sums [E, N] of the binned rows (bias gradients).  One launch has E * ceil(N / 512) workgroups; with few
    experts (the reference's LLaVA configs use 4) that is a few dozen on a 256-CU chip (0.5 TB/s), so each expert's rows are cut
    into chunks handled as pseudo-experts (fp32 partials, deterministic) and summed."""
    N = g.shape[1]
    dev = g.device
    wgs = E * ((N + 511) // 512)
    if wgs >= 256 or g.shape[0] < 4096:
        out = torch.empty(E, N, dtype=pd, device=dev)
        es = out.element_size()
        ops.grouped_colsum(g, offsets, E, out, ops.ptr_table(out, E, N * es))
        return out
    P = min(64, max(2, 512 // wgs))
    chunk_off = ops.chunk_offsets(offsets, E, P)              # chunk (e, j) ends where (e, j+1) starts
    part = torch.empty(E * P, N, dtype=torch.float32, device=dev)
    ops.grouped_colsum(g, chunk_off, E * P, part, ops.ptr_table(part, E * P, N * 4))
    return ops.sum_partials(part, E, P, pd)


def _gate_wgrad(dlogits: torch.Tensor, x2: torch.Tensor, w_dtype, small: bool) -> torch.Tensor:
    """d w_gate = dlogits^T @ x in the dtype the product ran in, THEN cast to the parameter's dtype: with an fp32 master under bf16
    autocast `F.linear` multiplies a bf16 copy of w_gate, autograd's matmul backward returns a bf16 gradient for that copy and the
    cast's backward widens it (moe_pretrain_model/layers/moe/moe.py:121 under simple_task.py:295) -- the gate gradient the
    reference's optimizer sees is bf16-rounded (observed: 3e-3 from an unrounded fp32 product, 1e-4 once rounded the same way)."""
    op = x2.dtype
    g = ops.gate_bwd_dw(dlogits, x2, op) if small else _chunked_dense_wgrad(dlogits, x2, op)
    return g if w_dtype == op else g.to(w_dtype)


class GateLogits(torch.autograd.Function):
    """logits = x @ w_gate^T rounded to x.dtype -- `self.gate(x)` (moe_model/model/moe/smoe.py:42) /
    `F.linear(x, self.w_gate)` (moe_pretrain_model/layers/moe/moe.py:121)."""

    @staticmethod
    def forward(ctx, x2: torch.Tensor, w_gate: torch.Tensor):
        x2 = x2.contiguous()
        wg = w_gate.contiguous()
        if wg.dtype != x2.dtype:
            wg = wg.to(x2.dtype)
        ctx.save_for_backward(x2, wg)
        ctx.w_dtype = w_gate.dtype
        return ops.gate_logits(x2, wg)

    @staticmethod
    def backward(ctx, dlogits):
        x2, wg = ctx.saved_tensors
        dlogits = dlogits.contiguous()
        dx = dw = None
        small = ops.gate_bwd_small_ok(x2.shape[1], wg.shape[0], x2.dtype)      # few experts: row passes, not MFMA tiles
        if ctx.needs_input_grad[0]:
            dx = ops.gate_bwd_dx(dlogits, wg) if small else ops.dense_gemm(dlogits, wg, L.B_KN)         # [T,E] @ [E,D]
        if ctx.needs_input_grad[1]:
            dw = _gate_wgrad(dlogits, x2, ctx.w_dtype, small)
        return dx, dw


class LayerNormGate(torch.autograd.Function):
    """The read side of the block around the layer: xn = LayerNorm(x), logits = xn @ w_gate^T in one
    launch, plus x itself handed on as the residual so that ITS gradient comes back here and is added inside the LayerNorm
    backward kernel instead of by a separate accumulation pass.
    `layer_norm2` + `self.gate(x)` + `residual + results` of SiglipEncoderMoELayer.forward (siglip_smoe.py:152-155, smoe.py:42)."""

    @staticmethod
    def forward(ctx, x2, gamma, beta, eps: float, w_gate, act_dtype=None):
        """`act_dtype` = torch.bfloat16 with fp32 x: the pretrain stack's fp32 residual stream under bf16 autocast
        (relative_moe_transformer.py:153-161): LayerNorm in fp32, xn / logits leave as bf16 (csmoe_layernorm_gate_mixed)."""
        x2 = x2.contiguous()
        ctx.mixed = act_dtype == torch.bfloat16 and x2.dtype == torch.float32
        od = torch.bfloat16 if ctx.mixed else x2.dtype
        wg = w_gate.contiguous()
        if wg.dtype != od:
            wg = wg.to(od)
        g = None if gamma is None else gamma.to(x2.dtype).contiguous()
        b = None if beta is None else beta.to(x2.dtype).contiguous()
        if ctx.mixed:
            xn, mean, rstd, logits = ops.layernorm_gate_mixed(x2, g, b, eps, wg)
        else:
            xn, mean, rstd, logits = ops.layernorm_gate(x2, g, b, eps, wg)
        ctx.save_for_backward(x2, g, mean, rstd, xn, wg)
        ctx.dtypes = (None if gamma is None else gamma.dtype, None if beta is None else beta.dtype, w_gate.dtype)
        return xn, logits, x2.view_as(x2)

    @staticmethod
    def backward(ctx, dxn, dlogits, dres):
        x2, g, mean, rstd, xn, wg = ctx.saved_tensors
        gd, bd, wd = ctx.dtypes
        dxn_gate = dwg = None
        if dlogits is not None:
            dlogits = dlogits.contiguous()
            small = ops.gate_bwd_small_ok(xn.shape[1], wg.shape[0], xn.dtype)
            dxn_gate = ops.gate_bwd_dx(dlogits, wg) if small else ops.dense_gemm(dlogits, wg, L.B_KN)     # [T,E] @ [E,D]
            if ctx.needs_input_grad[4]:
                dwg = _gate_wgrad(dlogits, xn, wd, small)
        a, b2 = dxn, dxn_gate
        if a is None:
            a, b2 = dxn_gate, None
        if a is None:                                                      # only the residual carries a gradient
            return dres, None, None, None, dwg, None
        want_affine = (gd is not None and ctx.needs_input_grad[1]) or (bd is not None and ctx.needs_input_grad[2])
        bwd = ops.layernorm_bwd_mixed if ctx.mixed else ops.layernorm_bwd
        dx, dgamma, dbeta = bwd(a.contiguous(), x2, g, mean, rstd, add=None if dres is None else dres.contiguous(),
                                want_affine_grads=want_affine, dxn2=None if b2 is None else b2.contiguous())
        dg = dgamma.to(gd) if (want_affine and gd is not None) else None
        db = dbeta.to(bd) if (want_affine and bd is not None) else None
        return dx, dg, db, None, dwg, None


# ======================================================================================================== router
class RouterSelect(torch.autograd.Function):
    """softmax / top-k / renormalise in one wave-per-token kernel.  Returns (softmax fp32 [T,E], idx int32 [T,K],
    w fp32 [T,K]).  moe_model/model/moe/moe.py:113-132, smoe.py:44, competesmoe.py:246-255 and the pretrain
    variants (deepseekv2.py:140-142, deepseekv3.py:147-151)."""

    @staticmethod
    def forward(ctx, scores: torch.Tensor, K: int, mode: int, round_sum_bf16: bool, param: float = 1.0):
        """`param`: the scale of SEL_TOPK_SIGMOID (`args.scale_weight`, pretrain competesmoe.py:476-483)."""
        scores = scores.contiguous()
        sm, idx, w = ops.router_select(scores, K, mode, round_sum_bf16, want_softmax=True, param=param)
        ctx.save_for_backward(scores, sm, idx, w)
        ctx.cfg = (K, mode, round_sum_bf16, param)
        ctx.mark_non_differentiable(idx)
        return sm, idx, w

    @staticmethod
    def backward(ctx, dsm, _didx, dw):
        scores, sm, idx, w = ctx.saved_tensors
        K, mode, rb, param = ctx.cfg
        dsm = None if dsm is None else dsm.contiguous().float()
        dw = None if dw is None else dw.contiguous().float()
        ds = ops.router_select_bwd(scores, K, mode, rb, sm, idx, w, dw, dsm, param=param)
        return ds, None, None, None, None


class OperandFork(torch.autograd.Function):
    """fp32 x [T, D] -> `n` bf16 operands out of ONE cast (the tensors share their storage), for the ops of a layer that each read x
    under bf16 autocast: the reference's gate (F.linear), cvmm and shared expert cast x themselves (moe.py:121, cvmm.py:29-32,445,
    deepseekv2.py:154-165), every cast's backward widens its bf16 gradient and the engine adds the fp32 streams.  Same values here
    from one cast forward and one pass backward (csmoe_widen_sum: the streams in fp32, later-created consumer first, as the engine
    meets them) instead of a cast per consumer each way and an fp32 add per pair."""

    @staticmethod
    def forward(ctx, x, n: int):
        ctx.set_materialize_grads(False)
        xb = x.to(torch.bfloat16)
        twins = [torch.empty(0, dtype=xb.dtype, device=xb.device).set_(xb.untyped_storage(), xb.storage_offset(), xb.shape, xb.stride())
                 for _ in range(n - 1)]
        return (xb, *twins)

    @staticmethod
    def backward(ctx, *gs):
        live = [g for g in reversed(gs) if g is not None]
        if not live:
            return None, None
        return ops.widen_sum(live), None


class GateSelect(torch.autograd.Function):
    """GateLogits + RouterSelect in ONE launch that reads x once (csmoe_gate_select): returns (logits, softmax fp32, idx int32,
    w fp32) with the bits the two separate functions give.  `self.gate(x)` + `topk_expert` (moe_model/model/moe/smoe.py:42-44) /
    `F.linear(x, w_gate)` + top-k (moe_pretrain_model/layers/moe/moe.py:121, smoe.py:30-40).  Backward: the selection's gradient plus
    whatever the losses put on the logits, then the two gate products of GateLogits.backward."""

    @staticmethod
    def forward(ctx, x2: torch.Tensor, w_gate: torch.Tensor, K: int, mode: int, round_sum_bf16: bool, param: float = 1.0):
        x2 = x2.contiguous()
        wg = w_gate.contiguous()
        if wg.dtype != x2.dtype:
            wg = wg.to(x2.dtype)
        logits, sm, idx, w = ops.gate_select(x2, wg, K, mode, round_sum_bf16, want_softmax=True, param=param)
        ctx.save_for_backward(x2, wg, logits, sm, idx, w)
        ctx.cfg = (K, mode, round_sum_bf16, param)
        ctx.w_dtype = w_gate.dtype
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)
        return logits, sm, idx, w

    @staticmethod
    def backward(ctx, dlogits, dsm, _didx, dw):
        x2, wg, logits, sm, idx, w = ctx.saved_tensors
        K, mode, rb, param = ctx.cfg
        ds = None
        if dsm is not None or dw is not None:
            dsm = None if dsm is None else dsm.contiguous().float()
            dw = None if dw is None else dw.contiguous().float()
            ds = ops.router_select_bwd(logits, K, mode, rb, sm, idx, w, dw, dsm, param=param)
        if dlogits is not None:
            ds = dlogits.contiguous() if ds is None else ds + dlogits
        if ds is None:
            return None, None, None, None, None, None
        dx = dwg = None
        small = ops.gate_bwd_small_ok(x2.shape[1], wg.shape[0], x2.dtype)      # few experts: row passes, not MFMA tiles
        if ctx.needs_input_grad[0]:
            dx = ops.gate_bwd_dx(ds, wg) if small else ops.dense_gemm(ds, wg, L.B_KN)                     # [T,E] @ [E,D]
        if ctx.needs_input_grad[1]:
            dwg = _gate_wgrad(ds, x2, ctx.w_dtype, small)
        return dx, dwg, None, None, None, None


class RouterAux(torch.autograd.Function):
    """(balance loss fp32, z-loss in logits.dtype) of the LLaVA stack in two launches, one for the backward --
    `balanceloss` (moe_model/model/moe/moe.py:90-110: mean_n softmax x mean_n one_hot(top-1), mean over (b, e), x E^2) and
    `zloss` (moe.py:71-88: mean(logsumexp(logits)^2) evaluated in logits.dtype).  `sm` must be the fp32 softmax of `logits` and
    idx[..., 0] its arg-max (what `topk_expert` returns); logits = None gives the balance loss alone (z = 0)."""

    @staticmethod
    def forward(ctx, logits, sm, idx):
        sm = sm.contiguous()
        idx = idx.contiguous()
        if logits is not None:
            logits = logits.contiguous()
        out2, dens, lse = ops.router_aux(logits, sm, idx)
        ctx.save_for_backward(sm, dens, lse)
        ctx.ldtype = None if logits is None else logits.dtype
        z = out2[1] if logits is None or logits.dtype == torch.float32 else out2[1].to(logits.dtype)
        return out2[0], z

    @staticmethod
    def backward(ctx, g_bal, g_z):
        sm, dens, lse = ctx.saved_tensors
        gb = g_bal.float().reshape(1) if (g_bal is not None and ctx.needs_input_grad[1]) else None
        gz = g_z.float().reshape(1) if (g_z is not None and ctx.ldtype is not None and ctx.needs_input_grad[0]) else None
        dsm, dlogits = ops.router_aux_bwd(sm, dens, lse, gb, gz, ctx.ldtype or torch.float32)
        return dlogits, dsm, None


# ======================================================================================================== expert FFN
@dataclass
class ExpertTable:
    """Device pointer tables + static description of E two-matrix experts."""
    E: int
    D: int            # input dim
    F: int            # hidden dim
    Dout: int         # output dim
    layout: int       # B_NK: w1 [F,D], w2 [Dout,F] (nn.Linear);  B_KN: w1 [D,F], w2 [F,Dout] (cvmm keys/values)
    act: int
    w1_ptrs: torch.Tensor
    w2_ptrs: torch.Tensor
    b1_ptrs: Optional[torch.Tensor] = None
    b2_ptrs: Optional[torch.Tensor] = None
    param_dtype: torch.dtype = torch.float32   # dtype of the gradients handed back
    epi1: int = L.EPI_BIAS_ACT                 # epilogue of the first GEMM (EPI_ROUND_BIAS32_ACT: fp32 b1 table, pretrain autocast)
    scale_after_gemm: bool = False             # backward of the weighted second GEMM multiplies by the weight AFTER the product is
                                               # rounded (the pretrain stack's cvmm with reduction weights, cvmm.py:527-543)


# Operand-dtype copies of fp32 master weights kept across calls while the parameter is unchanged (same storage, same autograd
# version counter).  In-place updates through the parameter itself (optimizer.step(), p.mul_()) bump that counter; writes through
# `p.data` / `p.detach()` (DeepSpeed / apex bf16 optimizers, EMA, weight clipping) do NOT -- `.data` carries its own counter -- so
# with such a writer call `invalidate_weight_cache()` after every update (e.g. from an optimizer-step post hook).  The reference's pretrain task runs several micro-batches per
# optimizer step (simple_task.py:286-320) and evaluates between steps: all but the first forward after an update then skip the
# HBM-bound cast (34.5 GB at the headline shape).  Opt-in -- CSMOE_WEIGHT_CACHE=1 or `weight_cache(True)` -- because it holds one
# bf16 copy of the expert weights per layer for as long as the layer lives, and because a writer that goes around torch (a raw
# pointer write from another library) would not be seen.
_WEIGHT_CACHE_ON = os.environ.get("CSMOE_WEIGHT_CACHE", "0") == "1"
_WEIGHT_CACHE = {}


def weight_cache(enabled: bool) -> None:
    global _WEIGHT_CACHE_ON
    _WEIGHT_CACHE_ON = bool(enabled)
    if not enabled:
        _WEIGHT_CACHE.clear()


def invalidate_weight_cache() -> None:
    """Drop every cached operand copy (for writers that bypass the autograd version counter, see above)."""
    _WEIGHT_CACHE.clear()


def _cached_copy(t: torch.Tensor, op):
    """(operand copy of parameter t, hit) from the cache: the SAME tensor object (a freed parameter's address can be handed to a new
    one), unchanged since the copy was made."""
    ent = _WEIGHT_CACHE.get((t.data_ptr(), op))
    if ent is not None and ent[0]() is t and ent[1] == t._version and ent[2].shape == t.shape:
        return ent[2], True
    return None, False


def _remember(t: torch.Tensor, op, copy: torch.Tensor) -> None:
    for k in [k for k, e in _WEIGHT_CACHE.items() if e[0]() is None]:       # parameters that no longer exist
        del _WEIGHT_CACHE[k]
    _WEIGHT_CACHE[(t.data_ptr(), op)] = (weakref.ref(t), t._version, copy)


_CAST_OVERLAP = os.environ.get("CSMOE_CAST_OVERLAP", "1") != "0"
_F32W_ON = os.environ.get("CSMOE_F32W", "1") != "0"          # A/B: 0 = cast the fp32 masters to bf16 first (round 1's path)
_SIDE_STREAMS = {}


def _side_stream(dev) -> "torch.cuda.Stream":
    key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


def _ffn_forward(x2, w, idx, tab: ExpertTable, combine_mode: int, obias, residual=None, before_gemm2=None, masters=None):
    """`before_gemm2`: a stream the launch stream must wait for before the second grouped GEMM (operand cast running beside GEMM 1).
    `masters` = (keys fp32, k_copy bf16, values fp32, v_copy bf16): the forward GEMMs read the fp32 masters and convert inside
    their tile fill (csmoe_grouped_gemm_f32w), writing the bf16 copies the backward GEMMs read through `tab`'s pointer tables."""
    T = x2.shape[0]
    bins = ops.bin_tokens(idx, tab.E)
    xs = ops.dispatch_tokens(x2, bins)
    if masters is not None:
        keys, k_copy, values, v_copy = masters
        hpre, hact = ops.grouped_gemm_f32w(xs, keys, bins.offsets, copy=k_copy, bias_ptrs=tab.b1_ptrs, epilogue=tab.epi1, act=tab.act,
                                           want_c2=True, want_c=tab.act != L.ACT_RELU)
        y = ops.grouped_gemm_f32w(hact, values, bins.offsets, copy=v_copy)
        out = ops.combine(y, bins, idx, w, combine_mode, T, obias=obias, residual=residual)
        return out, (bins, xs, hpre, hact, y, combine_mode, idx)
    ld1 = tab.D if tab.layout == L.B_NK else tab.F
    ld2 = tab.F if tab.layout == L.B_NK else tab.Dout
    # ReLU: act'(pre) = (act(pre) > 0), so the pre-activation is neither written (1.44 GB at the headline shape) nor re-read by the
    # dH epilogue; every other activation keeps it
    hpre, hact = ops.grouped_gemm(xs, tab.w1_ptrs, tab.layout, ld1, tab.F, bins.offsets, tab.E, bias_ptrs=tab.b1_ptrs,
                                  epilogue=tab.epi1, act=tab.act, want_c2=True, want_c=tab.act != L.ACT_RELU)
    if before_gemm2 is not None:
        torch.cuda.current_stream().wait_stream(before_gemm2)
    y = ops.grouped_gemm(hact, tab.w2_ptrs, tab.layout, ld2, tab.Dout, bins.offsets, tab.E, bias_ptrs=tab.b2_ptrs,
                         epilogue=L.EPI_BIAS if tab.b2_ptrs is not None else L.EPI_PLAIN)
    out = ops.combine(y, bins, idx, w, combine_mode, T, obias=obias, residual=residual)
    return out, (bins, xs, hpre, hact, y, combine_mode, idx)


_WGRAD_SPLIT = os.environ.get("CSMOE_WGRAD_SPLIT", "1") != "0"       # A/B switch
_SCALE_AFTER_ON = os.environ.get("CSMOE_SCALE_AFTER_GEMM", "1") != "0"   # A/B switch: 0 = dH from dy = round(w * dout) in every stack


def _grouped_wgrad(a: torch.Tensor, b: torch.Tensor, bins, E: int, pd) -> torch.Tensor:
    """out[e] = a_e^T @ b_e -> [E, Na, Nb] in dtype pd.  E * ceil(Na/256) * ceil(Nb/256) output tiles: 340 at the reference's SigLIP
    layer (4 experts, 4304 x 1152) on a 256-CU chip, i.e. a second round that is one third full.  Then every expert's rows are cut
    into P chunks handled as pseudo-experts of the same persistent kernel (fp32 partials, fixed chunking: deterministic) and the P
    partials are summed -- split-K for the grouped weight gradient, as `_dense_wgrad` does for one dense expert."""
    Na, Nb = a.shape[1], b.shape[1]
    dev = a.device
    tiles = E * ((Na + 255) // 256) * ((Nb + 255) // 256)
    P = min(4, (640 + tiles - 1) // tiles, max(1, a.shape[0] // (E * 1024)))      # about 2.5 tiles per CU; targets of 1000 / 1300 measured slower
    if not _WGRAD_SPLIT or tiles >= 512 or P < 2:
        out = torch.empty(E, Na, Nb, dtype=pd, device=dev)
        ops.grouped_wgrad(a, b, bins.offsets, E, out, ops.ptr_table(out, E, Na * Nb * out.element_size()), xcd_order=bins.xcd_order)
        return out
    chunk_off = bins.chunk_offsets(P)
    part = torch.empty(E * P, Na, Nb, dtype=torch.float32, device=dev)
    ops.grouped_wgrad(a, b, chunk_off, E * P, part, ops.ptr_table(part, E * P, Na * Nb * 4), tag="grouped_wgrad_tn")
    return ops.sum_partials(part.view(E * P, Na * Nb), E, P, pd).view(E, Na, Nb)


class DxHandoff:
    """Carries the always-on expert's dx (DenseFFN.backward) to the routed step's gather-sum (MoEFFNModules.backward) of the SAME layer
    call, where it enters as the chain's first addend (csmoe_dispatch_rows_bwd `pre`): the reference's autograd adds the streams of x
    one expert at a time in x.dtype -- shared expert (created last, so its node runs first), routed experts E-1 .. 0, gate -- and the
    shared-expert fixtures' dx is exactly that chain (tests/test_llava_modules_gpu.py).  Summing the routed experts first and adding
    the shared stream afterwards, as two autograd nodes do, is 2.8e-3 away from it.  The engine runs the later-created node first; if
    it ever did not, both nodes fall back to plain accumulation (`routed_done`)."""
    __slots__ = ("dx", "routed_done", "has_consumer")

    def __init__(self):
        self.dx, self.routed_done, self.has_consumer = None, False, False


_PENDING_HANDOFF = None        # set by a shared-expert layer around its two calls, taken by the forward of the node it is meant for


def set_handoff(h):
    global _PENDING_HANDOFF
    _PENDING_HANDOFF = h


def _take_handoff():
    global _PENDING_HANDOFF
    h, _PENDING_HANDOFF = _PENDING_HANDOFF, None
    return h


def _ffn_backward(dout, w, tab: ExpertTable, saved, need_dx: bool, need_dw: bool, need_params: bool, dy_extra=None, dx_add=None,
                  handoff=None):
    """Returns dx2, dw, (gW1 [E,..], gb1 [E,F]|None, gW2 [E,..], gb2 [E,Dout]|None).
    `dx_add`: a callable dw -> [T, D] tensor (or None) added to dx INSIDE the gather-sum of the binned rows (csmoe_dispatch_rows_bwd's
    `add` input): the gate-path gradient of x, which depends on this function's dw (SparseMoEModules).
    `handoff`: DxHandoff of a shared-expert layer (its always-on expert's dx becomes the chain's first addend)."""
    if saved is None:
        raise RuntimeError("competesmoe_amd: the MoE layer's saved activations were freed by the first backward pass "
                           "(they are released early to bound memory); a second backward through the same graph "
                           "(retain_graph=True) is not supported -- run the forward again")
    bins, xs, hpre, hact, y, *rest = saved
    # competition steps hand in bf16 routing weights (COMBINE_SEQ_RW): the reference's `weights * out_exp` is then a bf16 product and
    # its autograd forms d w = (grad * out).sum(-1) with every product rounded to bf16 and the sum rounded once (moe.py:204)
    round_dw = bool(rest) and rest[0] == L.COMBINE_SEQ_RW
    T = dout.shape[0]
    dev = dout.device
    E = tab.E
    ld2 = tab.F if tab.layout == L.B_NK else tab.Dout
    scale_after = tab.scale_after_gemm and _SCALE_AFTER_ON
    # gradient of the reduction weights as the reference's weighted cvmm forms it, <unscaled rounded product, activated input>
    # (cvmm.py:544), out of the dH launch's epilogue -- when its aux operand IS the activated input (ReLU experts keep nothing else)
    dot_cols = 0
    if scale_after and need_dw and hpre is None:
        dot_cols = ops.rowdot_cols(bins.n, tab.F, tab.Dout, tab.Dout, ld2, tab.F, hact.dtype)
    dy, dw = ops.combine_bwd(dout.contiguous(), y if (need_dw and not dot_cols) else None, bins, w, want_dw=need_dw and not dot_cols,
                             act_dtype=hact.dtype, round_products=round_dw)
    if dy_extra is not None:                # gradient of the per-slot outputs (MoEFFNModulesSlots), already in binned order
        dy = dy + dy_extra
    ld1 = tab.D if tab.layout == L.B_NK else tab.F
    # dH = dY @ W2 (+ activation backward in the epilogue)
    if scale_after:
        # the reference's order for the weighted cvmm: the UNSCALED upstream rows go through the product, the bf16 result is
        # multiplied by the (bf16) reduction weight and rounded again, then the activation gradient (cvmm.py:527-543 + autograd
        # of `relu`): a gather of dout rows and a per-row scale in the epilogue instead of dy = round(w * dout) as the operand
        dyu = ops.dispatch_tokens(dout.contiguous().to(hact.dtype), bins)
        w_rows = w.reshape(-1)[bins.perm.long()].float().contiguous()
        dot = torch.empty(bins.n, dot_cols, dtype=torch.float32, device=dev) if dot_cols else None
        dh = ops.grouped_gemm(dyu, tab.w2_ptrs, _flip(tab.layout), ld2, tab.F, bins.offsets, E, epilogue=L.EPI_ACTGRAD_ROWSCALE,
                              act=tab.act, aux=hpre if hpre is not None else hact, row_scale=w_rows, row_dot=dot)
        del dyu
        if dot is not None:
            dw = ops.finish_row_dot(dot)[bins.slot_of.long()].view(T, bins.K)
        if dy_extra is not None:
            # the per-slot outputs (MoEFFNPackedSlots: the diversity loss's operands) are UNWEIGHTED rows of the second product, so
            # their gradient goes through W2^T and the activation gradient as it stands -- the reference backpropagates the
            # diversity loss through relu(x @ keys[e]) too (pretrain competesmoe.py:403-410) -- and carries no d w
            dh = dh + ops.grouped_gemm(dy_extra.contiguous(), tab.w2_ptrs, _flip(tab.layout), ld2, tab.F, bins.offsets, E,
                                       epilogue=L.EPI_ACTGRAD, act=tab.act, aux=hpre if hpre is not None else hact)
    else:
        dh = ops.grouped_gemm(dy, tab.w2_ptrs, _flip(tab.layout), ld2, tab.F, bins.offsets, E, epilogue=L.EPI_ACTGRAD,
                              act=tab.act, aux=hpre if hpre is not None else hact)
    grads = None
    if need_params:
        pd = tab.param_dtype
        if tab.layout == L.B_NK:
            gW2 = _grouped_wgrad(dy, hact, bins, E, pd)          # [E, Dout, F]
            gW1 = _grouped_wgrad(dh, xs, bins, E, pd)            # [E, F, D]
        else:
            gW2 = _grouped_wgrad(hact, dy, bins, E, pd)          # [E, F, Dout]
            gW1 = _grouped_wgrad(xs, dh, bins, E, pd)            # [E, D, F]
        gb1 = gb2 = None
        if tab.b2_ptrs is not None:
            gb2 = _grouped_colsum(dy, bins.offsets, E, pd)
        if tab.b1_ptrs is not None:
            gb1 = _grouped_colsum(dh, bins.offsets, E, pd)
        grads = (gW1, gb1, gW2, gb2)
    dx2 = None
    if need_dx:
        dxs = ops.grouped_gemm(dh, tab.w1_ptrs, _flip(tab.layout), ld1, tab.D, bins.offsets, E)
        pre = None
        if handoff is not None:
            pre, handoff.dx, handoff.routed_done = handoff.dx, None, True
        # the LLaVA stack's per-expert modules: sequential x.dtype accumulation in descending expert order (for K <= 2 without a
        # first addend that IS round(sum), so the fast gather-sum stays); the pretrain stack's cvmm reduces over K in fp32
        seq_idx = rest[1] if (len(rest) > 1 and rest[0] != L.COMBINE_DOT and (pre is not None or bins.K > 2)) else None
        if pre is not None and seq_idx is None:        # cannot happen for the LLaVA layers; keep the stream
            pre_lost, pre = pre, None
        else:
            pre_lost = None
        dx2 = ops.dispatch_rows_bwd(dxs, bins, T, add=None if dx_add is None else dx_add(dw),
                                    idx=None if seq_idx is None else seq_idx.contiguous(), pre=pre)
        if pre_lost is not None:
            dx2 = dx2 + pre_lost
    elif handoff is not None:
        handoff.routed_done = True
        if handoff.dx is not None:                     # the always-on expert already handed its dx over: it must not be dropped
            dx2, handoff.dx = handoff.dx, None
    return dx2, dw, grads


class MoEFFNModules(torch.autograd.Function):
    """dispatch -> grouped GEMM(+bias, act) -> grouped GEMM(+bias) -> combine for experts held as SEPARATE nn.Module
    parameters (LLaVA stack).  Replaces MoeLayer.compute_moe (moe_model/model/moe/moe.py:172-213) and its autograd.

    `params` = w1_0..w1_{E-1}, [b1_0..], w2_0.., [b2_0..]; they are only listed so autograd routes their gradients --
    the kernels read them through the pointer tables in `tab`."""

    @staticmethod
    def forward(ctx, x2, w, idx, tab: ExpertTable, combine_mode: int, *params):
        x2 = x2.contiguous()
        out, saved = _ffn_forward(x2, w, idx, tab, combine_mode, None)
        ctx.tab, ctx.saved, ctx.w = tab, saved, w
        ctx.n_params = len(params)
        ctx.handoff = _take_handoff()
        if ctx.handoff is not None:
            ctx.handoff.has_consumer = True          # without a routed node that will take it, the always-on expert keeps its dx
        return out

    @staticmethod
    def backward(ctx, dout):
        tab = ctx.tab
        need_params = any(ctx.needs_input_grad[5:])
        dx2, dw, grads = _ffn_backward(dout, ctx.w, tab, ctx.saved, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                       need_params, handoff=ctx.handoff)
        ctx.saved = None
        pg: List[Optional[torch.Tensor]] = [None] * ctx.n_params
        if grads is not None:
            gW1, gb1, gW2, gb2 = grads
            E = tab.E
            seq = [gW1] + ([gb1] if gb1 is not None else []) + [gW2] + ([gb2] if gb2 is not None else [])
            pg = [g[e] for g in seq for e in range(E)]
        return (dx2, dw, None, None, None, *pg)


class SparseMoEModules(torch.autograd.Function):
    """GateSelect + MoEFFNModules as ONE node -- router_policy / topk_expert + compute_moe of a sparse step (moe_model/model/moe/
    smoe.py:39-64, moe.py:113-213) -- so that the two gradient streams of x, through the gate and through the experts, meet inside
    the gather-sum of the binned rows (`csmoe_dispatch_rows_bwd(..., add)`: round(round(sum_k dxs) + dx_gate), the bits of autograd's
    separate accumulation pass) instead of in an elementwise add over [T, D] (0.13 ms and 0.8 GB at the headline shape).
    Returns (out, logits, softmax fp32, idx int32, w fp32): the losses are built on the last four outside, their gradients come
    back here.  Backward: expert path (which yields d w), then the selection's backward with d w and the losses' gradients, the
    gate's two products, and the gather-sum with the gate-path dx as its `add` input."""

    @staticmethod
    def forward(ctx, x2, w_gate, K: int, mode: int, round_sum_bf16: bool, tab: ExpertTable, combine_mode: int, *params):
        x2 = x2.contiguous()
        wg = w_gate.contiguous()
        if wg.dtype != x2.dtype:
            wg = wg.to(x2.dtype)
        logits, sm, idx, w = ops.gate_select(x2, wg, K, mode, round_sum_bf16, want_softmax=True)
        out, saved = _ffn_forward(x2, w, idx, tab, combine_mode, None)
        ctx.save_for_backward(x2, wg, logits, sm, idx, w)
        ctx.cfg = (K, mode, round_sum_bf16)
        ctx.w_dtype = w_gate.dtype
        ctx.tab, ctx.saved = tab, saved
        ctx.n_params = len(params)
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)
        return out, logits, sm, idx, w

    @staticmethod
    def backward(ctx, dout, dlogits, dsm, _didx, dw_ext):
        x2, wg, logits, sm, idx, w = ctx.saved_tensors
        K, mode, rb = ctx.cfg
        tab = ctx.tab
        need_x, need_wg = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_params = any(ctx.needs_input_grad[7:])
        small = ops.gate_bwd_small_ok(x2.shape[1], wg.shape[0], x2.dtype)
        box = {}

        def gate_path(dw):
            """d logits from the selection's backward (+ the losses' own gradient on the logits); the gate-path dx (or None)."""
            if dw_ext is not None:
                dw = dw_ext.contiguous().float() if dw is None else dw + dw_ext
            ds = None
            if dw is not None or dsm is not None:
                ds = ops.router_select_bwd(logits, K, mode, rb, sm, idx, w, None if dw is None else dw.contiguous().float(),
                                           None if dsm is None else dsm.contiguous().float())
            if dlogits is not None:
                ds = dlogits.contiguous() if ds is None else ds + dlogits
            box["ds"] = ds
            if ds is None or not need_x:
                return None
            return ops.gate_bwd_dx(ds, wg) if small else ops.dense_gemm(ds, wg, L.B_KN)

        if dout is None:                         # only the losses reach x: no expert path
            ctx.saved = None
            dx2 = gate_path(None)
            grads = None
        else:
            want_dw = need_x or need_wg          # d w only matters through the gate
            dx2, dw, grads = _ffn_backward(dout, w, tab, ctx.saved, need_x, want_dw, need_params, dx_add=gate_path if need_x else None)
            ctx.saved = None
            if not need_x:
                gate_path(dw)
        dwg = None
        if need_wg and box.get("ds") is not None:
            dwg = _gate_wgrad(box["ds"], x2, ctx.w_dtype, small)
        pg: List[Optional[torch.Tensor]] = [None] * ctx.n_params
        if grads is not None:
            gW1, gb1, gW2, gb2 = grads
            E = tab.E
            seq = [gW1] + ([gb1] if gb1 is not None else []) + [gW2] + ([gb2] if gb2 is not None else [])
            pg = [g[e] for g in seq for e in range(E)]
        return (dx2, dwg, None, None, None, None, None, *pg)


class _SlotMap:
    """Adapters so ops.dispatch_rows moves rows between the binned order and the flat (token, k) order of one binning."""
    def __init__(self, index, n):
        self.perm, self.K, self.n = index, 1, n


class MoEFFNModulesSlots(torch.autograd.Function):
    """MoEFFNModules that also returns the expert outputs per (token, k) slot, y_tk [T, K, Dout] -- what
    `torch.gather(expert_outputs, 2, idx)` takes out of the dense competition pass for the diversity loss
    (moe_model/model/moe/competesmoe.py:256-258), from the sparse step's own rows (same bits), so that the dense pass need not keep
    its outputs (CompetitionAffinity)."""

    @staticmethod
    def forward(ctx, x2, w, idx, tab: ExpertTable, combine_mode: int, *params):
        x2 = x2.contiguous()
        out, saved = _ffn_forward(x2, w, idx, tab, combine_mode, None)
        bins, y = saved[0], saved[4]
        y_tk = ops.dispatch_rows(y, _SlotMap(bins.slot_of, bins.n))           # y_tk[t*K+k] = y[slot_of[t*K+k]]
        ctx.tab, ctx.saved, ctx.w = tab, saved, w
        ctx.n_params = len(params)
        return out, y_tk.view(x2.shape[0], idx.shape[-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dout, dy_tk):
        tab = ctx.tab
        if ctx.saved is None:
            _ffn_backward(dout, ctx.w, tab, None, False, False, False)        # raises the second-backward error
        bins = ctx.saved[0]
        extra = None
        if dy_tk is not None:
            extra = ops.dispatch_rows(dy_tk.reshape(bins.n, -1).contiguous(), _SlotMap(bins.perm, bins.n))   # binned row m <- slot perm[m]
        if dout is None:
            dout = torch.zeros(ctx.w.shape[0], tab.Dout, dtype=ctx.saved[4].dtype, device=ctx.w.device)
        need_params = any(ctx.needs_input_grad[5:])
        dx2, dw, grads = _ffn_backward(dout, ctx.w, tab, ctx.saved, ctx.needs_input_grad[0], ctx.needs_input_grad[1], need_params,
                                       dy_extra=extra)
        ctx.saved = None
        pg: List[Optional[torch.Tensor]] = [None] * ctx.n_params
        if grads is not None:
            gW1, gb1, gW2, gb2 = grads
            E = tab.E
            seq = [gW1] + ([gb1] if gb1 is not None else []) + [gW2] + ([gb2] if gb2 is not None else [])
            pg = [g[e] for g in seq for e in range(E)]
        return (dx2, dw, None, None, None, *pg)


class MoEFFNModulesResidual(torch.autograd.Function):
    """MoEFFNModules with the block's residual added in the combine epilogue: out = round(moe_out + residual)
    (`hidden_states = residual + results`, moe_model/model/multimodal_encoder/siglip_smoe.py:155).  The gradient of the
    residual input is dout itself."""

    @staticmethod
    def forward(ctx, x2, residual, w, idx, tab: ExpertTable, combine_mode: int, *params):
        x2 = x2.contiguous()
        out, saved = _ffn_forward(x2, w, idx, tab, combine_mode, None, residual=residual.contiguous())
        ctx.tab, ctx.saved, ctx.w = tab, saved, w
        ctx.n_params = len(params)
        return out

    @staticmethod
    def backward(ctx, dout):
        tab = ctx.tab
        need_params = any(ctx.needs_input_grad[6:])
        dx2, dw, grads = _ffn_backward(dout, ctx.w, tab, ctx.saved, ctx.needs_input_grad[0], ctx.needs_input_grad[2],
                                       need_params)
        ctx.saved = None
        pg: List[Optional[torch.Tensor]] = [None] * ctx.n_params
        if grads is not None:
            gW1, gb1, gW2, gb2 = grads
            E = tab.E
            seq = [gW1] + ([gb1] if gb1 is not None else []) + [gW2] + ([gb2] if gb2 is not None else [])
            pg = [g[e] for g in seq for e in range(E)]
        return (dx2, dout if ctx.needs_input_grad[1] else None, dw, None, None, None, *pg)


class MoEFFNPacked(torch.autograd.Function):
    """Same pipeline for PACKED expert weights keys[E,D,F], values[E,F,Dout] (pretrain stack): the two `cvmm` calls of
    MoE.compute_scores / the reduction-weight cvmm (moe_pretrain_model/layers/moe/moe.py:397-435, smoe.py:240-248) and
    CVMM.backward (layers/cvmm.py:490-551).  keys/values may be fp32 masters while x is bf16 (autocast): operands are
    cast once per call, gradients come back in the master dtype."""

    @staticmethod
    def forward(ctx, x2, w, idx, keys, values, bias, o_bias, act: int, combine_mode: int, residual=None, stats=None):
        """`residual` [T, Dout] (x2.dtype, or fp32 around bf16 rows): added in the combine epilogue, out = residual + MoE(x2) in
        the residual's dtype -- the block around the layer (pretrain/block.py); its gradient is the upstream gradient itself."""
        x2 = x2.contiguous()
        op = x2.dtype
        if op == torch.bfloat16:
            # `reduction_weight.type_as(res) @ res` (cvmm.py:483, :499): the K weights ENTER the products as bf16 values, but the cast
            # sits inside the reference's autograd function -- the gradient it returns for the fp32 weights is not rounded.  (Casting
            # outside, `w.to(bf16).float()`, makes autograd round d w to bf16 on the way back: 1.6e-3 off, tools/grad_stage_probe.py.)
            w = w.to(op).float()
        E, D, F = keys.shape
        Dout = values.shape[2]
        dev = x2.device
        k_hit = v_hit = False
        k_op = v_op = None
        n_rows = x2.shape[0] * idx.shape[-1]
        masters = None
        if (_F32W_ON and not _WEIGHT_CACHE_ON and op == torch.bfloat16 and keys.dtype == torch.float32 and values.dtype == torch.float32
                and keys.is_contiguous() and values.is_contiguous() and ops.f32w_ok(n_rows, F, D) and ops.f32w_ok(n_rows, Dout, F)):
            # fp32 masters converted inside the forward GEMMs' tile fill (the reference's Triton kernel does the same per tile,
            # cvmm.py:126-140); the bf16 copies the backward needs are a side output of those launches.  Not with the operand
            # cache: a copy written this way lacks the experts that received no rows.
            k_op = torch.empty(keys.shape, dtype=op, device=dev)
            v_op = torch.empty(values.shape, dtype=op, device=dev)
            masters = (keys, k_op, values, v_op)
            k_hit = v_hit = True
        if _WEIGHT_CACHE_ON and keys.dtype != op:
            k_op, k_hit = _cached_copy(keys, op)
        if _WEIGHT_CACHE_ON and values.dtype != op:
            v_op, v_hit = _cached_copy(values, op)
        if not k_hit:
            k_op = keys.contiguous() if keys.dtype == op else keys.to(op)
            if _WEIGHT_CACHE_ON and keys.dtype != op:
                _remember(keys, op, k_op)
        side = None
        if v_hit:
            pass
        elif values.dtype == op:
            v_op = values.contiguous()
        elif _CAST_OVERLAP and values.is_contiguous() and values.numel() >= (1 << 24):
            # fp32 master -> bf16 operand cast of `values` (HBM-bound) on a side stream, under the first grouped GEMM (MFMA-bound),
            # which only needs `keys`; the buffer is allocated on the launch stream, so its lifetime follows that stream
            main = torch.cuda.current_stream()
            side = _side_stream(dev)
            v_op = torch.empty(values.shape, dtype=op, device=dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                v_op.copy_(values)
        else:
            v_op = values.to(op)
        if _WEIGHT_CACHE_ON and not v_hit and values.dtype != op:
            _remember(values, op, v_op)
        es = k_op.element_size()
        b_op = None
        b1 = None
        epi1 = L.EPI_BIAS_ACT
        if bias is not None and bias.dtype == torch.float32 and op == torch.bfloat16:
            # `scores = cvmm(...) + self.bias[sel]` under autocast (moe.py:400-405): the fp32 master bias is added to the ALREADY
            # ROUNDED bf16 product in fp32, the activation sees that sum, the next cvmm rounds it
            b_op = bias.contiguous()
            b1 = ops.ptr_table(b_op, E, F * 4)
            epi1 = L.EPI_ROUND_BIAS32_ACT
        elif bias is not None:
            b_op = bias.contiguous() if bias.dtype == op else bias.to(op)
            b1 = ops.ptr_table(b_op, E, F * es)
        ob = None
        if o_bias is not None:
            ob = o_bias.contiguous() if o_bias.dtype == op else o_bias.to(op)
        tab = ExpertTable(E=E, D=D, F=F, Dout=Dout, layout=L.B_KN, act=act,
                          w1_ptrs=ops.ptr_table(k_op, E, D * F * es), w2_ptrs=ops.ptr_table(v_op, E, F * Dout * es),
                          b1_ptrs=b1, b2_ptrs=None, param_dtype=keys.dtype, epi1=epi1, scale_after_gemm=combine_mode == L.COMBINE_DOT)
        out, saved = _ffn_forward(x2, w, idx, tab, combine_mode, ob, residual=None if residual is None else residual.contiguous(),
                                  before_gemm2=side, masters=masters)
        ctx.has_residual = residual is not None
        if stats is not None:               # the activated scores, for the caller's `relu_pass_rate` log (moe.py:406-414)
            stats["hact"] = saved[3]
        ctx.tab, ctx.saved, ctx.w = tab, saved, w
        ctx.keep = (k_op, v_op, b_op)       # keep the cast copies alive: the pointer tables reference them
        ctx.has = (bias is not None, o_bias is not None)
        ctx.bias_dtype = None if bias is None else bias.dtype
        ctx.ob_dtype = None if o_bias is None else o_bias.dtype
        return out

    @staticmethod
    def _backward(ctx, dout, dy_extra=None):
        """(dx2, dw, gk, gv, gb, gob) -- shared with MoEFFNPackedSlots, whose per-slot output gradient arrives as `dy_extra`."""
        tab = ctx.tab
        need_params = ctx.needs_input_grad[3] or ctx.needs_input_grad[4] or ctx.needs_input_grad[5]
        dx2, dw, grads = _ffn_backward(dout, ctx.w, tab, ctx.saved, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                       need_params, dy_extra=dy_extra)
        ctx.saved = None
        gk = gv = gb = gob = None
        if grads is not None:
            gk, gb1, gv, _ = grads
            if ctx.has[0]:
                gb = gb1 if gb1.dtype == ctx.bias_dtype else gb1.to(ctx.bias_dtype)
        if ctx.has[1] and ctx.needs_input_grad[6]:
            gob = _chunked_dense_colsum(dout.contiguous(), ctx.ob_dtype)
        return dx2, dw, gk, gv, gb, gob

    @staticmethod
    def backward(ctx, dout):
        dx2, dw, gk, gv, gb, gob = MoEFFNPacked._backward(ctx, dout)
        return dx2, dw, None, gk, gv, gb, gob, None, None, (dout if (ctx.has_residual and ctx.needs_input_grad[9]) else None), None


class MoEFFNPackedSlots(torch.autograd.Function):
    """MoEFFNPacked that also returns the expert outputs per (token, k) slot, [T, K, Dout]: the `topk` outputs the pretrain
    competition step takes out of its dense pass for the diversity loss (moe_pretrain_model/layers/moe/competesmoe.py:403-410), from
    the sparse step's own rows -- so that the dense pass need not keep its outputs (CompetitionAffinityPacked)."""

    @staticmethod
    def forward(ctx, x2, w, idx, keys, values, bias, o_bias, act: int, combine_mode: int):
        out = MoEFFNPacked.forward(ctx, x2, w, idx, keys, values, bias, o_bias, act, combine_mode, None, None)
        bins, y = ctx.saved[0], ctx.saved[4]
        y_tk = ops.dispatch_rows(y, _SlotMap(bins.slot_of, bins.n))
        return out, y_tk.view(w.shape[0], idx.shape[-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dout, dy_tk):
        if ctx.saved is None:
            _ffn_backward(dout, ctx.w, ctx.tab, None, False, False, False)    # raises the second-backward error
        bins = ctx.saved[0]
        extra = None
        if dy_tk is not None:
            extra = ops.dispatch_rows(dy_tk.reshape(bins.n, -1).contiguous(), _SlotMap(bins.perm, bins.n))
        if dout is None:
            dout = torch.zeros(ctx.w.shape[0], ctx.tab.Dout, dtype=ctx.saved[4].dtype, device=ctx.w.device)
        dx2, dw, gk, gv, gb, gob = MoEFFNPacked._backward(ctx, dout, extra)
        return dx2, dw, None, gk, gv, gb, gob, None, None


# ======================================================================================================== dense FFN
class DenseFFN(torch.autograd.Function):
    """One always-on expert over all tokens: Linear(+b) -> act -> Linear(+b).  The shared expert of `smoe_share` /
    `deepseekv3` (moe_model/model/moe/shard_smoe.py:53, deepseekv3.py:44), the pretrain shared cvmm with an all-zero
    selection (deepseekv2.py:154-165) and each expert of the dense competition pass (competesmoe.py:240-243)."""

    @staticmethod
    def forward(ctx, x2, w1, b1, w2, b2, act: int, layout: int):
        x2 = x2.contiguous()
        op = x2.dtype
        def cast(t):
            if t is None:
                return None
            if t.dtype == op:
                return t.contiguous()
            if _WEIGHT_CACHE_ON and t._base is None:      # fresh slices (keys[e]) die with the call: their entries could never hit
                c, hit = _cached_copy(t, op)
                if not hit:
                    c = t.to(op)
                    _remember(t, op, c)
                return c
            return t.to(op)

        epi1 = L.EPI_BIAS_ACT
        if b1 is not None and b1.dtype == torch.float32 and op == torch.bfloat16 and layout == L.B_KN:
            # the pretrain stack's shared expert: cvmm output + fp32 master bias (see MoEFFNPacked)
            w1o, b1o, w2o, b2o = cast(w1), b1.contiguous(), cast(w2), cast(b2)
            epi1 = L.EPI_ROUND_BIAS32_ACT
        else:
            w1o, b1o, w2o, b2o = cast(w1), cast(b1), cast(w2), cast(b2)
        hpre, hact = ops.dense_gemm(x2, w1o, layout, bias=b1o, epilogue=epi1, act=act, want_c2=True, want_c=act != L.ACT_RELU)
        y = ops.dense_gemm(hact, w2o, layout, bias=b2o, epilogue=L.EPI_BIAS if b2o is not None else L.EPI_PLAIN)
        ctx.save_for_backward(x2, w1o, w2o, hpre, hact)
        ctx.cfg = (act, layout, w1.dtype, None if b1 is None else b1.dtype, w2.dtype, None if b2 is None else b2.dtype)
        ctx.handoff = _take_handoff()
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, w1o, w2o, hpre, hact = ctx.saved_tensors
        act, layout, dt_w1, dt_b1, dt_w2, dt_b2 = ctx.cfg
        dy = dy.contiguous()
        dh = ops.dense_gemm(dy, w2o, _flip(layout), epilogue=L.EPI_ACTGRAD, act=act, aux=hpre if hpre is not None else hact)
        gw1 = gb1 = gw2 = gb2 = dx = None
        if ctx.needs_input_grad[3]:
            gw2 = _dense_wgrad(dy, hact, dt_w2) if layout == L.B_NK else _dense_wgrad(hact, dy, dt_w2)
        if dt_b2 is not None and ctx.needs_input_grad[4]:
            gb2 = _chunked_dense_colsum(dy, dt_b2)
        if ctx.needs_input_grad[1]:
            gw1 = _dense_wgrad(dh, x2, dt_w1) if layout == L.B_NK else _dense_wgrad(x2, dh, dt_w1)
        if dt_b1 is not None and ctx.needs_input_grad[2]:
            gb1 = _chunked_dense_colsum(dh, dt_b1)
        if ctx.needs_input_grad[0]:
            dx = ops.dense_gemm(dh, w1o, _flip(layout))
            h = ctx.handoff
            if h is not None and h.has_consumer and not h.routed_done:
                # the always-on expert of a shared-expert layer: its dx is the FIRST addend of the routed step's chain (DxHandoff);
                # this node hands nothing to the engine (None = zero)
                h.dx, dx = dx, None
        return dx, gw1, gb1, gw2, gb2, None, None


# ======================================================================================================== affinity
class SoftplusMean(torch.autograd.Function):
    """aff[r] = mean_d softplus(y[r,d]) -- `torch.mean(F.softplus(out_i), dim=-1)` (moe_model/model/moe/competesmoe.py:242;
    pretrain competesmoe.py:401).  In y.dtype for the LLaVA stack (x.dtype tensor ops); `fp32_affinity` for the pretrain stack under
    CUDA autocast, where F.softplus is on the fp32 cast list: fp32 softplus / mean of the bf16 expert outputs, fp32 affinities."""

    @staticmethod
    def forward(ctx, y2, fp32_affinity: bool = False):
        y2 = y2.contiguous()
        ctx.save_for_backward(y2)
        ctx.aff_dtype = torch.float32 if fp32_affinity else y2.dtype
        return ops.softplus_mean(y2, ctx.aff_dtype)

    @staticmethod
    def backward(ctx, daff):
        (y2,) = ctx.saved_tensors
        return ops.softplus_mean_bwd(y2, daff.contiguous().to(ctx.aff_dtype)), None


class CompetitionAffinity(torch.autograd.Function):
    """aff[t, e] = mean_d softplus(expert_e(x[t]))[d] for EVERY expert over all tokens -- the dense pass of `competition_policy`
    (moe_model/model/moe/competesmoe.py:237-243; pretrain competesmoe.py:395-401) -- without its [T, E, D] outputs and without saved
    activations: the second GEMM of each expert reduces softplus(y) in its epilogue (y is never stored), and the backward recomputes
    each expert's two GEMMs, turns the recomputed y into dy = d aff / D * sigmoid(y) in the epilogue and runs the four gradient
    GEMMs of DenseFFN.backward.  8 GEMM passes per expert instead of 6, and O(T * (F + D)) scratch instead of O(T * E * (2F + D))
    kept alive (130 GB at 64 experts x 32k tokens x 4096 / 11008).  The K selected experts' OUTPUTS (the weighted sum, the
    diversity loss) come from the sparse step on the same x, which produces the same bits.

    params = w1_0.., [b1_0..], w2_0.., [b2_0..] (E each; biases all-or-none per group), laid out as `layout` says."""

    @staticmethod
    def forward(ctx, x2, E: int, act: int, layout: int, has_b1: bool, has_b2: bool, fp32_affinity: bool, *params):
        x2 = x2.contiguous()
        op = x2.dtype
        w1 = params[:E]
        b1 = params[E:2 * E] if has_b1 else [None] * E
        o = 2 * E if has_b1 else E
        w2 = params[o:o + E]
        b2 = params[o + E:o + 2 * E] if has_b2 else [None] * E
        T = x2.shape[0]
        aff = torch.empty(T, E, dtype=torch.float32 if fp32_affinity else op, device=x2.device)
        cast = lambda t: None if t is None else (t if t.dtype == op else t.to(op)).contiguous()
        for e in range(E):
            hact = ops.dense_gemm(x2, cast(w1[e]), layout, bias=cast(b1[e]), epilogue=L.EPI_BIAS_ACT, act=act, want_c2=True,
                                  want_c=False)[1]
            ops.dense_gemm_affinity(hact, cast(w2[e]), layout, cast(b2[e]), aff[:, e], rounded=not fp32_affinity)
        ctx.save_for_backward(x2, *params)
        ctx.cfg = (E, act, layout, has_b1, has_b2, fp32_affinity)
        return aff

    @staticmethod
    def backward(ctx, daff):
        x2, *params = ctx.saved_tensors
        E, act, layout, has_b1, has_b2, fp32_affinity = ctx.cfg
        op = x2.dtype
        w1 = params[:E]
        b1 = params[E:2 * E] if has_b1 else [None] * E
        o = 2 * E if has_b1 else E
        w2 = params[o:o + E]
        b2 = params[o + E:o + 2 * E] if has_b2 else [None] * E
        cast = lambda t: None if t is None else (t if t.dtype == op else t.to(op)).contiguous()
        daff = daff.contiguous()
        need = ctx.needs_input_grad
        dx = None
        gw1, gb1, gw2, gb2 = [None] * E, [None] * E, [None] * E, [None] * E
        for e in range(E):
            w1o, b1o, w2o, b2o = cast(w1[e]), cast(b1[e]), cast(w2[e]), cast(b2[e])
            hpre, hact = ops.dense_gemm(x2, w1o, layout, bias=b1o, epilogue=L.EPI_BIAS_ACT, act=act, want_c2=True,
                                        want_c=act != L.ACT_RELU)
            # d aff / D is formed in the epilogue as the reference's autograd forms it: rounded to x.dtype for the LLaVA stack
            dy = ops.dense_gemm_affinity_grad(hact, w2o, layout, b2o, daff[:, e].float().contiguous(), rounded=not fp32_affinity)
            dh = ops.dense_gemm(dy, w2o, _flip(layout), epilogue=L.EPI_ACTGRAD, act=act, aux=hpre if hpre is not None else hact)
            k = 7 + e
            if need[k + o]:
                gw2[e] = _dense_wgrad(dy, hact, w2[e].dtype) if layout == L.B_NK else _dense_wgrad(hact, dy, w2[e].dtype)
            if has_b2 and need[k + o + E]:
                gb2[e] = _chunked_dense_colsum(dy, b2[e].dtype)
            if need[k]:
                gw1[e] = _dense_wgrad(dh, x2, w1[e].dtype) if layout == L.B_NK else _dense_wgrad(x2, dh, w1[e].dtype)
            if has_b1 and need[k + E]:
                gb1[e] = _chunked_dense_colsum(dh, b1[e].dtype)
            if need[0]:
                dxe = ops.dense_gemm(dh, w1o, _flip(layout))
                dx = dxe if dx is None else dx.add_(dxe)
            del hpre, hact, dy, dh
        grads = list(gw1) + (list(gb1) if has_b1 else []) + list(gw2) + (list(gb2) if has_b2 else [])
        return (dx, None, None, None, None, None, None, *grads)


class CompetitionAffinityPacked(torch.autograd.Function):
    """CompetitionAffinity for PACKED experts keys [E, D, F] / values [E, F, Dout] (pretrain stack, `competition_policy_mlp_faster`,
    moe_pretrain_model/layers/moe/competesmoe.py:395-401: relu(x @ keys[e]) @ values[e], no biases): same epilogues, same recompute
    backward; the gradients are written expert by expert into packed tensors, so autograd never slices the parameters."""

    @staticmethod
    def forward(ctx, x2, keys, values, act: int, fp32_affinity: bool):
        x2 = x2.contiguous()
        op = x2.dtype
        E = keys.shape[0]
        aff = torch.empty(x2.shape[0], E, dtype=torch.float32 if fp32_affinity else op, device=x2.device)
        for e in range(E):
            k_e = keys[e] if keys.dtype == op else keys[e].to(op)
            v_e = values[e] if values.dtype == op else values[e].to(op)
            hact = ops.dense_gemm(x2, k_e.contiguous(), L.B_KN, epilogue=L.EPI_BIAS_ACT, act=act, want_c2=True, want_c=False)[1]
            ops.dense_gemm_affinity(hact, v_e.contiguous(), L.B_KN, None, aff[:, e], rounded=not fp32_affinity)
        ctx.save_for_backward(x2, keys, values)
        ctx.cfg = (act, fp32_affinity)
        return aff

    @staticmethod
    def backward(ctx, daff):
        x2, keys, values = ctx.saved_tensors
        act, fp32_affinity = ctx.cfg
        op = x2.dtype
        E = keys.shape[0]
        daff = daff.contiguous()
        gk = torch.empty_like(keys) if ctx.needs_input_grad[1] else None
        gv = torch.empty_like(values) if ctx.needs_input_grad[2] else None
        dx = None
        for e in range(E):
            k_e = (keys[e] if keys.dtype == op else keys[e].to(op)).contiguous()
            v_e = (values[e] if values.dtype == op else values[e].to(op)).contiguous()
            hpre, hact = ops.dense_gemm(x2, k_e, L.B_KN, epilogue=L.EPI_BIAS_ACT, act=act, want_c2=True, want_c=act != L.ACT_RELU)
            dy = ops.dense_gemm_affinity_grad(hact, v_e, L.B_KN, None, daff[:, e].float().contiguous(), rounded=not fp32_affinity)
            dh = ops.dense_gemm(dy, v_e, _flip(L.B_KN), epilogue=L.EPI_ACTGRAD, act=act, aux=hpre if hpre is not None else hact)
            if gv is not None:
                gv[e] = _dense_wgrad(hact, dy, values.dtype)
            if gk is not None:
                gk[e] = _dense_wgrad(x2, dh, keys.dtype)
            if ctx.needs_input_grad[0]:
                dxe = ops.dense_gemm(dh, k_e, _flip(L.B_KN))
                dx = dxe if dx is None else dx.add_(dxe)
            del hpre, hact, dy, dh
        return dx, gk, gv, None, None


# ======================================================================================================== diversity loss
class DiversityLoss(torch.autograd.Function):
    """mean over T*K*K of the off-diagonal cosine similarities of the K selected experts' outputs (diagonal zeroed, counted in
    the mean) -- `experts_diversity_loss` (moe_model/model/moe/moe.py:133-171, competesmoe.py:180-218; pretrain
    competesmoe.py).  fp32 math on x.dtype inputs like the reference's `.to(torch.float32)`; K <= 8."""

    @staticmethod
    def forward(ctx, topk_out):
        y3 = topk_out.reshape(-1, topk_out.shape[-2], topk_out.shape[-1]).contiguous()
        T, K, _ = y3.shape
        ctx.save_for_backward(y3)
        ctx.shape = topk_out.shape
        return ops.pair_cosine(y3).sum() / float(T * K * K)

    @staticmethod
    def backward(ctx, g):
        (y3,) = ctx.saved_tensors
        T, K, _ = y3.shape
        gs = (g.float() / float(T * K * K)).reshape(1).contiguous()
        return ops.pair_cosine_bwd(y3, gs).view(ctx.shape)


# ======================================================================================================== MXFP8 experts
# Quantised master weights, kept while the parameter is unchanged (`args.fp8_weight_cache`, off by default): with gradient
# accumulation the same weights serve every micro-batch of an optimizer step, and quantising a 128-expert table costs more than a
# GEMM over it (69 GB of traffic for BASELINE config 5's two tables).  Keyed by the tensor's storage and autograd version counter:
# every in-place update the optimizer (or load_state_dict) makes bumps the version and the next forward quantises again.  Writes
# through `.data` bypass the counter -- call fp8_weight_cache_clear() after those.  One entry per tensor; the entry of a tensor
# that is freed goes with its finaliser.
_FP8_WCACHE = {}


def fp8_weight_cache_clear():
    _FP8_WCACHE.clear()


def _quantize_weight_both(t: torch.Tensor, cache: bool):
    """((q, s) blocks along the last dim, (qt, st) blocks along the second-to-last) of a master weight tensor [.., R, C]."""
    if not cache:
        return ops.quantize_mxfp8_both(t)
    key = (t.data_ptr(), tuple(t.shape), t.dtype)
    hit = _FP8_WCACHE.get(key)
    if hit is not None and hit[0] == t._version:
        return hit[1]
    out = ops.quantize_mxfp8_both(t)
    if hit is None:                                            # first entry under this key: it goes when its owner is freed
        import weakref
        owner = t._base if t._base is not None else t          # `keys_shared[0]` is a fresh view per call: the parameter owns the entry
        weakref.finalize(owner, _FP8_WCACHE.pop, key, None)
    _FP8_WCACHE[key] = (t._version, out)
    return out


class MoEFFNPackedFP8(torch.autograd.Function):
    """MoEFFNPacked with the four row-space expert GEMMs of a step (GEMM 1, GEMM 2, dH, dXs) on the block-scaled fp8 matrix pipe
    (BASELINE config 5; `csmoe_grouped_gemm_mxfp8`): operands quantised to MXFP8 (e4m3 + one e8m0 scale per 32 elements along each
    GEMM's reduction dim), fp32 accumulation, bf16 results.  The weight gradients stay bf16 products of the bf16 activations
    (`csmoe_grouped_wgrad`).  Master weights (fp32 or bf16) are quantised directly, once per use and orientation -- there is no
    bf16 operand copy -- or, with `cache`, once per parameter version (_quantize_weight_both).  No counterpart upstream (the
    reference has no fp8): checked against oracle/mxfp8.py, tests/test_fp8_gpu.py."""

    @staticmethod
    def forward(ctx, x2, w, idx, keys, values, bias, act: int, combine_mode: int, cache: bool = False):
        x2 = x2.contiguous()
        if x2.dtype != torch.bfloat16:
            raise ValueError("competesmoe_amd: the fp8 expert path takes bf16 activations (run under bf16 autocast)")
        w = w.to(torch.bfloat16).float()          # the K weights enter as bf16 values (as in MoEFFNPacked); d w comes back unrounded
        E, D, F = keys.shape
        T = x2.shape[0]
        bins = ops.bin_tokens(idx, E)
        xs = ops.dispatch_tokens(x2, bins)
        xq, xsc = ops.quantize_mxfp8(xs)
        (kq_b, ks_b), (kq, ks) = _quantize_weight_both(keys, cache)        # backward pair [E, D, F]; forward pair [E, F, D] (reduction over D)
        b_op = b1 = None
        if bias is not None:
            b_op = bias.to(torch.bfloat16).contiguous()
            b1 = ops.ptr_table(b_op, E, F * 2)
        hpre, hact = ops.grouped_gemm_mxfp8(xq, xsc, kq, ks, bins.offsets, bias_ptrs=b1, epilogue=L.EPI_BIAS_ACT, act=act,
                                            want_c2=True, want_c=act != L.ACT_RELU)
        del xq, xsc, kq, ks
        hq, hs = ops.quantize_mxfp8(hact)
        (vq_b, vs_b), (vq, vs) = _quantize_weight_both(values, cache)      # backward pair [E, F, Dout]; forward pair [E, Dout, F]
        y = ops.grouped_gemm_mxfp8(hq, hs, vq, vs, bins.offsets)
        out = ops.combine(y, bins, idx, w, combine_mode, T)
        ctx.saved = (bins, xs, hpre, hact, y, b_op)
        ctx.wq = (kq_b, ks_b, vq_b, vs_b)        # each master tensor is read ONCE per step: the backward's operands are kept
        ctx.w, ctx.keys, ctx.values = w, keys, values
        ctx.cfg = (act, None if bias is None else bias.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.saved is None:
            raise RuntimeError("competesmoe_amd: saved activations were freed by the first backward pass; run the forward again")
        bins, xs, hpre, hact, y, _ = ctx.saved
        ctx.saved = None
        act, bias_dtype = ctx.cfg
        keys, values = ctx.keys, ctx.values
        E, D, F = keys.shape
        T = dout.shape[0]
        need_dw = ctx.needs_input_grad[1]
        dy, dw = ops.combine_bwd(dout.contiguous(), y if need_dw else None, bins, ctx.w, want_dw=need_dw, act_dtype=torch.bfloat16)
        kq, ks, vq, vs = ctx.wq
        ctx.wq = None
        dyq, dys = ops.quantize_mxfp8(dy)
        dh = ops.grouped_gemm_mxfp8(dyq, dys, vq, vs, bins.offsets, epilogue=L.EPI_ACTGRAD, act=act,                 # values as stored
                                    aux=hpre if hpre is not None else hact)
        del dyq, dys, vq, vs
        gk = gv = gb = None
        pd = keys.dtype
        if ctx.needs_input_grad[4]:
            gv = _grouped_wgrad(hact, dy, bins, E, pd)                     # [E, F, Dout]
        if ctx.needs_input_grad[3]:
            gk = _grouped_wgrad(xs, dh, bins, E, pd)                       # [E, D, F]
        if bias_dtype is not None and ctx.needs_input_grad[5]:
            gb = _grouped_colsum(dh, bins.offsets, E, bias_dtype)
        dx2 = None
        if ctx.needs_input_grad[0]:
            dhq, dhs = ops.quantize_mxfp8(dh)
            dxs = ops.grouped_gemm_mxfp8(dhq, dhs, kq, ks, bins.offsets)   # keys as stored: dxs = dh @ keys[e]^T
            dx2 = ops.dispatch_rows_bwd(dxs, bins, T)
        return dx2, dw, None, gk, gv, gb, None, None, None


class DenseFFNFP8(torch.autograd.Function):
    """The always-on shared expert on the same fp8 pipe: x [T, D] -> act(x @ w1 [D, Fs] (+ b1)) @ w2 [Fs, Dout] (packed layout of
    the pretrain stack's `keys_shared[0]` / `values_shared[0]`, deepseekv2.py:97-105)."""

    @staticmethod
    def forward(ctx, x2, w1, b1, w2, act: int, cache: bool = False):
        x2 = x2.contiguous()
        xq, xsc = ops.quantize_mxfp8(x2)
        (w1q_b, w1s_b), (w1q, w1s) = _quantize_weight_both(w1, cache)      # backward pair [D, Fs]; forward pair [Fs, D]
        b1o = None if b1 is None else b1.to(torch.bfloat16).contiguous()
        hpre, hact = ops.dense_gemm_mxfp8(xq, xsc, w1q, w1s, bias=b1o, epilogue=L.EPI_BIAS_ACT, act=act, want_c2=True,
                                          want_c=act != L.ACT_RELU)
        hq, hs = ops.quantize_mxfp8(hact)
        (w2q_b, w2s_b), (w2q, w2s) = _quantize_weight_both(w2, cache)      # backward pair [Fs, Dout]; forward pair [Dout, Fs]
        y = ops.dense_gemm_mxfp8(hq, hs, w2q, w2s)
        ctx.save_for_backward(x2, hpre, hact, w1, w2, w1q_b, w1s_b, w2q_b, w2s_b)
        ctx.cfg = (act, None if b1 is None else b1.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, hpre, hact, w1, w2, w1q, w1s, w2q, w2s = ctx.saved_tensors
        act, b1_dtype = ctx.cfg
        dy = dy.contiguous()
        dyq, dys = ops.quantize_mxfp8(dy)
        dh = ops.dense_gemm_mxfp8(dyq, dys, w2q, w2s, epilogue=L.EPI_ACTGRAD, act=act, aux=hpre if hpre is not None else hact)
        gw1 = gb1 = gw2 = dx = None
        if ctx.needs_input_grad[3]:
            gw2 = _dense_wgrad(hact, dy, w2.dtype)                         # [Fs, Dout]
        if ctx.needs_input_grad[1]:
            gw1 = _dense_wgrad(x2, dh, w1.dtype)                           # [D, Fs]
        if b1_dtype is not None and ctx.needs_input_grad[2]:
            gb1 = _chunked_dense_colsum(dh, b1_dtype)
        if ctx.needs_input_grad[0]:
            dhq, dhs = ops.quantize_mxfp8(dh)
            dx = ops.dense_gemm_mxfp8(dhq, dhs, w1q, w1s)
        return dx, gw1, gb1, gw2, None, None
