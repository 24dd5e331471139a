"""`cvmm` ("conditional vector-matrix multiply") with the reference's API, on the HIP grouped GEMM.

Same surface as moe_pretrain_model/layers/cvmm.py: `CVMMSel` (:11-20), `cvmm_prepare_sel2(sel, w)` (:580-593) and
`cvmm(x, sel, keys)` (:555-577).  Differences: the sort is a stable counting sort on the GPU (the reference's `sort()` is
unstable; any order inside an expert gives the same outputs), gradients of the weights are deterministic (no atomics).
The MoE layer classes do NOT go through this generic op -- they use the fused pipeline `functional.MoEFFNPacked` -- it exists
for callers of the reference's `cvmm` API (e.g. MoE attention projections)."""
from dataclasses import dataclass
from typing import Optional

import torch

from .. import _lib as L
from .. import ops


@dataclass
class CVMMSel:
    raw_sel: torch.Tensor
    sel: torch.Tensor
    sel_index: torch.Tensor
    out_index: Optional[torch.Tensor] = None
    reduction_weight: Optional[torch.Tensor] = None
    bins: Optional[ops.Bins] = None          # binned row space of raw_sel (ours)

    def clone(self) -> "CVMMSel":
        return CVMMSel(self.raw_sel, self.sel, self.sel_index, self.out_index, self.reduction_weight, self.bins)


def get_dtype():
    return torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else torch.float32


def cvmm_prepare_sel2(sel: torch.Tensor, w: Optional[torch.Tensor] = None, n_experts: Optional[int] = None) -> CVMMSel:
    K = sel.shape[-1]
    flat = sel.reshape(-1, K).int().contiguous()
    E = int(n_experts) if n_experts is not None else int(flat.max().item()) + 1
    b = ops.bin_tokens(flat, E)
    ssel = flat.flatten()[b.perm.long()]
    return CVMMSel(sel, ssel.view_as(sel), torch.div(b.perm, K, rounding_mode="floor"), b.perm, w, b)


class _CVMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, keys, weight, bins, rows_are_slots: bool):
        op = get_dtype()
        E, Din, Dout = keys.shape
        xf = x.reshape(-1, x.shape[-1]).to(op).contiguous()
        k_op = keys.to(op).contiguous()
        ar = ops.cached_arange(E, x.device)
        ptrs = k_op.data_ptr() + ar * (Din * Dout * k_op.element_size())
        if rows_are_slots:      # x has one row per (token, k) slot in flat order -> bring to the binned order
            xs = ops.dispatch_rows(xf, ops.Bins(None, None, bins.perm, None, bins.n, bins.E, 1))
        else:                   # x has one row per token
            xs = ops.dispatch_rows(xf, bins)
        ys = ops.grouped_gemm(xs, ptrs, L.B_KN, Dout, Dout, bins.offsets, E)
        T = bins.n // bins.K
        rbins = None
        if weight is not None:
            # `res.view(*w.shape, N)`, `w.unsqueeze(-2).type_as(res) @ res` (cvmm.py:481-483): the reduction runs over the LAST dim of
            # the weights -- all K selections of a token in the MoE layers, the K experts of ONE head in the MoE attention
            # projections (w [B, N, heads, K]: flat slot (t * heads + h) * K + k) -- and the weights enter in the op dtype
            Kr = weight.shape[-1]
            rbins = bins if Kr == bins.K else ops.Bins(bins.counts, bins.offsets, bins.perm, bins.slot_of, bins.n, bins.E, Kr)
            out = ops.combine(ys, rbins, None, weight.reshape(-1, Kr).to(op).float().contiguous(), L.COMBINE_DOT, bins.n // Kr)
        else:                   # back to the flat (t*K+k) order
            out = ops.dispatch_rows(ys, ops.Bins(None, None, bins.slot_of, None, bins.n, bins.E, 1))
        ctx.save_for_backward(xs, k_op, None, weight)
        ctx.meta = (bins, rows_are_slots, ptrs, keys.dtype, x.shape, x.dtype, T, rbins)
        return out

    @staticmethod
    def backward(ctx, g):
        xs, k_op, ys, weight = ctx.saved_tensors
        bins, rows_are_slots, ptrs, kd, xshape, xdt, T, rbins = ctx.meta
        E, Din, Dout = k_op.shape
        g = g.reshape(-1, Dout).to(k_op.dtype).contiguous()
        dw = None
        ar = ops.cached_arange(E, g.device)
        if weight is not None:
            # CVMM.backward of the weighted call (cvmm.py:497-547), rounding points included: the weight gradient multiplies the rows
            # of x with round(w * g); the input gradient sends the UNSCALED upstream rows through the product, rounds, multiplies by
            # the op-dtype weight and rounds again; d w = <unscaled rounded product, x row>, returned unrounded.
            Kr = rbins.K
            wq = weight.reshape(-1, Kr).to(k_op.dtype).float().contiguous()
            gs, _ = ops.combine_bwd(g, None, rbins, wq, want_dw=False)
            gu = ops.dispatch_tokens(g, rbins)
            w_rows = wq.reshape(-1)[rbins.perm.long()].contiguous()
            dot_cols = ops.rowdot_cols(bins.n, Din, Dout, Dout, Dout, Din, k_op.dtype)
            if dot_cols:
                dot = torch.empty(bins.n, dot_cols, dtype=torch.float32, device=g.device)
                dxs = ops.grouped_gemm(gu, ptrs, L.B_NK, Dout, Din, bins.offsets, E, epilogue=L.EPI_ACTGRAD_ROWSCALE, act=L.ACT_NONE,
                                       aux=xs, row_scale=w_rows, row_dot=dot)
                dwr = ops.finish_row_dot(dot)
            else:               # fp32 / unaligned shapes (generic kernel: no dot table): the same arithmetic in three steps
                prod = ops.grouped_gemm(gu, ptrs, L.B_NK, Dout, Din, bins.offsets, E)
                dwr = (prod.float() * xs.float()).sum(-1)
                dxs = (prod.float() * w_rows.unsqueeze(-1)).to(prod.dtype)
            dw = dwr[rbins.slot_of.long()].view_as(weight).to(weight.dtype)
        else:
            gs = ops.dispatch_rows(g, ops.Bins(None, None, bins.perm, None, bins.n, bins.E, 1))
            dxs = ops.grouped_gemm(gs, ptrs, L.B_NK, Dout, Din, bins.offsets, E)
        gk = torch.empty(E, Din, Dout, dtype=kd, device=g.device)
        ops.grouped_wgrad(xs, gs, bins.offsets, E, gk, gk.data_ptr() + ar * (Din * Dout * gk.element_size()))
        if rows_are_slots:
            dx = ops.dispatch_rows(dxs, ops.Bins(None, None, bins.slot_of, None, bins.n, bins.E, 1))
        else:       # an fp32 x under bf16 autocast takes the widened gradient straight from the gather-sum
            dx = ops.dispatch_rows_bwd(dxs, bins, T, out_f32=(xdt == torch.float32 and dxs.dtype == torch.bfloat16 and dxs.shape[1] % 8 == 0))
        return dx.view(xshape).to(xdt), gk, dw, None, None


def cvmm(x: torch.Tensor, sel: CVMMSel, keys: torch.Tensor) -> torch.Tensor:
    """out[..] = x[sel_index] @ keys[sel] scattered to out_index (and reduced with reduction_weight if given)."""
    if not isinstance(sel, CVMMSel):
        sel = cvmm_prepare_sel2(sel.unsqueeze(-1) if sel.dim() == x.dim() - 1 else sel, n_experts=keys.shape[0])
    if sel.bins is None:
        raise ValueError("cvmm: selection was not produced by cvmm_prepare_sel2")
    rows_are_slots = sel.out_index is None      # second-call convention: sel_index <- out_index, out_index <- None
    out = _CVMM.apply(x, keys, sel.reduction_weight, sel.bins, rows_are_slots)
    if sel.reduction_weight is not None:
        return out.view(*sel.reduction_weight.shape[:-1], keys.shape[-1])      # [.., heads, out] for [B, N, heads, K] weights
    return out.view(*sel.sel.shape, keys.shape[-1])
