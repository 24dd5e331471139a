"""CPU ORACLE -- test infrastructure, NOT product code.

A from-scratch CPU restatement (plain PyTorch CPU ops, fp32 and bf16) of the sparse-MoE hot path of
Fsoft-AIC/CompeteSMoE.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this module, and only as the checker / the timed CPU baseline.  The shipped path
(`competesmoe_amd`) never imports it and fails loudly when its HIP library is missing.

Pinned by: `tests/golden/*.pt`, produced by running the reference classes themselves in the build
container (`tests/golden/make_golden_{llava,pretrain,block,pretrain_block}.py`; the pretrain fixtures run the
reference's own Triton cvmm kernels under the Triton interpreter and, for bf16, the CUDA autocast policy the
reference trains under -- tests/golden/ref_env.py); `tests/test_oracle_golden.py` checks every function below
against them.  The reference has no unit tests / golden vectors of its own for this
path, so those captured outputs are the pin.

Every function cites the reference file:line it restates (paths relative to the reference root).
Gradients come from torch autograd over these differentiable restatements.
"""
from __future__ import annotations

import math
from typing import Dict, Sequence, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# Router  (moe_model/model/moe/moe.py:113-132, smoe.py:42-44, competesmoe.py:301-320)
# --------------------------------------------------------------------------------------------
def gate_logits(x: torch.Tensor, w_gate: torch.Tensor) -> torch.Tensor:
    """`self.gate(x)`: Linear(D->E, bias=False), output in x.dtype (smoe.py:42)."""
    return F.linear(x, w_gate)


def topk_lowest_index(v: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Top-k, descending, ties broken by LOWEST index (the build's defined tie rule; torch.topk
    on CPU has no stable rule, SURVEY.md §7 hard parts).  Iterative arg-max so it is exact."""
    vals, idxs = [], []
    work = v.clone()
    for _ in range(k):
        m, i = work.max(dim=-1, keepdim=True)   # torch.max returns the first (lowest) index on ties
        vals.append(m)
        idxs.append(i)
        work = work.scatter(-1, i, float("-inf"))
    return torch.cat(vals, -1), torch.cat(idxs, -1)


def router_topk(logits: torch.Tensor, k: int, x_dtype: torch.dtype, use_torch_topk: bool = False):
    """softmax(fp32) -> top-k -> renormalise (moe.py:129-130, smoe.py:44).

    The denominator is rounded to x.dtype first (`torch.sum(...).to(x.dtype)`), the quotient stays fp32."""
    sm = F.softmax(logits, dim=-1, dtype=torch.float32)
    if use_torch_topk:
        w, idx = torch.topk(sm, k)
    else:
        idx = topk_lowest_index(sm.detach(), k)[1]
        w = torch.gather(sm, -1, idx)
    w = w / torch.sum(w, dim=-1, keepdim=True).to(x_dtype)
    return w, idx, sm


# --------------------------------------------------------------------------------------------
# Aux losses (moe.py:71-110, 214-226; competesmoe.py:180-218, 322-335)
# --------------------------------------------------------------------------------------------
def zloss(logits: torch.Tensor) -> torch.Tensor:
    """moe.py:71-88: mean(logsumexp(logits)^2), in the logits' dtype."""
    return torch.square(torch.logsumexp(logits, dim=-1)).mean()


def balanceloss(selected: torch.Tensor, softmax: torch.Tensor, n_experts: int) -> torch.Tensor:
    """moe.py:90-110: mean_n(softmax) * mean_n(onehot(top-1 column)), mean over (b,e), * E^2."""
    proxy = softmax.mean(dim=-2)
    onehot = F.one_hot(selected[..., 0], n_experts).float()
    dens = onehot.mean(dim=-2)
    return (proxy * dens).mean() * float(n_experts ** 2)


def combine_loss(selected, softmax, logits, n_experts, balance_coef, z_coef):
    """moe.py:214-226."""
    bal = balanceloss(selected, softmax, n_experts)
    z = zloss(logits)
    return bal * balance_coef + z * z_coef, bal, z


def router_loss(gate_softmax, affinity_softmax):
    """competesmoe.py:322-335: MSE."""
    return F.mse_loss(gate_softmax, affinity_softmax)


def experts_diversity_loss(topk_out: torch.Tensor) -> torch.Tensor:
    """competesmoe.py:180-218: mean over T*K*K of off-diagonal cosine similarities (diag zeroed, counted)."""
    eo = topk_out.to(torch.float32)
    B, N, K, D = eo.shape
    nrm = F.normalize(eo, p=2, dim=-1).view(B * N, K, D)
    sim = torch.bmm(nrm, nrm.transpose(1, 2))
    sim = sim * (1 - torch.eye(K))
    return sim.mean()


def entropy_balance(logits: torch.Tensor) -> torch.Tensor:
    """moe_pretrain_model/layers/moe/moe.py:323-332 with framework/utils/entropy.py:21-22 and
    distributed_ops.py:47-58 (non-distributed branch): -H(logmeanexp_n log_softmax(logits)), mean over batch."""
    sel = logits.flatten(1, -2)
    ls = F.log_softmax(sel.float(), dim=-1)      # bf16 logits under CUDA autocast: log_softmax is an fp32-policy op
    lm = ls.logsumexp(-2) - math.log(ls.shape[-2])
    ent = -(lm * lm.exp()).sum(-1)
    return -ent.mean()


# --------------------------------------------------------------------------------------------
# Expert FFN + compute_moe (LLaVA stack: moe.py:172-213; experts: siglip_smoe.py:85-97, builder.py:61-65)
# --------------------------------------------------------------------------------------------
ACTS = {
    "gelu": lambda h: F.gelu(h),
    "gelu_tanh": lambda h: F.gelu(h, approximate="tanh"),
    "relu": lambda h: F.relu(h),
    "silu": lambda h: F.silu(h),
    "quick_gelu": lambda h: h * torch.sigmoid(1.702 * h),      # transformers QuickGELUActivation (CLIP towers, clip_smoe.py CLIPMLP)
    "none": lambda h: h,
}


def expert_ffn(x, w1, b1, w2, b2, act: str):
    """Linear(D,F)+b -> act -> Linear(F,Dout)+b  (every intermediate rounded to x.dtype, as nn.Linear does)."""
    h = F.linear(x, w1, b1)
    a = ACTS[act](h)
    return F.linear(a, w2, b2)


def compute_moe(x, selected, weights, experts: Sequence[Tuple], act: str, out_dim: int):
    """moe.py:172-213: loop over experts in index order; results[b,t] += w[b,t,k] * expert(x[b,t]).

    `results` lives in x.dtype: every `+=` computes in fp32 (fp32 weight * x.dtype out) and rounds back."""
    B, N, D = x.shape
    results = torch.zeros(B, N, out_dim, dtype=x.dtype)
    for i, (w1, b1, w2, b2) in enumerate(experts):
        bi, ti, ki = torch.where(selected == i)
        out = expert_ffn(x[bi, ti], w1, b1, w2, b2, act)
        contrib = weights[bi, ti, ki].unsqueeze(0).T * out
        results = results.index_put((bi, ti), (results[bi, ti] + contrib).to(x.dtype))
    return results


# --------------------------------------------------------------------------------------------
# Competition policy (competesmoe.py:219-259)
# --------------------------------------------------------------------------------------------
def competition_policy(x, experts, act, k, norm_sigmoid=False, use_torch_topk=False):
    """All experts densely; affinity = mean_D softplus(out_i) in x.dtype; softmax fp32; top-k on RAW affinity."""
    B, N, D = x.shape
    E = len(experts)
    outs = [expert_ffn(x, *experts[i], act) for i in range(E)]
    aff = torch.stack([torch.mean(F.softplus(o), dim=-1) for o in outs], dim=-1).to(x.dtype)
    aff_sm = F.softmax(aff, dim=-1, dtype=torch.float32)
    score = torch.sigmoid(aff) if norm_sigmoid else aff
    if use_torch_topk:
        w, idx = torch.topk(score, k)
    else:
        idx = topk_lowest_index(score.detach(), k)[1]
        w = torch.gather(score, -1, idx)
    w = w / torch.sum(w, dim=-1, keepdim=True).to(x.dtype)
    all_out = torch.stack(outs, dim=2)                       # [B,N,E,Dout]
    topk_out = torch.gather(all_out, 2, idx.unsqueeze(-1).expand(B, N, k, all_out.size(-1)))
    return w, idx, aff_sm, aff, topk_out


# --------------------------------------------------------------------------------------------
# LLaVA-stack layer forwards
# --------------------------------------------------------------------------------------------
def llava_smoe_forward(x, w_gate, experts, act, k, args, out_dim=None, forced_idx=None):
    """SMoeLayer.forward (smoe.py:39-64).  Returns (output, aux, infor_aux, stages)."""
    E = w_gate.shape[0]
    out_dim = out_dim or experts[0][2].shape[0]
    lg = gate_logits(x, w_gate)
    w, idx, sm = router_topk(lg, k, x.dtype)
    if forced_idx is not None:   # evaluate with the reference's own (tie-broken) indices
        idx = forced_idx
        w = torch.gather(sm, -1, idx)
        w = w / torch.sum(w, dim=-1, keepdim=True).to(x.dtype)
    out = compute_moe(x, idx, w, experts, act, out_dim)
    aux = torch.tensor(0.0, dtype=x.dtype)
    infor = {}
    if x.requires_grad:
        aux, bal, z = combine_loss(idx, sm, lg, E, args.balance_loss_coef, args.router_z_loss_coef)
        infor = {"balance_loss": bal.detach(), "router_z_loss": z.detach()}
    return out, aux, infor, dict(gate_logits=lg, gate_softmax=sm, selected_experts=idx, weights=w)


def llava_competesmoe_forward(x, w_gate, experts, act, k, args, competing: bool, out_dim=None,
                              forced_idx=None, forced_aff_idx=None):
    """CompeteSMoE.forward (competesmoe.py:337-415); `competing` = the scheduled-branch test (:347)."""
    E = w_gate.shape[0]
    out_dim = out_dim or experts[0][2].shape[0]
    lg = gate_logits(x, w_gate)
    gw, gidx, gsm = router_topk(lg, k, x.dtype)
    if forced_idx is not None:
        gidx = forced_idx
        gw = torch.gather(gsm, -1, gidx)
        gw = gw / torch.sum(gw, dim=-1, keepdim=True).to(x.dtype)
    stages = dict(gate_logits=lg, gate_softmax=gsm, selected_experts=gidx, weights=gw)
    aux = torch.tensor(0.0, dtype=x.dtype)
    infor = {}
    if x.requires_grad and competing:
        aw, aidx, asm, aff, topk_out = competition_policy(x, experts, act, k, getattr(args, "norm_sigmoid", False))
        if forced_aff_idx is not None:
            aidx = forced_aff_idx
            score = torch.sigmoid(aff) if getattr(args, "norm_sigmoid", False) else aff
            aw = torch.gather(score, -1, aidx)
            aw = aw / torch.sum(aw, dim=-1, keepdim=True).to(x.dtype)
            allo = torch.stack([expert_ffn(x, *experts[i], act) for i in range(E)], dim=2)
            topk_out = torch.gather(allo, 2, aidx.unsqueeze(-1).expand(*aidx.shape, allo.size(-1)))
        rl = router_loss(gsm, asm.detach())
        if getattr(args, "hybrid", False):
            g_top = torch.gather(gsm, -1, aidx)
            a_top = torch.gather(asm, -1, aidx)
            rl = rl + router_loss(g_top, a_top.detach()) * args.router_theta
        div = experts_diversity_loss(topk_out)
        bal = balanceloss(aidx, asm, E)
        aux = rl * args.router_loss_coef + div * args.diversity_loss_coef + bal * args.bal_comp_loss_coef
        out = compute_moe(x, aidx, aw, experts, act, out_dim)
        infor = {"balance_loss": bal.detach(), "diversity_loss": div.detach(), "routerloss": rl.detach()}
        stages.update(aff_weights=aw, aff_selected=aidx, aff_softmax=asm, aff_scores=aff, aff_topk_out=topk_out)
    else:
        out = compute_moe(x, gidx, gw, experts, act, out_dim)
        if x.requires_grad:
            aux, bal, z = combine_loss(gidx, gsm, lg, E, args.balance_loss_coef, args.router_z_loss_coef)
            infor = {"balance_loss": bal.detach(), "router_z_loss": z.detach()}
    return out, aux, infor, stages


def llava_shared_forward(x, w_gate, experts, act, k, args, mode: str, out_dim=None, forced_idx=None):
    """MoEShareLayer.forward: `smoe_share` (shard_smoe.py:38-67, 0.5/0.5 mix) and `deepseekv3`
    (deepseekv3.py:35-56, shared + routed; aux always computed).  w_gate has E-1 rows; top-(K-1)."""
    Er = w_gate.shape[0]
    out_dim = out_dim or experts[0][2].shape[0]
    lg = gate_logits(x, w_gate)
    w, idx, sm = router_topk(lg, k - 1, x.dtype)
    if forced_idx is not None:
        idx = forced_idx
        w = torch.gather(sm, -1, idx)
        w = w / torch.sum(w, dim=-1, keepdim=True).to(x.dtype)
    routed = compute_moe(x, idx, w, experts[:Er], act, out_dim)
    shared = expert_ffn(x, *experts[Er], act)
    out = torch.zeros_like(routed)
    if mode == "smoe_share":
        out = out + (shared * 0.5 + routed * 0.5)
    else:
        out = out + (shared + routed)
    aux = torch.tensor(0.0, dtype=x.dtype)
    infor = {}
    if x.requires_grad or mode == "deepseekv3":
        aux, bal, z = combine_loss(idx, sm, lg, Er, args.balance_loss_coef, args.router_z_loss_coef)
        infor = {"balance_loss": bal.detach(), "router_z_loss": z.detach()}
    return out, aux, infor, dict(gate_logits=lg, gate_softmax=sm, selected_experts=idx, weights=w)


# --------------------------------------------------------------------------------------------
# Competition schedule (competesmoe.py:35-176 / pretrain competesmoe.py:123-273)
# --------------------------------------------------------------------------------------------
def llava_block_forward(x, ln_weight, ln_bias, eps, moe_forward):
    """The MoE half of SiglipEncoderMoELayer.forward (moe_model/model/multimodal_encoder/siglip_smoe.py:152-155):
    residual + moelayer(layer_norm2(residual)).  `moe_forward(xn)` is one of the llava_*_forward functions above (closure over
    its weights); returns (hidden, aux, infor, stages, xn)."""
    xn = F.layer_norm(x, (x.shape[-1],), ln_weight, ln_bias, eps)
    out, aux, infor, st = moe_forward(xn)
    return x + out, aux, infor, st, xn


def make_prob_flips(flip_steps: int, rate_flip: float, max_compete_in_iter: int,
                    prev: Dict[int, torch.Tensor]) -> torch.Tensor:
    """One `torch.rand(1)` per slot; cap layers competing per step by shifting left then right (:108-130)."""
    freq = torch.zeros(flip_steps, dtype=torch.int)
    for v in prev.values():
        freq += v.int()
    cur = [False] * flip_steps
    for i in range(flip_steps):
        if torch.rand(1).item() < rate_flip:
            if freq[i] < max_compete_in_iter:
                cur[i] = True
                freq[i] += 1
            else:
                found = False
                for j in range(i - 1, -1, -1):
                    if freq[j] < max_compete_in_iter and not cur[j]:
                        cur[j] = True
                        freq[j] += 1
                        found = True
                        break
                if not found:
                    for j in range(i + 1, flip_steps):
                        if freq[j] < max_compete_in_iter and not cur[j]:
                            cur[j] = True
                            freq[j] += 1
                            break
    return torch.tensor(cur, dtype=torch.bool)


# --------------------------------------------------------------------------------------------
# Pretrain stack: cvmm index semantics (moe_pretrain_model/layers/cvmm.py)
# --------------------------------------------------------------------------------------------
def bin_tokens(sel: torch.Tensor, n_experts: int):
    """Stable counting sort of the flattened [T,K] expert ids -- the build's deterministic replacement for the
    reference's unstable `fsel.sort()` (cvmm.py:580-593).  Returns (counts[E], offsets[E+1], perm[T*K]) where
    perm[m] is the flat (t*K+k) index held by sorted slot m; in_index = perm // K, out_index = perm."""
    f = sel.flatten().long()
    perm = torch.sort(f, stable=True).indices
    counts = torch.bincount(f, minlength=n_experts)
    offsets = torch.zeros(n_experts + 1, dtype=torch.long)
    offsets[1:] = counts.cumsum(0)
    return counts, offsets, perm


def _cvmm_rows(x2, in_index, sel_sorted, keys, out_index, op_dtype):
    """cvmm_kernel (cvmm.py:61-168): out[out_index[m]] = round_op(x2[in_index[m]] @ keys[sel_sorted[m]]), operands cast to the op
    dtype per tile, fp32 accumulate.  x2 [R, Din], keys [E, Din, Dout] -> [M, Dout] in the op dtype."""
    rows = x2[in_index.long()].to(op_dtype)
    M, Dout = in_index.shape[0], keys.shape[-1]
    prod = torch.zeros(M, Dout, dtype=op_dtype)
    ss = sel_sorted.flatten().long()
    for e in range(keys.shape[0]):
        m = (ss == e).nonzero().squeeze(-1)
        if m.numel():
            prod = prod.index_put((m,), (rows[m].float() @ keys[e].to(op_dtype).float()).to(op_dtype))
    return torch.zeros(M, Dout, dtype=op_dtype).index_put((out_index.long(),), prod)


class _CvmmFn(torch.autograd.Function):
    """CVMM.forward / CVMM.backward (cvmm.py:460-551) restated with the reference's rounding points: the op dtype is the autocast
    dtype (bf16) or fp32; the K weights enter as op-dtype values (`type_as`); the weight gradient multiplies op-dtype operands
    into an fp32 (master dtype) accumulator (cvmm_backward_kernel3, :194-345); the input gradient is the UNSCALED grouped product
    rounded to the op dtype, then scaled by the op-dtype weights (:538-547)."""

    @staticmethod
    def forward(ctx, x, sel_sorted, in_index, out_index, keys, reduction_weight, op_dtype):
        ctx.save_for_backward(x, keys, sel_sorted, in_index, out_index, reduction_weight)
        ctx.op = op_dtype
        x2 = x.flatten(end_dim=-2)
        res = _cvmm_rows(x2, in_index, sel_sorted, keys, in_index if out_index is None else out_index, op_dtype)
        if reduction_weight is not None:
            w = reduction_weight
            res = res.view(*w.shape, res.shape[-1])
            res = (w.unsqueeze(-2).type_as(res) @ res).squeeze(-2)
        return res

    @staticmethod
    def backward(ctx, g):
        x, keys, sel_sorted, in_index, out_index, w = ctx.saved_tensors
        op = ctx.op
        gw = (w.unsqueeze(-1).type_as(g) @ g.unsqueeze(-2)) if w is not None else g
        # weight gradient (cvmm_triton_backward, :421-457): A = x rows by sel_index, B = grad rows by out_index (or sel_index)
        x2 = x.flatten(end_dim=-2)
        g2 = gw.flatten(end_dim=-2)
        a = x2[in_index.long()].to(op).float()
        b = g2[(in_index if out_index is None else out_index).long()].to(op).float()
        ss = sel_sorted.flatten().long()
        gk = torch.zeros(keys.shape, dtype=torch.float32)
        for e in range(keys.shape[0]):
            m = (ss == e).nonzero().squeeze(-1)
            if m.numel():
                gk[e] = a[m].t() @ b[m]
        gk = gk.to(keys.dtype)
        # input gradient (:519-549)
        bw_index = in_index if out_index is None else out_index
        bw_out = bw_index
        if w is not None:
            bw_index = bw_index // w.shape[-1]
        gxf = _cvmm_rows(g.flatten(end_dim=-2), bw_index, sel_sorted, keys.transpose(1, 2), bw_out, op)
        gxf = gxf.view(*x.shape[:-1], -1, x.shape[-1])
        gwo = None
        if w is not None:
            gx = (w.view(*gxf.shape[:-1]).unsqueeze(-2).type_as(gxf) @ gxf).squeeze(-2)
            gwo = (gxf.type_as(w) @ x.unsqueeze(-1).type_as(w)).squeeze(-1).view_as(w)
        elif gxf.shape[-2] != 1:
            gx = gxf.sum(-2)
        else:
            gx = gxf
        return gx.view_as(x).to(x.dtype), None, None, None, gk, gwo, None


def cvmm_ref(x, sel_sorted, in_index, out_index, keys, op_dtype, reduction_weight=None):
    """`cvmm(x, sel, keys)` (cvmm.py:555-577) for a CVMMSel(sel=sel_sorted, sel_index=in_index, out_index=out_index (may be None),
    reduction_weight).  Forward AND backward follow the reference's kernels / autograd function (see _CvmmFn)."""
    return _CvmmFn.apply(x, sel_sorted, in_index, out_index, keys, reduction_weight, op_dtype)


def pretrain_ffn(x, idx, weights, keys, values, act, op_dtype, bias=None, o_bias=None):
    """compute_scores + second cvmm with reduction_weight (moe.py:397-416, smoe.py:240-248)."""
    B, N, D = x.shape
    K = idx.shape[-1]
    E = keys.shape[0]
    _, _, perm = bin_tokens(idx, E)
    ssel = idx.flatten()[perm]
    scores = cvmm_ref(x, ssel, perm // K, perm, keys, op_dtype).view(B, N, K, -1)
    if bias is not None:
        scores = scores + bias[idx.long()]
    scores = ACTS[act](scores)
    # second call (smoe.py:240-248): reduction_weight = weights, sel_index <- out_index, out_index <- None
    out = cvmm_ref(scores, ssel, perm, None, values, op_dtype, reduction_weight=weights)
    out = out.view(B, N, -1)
    if o_bias is not None:
        out = out + o_bias
    return out


def pretrain_dense_affinity(x, keys, values, act, k, x_dtype, op_dtype=None):
    """competition_policy_mlp_faster (pretrain competesmoe.py:381-414).  `op_dtype` = torch.bfloat16 restates the run under CUDA
    autocast (simple_task.py:295): both matmuls cast their operands to bf16 and return bf16; F.softplus is an fp32-policy op, so
    softplus, mean, the affinities, the top-k and the renormalised weights are fp32."""
    B, N, D = x.shape
    if op_dtype is not None:
        x, keys, values = x.to(op_dtype), keys.to(op_dtype), values.to(op_dtype)
    eo = torch.matmul(x.reshape(-1, D), keys)
    eo = ACTS[act](eo)
    eo = torch.matmul(eo, values).transpose(1, 0)        # [T,E,D]
    aff = torch.mean(F.softplus(eo.float() if op_dtype is not None else eo), dim=-1).view(B, N, -1)
    asm = F.softmax(aff, dim=-1, dtype=torch.float32)
    idx = topk_lowest_index(aff.detach(), k)[1]
    w = torch.gather(aff, -1, idx)
    w = w / torch.sum(w, dim=-1, keepdim=True).to(x_dtype)
    eo = eo.reshape(B, N, *eo.shape[1:])
    topk_out = torch.gather(eo, 2, idx.unsqueeze(-1).expand(B, N, k, eo.size(-1)))
    return w, idx, asm, aff, topk_out


def pretrain_diversity_loss(topk_out: torch.Tensor, op_dtype=None) -> torch.Tensor:
    """experts_diversity_loss of the pretrain stack (pretrain competesmoe.py:330-372): F.normalize -> bmm -> zeroed diagonal ->
    mean, WITHOUT the LLaVA stack's `.to(float32)`.  Under CUDA autocast (`op_dtype` = bf16): `norm` is an fp32-policy op, so the
    bf16 outputs are normalised in fp32; `bmm` is a lower-precision op, so the fp32 unit vectors are rounded to bf16 and the K x K
    products come back in bf16; the fp32 mask promotes them to fp32 for the mean."""
    eo = topk_out
    B, N, K, D = eo.shape
    if op_dtype is not None and eo.dtype == op_dtype and op_dtype != torch.float32:
        denom = eo.float().norm(2, -1, keepdim=True).clamp_min(1e-12).expand_as(eo)
        nrm = (eo / denom).view(B * N, K, D).to(op_dtype)
    else:
        nrm = F.normalize(eo, p=2, dim=-1).view(B * N, K, D)
    sim = torch.bmm(nrm, nrm.transpose(1, 2)) * (1 - torch.eye(K))
    return sim.mean()


def pretrain_deepseek_forward(x, w_gate, keys, values, keys_shared, values_shared, k, mode: str, op_dtype, x_dtype, forced_idx=None,
                              ffn=None):
    """DeepSeekV2.forward (moe_pretrain_model/layers/moe/deepseekv2.py:135-181: top-k of the logits, softmax over the K)
    and DeepSeekV3.forward (deepseekv3.py:142-190: top-k of sigmoid(logits), w / (sum + 1e-20)), both plus the always-on shared
    expert (a 1-expert cvmm with an all-zero selection and unit weight = a dense FFN).  Returns (out, gate_logits).
    `forced_idx`: evaluate with given (tie-broken) indices -- the reference's own, or the kernel's -- instead of the lowest-index rule.
    `ffn`: what computes the experts (default pretrain_ffn, the reference's two cvmm calls; oracle/mxfp8.py's for the fp8 build)."""
    ffn = ffn or pretrain_ffn
    B, N, D = x.shape
    xx = x.to(op_dtype)
    lg = F.linear(xx, w_gate.to(op_dtype))
    # bf16 logits = the run under CUDA autocast: softmax and sum are fp32-policy ops there, sigmoid is not
    if mode == "deepseekv2":
        idx = topk_lowest_index(lg.detach().float(), k)[1] if forced_idx is None else forced_idx
        w = F.softmax(torch.gather(lg, -1, idx).float(), dim=-1).to(x_dtype)
    else:
        sg = torch.sigmoid(lg)
        idx = topk_lowest_index(sg.detach().float(), k)[1] if forced_idx is None else forced_idx
        w = torch.gather(sg, -1, idx)
        w = w / (w.float().sum(dim=-1, keepdim=True) + 1e-20)
    out = ffn(x, idx, w, keys, values, "relu", op_dtype)
    zero = torch.zeros(B, N, 1, dtype=torch.long)
    one = torch.ones(B, N, 1)
    shared = ffn(x, zero, one, keys_shared, values_shared, "relu", op_dtype)
    return out + shared, lg


def pretrain_block_forward(x, ln_weight, ln_bias, eps, moe_forward):
    """The MoE half of RelativeMoeTransformerEncoderLayer.forward, preln (moe_pretrain_model/layers/transformer/
    relative_moe_transformer.py:153-161 with dropout 0): src + pkm(norm2(src)).  Under bf16 autocast on an fp32 stream LayerNorm
    stays an fp32 op, `moe_forward(xn)` (a closure over the pretrain_* functions above, which cast to the op dtype like the gate's
    F.linear and cvmm do) returns bf16, and the sum is fp32 by type promotion.  Returns (src_out, moe_out, xn)."""
    xn = F.layer_norm(x, (x.shape[-1],), ln_weight, ln_bias, eps)
    out = moe_forward(xn)
    return x + out, out, xn


# --------------------------------------------------------------------------------------------
# smoe_perturbed and the MoE attention projections (moe_pretrain_model/layers/moe/smoe_perturbed.py;
# layers/transformer/full_moe_relative_attention.py:267-300, 355-388) -- SURVEY.md section 8 (f4)
# --------------------------------------------------------------------------------------------
def perturbed_gate(x, expert_sel, embeddings_renormed, op_dtype=None, theta=0.1):
    """compute_gate (smoe_perturbed.py:148-160) AFTER the in-place, no-grad renormalisation of `expert_embeddings`
    (`emb *= 1.5 / (|emb| + theta)`), which the caller applies to the parameter first (`renorm_embeddings`).  `op_dtype` = bf16
    restates the CUDA-autocast run: F.linear and torch.matmul round their operands and results to bf16, `norm` is an fp32-policy
    op, `.type_as(mat1)` makes the logits bf16."""
    if op_dtype is not None:
        reduced = F.linear(x.to(op_dtype), expert_sel.to(op_dtype))
        m1 = reduced.float() / (reduced.float().norm(p=2, dim=-1, keepdim=True) + theta)
        logits = torch.matmul(m1.to(op_dtype), embeddings_renormed.float().transpose(0, 1).to(op_dtype)).type_as(reduced)
    else:
        reduced = F.linear(x, expert_sel)
        m1 = reduced.float() / (reduced.norm(p=2, dim=-1, keepdim=True) + theta)
        logits = torch.matmul(m1, embeddings_renormed.float().transpose(0, 1)).type_as(reduced)
    ok = logits.isfinite()
    return torch.where(ok, logits, torch.where(ok, logits, torch.full_like(logits, float("inf"))).min())


def renorm_embeddings(emb, theta=0.1):
    return emb * (1.5 / (emb.norm(p=2.0, dim=-1, keepdim=True) + theta))


def top_softmax(gate_softmax, k):
    """`_keepTopk` / att_forward :209-214: top-k of the softmax VALUES (indices without gradient), softmax over the k."""
    idx = topk_lowest_index(gate_softmax.detach(), k)[1]
    return torch.softmax(torch.gather(gate_softmax, -1, idx), dim=-1), idx


def attention_projection(x, expert_sel, embeddings_renormed, experts, heads, n_experts, k, op_dtype, temperature=0.3,
                         forced_index=None):
    """att_forward + compute_moe (smoe_perturbed.py:199-226): per head the top-k of softmax(gate / T) over that head's experts;
    out[b,n,h] = sum_k val[b,n,h,k] * x[b,n] @ experts[h * E + index[b,n,h,k]] as ONE cvmm with reduction weights [B,N,heads,k]."""
    B, N, _ = x.shape
    logits = perturbed_gate(x, expert_sel, embeddings_renormed, None if op_dtype == torch.float32 else op_dtype)
    logits = logits.view(B, N, heads, -1)
    sm = F.softmax((logits / temperature).float(), dim=-1).to(x.dtype)
    val, index = top_softmax(sm, k)
    if forced_index is not None:
        index = forced_index
        val = torch.softmax(torch.gather(sm, -1, index), dim=-1)
    gid = (torch.arange(heads) * n_experts).view(1, 1, heads, 1) + index            # global expert ids
    flat = gid.flatten(-2, -1)                                                      # [B, N, heads*k]
    _, _, perm = bin_tokens(flat, experts.shape[0])
    ssel = flat.flatten()[perm]
    out = cvmm_ref(x, ssel, perm // (heads * k), perm, experts, op_dtype, reduction_weight=val)
    return out, val, index, logits
