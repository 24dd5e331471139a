"""MoeLayer: base class of the LLaVA-stack MoE layers, HIP-backed.

Same constructor, attributes, state-dict names and method names as moe_model/model/moe/moe.py:8-245 so it is a
drop-in for `SiglipEncoderMoELayer` / `CLIPEncoderMoELayer` / `MLPMoE` (siglip_smoe.py:108-141, builder.py:56-71):
`experts` stays an nn.ModuleList of per-expert modules (weight upcycling writes through
`moelayer.experts[i].load_state_dict`, llava_arch.py:115-143), `gate` stays `nn.Linear(D, E, bias=False)`.

What changed underneath: `compute_moe` is one binning pass + dispatch + two grouped MFMA GEMMs + combine
(competesmoe_amd.functional.MoEFFNModules) instead of a Python loop of E x (torch.where + gather + 2 GEMM + index_put).
"""
from __future__ import annotations

import copy
import os
from typing import List, NamedTuple, Optional, Tuple

import torch
import torch.nn as nn

from .. import _lib as L
from ..functional import (CompetitionAffinity, ExpertTable, GateLogits, GateSelect, MoEFFNModules, MoEFFNModulesResidual,
                          MoEFFNModulesSlots, RouterSelect, DenseFFN, DiversityLoss,
                          RouterAux, SparseMoEModules)
from .. import ops


_ACT_PROBE = torch.tensor([-30.0, -8.0, -4.0, -1.5, -0.3, 0.0, 0.4, 1.1, 3.0, 8.0, 30.0])
_ACT_CACHE = {}


def _act_code(mod) -> int:
    """Map an activation module / callable to the kernel's activation code by what it COMPUTES (class names lie: transformers'
    ACT2FN has GELUActivation, PytorchGELUTanh, NewGELUActivation, FastGELUActivation, QuickGELUActivation, ... and an `nn.Tanh`
    must not be taken for a tanh-GELU): the callable is probed once on a few points and compared with the supported functions."""
    key = id(type(mod)) if isinstance(mod, nn.Module) else id(mod)
    hit = _ACT_CACHE.get((key, getattr(mod, "approximate", None)))
    if hit is not None:
        return hit
    import torch.nn.functional as F
    with torch.no_grad():
        out = mod(_ACT_PROBE.clone()).float()
    refs = ((L.ACT_RELU, F.relu(_ACT_PROBE)), (L.ACT_GELU, F.gelu(_ACT_PROBE)), (L.ACT_GELU_TANH, F.gelu(_ACT_PROBE, approximate="tanh")),
            (L.ACT_SILU, F.silu(_ACT_PROBE)), (L.ACT_QUICK_GELU, _ACT_PROBE * torch.sigmoid(1.702 * _ACT_PROBE)), (L.ACT_NONE, _ACT_PROBE))
    for code, ref in refs:
        if out.shape == ref.shape and torch.allclose(out, ref, atol=2e-6, rtol=0):
            _ACT_CACHE[(key, getattr(mod, "approximate", None))] = code
            return code
    raise NotImplementedError(f"competesmoe_amd: unsupported expert activation {mod!r} (supported: ReLU, GELU (erf / tanh), "
                              "SiLU, quick-GELU, identity)")


def parse_expert(expert: nn.Module) -> Tuple[nn.Linear, int, nn.Linear]:
    """(fc1, act code, fc2) of a two-matrix expert: `Sequential(Linear, act, Linear)` (moe.py:36-38, builder.py:61-65)
    or a module with fc1 / activation_fn / fc2 (SiglipMLP siglip_smoe.py:85-97, CLIPMLP clip_smoe.py:94-105)."""
    if isinstance(expert, nn.Sequential) and len(expert) == 3 and isinstance(expert[0], nn.Linear) and isinstance(expert[2], nn.Linear):
        return expert[0], _act_code(expert[1]), expert[2]
    if hasattr(expert, "fc1") and hasattr(expert, "fc2"):
        act = getattr(expert, "activation_fn", None) or getattr(expert, "act", None) or getattr(expert, "act_fn", None)
        if act is None:
            raise NotImplementedError("competesmoe_amd: expert has fc1/fc2 but no activation_fn/act attribute")
        return expert.fc1, _act_code(act), expert.fc2
    raise NotImplementedError(
        f"competesmoe_amd: expert module {type(expert).__name__} is not a Linear-act-Linear FFN; the HIP path supports "
        "nn.Sequential(Linear, act, Linear) and modules exposing fc1 / activation_fn / fc2")


class Route(NamedTuple):
    """One routing decision of the gate: x.dtype logits, fp32 softmax, int32 top-K indices, fp32 renormalised weights."""
    logits: torch.Tensor
    softmax: torch.Tensor
    idx: torch.Tensor
    w: torch.Tensor


class MoeLayer(nn.Module):

    def __init__(self, in_embed_dim=768, out_embed_dim=768, num_of_experts=4, num_selected=2, expert=None, args=None):
        super().__init__()
        self.in_embed_dim = in_embed_dim
        self.out_embed_dim = out_embed_dim
        self.num_of_experts = num_of_experts
        self.num_selected = num_selected
        self.aux_loss = {"zloss": self.zloss, "balanceloss": self.balanceloss}
        if expert is None:
            self.experts = nn.ModuleList([
                nn.Sequential(nn.Linear(in_embed_dim, out_embed_dim), nn.GELU(), nn.Linear(out_embed_dim, out_embed_dim))
                for _ in range(num_of_experts)])
        elif isinstance(expert, nn.ModuleList):
            self.experts = expert
        else:
            self.experts = nn.ModuleList([copy.deepcopy(expert) for _ in range(self.num_of_experts)])
        self.gate = nn.Linear(in_embed_dim, num_of_experts, bias=False)
        self.args = args
        self.is_vision = False
        self.log_metrics = {}
        self._table_key = None
        self._table = None

    # ------------------------------------------------------------------ init (moe.py:50-70)
    def init_gate_weights(self, std=0.02):
        if getattr(self.args, "init_weight", True) is False:
            return
        device = self.gate.weight.device if self.gate.weight.device != torch.device("meta") else torch.device("cpu")
        gen = torch.Generator(device=device)
        gen.manual_seed(42)
        nn.init.normal_(self.gate.weight, mean=0.0, std=std, generator=gen)
        if self.gate.bias is not None:
            nn.init.constant_(self.gate.bias, 0.0)

    # ------------------------------------------------------------------ aux losses (tiny [T,E] math; moe.py:71-110, 214-226)
    def zloss(self, gate_logits, gate_softmax=None):
        return torch.square(torch.logsumexp(gate_logits, dim=-1)).mean()

    def balanceloss(self, selected_experts, gate_softmax):
        if self._fusable(selected_experts, gate_softmax):
            return RouterAux.apply(None, gate_softmax, selected_experts)[0]
        E = gate_softmax.shape[-1]
        density_1_proxy = gate_softmax.mean(dim=-2)
        # F.one_hot range-checks its input with .item() (a device sync that drains the launch queue): compare instead
        top1 = selected_experts[..., 0].long().unsqueeze(-1)
        one_hot = (top1 == ops.cached_arange(E, top1.device)).float()
        density_1 = one_hot.mean(dim=-2)
        return (density_1_proxy * density_1).mean() * float(E ** 2)

    @staticmethod
    def _fusable(selected_experts, gate_softmax, gate_logits=None) -> bool:
        """The two-launch kernel path (functional.RouterAux) takes what `topk_expert` returns: [B,N,E] fp32 softmax, int32 indices."""
        return (gate_softmax.is_cuda and gate_softmax.dim() == 3 and gate_softmax.dtype == torch.float32 and gate_softmax.numel() > 0
                and gate_softmax.shape[-1] <= 1024 and selected_experts.dtype == torch.int32
                and selected_experts.shape[:-1] == gate_softmax.shape[:-1]
                and (gate_logits is None or (gate_logits.shape == gate_softmax.shape
                                             and gate_logits.dtype in (torch.float32, torch.bfloat16))))

    def combine_loss(self, selected_experts, gate_softmax, gate_logits, acitve_zloss=True):
        if self._fusable(selected_experts, gate_softmax, gate_logits if acitve_zloss else None):
            balance_loss, router_z_loss = RouterAux.apply(gate_logits if acitve_zloss else None, gate_softmax, selected_experts)
            if acitve_zloss:
                return (balance_loss * self.args.balance_loss_coef + router_z_loss * self.args.router_z_loss_coef, balance_loss,
                        router_z_loss)
            return balance_loss * self.args.balance_loss_coef, balance_loss, router_z_loss
        balance_loss = self.balanceloss(selected_experts=selected_experts, gate_softmax=gate_softmax)
        router_z_loss = gate_softmax.new_zeros(())
        if acitve_zloss:
            router_z_loss = self.zloss(gate_logits, gate_softmax)
            auxiliary_loss = balance_loss * self.args.balance_loss_coef + router_z_loss * self.args.router_z_loss_coef
        else:
            auxiliary_loss = balance_loss * self.args.balance_loss_coef
        return auxiliary_loss, balance_loss, router_z_loss

    def experts_diversity_loss(self, expert_outputs):
        """moe.py:133-171 / competesmoe.py:180-218."""
        if expert_outputs.shape[-2] <= 8:          # one kernel instead of normalize + 32k tiny batched matmuls
            return DiversityLoss.apply(expert_outputs)
        eo = expert_outputs.to(torch.float32)
        B, N, K, D = eo.shape
        nrm = nn.functional.normalize(eo, p=2, dim=-1).view(B * N, K, D)
        sim = torch.bmm(nrm, nrm.transpose(1, 2))
        sim = sim * (1 - torch.eye(K, device=eo.device))
        return sim.mean()

    # ------------------------------------------------------------------ router (moe.py:113-132; smoe.py:42-44)
    def gate_logits(self, x):
        B, N, D = x.shape
        pre = getattr(self, "_pre_logits", None)
        if pre is not None:                       # computed with the LayerNorm by the block around the layer (moe/block.py)
            self._pre_logits = None
            return pre.view(B, N, self.gate.weight.shape[0])
        return GateLogits.apply(x.reshape(B * N, D), self.gate.weight).view(B, N, self.gate.weight.shape[0])

    def topk_expert(self, gate_logits, num_selected=None):
        """Returns (weights fp32 RENORMALISED, selected_experts int32, gate_softmax fp32).  The reference returns the
        un-normalised top-k values and divides at every call site (smoe.py:44, competesmoe.py:318); the kernel fuses
        the division (denominator rounded to the activation dtype first, as `.to(x.dtype)` does)."""
        K = num_selected or self.num_selected
        shp = gate_logits.shape
        sm, idx, w = RouterSelect.apply(gate_logits.reshape(-1, shp[-1]), K, L.SEL_SOFTMAX,
                                        gate_logits.dtype == torch.bfloat16)
        return w.view(*shp[:-1], K), idx.view(*shp[:-1], K), sm.view(shp)

    # ------------------------------------------------------------------ helpers shared by the sparse layers
    def _route(self, x) -> "Route":
        """Gate projection + softmax / top-K / renormalisation (smoe.py:42-44) as one record."""
        B, N, D = x.shape
        x2 = x.reshape(B * N, D)
        K, E = self.num_selected, self.gate.weight.shape[0]
        if getattr(self, "_pre_logits", None) is None and ops.gate_select_ok(x2, self.gate.weight, K):
            # one launch that reads x once: logits, softmax, top-K and the binning histogram (csmoe_gate_select)
            logits, sm, idx, w = GateSelect.apply(x2, self.gate.weight, K, L.SEL_SOFTMAX, x.dtype == torch.bfloat16)
            return Route(logits.view(B, N, E), sm.view(B, N, E), idx.view(B, N, K), w.view(B, N, K))
        logits = self.gate_logits(x)
        w, idx, sm = self.topk_expert(gate_logits=logits)
        return Route(logits, sm, idx, w)

    def _route_and_compute(self, x):
        """`_route` + `compute_moe` of a sparse step.  Where the one-pass router takes the shape and nothing else is hooked in (no
        logits / residual handed over by the block around the layer) both run as ONE autograd node (SparseMoEModules): same bits,
        and the gate-path and expert-path gradients of x meet inside the backward's gather-sum instead of in an autograd add."""
        B, N, D = x.shape
        K, E = self.num_selected, self.gate.weight.shape[0]
        x2 = x.reshape(B * N, D)
        if (os.environ.get("CSMOE_FUSED_STEP", "1") != "0" and getattr(self, "_pre_logits", None) is None
                and getattr(self, "_residual", None) is None and E == len(self.experts) and ops.gate_select_ok(x2, self.gate.weight, K)):
            tab, params = self._expert_table(E, x.dtype, x.device)
            out, logits, sm, idx, w = SparseMoEModules.apply(x2, self.gate.weight, K, L.SEL_SOFTMAX, x.dtype == torch.bfloat16, tab,
                                                             L.COMBINE_SEQ, *params)
            return Route(logits.view(B, N, E), sm.view(B, N, E), idx.view(B, N, K), w.view(B, N, K)), out.view(B, N, out.shape[-1])
        route = self._route(x)
        return route, self.compute_moe(route.idx, route.w, None, x)

    def _router_aux(self, route: "Route", wanted: bool, like: torch.Tensor, keep_metrics: bool = False):
        """(auxiliary loss, infor_aux) of a sparse step: balance + z-loss when `wanted` (smoe.py:51-62), else a zero and {}.
        `keep_metrics` stores the tensors the reference keeps in `log_metrics` for analysis -- as tensors, without its two
        `.item()` host syncs."""
        if not wanted:
            return like.new_zeros(()), {}       # a fill kernel: torch.tensor(0.0, device=...) would be a blocking H2D copy
        aux, bal, z = self.combine_loss(route.idx, route.softmax, route.logits)
        infor = {"balance_loss": bal.clone().detach(), "router_z_loss": z.clone().detach()}
        if keep_metrics:
            # detached: a tensor with a grad_fn kept on the module keeps the WHOLE autograd graph of this step alive until the next
            # forward -- its saved activations, and the parameters' AccumulateGrad nodes, which then carry a stale stream into a
            # later hipGraph capture (the cause of round 1's capture crash, competesmoe_amd/graphs.py)
            self.log_metrics.update(weights=route.w.detach(), balance_loss=infor["balance_loss"], router_z_loss=infor["router_z_loss"],
                                    gate_softmax=route.softmax.detach(), selected_experts=route.idx)
        return aux, infor

    # ------------------------------------------------------------------ expert pointer table
    def _expert_table(self, n_experts: int, dtype, device) -> Tuple[ExpertTable, List[torch.Tensor]]:
        parsed = [parse_expert(self.experts[i]) for i in range(n_experts)]
        acts = {p[1] for p in parsed}
        if len(acts) != 1:
            raise NotImplementedError("competesmoe_amd: experts with different activations in one layer")
        w1 = [p[0].weight for p in parsed]
        w2 = [p[2].weight for p in parsed]
        b1 = [p[0].bias for p in parsed]
        b2 = [p[2].bias for p in parsed]
        has_b1 = all(b is not None for b in b1)
        has_b2 = all(b is not None for b in b2)
        if (not has_b1 and any(b is not None for b in b1)) or (not has_b2 and any(b is not None for b in b2)):
            raise NotImplementedError("competesmoe_amd: experts must all have (or all lack) a bias")
        if any(w.shape != w1[0].shape for w in w1) or any(w.shape != w2[0].shape for w in w2) or w2[0].shape[1] != w1[0].shape[0]:
            raise NotImplementedError("competesmoe_amd: all experts of a layer must have the same Linear shapes (fc1 [F,D], fc2 [Dout,F])")
        params = w1 + (b1 if has_b1 else []) + w2 + (b2 if has_b2 else [])
        for p in params:
            if p.dtype != dtype or p.device != device or not p.is_contiguous():
                raise ValueError(f"competesmoe_amd: expert parameters must be contiguous {dtype} tensors on {device} "
                                 f"(got {p.dtype} on {p.device}); call layer.to(x.dtype)")
        key = tuple(p.data_ptr() for p in params)
        if key != self._table_key:
            F_, D_ = w1[0].shape
            Dout = w2[0].shape[0]
            self._table = ExpertTable(
                E=n_experts, D=D_, F=F_, Dout=Dout, layout=L.B_NK, act=acts.pop(),
                w1_ptrs=ops.ptr_array(w1, device), w2_ptrs=ops.ptr_array(w2, device),
                b1_ptrs=ops.ptr_array(b1, device) if has_b1 else None,
                b2_ptrs=ops.ptr_array(b2, device) if has_b2 else None, param_dtype=dtype)
            self._table_key = key
        return self._table, params

    # ------------------------------------------------------------------ dispatch + FFN + combine (moe.py:172-213)
    def compute_moe(self, selected_experts, weights, results, x, expert_outputs=None, return_topk_outputs=False,
                    n_experts: Optional[int] = None, weights_rounded: bool = False):
        """out[b,t] = sum_k w[b,t,k] * expert_{idx[b,t,k]}(x[b,t]) with the reference's accumulation order/rounding.
        `results` is accepted for signature compatibility (the reference accumulates into it; callers pass zeros)."""
        B, N, D = x.shape
        n_experts = n_experts or len(self.experts)
        tab, params = self._expert_table(n_experts, x.dtype, x.device)
        mode = L.COMBINE_SEQ_RW if weights_rounded else L.COMBINE_SEQ
        idx2 = selected_experts.reshape(B * N, selected_experts.shape[-1])
        if idx2.dtype != torch.int32:
            idx2 = idx2.int()
        w2 = weights.reshape(B * N, weights.shape[-1])
        if w2.dtype != torch.float32:
            w2 = w2.float()
        res = getattr(self, "_residual", None)
        if res is not None and res.shape[0] == B * N and res.shape[1] == tab.Dout:
            # the block's residual, added in the combine epilogue (moe/block.py); consumed once
            self._residual = None
            out = MoEFFNModulesResidual.apply(x.reshape(B * N, D), res, w2.contiguous(), idx2.contiguous(), tab, mode, *params)
        else:
            out = MoEFFNModules.apply(x.reshape(B * N, D), w2.contiguous(), idx2.contiguous(), tab, mode, *params)
        return out.view(B, N, out.shape[-1])

    def compute_moe_slots(self, selected_experts, weights, x, weights_rounded: bool = False):
        """compute_moe plus the per-slot expert outputs [B, N, K, Dout] of the same pass (MoEFFNModulesSlots)."""
        B, N, D = x.shape
        tab, params = self._expert_table(len(self.experts), x.dtype, x.device)
        mode = L.COMBINE_SEQ_RW if weights_rounded else L.COMBINE_SEQ
        K = selected_experts.shape[-1]
        idx2 = selected_experts.reshape(B * N, K).int().contiguous()
        w2 = weights.reshape(B * N, K).float().contiguous()
        out, y_tk = MoEFFNModulesSlots.apply(x.reshape(B * N, D), w2, idx2, tab, mode, *params)
        return out.view(B, N, out.shape[-1]), y_tk.view(B, N, K, y_tk.shape[-1])

    def dense_affinities(self, x, fp32_affinity: bool = False):
        """[B*N, E] affinities of the dense competition pass without its outputs (CompetitionAffinity), or None when the shapes
        are not ones its kernels take."""
        B, N, D = x.shape
        x2 = x.reshape(B * N, D)
        parsed = [parse_expert(m) for m in self.experts]
        fc1s, acts, fc2s = zip(*parsed)
        if len(set(acts)) != 1 or not ops.affinity_ok(x2, fc1s[0].weight.shape[0], fc2s[0].weight.shape[0]):
            return None
        has_b1 = all(f.bias is not None for f in fc1s)
        has_b2 = all(f.bias is not None for f in fc2s)
        if (not has_b1 and any(f.bias is not None for f in fc1s)) or (not has_b2 and any(f.bias is not None for f in fc2s)):
            return None
        params = [f.weight for f in fc1s] + ([f.bias for f in fc1s] if has_b1 else []) + [f.weight for f in fc2s] \
            + ([f.bias for f in fc2s] if has_b2 else [])
        return CompetitionAffinity.apply(x2, len(self.experts), acts[0], L.B_NK, has_b1, has_b2, fp32_affinity, *params)

    def dense_expert(self, i: int, x):
        """experts[i](x) over all tokens (shared expert / competition pass)."""
        fc1, act, fc2 = parse_expert(self.experts[i])
        B, N, D = x.shape
        y = DenseFFN.apply(x.reshape(B * N, D), fc1.weight, fc1.bias, fc2.weight, fc2.bias, act, L.B_NK)
        return y.view(B, N, y.shape[-1])

    def forward(self, x, return_id_experts=False):
        gate_logits = self.gate_logits(x)
        weights, selected_experts, gate_softmax = self.topk_expert(gate_logits=gate_logits)
        output = self.compute_moe(selected_experts, weights, None, x)
        auxiliary_loss, balance_loss, router_z_loss = self.combine_loss(selected_experts, gate_softmax, gate_logits)
        infor_aux = {"balance_loss": balance_loss.clone().detach(), "router_z_loss": router_z_loss.clone().detach()}
        if return_id_experts:
            return output, auxiliary_loss, gate_softmax
        return output, auxiliary_loss, None, infor_aux
