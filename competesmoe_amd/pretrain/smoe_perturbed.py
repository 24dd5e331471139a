"""Pretrain `smoe_perturbed` (moe_pretrain_model/layers/moe/smoe_perturbed.py:38-226): cosine gating against learned expert
embeddings in a reduced space, and -- the reason it is built here -- the only upstream layer that implements the MoE ATTENTION
projections (`att_forward` / `compute_moe`, :199-226) which `FullMoeRelativeAttentionCore` calls for its q / k / v / o maps
(layers/transformer/full_moe_relative_attention.py:267-300, 355-388): per head, a top-K mixture of expert projection matrices
`experts[h * E + e]` of shape [inp_expert, out_expert], evaluated by ONE `cvmm` with reduction weights [B, N, heads, K].

Gate math ([T, E/2]- and [T, E]-sized) stays torch ops like upstream; selection is the router kernel (RouterSelect,
SEL_TOPK_SOFTMAX: top-K of the softmax values, softmax over the K); the expert products are the grouped GEMM behind
`pretrain.cvmm`."""
from collections import namedtuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from ..functional import RouterSelect
from .cvmm import cvmm, cvmm_prepare_sel2
from .moe import MoE
from .register import register_moe

Selection = namedtuple("Selection", ["raw_sel", "sel_val", "raw_sel_index", "sel_index"])


@register_moe("smoe_perturbed")
class MoEPerturbedCosingGating(MoE):
    _fuses_residual = True

    def __init__(self, *a, std=1, **kw):
        sel_bias = kw.get("sel_bias", False)
        super().__init__(*a, **kw)
        self.reduction_dim = int(self.n_experts / 2)
        # upstream leaves `expert_embeddings` uninitialised (torch.empty, :100-104) and renormalises it in every forward
        self.expert_embeddings = nn.Parameter(torch.empty(self.num_of_experts, self.reduction_dim))
        self.temperature = 0.3
        self.bias = None
        self.expert_sel = nn.Parameter(torch.empty(self.reduction_dim, self.k_vec_dim))
        self.sel_bias = nn.Parameter(torch.zeros(self.reduction_dim)) if sel_bias else None
        nn.init.normal_(self.expert_sel, std=self.k_vec_dim ** -0.5 * self.sel_weight_scale)
        self.theta = 0.1
        self.total_selections, self.total_gate_softmax, self.total_gate_logits = [], [], []

    def _plain_gate(self) -> bool:
        return False

    # ------------------------------------------------------------------ gate (:148-160)
    def inp_reduction(self, x):
        return F.linear(x, self.expert_sel, self.sel_bias)

    def _cosine(self, mat1, mat2):
        m1 = mat1.float() / (mat1.norm(p=2, dim=-1, keepdim=True) + self.theta)
        return torch.matmul(m1, mat2.float().transpose(0, 1)).type_as(mat1)

    def _make_finite(self, scores):
        ok = scores.isfinite()          # upstream branches on ok.all() (a host sync) and assigns in place; same values
        low = torch.where(ok, scores, torch.full_like(scores, float("inf"))).min()
        return torch.where(ok, scores, low)

    def compute_gate(self, x):
        reduced = self.inp_reduction(x)
        with torch.no_grad():           # the parameter is renormalised IN PLACE on every call, as upstream
            nrm = self.expert_embeddings.norm(p=2.0, dim=-1, keepdim=True)
            self.expert_embeddings.mul_(1.5 / (nrm + self.theta))
        return self._make_finite(self._cosine(reduced, self.expert_embeddings))

    def _top_softmax(self, gate_softmax):
        """top-K of the softmax values, softmax over the K (`_keepTopk`, :119-122; att_forward :209-214)."""
        shp = gate_softmax.shape
        _, idx, w = RouterSelect.apply(gate_softmax.reshape(-1, shp[-1]), self.num_selected, L.SEL_TOPK_SOFTMAX, False)
        K = self.num_selected
        return w.view(*shp[:-1], K), idx.view(*shp[:-1], K)

    # ------------------------------------------------------------------ FFN form (:162-197)
    def forward(self, x, return_id_experts=False, return_full=True, *args, **kwargs):
        gate_logits = self.compute_gate(x)
        gate_softmax = F.softmax(gate_logits / self.temperature, dim=-1, dtype=torch.float).to(x.dtype)
        weights, selected_experts = self._top_softmax(gate_softmax)
        out = self.ffn(x, selected_experts, weights)
        bal = self.entropy_balance(gate_logits) * (self.args.balance_loss_coef / self.div)
        self.add_reg(lambda: bal, f"{self.name_moe}_ebalance")
        return self._finish(out, x)

    # ------------------------------------------------------------------ attention projections (:199-226)
    def pre_train_forward(self):
        self.total_selections, self.total_gate_softmax, self.total_gate_logits = [], [], []

    def update_aux_statistics(self, gate_logits, gate_softmax, selected_experts):
        self.total_selections.append(selected_experts)
        self.total_gate_logits.append(gate_logits)
        self.total_gate_softmax.append(gate_softmax)

    def before_loss(self):
        self.pre_train_forward()        # add_perplexity_reg upstream only resets the history (moe.py:340-358)
        if self.training:
            self.iter += 1

    def att_forward(self, x, n_experts, n_copies, return_full=True, *args, **kwargs):
        """x [B, N, dmodel] -> Selection: per head (n_copies) the top-K of softmax(gate / T) over that head's `n_experts`, the K
        values re-normalised by a softmax, and the cvmm selection over the GLOBAL expert ids h * n_experts + e."""
        if self.selection_dropout > 0 and self.training:
            x = F.dropout(x, self.selection_dropout)
        gate_logits = self.compute_gate(x)
        gate_logits = gate_logits.view(*gate_logits.shape[:-1], n_copies, -1)
        gate_softmax = F.softmax(gate_logits / self.temperature, dim=-1, dtype=torch.float).to(x.dtype)
        val, index = self._top_softmax(gate_softmax)                              # [B, N, heads, K]
        shift = (torch.arange(n_copies, device=index.device, dtype=index.dtype) * n_experts).unsqueeze(-1)
        sel_pp = cvmm_prepare_sel2((shift + index).flatten(-2, -1).int(), val, n_experts=self.n_experts)
        if self.training:
            self.update_aux_statistics(gate_logits=gate_logits, gate_softmax=gate_softmax, selected_experts=index)
        return Selection(gate_logits, val, index, sel_pp)

    def compute_moe(self, x, sel: Selection):
        """[B, N, inp_expert] -> [B, N, heads, out_expert]: sum_k val[b,n,h,k] * x[b,n] @ experts[h * E + index[b,n,h,k]]."""
        return cvmm(x, sel.sel_index, self.experts)
