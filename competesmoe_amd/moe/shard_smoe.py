"""Shared-expert variants: E-1 routed experts (top K-1) + the last expert always on.
`smoe_share` mixes 0.5/0.5 (moe_model/model/moe/shard_smoe.py:12-67); `deepseekv3` adds them and always returns the
aux loss (moe_model/model/moe/deepseekv3.py:12-56 -- not imported by the reference's __init__, registered here)."""
import copy

import torch.nn as nn

import torch

from .register import register_moe
from .moe import MoeLayer
from ..functional import DxHandoff, set_handoff


class _SharedBase(MoeLayer):
    def __init__(self, in_embed_dim=768, out_embed_dim=768, num_of_experts=4, num_selected=2, expert=None, args=None):
        # the reference calls MoeLayer.__init__() with defaults and then overwrites everything (shard_smoe.py:15-34)
        super().__init__(in_embed_dim, out_embed_dim, num_of_experts, num_selected, None, args)
        if expert is None:
            self.experts = nn.ModuleList([
                nn.Sequential(nn.Linear(in_embed_dim, out_embed_dim), nn.GELU(), nn.Linear(out_embed_dim, out_embed_dim))
                for _ in range(num_of_experts)])
        else:
            self.experts = nn.ModuleList([copy.deepcopy(expert) for _ in range(num_of_experts)])
        self.num_selected, self.num_of_experts = num_selected - 1, num_of_experts - 1
        self.gate = nn.Linear(in_embed_dim, self.num_of_experts, bias=False)
        self.init_gate_weights()

    def _routed_and_shared(self, x):
        """(routed output over the E-1 gated experts, dense output of the always-on last expert, the routing record)."""
        route = self._route(x)
        # bf16 training: the always-on expert's dx joins the routed experts' gather-sum as its first addend, which is the order the
        # reference's autograd accumulates x's gradient streams in (functional.DxHandoff)
        h = DxHandoff() if (x.requires_grad and torch.is_grad_enabled() and x.dtype != torch.float32) else None
        try:
            set_handoff(h)
            routed = self.compute_moe(route.idx, route.w, None, x, n_experts=self.num_of_experts)
            set_handoff(h)
            shared = self.dense_expert(self.num_of_experts, x)
        finally:
            set_handoff(None)
        return routed, shared, route


@register_moe("smoe_share")
class MoEShareLayer(_SharedBase):
    def forward(self, x, return_id_experts=False, is_vision=False):
        routed, shared, route = self._routed_and_shared(x)
        aux, infor_aux = self._router_aux(route, x.requires_grad, x)
        return shared * 0.5 + routed * 0.5, aux, None, infor_aux


@register_moe("deepseekv3")
class DeepSeekV3ShareLayer(_SharedBase):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.routed_scaling_factor = 2.5   # declared, unused (deepseekv3.py:21)

    def forward(self, x, return_id_experts=False, is_vision=False):
        routed, shared, route = self._routed_and_shared(x)
        aux, infor_aux = self._router_aux(route, True, x)          # this variant computes the losses unconditionally
        if return_id_experts:
            return shared + routed, aux, route.softmax
        return shared + routed, aux, None, infor_aux
