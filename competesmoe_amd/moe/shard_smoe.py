"""Shared-expert variants: E-1 routed experts (top K-1) + the last expert always on.
`smoe_share` mixes 0.5/0.5 (moe_model/model/moe/shard_smoe.py:12-67); `deepseekv3` adds them and always returns the
aux loss (moe_model/model/moe/deepseekv3.py:12-56 -- not imported by the reference's __init__, registered here)."""
import copy

import torch.nn as nn

from .register import register_moe
from .moe import MoeLayer


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
        gate_logits = self.gate_logits(x)
        weights, selected_experts, gate_softmax = self.topk_expert(gate_logits=gate_logits)
        routed = self.compute_moe(selected_experts, weights, None, x, n_experts=self.num_of_experts)
        shared = self.dense_expert(self.num_of_experts, x)
        return routed, shared, selected_experts, gate_softmax, gate_logits


@register_moe("smoe_share")
class MoEShareLayer(_SharedBase):
    def forward(self, x, return_id_experts=False, is_vision=False):
        routed, shared, selected_experts, gate_softmax, gate_logits = self._routed_and_shared(x)
        output = shared * 0.5 + routed * 0.5
        auxiliary_loss = x.new_zeros(())        # a fill kernel: torch.tensor(0.0, device=...) is a blocking H2D copy
        infor_aux = {}
        if x.requires_grad:
            auxiliary_loss, balance_loss, router_z_loss = self.combine_loss(selected_experts, gate_softmax, gate_logits)
            infor_aux = {"balance_loss": balance_loss.clone().detach(), "router_z_loss": router_z_loss.clone().detach()}
        return output, auxiliary_loss, None, infor_aux


@register_moe("deepseekv3")
class DeepSeekV3ShareLayer(_SharedBase):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.routed_scaling_factor = 2.5   # declared, unused (deepseekv3.py:21)

    def forward(self, x, return_id_experts=False, is_vision=False):
        routed, shared, selected_experts, gate_softmax, gate_logits = self._routed_and_shared(x)
        output = shared + routed
        auxiliary_loss, balance_loss, router_z_loss = self.combine_loss(selected_experts, gate_softmax, gate_logits)
        infor_aux = {"balance_loss": balance_loss.clone().detach(), "router_z_loss": router_z_loss.clone().detach()}
        if return_id_experts:
            return output, auxiliary_loss, gate_softmax
        return output, auxiliary_loss, None, infor_aux
