"""`smoe`: softmax -> top-K -> renormalise router + sparse FFN (moe_model/model/moe/smoe.py:11-64)."""

from .register import register_moe
from .moe import MoeLayer


@register_moe("smoe")
class SMoeLayer(MoeLayer):
    _fuses_residual = True        # output = one combine: a block around the layer may add its residual there (moe/block.py)

    def __init__(self, in_embed_dim=768, out_embed_dim=768, num_of_experts=4, num_selected=2, expert=None, args=None):
        super().__init__(in_embed_dim, out_embed_dim, num_of_experts, num_selected, expert, args)
        self.log_metrics = {}
        self.is_vision = False
        self.init_gate_weights()

    def forward(self, x, return_id_experts=False, is_vision=False):
        self.is_vision = is_vision
        gate_logits = self.gate_logits(x)
        weights, selected_experts, gate_softmax = self.topk_expert(gate_logits=gate_logits)
        output = self.compute_moe(selected_experts, weights, None, x)
        auxiliary_loss = x.new_zeros(())        # a fill kernel: torch.tensor(0.0, device=...) is a blocking H2D copy
        infor_aux = {}
        if x.requires_grad or return_id_experts:
            auxiliary_loss, balance_loss, router_z_loss = self.combine_loss(selected_experts, gate_softmax, gate_logits)
            infor_aux = {"balance_loss": balance_loss.clone().detach(), "router_z_loss": router_z_loss.clone().detach()}
            # kept for analysis like the reference (smoe.py:58-62) but WITHOUT the .item() host syncs
            self.log_metrics["weights"] = weights
            self.log_metrics["balance_loss"] = infor_aux["balance_loss"]
            self.log_metrics["router_z_loss"] = infor_aux["router_z_loss"]
            self.log_metrics["gate_softmax"] = gate_softmax
            self.log_metrics["selected_experts"] = selected_experts
        return output, auxiliary_loss, None, infor_aux
