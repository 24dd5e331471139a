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
        """(output, auxiliary_loss, None, infor_aux) -- smoe.py:39-64."""
        self.is_vision = is_vision
        route, output = self._route_and_compute(x)
        aux, infor_aux = self._router_aux(route, x.requires_grad or return_id_experts, x, keep_metrics=True)
        return output, aux, None, infor_aux
