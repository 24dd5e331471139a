"""Pretrain `smoe` (moe_pretrain_model/layers/moe/smoe.py:38-263)."""
from .. import _lib as L
from .moe import MoE
from .register import register_moe


@register_moe("smoe")
class SMoeLayer(MoE):
    _fuses_residual = True

    def forward(self, x, return_id_experts=False, return_full=True, *args, **kwargs):
        gate_logits, weights, selected_experts, gate_softmax = self.gate_and_select(x, L.SEL_SOFTMAX, x.dtype)
        out = self.ffn(x, selected_experts, weights)
        bal = self.entropy_balance(gate_logits) * (self.args.balance_loss_coef / self.div)
        self.add_reg(lambda: bal, f"{self.name_moe}_ebalance")
        self._test_stats(selected_experts, weights, gate_softmax)
        return self._finish(out, x)
