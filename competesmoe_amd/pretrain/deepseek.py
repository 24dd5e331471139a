"""Pretrain shared-expert variants: `deepseekv2` (top-k of the logits, softmax over the K;
moe_pretrain_model/layers/moe/deepseekv2.py:38-181) and `deepseekv3` (top-k of sigmoid, w/(sum+1e-20);
deepseekv3.py:38-190).  Both add one always-on shared expert `keys_shared [1,D,F*n_shared]`, `values_shared [1,F*n_shared,D]`
(n_shared_experts hard-coded 1 upstream)."""
import torch
import torch.nn as nn

from .. import _lib as L
from .moe import MoE
from .register import register_moe


class _SharedBase(MoE):
    SEL = L.SEL_TOPK_SOFTMAX

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        ws = kw.get("weight_scale", 1.0)
        # upstream hard-codes one shared expert (deepseekv2.py:97); `args.n_shared_experts` widens it to n x F (BASELINE config 5:
        # 128 routed + 2 shared), which is how its own comment describes more than one
        self.n_shared_experts = int(getattr(self.args, "n_shared_experts", 1))
        fs = self.expert_size * self.n_shared_experts
        self.values_shared = nn.Parameter(torch.empty(1, fs, self.v_dim))
        nn.init.normal_(self.values_shared, std=fs ** -0.5 * ws)
        self._extra_init()
        self.keys_shared = nn.Parameter(torch.empty(1, self.k_vec_dim, fs))
        nn.init.normal_(self.keys_shared, std=self.k_dim ** -0.5 * ws)
        self.bias_shared = nn.Parameter(torch.zeros(1, fs)) if self.bias is not None else None

    def _extra_init(self):
        pass

    def forward(self, x, return_id_experts=False, return_full=True, *args, **kwargs):
        gate_logits, weights, selected_experts, gate_softmax = self.gate_and_select(x, self.SEL, x.dtype)
        out = self.ffn(x, selected_experts, weights)
        out = out + self.shared_ffn(x, self.keys_shared, self.values_shared, self.bias_shared)
        bal = self.entropy_balance(gate_logits) * (self.args.balance_loss_coef / self.div)
        self.add_reg(lambda: bal, f"{self.name_moe}_ebalance")
        self._test_stats(selected_experts, weights, gate_softmax)
        return self._finish(out, x)


@register_moe("deepseekv2")
class DeepSeekV2(_SharedBase):
    SEL = L.SEL_TOPK_SOFTMAX


@register_moe("deepseekv3")
class DeepSeekV3(_SharedBase):
    SEL = L.SEL_SIGMOID

    def _extra_init(self):
        # declared, unused upstream (deepseekv3.py:105-109); kept so checkpoints load
        self.e_score_correction_bias = nn.Parameter(torch.zeros(self.n_experts))
        self.n_group, self.topk_group, self.routed_scaling_factor = 8, 4, 1
