"""Pretrain-stack MoE base class on the HIP path.

Drop-in for `moe_pretrain_model/layers/moe/moe.py:35-454` as constructed by
`RelativeMoeTransformerEncoderLayer` (`layers/transformer/relative_moe_transformer.py:82-95`): same constructor keywords,
same parameter names / shapes / initialisers (`w_gate [E,D]`, `keys [E,D,F]`, `values [E,F,D]`, optional `bias [E,F]`,
`o_bias [D]`), `forward(x, id_layer=...) -> out`, aux losses through `RegularizedLayer.add_reg`, `num_selected = n_heads`
(:128).  The two `cvmm` calls (compute_scores :397-416 and the reduction-weight call :427-435) are ONE fused pipeline here:
bin -> dispatch -> grouped GEMM(+bias, act) -> grouped GEMM -> weighted combine (functional.MoEFFNPacked).
The MoE-attention branch (`is_att=True`, only used by `smoe_perturbed` upstream) is out of scope."""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from .. import ops
from ..functional import (DenseFFN, DenseFFNFP8, GateLogits, GateSelect, MoEFFNPacked, MoEFFNPackedFP8, MoEFFNPackedSlots, OperandFork,
                          RouterSelect)
from .framework_layers import LoggingLayer, OncePerIterLayer, RegularizedLayer


def _act_code(fn) -> int:
    if fn is None:
        return L.ACT_RELU
    for name, code in (("relu", L.ACT_RELU), ("gelu", L.ACT_GELU), ("silu", L.ACT_SILU)):
        if fn is getattr(F, name) or getattr(fn, "__name__", "") == name:
            return code
    # the reference default is `lambda x: F.relu(x, inplace=True)` (moe.py:48): probe the callable once
    probe = torch.tensor([-1.0, 0.5, 2.0])
    out = fn(probe.clone())
    for code, ref in ((L.ACT_RELU, F.relu(probe)), (L.ACT_GELU, F.gelu(probe)), (L.ACT_SILU, F.silu(probe)),
                      (L.ACT_GELU_TANH, F.gelu(probe, approximate="tanh")), (L.ACT_NONE, probe)):
        if torch.allclose(out, ref, atol=1e-6):
            return code
    raise NotImplementedError("competesmoe_amd: unsupported expert activation callable")


def _operand_fork() -> bool:
    import os
    return os.environ.get("CSMOE_OPERAND_FORK", "1") != "0"


def op_dtype(x: torch.Tensor) -> torch.dtype:
    """cvmm.get_dtype() (cvmm.py:29-32): the autocast dtype, else fp32 (bf16 inputs outside autocast stay bf16)."""
    if torch.is_autocast_enabled("cuda"):
        return torch.get_autocast_dtype("cuda")
    return x.dtype if x.dtype == torch.bfloat16 else torch.float32


class MoE(LoggingLayer, RegularizedLayer, OncePerIterLayer, nn.Module):
    def __init__(self, dmodel: int, n_experts: int, expert_size: int, n_heads: int, std_gate: float = 1.0,
                 std_expert: float = 1.0, topk=2, dropout: float = 0, weight_scale: float = 1.0,
                 selection_mode: str = "sigmoid", perplexity_reg: float = 0.0, perplexity_reg_mode: str = "step",
                 activation_after_topk: bool = False, activation=F.relu, sel_bias: bool = False, bias: bool = False,
                 v_dim: Optional[int] = None, expert_dropout: float = 0.0, sync_distributed: bool = False,
                 selection_dropout: float = 0.0, log_interval: Optional[int] = 100, args=None, is_att=False,
                 out_dmodel=None, inp_expert=None, out_expert=None):
        super().__init__()
        self.is_att = bool(is_att)
        self.iter = 0
        self.k_dim = dmodel
        self.v_dim = v_dim if v_dim is not None else dmodel
        self.n_experts = n_experts
        self.expert_size = expert_size
        self.size = n_experts * expert_size
        self.dropout = dropout
        self.selection_mode = selection_mode
        self.k_vec_dim = dmodel
        self.n_heads = n_heads
        self.activation = activation
        self.act_code = _act_code(activation)
        self.weight_scale = weight_scale
        self.layer = 0
        self.was_training = True
        self.log_interval = log_interval
        self.out_dmodel = out_dmodel if out_dmodel is not None else dmodel
        self.div = 1
        self.name_moe = "mlp"
        self.args = args
        self.training = False            # the reference leaves the flag False until .train() (moe.py:104)
        self.num_experts = self.num_of_experts = n_experts
        self.real_n_experts = 1
        # BASELINE config 5 (no counterpart upstream): the four row-space expert GEMMs of a step on the MXFP8 matrix pipe
        self.fp8_experts = bool(getattr(args, "fp8_experts", False))
        # keep the quantised weights while the parameters are unchanged (micro-batches of one optimizer step): functional._FP8_WCACHE
        self.fp8_weight_cache = bool(getattr(args, "fp8_weight_cache", False))
        self.selection_dropout = selection_dropout
        self.expert_dropout = expert_dropout
        self.sel_weight_scale = weight_scale
        self.num_selected = topk
        if self.is_att:
            # MoE attention projection (moe.py:111-117; built by full_moe_relative_attention.py:267-300 with n_experts = experts
            # per head x heads, expert_size = 1): a gate over all heads' experts and ONE [inp_expert, out_expert] matrix per expert
            self.w_gate = nn.Parameter(torch.randn(n_experts, dmodel) * std_gate)
            self.renorm_rows(self.w_gate)
            self.div = 10
            self.real_n_experts = n_heads
            self.experts = nn.Parameter(torch.randn(n_experts, inp_expert, out_expert) * std_expert)
        else:
            self.w_gate = nn.Parameter(torch.empty(n_experts, dmodel))
            nn.init.normal_(self.w_gate, std=dmodel ** -0.5 * weight_scale)
            n_held = self._n_held_experts(n_experts)     # expert-parallel layers hold their own slice of the packed tensors only
            self.values = nn.Parameter(torch.empty(n_held, expert_size, self.v_dim))
            self.keys = nn.Parameter(torch.empty(n_held, dmodel, expert_size))
            nn.init.normal_(self.keys, std=dmodel ** -0.5 * weight_scale)
            nn.init.normal_(self.values, std=self.size ** -0.5 * weight_scale)
            self.num_selected = n_heads      # "with MLP we get number of expert is n_head" (moe.py:128)
        if bias:
            self.bias = nn.Parameter(torch.zeros(self._n_held_experts(n_experts), expert_size))
            self.o_bias = nn.Parameter(torch.zeros(self.v_dim))
        else:
            self.bias = None
            self.o_bias = None
        self.dist_experts = None
        self.entropy_expert_selected = []
        self.entropy_expert_all = []
        # handed over by the block around the layer (pretrain/block.py), each consumed by the next forward
        self._pre_logits = None       # gate logits computed with the LayerNorm
        self._residual = None         # the block's residual stream, added in the combine epilogue
        self._stream_dtype = None     # dtype of the tensor the reference layer would have been called with (its `x.dtype`)
        self._twins = None            # spare bf16 operands of this forward's x (operand())
        self._forking = False         # inside forward(): only there is x cast once for all its consumers, and the spares never outlive it
        self.register_forward_pre_hook(MoE._enter_forward)
        self.register_forward_hook(MoE._leave_forward)

    _fuses_residual = False           # True on layers whose output IS one combine (smoe, competesmoe)

    def _n_held_experts(self, n_experts: int) -> int:
        """Experts whose weights this module holds (all of them; pretrain/smoe_ep.py: this rank's share)."""
        return n_experts

    def _plain_gate(self) -> bool:
        """compute_gate is F.linear(x, w_gate) (what the block's fused LayerNorm + gate launch computes)."""
        return True

    # ------------------------------------------------------------------ gate / selection
    def gate(self, x):
        return F.linear(x, self.w_gate, None)

    def compute_gate(self, x):
        """F.linear(x, w_gate) in the op dtype (logits are bf16 under autocast), HIP skinny GEMM."""
        shp = x.shape
        pre = self._pre_logits
        if pre is not None:
            self._pre_logits = None
            return pre.view(*shp[:-1], -1)
        lg = GateLogits.apply(self.operand(x), self.w_gate)
        return lg.view(*shp[:-1], -1)

    def select(self, scores, mode, x_dtype):
        if self._stream_dtype is not None:
            x_dtype = self._stream_dtype
        shp = scores.shape
        sm, idx, w = RouterSelect.apply(scores.reshape(-1, shp[-1]), self.num_selected, mode, x_dtype == torch.bfloat16)
        K = self.num_selected
        return w.view(*shp[:-1], K), idx.view(*shp[:-1], K), sm.view(shp)

    @staticmethod
    def _enter_forward(module, args):
        module._twins, module._forking = None, True

    @staticmethod
    def _leave_forward(module, args, output):
        module._twins, module._forking = None, False

    def operand(self, x):
        """x as the [T, D] operand of a kernel in the op dtype.  An fp32 x under bf16 autocast that takes gradients is cast ONCE per
        forward: the first call forks it (functional.OperandFork), the gate / experts / shared expert each take one of the bf16
        tensors, and x receives the fp32 sum of their gradients in one pass -- the values of the reference's separate casts.
        Only inside forward() (hooks above): a method called on its own (att_forward, a test calling ffn) casts per call, so no spare
        tensor -- and with it the graph behind x -- outlives a forward.  CSMOE_OPERAND_FORK=0: a cast per call."""
        op = op_dtype(x)
        x2 = x.reshape(-1, x.shape[-1])
        if x2.dtype == op:
            return x2
        if not (self._forking and op == torch.bfloat16 and x2.dtype == torch.float32 and x2.is_cuda and torch.is_grad_enabled()
                and x2.requires_grad and _operand_fork()):
            return x2.to(op)
        tw = self._twins
        if tw is not None and tw[0] is x and tw[1] == x._version and tw[2]:
            return tw[2].pop()
        outs = OperandFork.apply(x2.contiguous(), 4)
        self._twins = (x, x._version, list(outs[1:]))
        return outs[0]

    def gate_and_select(self, x, mode, x_dtype):
        """compute_gate + select -> (gate_logits, weights, selected_experts, gate_softmax); one launch that reads x once
        (csmoe_gate_select, same bits) when the gate is the plain F.linear and the block has not computed the logits already."""
        shp = x.shape
        op = op_dtype(x)
        K, E = self.num_selected, self.w_gate.shape[0]
        if self._pre_logits is None and self._plain_gate() and op == torch.bfloat16:
            x2 = self.operand(x)
            if ops.gate_select_ok(x2, self.w_gate, K):
                sd = self._stream_dtype if self._stream_dtype is not None else x_dtype
                lg, sm, idx, w = GateSelect.apply(x2, self.w_gate, K, mode, sd == torch.bfloat16)
                return lg.view(*shp[:-1], E), w.view(*shp[:-1], K), idx.view(*shp[:-1], K), sm.view(*shp[:-1], E)
            gate_logits = GateLogits.apply(x2, self.w_gate).view(*shp[:-1], E)      # compute_gate on the operand already in hand
            w, idx, sm = self.select(gate_logits, mode, x_dtype)
            return gate_logits, w, idx, sm
        gate_logits = self.compute_gate(x)
        w, idx, sm = self.select(gate_logits, mode, x_dtype)
        return gate_logits, w, idx, sm

    def topk_expert(self, gate_logits, x_dtype=torch.float32):
        """softmax(fp32) -> top-k -> renormalised weights (smoe.py:123-143 + :236)."""
        return self.select(gate_logits, L.SEL_SOFTMAX, x_dtype)

    # ------------------------------------------------------------------ fused FFN (the two cvmm calls)
    def ffn(self, x, selected_experts, weights, keys=None, values=None, bias=None):
        shp = x.shape
        op = op_dtype(x)
        x2 = self.operand(x)
        K = selected_experts.shape[-1]
        res = None
        if keys is None and self._residual is not None:
            res, self._residual = self._residual, None
        # `relu_pass_rate` every log_interval iterations (compute_scores, moe.py:406-414; upstream tests the bound method
        # `self.train`, which is always true, so evaluation logs too)
        stats = {} if (keys is None and self.log_interval is not None and self.iter % self.log_interval == 0) else None
        wk = weights.reshape(-1, K)         # fp32; the bf16 rounding of `reduction_weight.type_as(res)` happens inside the function
        if self.fp8_experts:
            if op != torch.bfloat16:
                raise ValueError("competesmoe_amd: args.fp8_experts needs bf16 activations (bf16 autocast or a bf16 layer)")
            out = MoEFFNPackedFP8.apply(x2, wk.float().contiguous(), selected_experts.reshape(-1, K).int().contiguous(),
                                        self.keys if keys is None else keys, self.values if values is None else values,
                                        self.bias if bias is None else bias, self.act_code, L.COMBINE_DOT,
                                        self.fp8_weight_cache and keys is None and values is None)
            if res is not None:
                out = res + out.view(res.shape)
            return out.view(*shp[:-1], -1)
        out = MoEFFNPacked.apply(x2, wk.float().contiguous(),
                                 selected_experts.reshape(-1, K).int().contiguous(),
                                 self.keys if keys is None else keys, self.values if values is None else values,
                                 self.bias if bias is None else bias, None, self.act_code, L.COMBINE_DOT, res, stats)
        if stats:
            with torch.no_grad():
                h = stats["hact"]
                self.log("relu_pass_rate", (h > 0).float().sum() / h.numel())
        return out.view(*shp[:-1], -1)

    def ffn_slots(self, x, selected_experts, weights):
        """ffn() plus the selected experts' outputs per (token, k) slot [..., K, Dout] of the same pass (MoEFFNPackedSlots)."""
        shp = x.shape
        x2 = self.operand(x)
        K = selected_experts.shape[-1]
        wk = weights.reshape(-1, K)
        out, y_tk = MoEFFNPackedSlots.apply(x2, wk.float().contiguous(), selected_experts.reshape(-1, K).int().contiguous(),
                                            self.keys, self.values, self.bias, None, self.act_code, L.COMBINE_DOT)
        return out.view(*shp[:-1], -1), y_tk.view(*shp[:-1], K, -1)

    def shared_ffn(self, x, keys_shared, values_shared, bias_shared=None):
        """The always-on shared expert: the reference routes every token to expert 0 of a 1-expert table with unit weight
        (deepseekv2.py:154-165); that is a dense FFN."""
        shp = x.shape
        op = op_dtype(x)
        b = None if bias_shared is None else bias_shared[0]
        if self.fp8_experts:
            y = DenseFFNFP8.apply(self.operand(x), keys_shared[0], b, values_shared[0], self.act_code, self.fp8_weight_cache)
            return y.view(*shp[:-1], -1)
        y = DenseFFN.apply(self.operand(x), keys_shared[0], b, values_shared[0], None, self.act_code, L.B_KN)
        return y.view(*shp[:-1], -1)

    # ------------------------------------------------------------------ losses ([B,N,E]-sized torch math)
    def renorm_rows(self, x: torch.Tensor):
        """moe.py:140-144: unit rows rescaled to the tensor's previous overall spread."""
        with torch.no_grad():
            std_t = x.std(dim=-1, keepdim=True)
            x.div_(x.norm(dim=-1, keepdim=True))
            x.mul_(std_t / x.std())

    def entropy_balance(self, sel):
        """moe.py:323-332 + framework/utils/entropy.py:21-22 + distributed_ops.py:47-58 (non-distributed branch); attention
        selections [B, N, heads, E] reduce over N per head (d = -3)."""
        d = -3 if self.is_att else -2
        if not self.is_att:
            sel = sel.flatten(1, -2)
        ls = F.log_softmax(sel, dim=-1)
        lm = ls.float().logsumexp(d) - math.log(ls.shape[d])
        return -(-(lm * lm.exp()).sum(-1)).mean()

    def zloss(self, gate_logits, gate_softmax=None):
        return torch.square(torch.logsumexp(gate_logits, dim=-1)).mean()

    def balanceloss(self, selected_experts, gate_softmax):
        E = self.num_of_experts
        proxy = gate_softmax.mean(dim=-2)
        top1 = selected_experts[..., 0].long().unsqueeze(-1)
        dens = (top1 == ops.cached_arange(E, top1.device)).float().mean(dim=-2)      # no F.one_hot: it syncs the device
        return (proxy * dens).mean() * float(E ** 2)

    # ------------------------------------------------------------------ eval-time statistics (moe.py:163-182)
    def entropy(self, prob_dist):
        return -torch.sum(prob_dist * torch.log(prob_dist + 1e-18), dim=-1)

    def get_dist_experts(self):
        return self.dist_experts

    def add_dist_experts(self, selection=None):
        assert selection is not None, "Selection must to not None"
        oh = F.one_hot(selection.reshape(-1, selection.shape[-1]).long(), num_classes=self.num_of_experts).sum(-2).sum(0)
        self.dist_experts = oh if self.dist_experts is None else self.dist_experts + oh

    def add_dist_weight(self, weight, is_all=False):
        (self.entropy_expert_all if is_all else self.entropy_expert_selected).append(self.entropy(weight).mean())

    def get_weight_dist(self):
        return {"entropy_all": torch.stack(self.entropy_expert_all).mean().item(),
                "entropy_topk": torch.stack(self.entropy_expert_selected).mean().item()}

    def pre_train_forward(self):
        pass

    def before_loss(self):
        if self.training:
            self.iter += 1

    def _finish(self, out, x):
        self._twins = None
        self.layer += 1
        self.was_training = self.training
        res = out.view(*x.shape[:-1], self.v_dim)
        if self.o_bias is not None:
            res = res + self.o_bias
        return res

    def _test_stats(self, selected, weights, softmax):
        if getattr(self.args, "test_only", False):
            self.add_dist_experts(selection=selected)
            self.add_dist_weight(weight=weights)
            self.add_dist_weight(weight=softmax, is_all=True)

    def forward(self, x, return_id_experts=False, return_full=True, *args, **kwargs):
        """Base forward (moe.py:418-454): top-k of the softmax WITHOUT renormalisation."""
        gate_logits = self.compute_gate(x)
        sm, idx, w = RouterSelect.apply(gate_logits.reshape(-1, gate_logits.shape[-1]), self.num_selected, L.SEL_SOFTMAX, False)
        K = self.num_selected
        idx = idx.view(*x.shape[:-1], K)
        vals = torch.gather(sm.view(*x.shape[:-1], -1), -1, idx.long())
        out = self.ffn(x, idx, vals)
        bal = self.entropy_balance(gate_logits) * (self.args.balance_loss_coef / self.div)
        self.add_reg(lambda: bal, f"{self.name_moe}_ebalance")
        return self._finish(out, x)
