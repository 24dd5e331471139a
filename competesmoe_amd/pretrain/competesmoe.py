"""Pretrain `competesmoe` (moe_pretrain_model/layers/moe/competesmoe.py:37-616): router policy on ordinary steps, dense
competition (`competition_policy_mlp_faster`, :381-414) on the steps scheduled in `prob_flips_final[id_layer]`."""
import torch
import torch.distributed as dist
import torch.nn.functional as F

from .. import _lib as L
from .. import ops
from ..functional import CompetitionAffinityPacked, DenseFFN, DiversityLoss, RouterSelect, SoftplusMean
from .moe import MoE, op_dtype
from ..schedule import draw_balanced_flips
from .register import register_moe


@register_moe("competesmoe")
class CompeteSMoE(MoE):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        args = self.args
        self.warm_up = args.warm_up
        self.rate_flip = args.rate_flip
        self.current_steps = 0
        self.step_warm = None
        self.is_prob_flips = True
        self.total_steps = args.stop_after
        assert args.stop_after > 0, f"Warning: stop_after {args.stop_after} < 1, You must setting stop_after > 0"
        self.prob_flips_final = {}
        self.max_compete_in_iter = args.max_compete_in_iter
        self.nb_diver = 0
        self._flips_host = {}

    # ------------------------------------------------------------------ schedule (:123-273)
    def set_total_steps(self, id_layer=0):
        self.step_warm = int(self.warm_up * self.total_steps)
        flip_steps = self.total_steps - self.step_warm
        self.flip_steps = flip_steps
        if flip_steps <= 0:
            raise ValueError("self.total_steps - self.step_warm must be greater than 0.")
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
        device = self.w_gate.device
        cap = self.max_compete_in_iter
        if rank == 0:
            cur = draw_balanced_flips(flip_steps, self.rate_flip, cap, self.prob_flips_final)
            probs_current = torch.tensor(cur, dtype=torch.bool, device=device)
        else:
            probs_current = torch.empty(flip_steps, dtype=torch.bool, device=device)
        if world > 1:
            dist.broadcast(probs_current, src=0)
        self.prob_flips_final[id_layer] = probs_current
        self._flips_host = {}
        self.is_prob_flips = False
        return self.prob_flips_final

    def set_current_steps(self, step):
        self.current_steps = step

    def _competing(self, x, id_layer) -> bool:
        if not x.requires_grad or self.step_warm is None or self.current_steps < self.step_warm:
            return False
        t = self.prob_flips_final[id_layer]
        key = (t.data_ptr(), t._version, t.numel())       # a checkpoint load / in-place edit of the schedule refreshes the host copy
        h = self._flips_host.get(id_layer)
        if h is None or h[0] != key:
            h = self._flips_host[id_layer] = (key, t.tolist())
        return bool(h[1][self.current_steps - self.step_warm] == 1)

    # ------------------------------------------------------------------ policies
    _fuses_residual = True

    def _plain_gate(self) -> bool:
        return not (getattr(self.args, "is_cosine", False) or getattr(self.args, "is_norm_weight", False))

    def compute_gate(self, x):
        a = self.args
        if getattr(a, "is_cosine", False) and not getattr(a, "is_norm_weight", False):
            return F.linear(F.normalize(x, p=2.0, dim=-1), F.normalize(self.w_gate, p=2.0, dim=-1))
        if getattr(a, "is_norm_weight", False):
            return F.linear(x, F.normalize(self.w_gate, p=2.0, dim=-1))
        return super().compute_gate(x)

    def router_policy(self, x, is_normal_mode=False):
        assert not (getattr(self.args, "is_cosine", False) and getattr(self.args, "is_norm_weight", False)), \
            "Can not active  both  Cosine and Norm Weigh. Just use one method - Cosine or Norm Weigh to Normalization"
        gate_logits = self.compute_gate(x)
        if getattr(self.args, "norm_sigmoid", False):
            # top-k of the logits, sigmoid(v / scale_weight), renormalised (:476-483): same kernel and tie rule as every other router
            shp = gate_logits.shape
            xd = self._stream_dtype or x.dtype
            sm, idx, w = RouterSelect.apply(gate_logits.reshape(-1, shp[-1]), self.num_selected, L.SEL_TOPK_SIGMOID,
                                            xd == torch.bfloat16, float(getattr(self.args, "scale_weight", 1.0)))
            K = self.num_selected
            return w.view(*shp[:-1], K), idx.view(*shp[:-1], K), sm.view(shp), gate_logits
        weights, selected_experts, gate_softmax = self.topk_expert(gate_logits, x.dtype)
        return weights, selected_experts, gate_softmax, gate_logits

    def competition_policy_mlp_faster(self, x):
        """relu(x @ keys[e]) @ values[e] for EVERY expert (dense GEMM kernels), affinity = mean softplus, top-K of the raw
        affinities, renormalised weights, the K selected dense outputs for the diversity loss."""
        B, N, D = x.shape
        op = op_dtype(x)
        x2 = self.operand(x)
        if self._lean_competition(x2):
            # without the [T, E, D] outputs: affinities from the second GEMM's epilogue, backward by recomputation, the selected
            # experts' outputs from the sparse step (forward below); functional.CompetitionAffinityPacked
            fp32_aff = op == torch.bfloat16 and torch.is_autocast_enabled("cuda")
            aff = CompetitionAffinityPacked.apply(x2, self.keys, self.values, self.act_code, fp32_aff)
            asm, idx, w = RouterSelect.apply(aff, self.num_selected, L.SEL_RAW, False)
            return w.view(B, N, -1), idx.view(B, N, -1), asm.view(B, N, -1), aff.view(B, N, -1), None
        outs = [DenseFFN.apply(x2, self.keys[e], None, self.values[e], None, self.act_code, L.B_KN) for e in range(self.n_experts)]
        # under CUDA autocast F.softplus is an fp32-policy op: fp32 softplus / mean / affinities from the bf16 expert outputs, top-K
        # and the renormalised weights on fp32 scores (the reference trains this way, simple_task.py:295); outside autocast the ops
        # stay in the operand dtype
        fp32_aff = op == torch.bfloat16 and torch.is_autocast_enabled("cuda")
        aff = torch.stack([SoftplusMean.apply(o, fp32_aff) for o in outs], dim=-1)       # [T,E] fp32 (autocast) / op dtype
        asm, idx, w = RouterSelect.apply(aff, self.num_selected, L.SEL_RAW, False)
        eo = torch.stack(outs, dim=1).view(B, N, self.n_experts, -1)
        idx_l = idx.view(B, N, -1).long()
        topk = torch.gather(eo, 2, idx_l.unsqueeze(-1).expand(B, N, self.num_selected, eo.size(-1)))
        return w.view(B, N, -1), idx.view(B, N, -1), asm.view(B, N, -1), aff.view(B, N, -1), topk

    def _lean_competition(self, x2) -> bool:
        """The dense pass without its stored outputs / activations?  CSMOE_COMPETITION_LEAN=1 / 0 forces it; default: when the
        stored form would keep more than 24 GiB alive (T * E * (2F + Dout) activations).  Needs the plain sparse step for the
        selected outputs (no fp8 experts, no residual handed in by the block)."""
        import os
        E, D, Fd = self.keys.shape
        Dout = self.values.shape[2]
        if self.fp8_experts or self._residual is not None or not ops.affinity_ok(x2, Fd, Dout):
            return False
        mode = os.environ.get("CSMOE_COMPETITION_LEAN", "auto")
        if mode in ("0", "1"):
            return mode == "1"
        return x2.shape[0] * E * (2 * Fd + Dout) * x2.element_size() > 24 * 2 ** 30

    def router_loss(self, gate_softmax, affinity_softmax):
        return F.mse_loss(gate_softmax, affinity_softmax)

    def experts_diversity_loss(self, expert_outputs):
        """Mean over T*K*K of the off-diagonal cosine similarities (:330-372) -- `csmoe_pair_cosine`, fp32 norms and dots of the
        expert outputs as stored.  (Under autocast upstream normalises in fp32 and runs the K x K bmm in bf16; the kernel keeps the
        fp32 dots.)  `nb_diver` counts the non-zero entries upstream and is never read back; K <= 8 here."""
        eo = expert_outputs
        if eo.dim() == 5:
            eo = eo.reshape(eo.shape[0], eo.shape[1] * eo.shape[2], *eo.shape[3:])
        B, N, K, D = eo.shape
        if K > 8:
            raise NotImplementedError("competesmoe_amd: diversity loss supports at most 8 selected experts")
        self.nb_diver += B * N * K * (K - 1)
        return DiversityLoss.apply(eo)

    def compute_moe_main(self, x, selected_experts, weights):
        return self.ffn(x, selected_experts, weights)

    def forward(self, x, return_id_experts=False, return_full=True, *args, **kwargs):
        id_layer = kwargs["id_layer"]
        assert id_layer is not None, "Layer Id must to not None"
        a = self.args
        is_comp = self._competing(x, id_layer)
        gate_weights, gate_selected_experts, gate_softmax, gate_logits = self.router_policy(x)
        if is_comp:
            aw, aidx, asm, aff, expert_outputs = self.competition_policy_mlp_faster(x)
            if expert_outputs is None:
                out, expert_outputs = self.ffn_slots(x, aidx, aw)
            else:
                out = self.compute_moe_main(x, aidx, aw)
            div = self.experts_diversity_loss(expert_outputs)
            self.add_reg(lambda: div * a.balance_loss_coef_comp / 2, self.name_moe + "_comp_diver_loss")
            if a.balance_affinity:
                bexp = self.entropy_balance(asm)
                self.add_reg(lambda: bexp * a.balance_loss_coef_comp / 2, f"{self.name_moe}_comp_ebalance")
            il = aidx.long()
            if a.in_topk:
                rl = self.router_loss(affinity_softmax=torch.gather(asm, -1, il).detach(), gate_softmax=torch.gather(gate_softmax, -1, il))
            elif a.hybrid or a.tribrid:
                rl = self.router_loss(affinity_softmax=asm.detach(), gate_softmax=gate_softmax) + self.router_loss(
                    affinity_softmax=torch.gather(asm, -1, il).detach(), gate_softmax=torch.gather(gate_softmax, -1, il)) * a.router_theta
                if a.tribrid and not a.hybrid:
                    gl = gate_selected_experts.long()
                    rl = rl + self.router_loss(affinity_softmax=torch.gather(asm, -1, gl).detach(),
                                               gate_softmax=torch.gather(gate_softmax, -1, gl)) * a.router_theta
            else:
                rl = self.router_loss(affinity_softmax=asm.detach(), gate_softmax=gate_softmax)
            self.add_reg(lambda: rl * a.router_loss_coef, f"{self.name_moe}_router_loss")
        else:
            out = self.compute_moe_main(x, gate_selected_experts, gate_weights)
            bal = self.entropy_balance(gate_logits) * (a.balance_loss_coef / self.div)
            self.add_reg(lambda: bal, f"{self.name_moe}_ebalance")
        self._test_stats(gate_selected_experts, gate_weights, gate_softmax)
        return self._finish(out, x)
