"""`competesmoe`: router policy on ordinary steps, competition policy (every expert runs densely, top-K by mean-softplus
affinity, router distilled towards it) on scheduled steps -- moe_model/model/moe/competesmoe.py:8-415."""
import os

import torch
import torch.distributed as dist
import torch.nn.functional as F

from .. import _lib as L
from ..functional import RouterSelect, SoftplusMean
from ..schedule import draw_balanced_flips
from .register import register_moe
from .moe import MoeLayer, parse_expert


@register_moe("competesmoe")
class CompeteSMoE(MoeLayer):
    _fuses_residual = True        # output = one combine: a block around the layer may add its residual there (moe/block.py)

    def __init__(self, in_embed_dim=768, out_embed_dim=768, num_of_experts=4, num_selected=2, expert=None, args=None):
        super().__init__(in_embed_dim, out_embed_dim, num_of_experts, num_selected, expert, args)
        if args is None or not hasattr(args, "rate_flip"):
            raise ValueError("The 'args' parameter must have the attribute 'rate_flip'.")
        if not hasattr(args, "warm_up"):
            raise ValueError("The 'args' parameter must include 'warm_up'.")
        self.warm_up = args.warm_up
        self.rate_flip = args.rate_flip
        self.total_steps = None
        self.current_steps = 0
        self.step_warm = None
        self.is_prob_flips = True
        self.register_buffer("prob_flips", torch.zeros(15801))   # same placeholder shape as competesmoe.py:32
        self._flips_host = None
        self._flips_key = None
        self.init_gate_weights()

    # ------------------------------------------------------------------ schedule (competesmoe.py:35-179)
    def set_total_steps(self, total_steps, id_layer, prob_flips_final):
        assert id_layer is not None, "You must setup id layer is not None"
        assert prob_flips_final is not None, "You must setup prob_flips_final is not None"
        self.total_steps = total_steps
        self.step_warm = int(self.warm_up * self.total_steps)
        flip_steps = self.total_steps - self.step_warm
        self.flip_steps = flip_steps
        if flip_steps <= 0:
            raise ValueError("self.total_steps - self.step_warm must be greater than 0.")
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
        device = self.gate.weight.device
        cap = self.args.max_compete_in_iter
        if rank == 0:
            cur = draw_balanced_flips(flip_steps, self.rate_flip, cap, prob_flips_final)
            probs_current = torch.tensor(cur, dtype=torch.bool, device=device)
        else:
            probs_current = torch.empty(flip_steps, dtype=torch.bool, device=device)
        if world > 1:
            dist.broadcast(probs_current, src=0)
        prob_flips_final[id_layer] = probs_current
        self.prob_flips = probs_current
        self._flips_host = None                        # host copy (no per-step device sync); rebuilt by _competing
        self.is_prob_flips = False
        return prob_flips_final

    def set_current_steps(self, step):
        self.current_steps = step

    def _competing(self, x) -> bool:
        """competesmoe.py:347 -- x.requires_grad and scheduled; read from the host copy instead of `.item()`."""
        if not x.requires_grad or self.step_warm is None or self.current_steps < self.step_warm:
            return False
        i = self.current_steps - self.step_warm
        t = self.prob_flips
        key = (t.data_ptr(), t._version, t.numel())           # load_state_dict copies IN PLACE: the version counter moves
        if self._flips_host is None or self._flips_key != key:
            self._flips_host, self._flips_key = t.tolist(), key
        return bool(self._flips_host[i] == 1)

    # ------------------------------------------------------------------ policies
    def router_policy(self, x):
        r = self._route(x)                # gate + softmax + top-K (+ binning histogram) in one launch where the shapes allow
        return r.w, r.idx, r.softmax, r.logits

    def competition_policy(self, x):
        """competesmoe.py:219-259: every expert densely (DenseFFN kernels), affinity = mean softplus (SoftplusMean kernel),
        softmax fp32 + top-K on the RAW affinities + renormalisation in x.dtype (RouterSelect, SEL_RAW)."""
        B, N, D = x.shape
        aff = self.dense_affinities(x) if self._lean_competition(x) else None
        lean = aff is not None
        if not lean:
            outs = [self.dense_expert(i, x) for i in range(self.num_of_experts)]
            aff = torch.stack([SoftplusMean.apply(o.reshape(B * N, o.shape[-1])) for o in outs], dim=-1)      # [T,E] x.dtype
        scores = torch.sigmoid(aff) if getattr(self.args, "norm_sigmoid", False) else aff
        if getattr(self.args, "norm_sigmoid", False):
            asm = F.softmax(aff, dim=-1, dtype=torch.float32)
            _, idx, w = RouterSelect.apply(scores, self.num_selected, L.SEL_RAW, False)
        else:
            asm, idx, w = RouterSelect.apply(scores, self.num_selected, L.SEL_RAW, False)
        E, K = self.num_of_experts, self.num_selected
        if lean:        # the selected experts' outputs come out of the sparse step (forward: compute_moe_slots), same bits
            return w.view(B, N, K), idx.view(B, N, K), asm.view(B, N, E), aff.view(B, N, E), None
        expert_outputs = torch.stack(outs, dim=2)                                                   # [B,N,E,Dout]
        idx_l = idx.view(B, N, self.num_selected).long()
        topk = torch.gather(expert_outputs, 2, idx_l.unsqueeze(-1).expand(B, N, self.num_selected, expert_outputs.size(-1)))
        return w.view(B, N, K), idx.view(B, N, K), asm.view(B, N, E), aff.view(B, N, E), topk

    def _lean_competition(self, x) -> bool:
        """Run the dense pass without keeping its outputs / activations (functional.CompetitionAffinity: 8 GEMM passes per expert
        instead of 6, O(T) scratch instead of O(T*E))?  CSMOE_COMPETITION_LEAN=1 / 0 forces it; default: when what the stored form
        would keep alive -- T*E*(2F + Dout) activations -- exceeds 24 GiB."""
        mode = os.environ.get("CSMOE_COMPETITION_LEAN", "auto")
        if mode in ("0", "1"):
            return mode == "1"
        fc1, _, fc2 = parse_expert(self.experts[0])
        T = x.shape[0] * x.shape[1]
        return T * self.num_of_experts * (2 * fc1.weight.shape[0] + fc2.weight.shape[0]) * x.element_size() > 24 * 2 ** 30

    def router_loss(self, gate_softmax, affinity_softmax):
        return F.mse_loss(gate_softmax, affinity_softmax)

    def forward(self, x, return_id_experts=False, is_vision=False):
        gate_weights, gate_selected_experts, gate_softmax, gate_logits = self.router_policy(x)
        auxiliary_loss = x.new_zeros(())        # a fill kernel: torch.tensor(0.0, device=...) is a blocking H2D copy
        infor_aux = {}
        if self._competing(x):
            aff_w, aff_idx, aff_softmax, aff_scores, expert_outputs = self.competition_policy(x)
            routerloss = self.router_loss(gate_softmax=gate_softmax, affinity_softmax=aff_softmax.detach())
            if getattr(self.args, "hybrid", False):
                il = aff_idx.long()
                routerloss = routerloss + self.router_loss(
                    affinity_softmax=torch.gather(aff_softmax, -1, il).detach(),
                    gate_softmax=torch.gather(gate_softmax, -1, il)) * self.args.router_theta
            # the reference re-runs the K selected experts from x (competesmoe.py:374-379); weights are already x.dtype
            if expert_outputs is None:
                output, expert_outputs = self.compute_moe_slots(aff_idx, aff_w, x, weights_rounded=True)
            else:
                output = self.compute_moe(aff_idx, aff_w, None, x, weights_rounded=True)
            diversity_loss = self.experts_diversity_loss(expert_outputs=expert_outputs)
            balance_loss = self.balanceloss(selected_experts=aff_idx, gate_softmax=aff_softmax)
            auxiliary_loss = (routerloss * self.args.router_loss_coef + diversity_loss * self.args.diversity_loss_coef
                              + balance_loss * self.args.bal_comp_loss_coef)
            infor_aux = {"balance_loss": balance_loss.clone().detach(), "diversity_loss": diversity_loss.clone().detach(),
                         "routerloss": routerloss.clone().detach()}
        else:
            output = self.compute_moe(gate_selected_experts, gate_weights, None, x)
            if x.requires_grad or return_id_experts:
                auxiliary_loss, balance_loss, router_z_loss = self.combine_loss(
                    selected_experts=gate_selected_experts, gate_softmax=gate_softmax, gate_logits=gate_logits)
                infor_aux = {"balance_loss": balance_loss.clone().detach(), "router_z_loss": router_z_loss.clone().detach()}
        return output, auxiliary_loss, None, infor_aux
