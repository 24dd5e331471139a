#!/usr/bin/env python3
"""Generate golden vectors for the LLaVA-stack MoE layers by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference).  Nothing from the
reference is copied: this script imports `moe_model.model.moe` from
/root/reference (with an empty `loguru` stub), feeds seeded
synthetic tensors through the reference classes on CPU and dumps inputs,
parameters, per-stage outputs and gradients as small .pt fixtures.

Usage:  python tests/golden/make_golden_llava.py   (writes tests/golden/llava_*.pt)
"""
import os
import sys
import types
import copy

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _import_reference():
    sys.modules.setdefault("loguru", types.ModuleType("loguru"))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import moe_model.model.moe as ref_moe  # noqa: F401  (registers smoe/competesmoe/smoe_share/...)
    import moe_model.model.moe.deepseekv3 as ref_dsv3  # not imported by the package __init__
    from moe_model.model.moe.register import get_moe
    from moe_model.model.multimodal_encoder.siglip_smoe import SiglipMLP
    return get_moe, SiglipMLP


def make_args(**kw):
    base = dict(
        moe_name="smoe", balance_loss_coef=0.01, router_z_loss_coef=0.001,
        rate_flip=1.0, warm_up=0.0, max_compete_in_iter=8,
        router_loss_coef=0.02, diversity_loss_coef=0.03, bal_comp_loss_coef=0.04,
        hybrid=False, router_theta=0.5, norm_sigmoid=False, init_weight=True,
    )
    base.update(kw)
    return types.SimpleNamespace(**base)


def build_experts(kind, E, D, F, Dout, seed, SiglipMLP):
    g = torch.Generator().manual_seed(seed)
    experts = []
    for _ in range(E):
        if kind == "seq_gelu":
            m = nn.Sequential(nn.Linear(D, F), nn.GELU(), nn.Linear(F, Dout))
        elif kind == "siglip_tanh":
            cfg = types.SimpleNamespace(hidden_act="gelu_pytorch_tanh", hidden_size=D, intermediate_size=F)
            m = SiglipMLP(cfg)
        else:
            raise ValueError(kind)
        for p in m.parameters():
            # weights N(0, 0.2/sqrt(fan)) keep activations O(1) at fixture size; biases N(0, 0.1)
            if p.dim() == 2:
                p.data = torch.randn(p.shape, generator=g) * (1.0 / p.shape[1] ** 0.5)
            else:
                p.data = torch.randn(p.shape, generator=g) * 0.1
        experts.append(m)
    return nn.ModuleList(experts)


def run_case(name, moe_name, dtype, *, B=2, N=64, D=64, F=128, Dout=None, E=8, K=2,
             expert_kind="seq_gelu", args_kw=None, competition=False, seed=0):
    get_moe, SiglipMLP = _import_reference()
    Dout = D if Dout is None else Dout
    args = make_args(moe_name=moe_name, **(args_kw or {}))
    experts = build_experts(expert_kind, E, D, F, Dout, seed + 1, SiglipMLP)
    cls = get_moe(moe_name)
    if moe_name in ("smoe_share", "deepseekv3"):
        # these classes deep-copy ONE expert; overwrite with the independently initialised ones afterwards
        layer = cls(D, Dout, E, K, experts[0], args)
        for i in range(E):
            layer.experts[i].load_state_dict(experts[i].state_dict())
    else:
        layer = cls(D, Dout, E, K, experts, args)
    layer = layer.to(dtype)
    layer.train()

    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, N, D, generator=g).to(dtype)
    dy = torch.randn(B, N, Dout, generator=g).to(dtype)

    fx = {"meta": dict(name=name, moe_name=moe_name, dtype=str(dtype).replace("torch.", ""),
                       B=B, N=N, D=D, F=F, Dout=Dout, E=E, K=K, expert_kind=expert_kind,
                       competition=competition, args=vars(args))}

    if "competesmoe" in moe_name:
        torch.manual_seed(1234)
        pf = layer.set_total_steps(10, 0, {})
        layer.set_current_steps(3)
        if not competition:
            # force the router branch on the scheduled step
            layer.prob_flips = torch.zeros_like(layer.prob_flips)
        fx["prob_flips"] = layer.prob_flips.clone()

    fx["state"] = {k: v.clone() for k, v in layer.state_dict().items()}
    fx["x"], fx["dy"] = x.clone(), dy.clone()

    # ---- per-stage intermediates (no grad) ----
    with torch.no_grad():
        if hasattr(layer, "router_policy"):
            w, idx, sm, lg = layer.router_policy(x)
        else:
            lg = layer.gate(x)
            w, idx, sm = layer.topk_expert(gate_logits=lg)
            w = w / torch.sum(w, dim=-1, keepdim=True).to(x.dtype)
        fx["gate_logits"], fx["gate_softmax"] = lg.clone(), sm.clone()
        fx["selected_experts"], fx["weights"] = idx.clone(), w.clone()
        if competition:
            aw, aidx, asm, ascore, topk_out = layer.competition_policy(x)
            fx["aff_weights"], fx["aff_selected"] = aw.clone(), aidx.clone()
            fx["aff_softmax"], fx["aff_scores"] = asm.clone(), ascore.clone()
            fx["aff_topk_out"] = topk_out.clone()

    # ---- full forward + backward ----
    xg = x.clone().requires_grad_(True)
    out, aux, _none, infor = layer(xg)
    fx["output"] = out.detach().clone()
    fx["aux_loss"] = aux.detach().clone()
    fx["infor_aux"] = {k: v.detach().clone() for k, v in infor.items()}
    loss = (out.float() * dy.float()).sum() + aux.float()
    loss.backward()
    fx["x_grad"] = xg.grad.clone()
    fx["grads"] = {k: (p.grad.clone() if p.grad is not None else None) for k, p in layer.named_parameters()}

    # ---- eval-mode forward (x.requires_grad False => no aux) ----
    with torch.no_grad():
        out_e, aux_e, _, infor_e = layer(x)
    fx["output_nograd"] = out_e.clone()
    fx["aux_loss_nograd"] = aux_e.clone()

    path = os.path.join(HERE, f"llava_{name}.pt")
    torch.save(fx, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB",
          "| out", tuple(out.shape), "aux", float(aux))


def schedule_case():
    """prob_flips tensors for a fixed torch.manual_seed: three layers chained (vision->projector order)."""
    get_moe, _ = _import_reference()
    args = make_args(moe_name="competesmoe", rate_flip=0.6, warm_up=0.25, max_compete_in_iter=2)
    cls = get_moe("competesmoe")
    layers = [cls(16, 16, 4, 2, None, args) for _ in range(4)]
    torch.manual_seed(7)
    final = {}
    for i, l in enumerate(layers):
        final = l.set_total_steps(40, i, final)
    fx = {"meta": dict(total_steps=40, seed=7, args=vars(args)),
          "prob_flips": {int(k): v.clone() for k, v in final.items()},
          "step_warm": layers[0].step_warm, "flip_steps": layers[0].flip_steps}
    path = os.path.join(HERE, "llava_schedule.pt")
    torch.save(fx, path)
    print("wrote", path)


def main():
    torch.set_num_threads(4)
    for dt, tag in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        run_case(f"smoe_{tag}", "smoe", dt)
        run_case(f"smoe_siglip_{tag}", "smoe", dt, expert_kind="siglip_tanh", D=64, F=96, E=4)
        run_case(f"smoe_proj_{tag}", "smoe", dt, D=48, F=80, Dout=80, E=4)  # projector: Din != Dout
        run_case(f"competesmoe_router_{tag}", "competesmoe", dt, competition=False)
        run_case(f"competesmoe_comp_{tag}", "competesmoe", dt, competition=True)
        run_case(f"competesmoe_comp_hybrid_{tag}", "competesmoe", dt, competition=True,
                 args_kw=dict(hybrid=True))
        run_case(f"smoe_share_{tag}", "smoe_share", dt, K=3)
        run_case(f"deepseekv3_{tag}", "deepseekv3", dt, K=3)
    # sigmoid-normalised competition scores (competesmoe.py:249-251), fp32
    run_case("competesmoe_comp_normsigmoid_fp32", "competesmoe", torch.float32, competition=True, args_kw=dict(norm_sigmoid=True))
    schedule_case()


if __name__ == "__main__":
    main()
