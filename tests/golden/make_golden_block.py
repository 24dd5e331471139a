#!/usr/bin/env python3
"""Golden vectors for the block around the layer by RUNNING THE REFERENCE's SiglipEncoderMoELayer.

Runs only in the build container (needs /root/reference; imported, never copied -- same loguru stub as
make_golden_llava.py).  The whole encoder layer (attention half included) runs on CPU; what is recorded is the MoE half:
the tensor entering `layer_norm2` (captured by a forward pre-hook, with its gradient), the LayerNorm and MoE parameters, the
layer's outputs and every gradient.  The attention half only serves to hand the MoE half a realistic, non-leaf input.

Usage:  python tests/golden/make_golden_block.py   (writes tests/golden/block_*.pt)
"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _import_reference():
    sys.modules.setdefault("loguru", types.ModuleType("loguru"))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import moe_model.model.moe  # noqa: F401  (registers the layers)
    from moe_model.model.multimodal_encoder.siglip_smoe import SiglipEncoderMoELayer
    return SiglipEncoderMoELayer


def run_case(name, moe_name, dtype, competition=False, B=2, N=48, D=64, F=128, E=8, K=2, seed=0):
    Layer = _import_reference()
    cfg = types.SimpleNamespace(hidden_size=D, num_attention_heads=4, attention_dropout=0.0, layer_norm_eps=1e-6,
                                moe_name=moe_name, num_experts=E, num_selected=K, hidden_act="gelu_pytorch_tanh",
                                intermediate_size=F)
    args = types.SimpleNamespace(sparse_upcycling=False, moe_name=moe_name, balance_loss_coef=0.01, router_z_loss_coef=0.001,
                                 rate_flip=1.0, warm_up=0.0, max_compete_in_iter=8, router_loss_coef=0.02,
                                 diversity_loss_coef=0.03, bal_comp_loss_coef=0.04, hybrid=False, router_theta=0.5,
                                 norm_sigmoid=False, init_weight=True)
    torch.manual_seed(seed + 11)
    layer = Layer(cfg, args)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in layer.named_parameters():
            if "layer_norm" in n:                      # non-trivial affine parameters
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g) if n.endswith("weight") else 0.1 * torch.randn(p.shape, generator=g))
            elif "experts" in n:
                p.copy_(torch.randn(p.shape, generator=g) * ((1.0 / p.shape[1] ** 0.5) if p.dim() == 2 else 0.1))
    layer = layer.to(dtype).train()
    if "competesmoe" in moe_name:
        torch.manual_seed(1234)
        layer.moelayer.set_total_steps(10, 0, {})
        layer.moelayer.set_current_steps(3)
        if not competition:
            layer.moelayer.prob_flips = torch.zeros_like(layer.moelayer.prob_flips)

    cap = {}

    def grab(mod, inp):
        cap["mid"] = inp[0]
        inp[0].retain_grad()

    h = layer.layer_norm2.register_forward_pre_hook(grab)
    x = torch.randn(B, N, D, generator=g).to(dtype).requires_grad_(True)
    dy = torch.randn(B, N, D, generator=g).to(dtype)
    out, aux, _ids, infor = layer(x)
    ((out.float() * dy.float()).sum() + aux.float()).backward()
    h.remove()
    fx = {"meta": dict(name=name, moe_name=moe_name, dtype=str(dtype).replace("torch.", ""), B=B, N=N, D=D, F=F, E=E, K=K,
                       competition=competition, eps=cfg.layer_norm_eps, args=vars(args)),
          "x_mid": cap["mid"].detach().clone(), "x_mid_grad": cap["mid"].grad.clone(), "dy": dy.clone(),
          "ln_state": {k: v.clone() for k, v in layer.layer_norm2.state_dict().items()},
          "moe_state": {k: v.clone() for k, v in layer.moelayer.state_dict().items()},
          "output": out.detach().clone(), "aux_loss": aux.detach().clone(),
          "infor_aux": {k: v.detach().clone() for k, v in infor.items()},
          "ln_grads": {k: p.grad.clone() for k, p in layer.layer_norm2.named_parameters()},
          "moe_grads": {k: (p.grad.clone() if p.grad is not None else None) for k, p in layer.moelayer.named_parameters()}}
    if "competesmoe" in moe_name:
        fx["prob_flips"] = layer.moelayer.prob_flips.clone()
    with torch.no_grad():
        xn = layer.layer_norm2(cap["mid"].detach())
        fx["xn"] = xn.clone()
        fx["gate_logits"] = layer.moelayer.gate(xn).clone()
        ml = layer.moelayer
        if hasattr(ml, "router_policy"):
            _w, idx, _sm, _lg = ml.router_policy(xn)
        else:
            _w, idx, _sm = ml.topk_expert(gate_logits=fx["gate_logits"])
        fx["selected_experts"] = idx.clone()
        if competition:
            _aw, aidx, _asm, ascore, _tk = ml.competition_policy(xn)
            fx["aff_selected"], fx["aff_scores"] = aidx.clone(), ascore.clone()
    path = os.path.join(HERE, f"block_{name}.pt")
    torch.save(fx, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB | aux", float(aux))


if __name__ == "__main__":
    for dt, tag in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        run_case(f"smoe_{tag}", "smoe", dt)
        run_case(f"competesmoe_router_{tag}", "competesmoe", dt, competition=False)
        run_case(f"competesmoe_comp_{tag}", "competesmoe", dt, competition=True)
