#!/usr/bin/env python3
"""bf16 competition step: d w (gradient of the affinity weights) and d aff (gradient of the affinities) of the HIP path against the
pinned oracle under the kernel's indices -- where the dense experts' gradient streams start to differ (GPU box)."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.golden_util import load, rel_l2
from tests.test_llava_modules_gpu import build_layer, args_of, unpack_experts, ACT_OF_KIND
from oracle import moe_oracle as O

fx = load(sys.argv[1] if len(sys.argv) > 1 else "llava_competesmoe_comp_bf16")
layer, dt = build_layer(fx)
m = fx["meta"]
got = {}
cp0 = layer.competition_policy

def cp(x):
    w, idx, asm, aff, topk = cp0(x)
    w.register_hook(lambda g: got.__setitem__("dw", g.detach().float().cpu()))
    aff.register_hook(lambda g: got.__setitem__("daff", g.detach().float().cpu()))
    got["w"], got["aff"], got["idx"] = w.detach().float().cpu(), aff.detach().float().cpu(), idx.detach().cpu().long()
    if topk is not None:
        topk.register_hook(lambda g: got.__setitem__("dtopk", g.detach().float().cpu()))
    return w, idx, asm, aff, topk
layer.competition_policy = cp
from competesmoe_amd import functional as Fn
from competesmoe_amd.moe import competesmoe as CM
rs_apply = CM.RouterSelect.apply

def rs(scores, *a):
    if scores.requires_grad:
        scores.register_hook(lambda g: got.__setitem__("daff", g.detach().float().cpu()))
    return rs_apply(scores, *a)
CM.RouterSelect = types.SimpleNamespace(apply=rs)
x = fx["x"].cuda().requires_grad_(True)
out, aux, _, _ = layer(x)
((out.float() * fx["dy"].cuda().float()).sum() + aux.float()).backward()

xo = fx["x"].clone().requires_grad_(True)
wg = fx["state"]["gate.weight"].clone().requires_grad_(True)
experts = [tuple(t.clone().requires_grad_(True) for t in e) for e in unpack_experts(fx)]
o, a, inf, st = O.llava_competesmoe_forward(xo, wg, experts, ACT_OF_KIND[m["expert_kind"]], m["K"], args_of(fx), competing=True,
                                            forced_aff_idx=got["idx"])
for k in ("aff_weights", "aff_scores", "aff_topk_out"):
    st[k].retain_grad()
((o.float() * fx["dy"].float()).sum() + a.float()).backward()
print("w", rel_l2(got["w"], st["aff_weights"].detach().float()), "aff", rel_l2(got["aff"], st["aff_scores"].detach().float()))
print("d w   :", rel_l2(got["dw"], st["aff_weights"].grad.float()), "elements differing", int((got["dw"].bfloat16() != st["aff_weights"].grad.bfloat16().reshape(got["dw"].shape)).sum()), "of", got["dw"].numel())
print("d aff :", rel_l2(got["daff"], st["aff_scores"].grad.float().reshape(got["daff"].shape)))
if "dtopk" in got:
    print("d topk:", rel_l2(got["dtopk"], st["aff_topk_out"].grad.float().reshape(got["dtopk"].shape)))
print("dx", rel_l2(x.grad.cpu(), xo.grad))
