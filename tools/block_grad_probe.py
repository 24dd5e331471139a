#!/usr/bin/env python3
"""LLaVA block, bf16 smoe fixture: where the 2e-3 of its expert gradients comes from (GPU box).  Compares the kernel's xn with the
fixture's, the block's expert gradients with the fixture's, and the LAYER alone on the fixture's own xn / upstream gradient."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.golden_util import load, rel_l2
from tests.test_block_gpu import build_block
from competesmoe_amd import ops

fx = load("block_smoe_bf16")
m = fx["meta"]
blk = build_block(fx, torch.bfloat16)
x = fx["x_mid"].cuda().requires_grad_(True)
out, aux, _, _ = blk(x)
dy = fx["dy"].cuda()
((out.float() * dy.float()).sum() + aux.float()).backward()
with torch.no_grad():
    xn, _, _, lg = ops.layernorm_gate(fx["x_mid"].cuda().reshape(-1, m["D"]), blk.layer_norm2.weight, blk.layer_norm2.bias, m["eps"], blk.moelayer.gate.weight)
print("xn elements differing from the fixture:", int((xn.cpu() != fx["xn"].reshape(-1, m["D"])).sum()), "of", xn.numel())
print("out", rel_l2(out.detach().cpu(), fx["output"]), "elements differing", int((out.detach().cpu() != fx["output"]).sum()))
params = dict(blk.moelayer.named_parameters())
for k, g in fx["moe_grads"].items():
    if g is not None and (k.startswith("experts.0.") or k.startswith("experts.3.") or k == "gate.weight"):
        print(k, rel_l2(params[k].grad.cpu(), g))
# the layer alone on the fixture's xn, with the gradient the block hands it (dy in bf16)
blk2 = build_block(fx, torch.bfloat16)
lay = blk2.moelayer
xn_f = fx["xn"].cuda().requires_grad_(True)
o2, a2, _, _ = lay(xn_f)
((o2.float() * dy.float()).sum() + a2.float()).backward()
p2 = dict(lay.named_parameters())
for k in ("experts.0.fc1.weight", "experts.0.fc2.weight", "experts.3.fc1.weight", "gate.weight"):
    print("layer alone on the fixture's xn:", k, rel_l2(p2[k].grad.cpu(), fx["moe_grads"][k]))
