import sys, torch
sys.path.insert(0, "/root/repo")
from tests.golden_util import load, rel_l2
import tests.test_llava_modules_gpu as TL
fx = load("llava_smoe_bf16")
layer, dt = TL.build_layer(fx)
x = fx["x"].to("cuda").requires_grad_(True)
out, aux, _, infor = layer(x)
((out.float() * fx["dy"].to("cuda").float()).sum() + aux.float()).backward()
print("dx", rel_l2(x.grad, fx["x_grad"].to("cuda")))
for k, p in layer.named_parameters():
    g = fx["grads"].get(k)
    if g is not None and p.grad is not None:
        d = (p.grad.float() - g.to("cuda").float())
        ne = (p.grad != g.to("cuda")).float().mean().item()
        print(f"{k:28s} rel {rel_l2(p.grad, g.to('cuda')):.2e}  elements differing {ne:.4f}  max|d|/max|g| {float(d.abs().max() / g.float().abs().max()):.2e}")
