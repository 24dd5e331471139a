"""In which order do the x-gradient streams of a HIP layer reach x?  (bf16 sums are order-sensitive: DESIGN.md section 4.)
Logs the execution order of the custom autograd nodes' backward for the shared-expert and competition fixtures, next to the
reference's order (pure-torch: later-created consumers first).  usage (GPU box): python tools/grad_order_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from competesmoe_amd import functional as Fn  # noqa: E402
from tests.golden_util import load, rel_l2  # noqa: E402
from tests.test_llava_modules_gpu import build_layer, oracle_grads  # noqa: E402

log = []
for name in ("DenseFFN", "MoEFFNModules", "GateSelect", "GateLogits", "RouterSelect", "SoftplusMean", "RouterAux"):
    cls = getattr(Fn, name)
    orig = cls.backward

    def make(orig, name):
        def bw(ctx, *a):
            log.append(name)
            return orig(ctx, *a)
        return staticmethod(bw)
    cls.backward = make(orig, name)

for case in ("smoe_share", "competesmoe_comp"):
    fx = load(f"llava_{case}_bf16")
    layer, dt = build_layer(fx)
    x = fx["x"].cuda().requires_grad_(True)
    out, aux, _, _ = layer(x)
    log.clear()
    ((out.float() * fx["dy"].cuda().float()).sum() + aux.float()).backward()
    print(case, "backward order:", log)
    print(case, "dx vs golden", rel_l2(x.grad.cpu(), fx["x_grad"]))

# hypothesis test: does creating the dense (shared / competition) experts' nodes in the other order reproduce the reference's bits?
from competesmoe_amd.moe import shard_smoe as SS, competesmoe as CS  # noqa: E402


def routed_and_shared_swapped(self, x):
    route = self._route(x)
    shared = self.dense_expert(self.num_of_experts, x)
    routed = self.compute_moe(route.idx, route.w, None, x, n_experts=self.num_of_experts)
    return routed, shared, route


SS._SharedBase._routed_and_shared = routed_and_shared_swapped
fx = load("llava_smoe_share_bf16")
layer, dt = build_layer(fx)
x = fx["x"].cuda().requires_grad_(True)
out, aux, _, _ = layer(x)
log.clear()
((out.float() * fx["dy"].cuda().float()).sum() + aux.float()).backward()
print("smoe_share, shared expert created BEFORE the routed step:", log, "dx vs golden", rel_l2(x.grad.cpu(), fx["x_grad"]))
