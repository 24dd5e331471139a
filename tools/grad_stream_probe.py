"""Which of the HIP competition step's dx streams differs from the oracle's, and how does the engine add them?  (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from competesmoe_amd import functional as Fn  # noqa: E402
from tests.golden_util import load, rel_l2  # noqa: E402
from tests.test_llava_modules_gpu import build_layer, oracle_dx_streams  # noqa: E402

rec = []
for name in ("DenseFFN", "MoEFFNModules", "GateSelect"):
    cls = getattr(Fn, name)
    orig = cls.backward

    def make(orig, name):
        def bw(ctx, *a):
            out = orig(ctx, *a)
            rec.append((name, out[0].detach().clone()))
            return out
        return staticmethod(bw)
    cls.backward = make(orig, name)

fx = load("llava_competesmoe_comp_bf16")
layer, dt = build_layer(fx)
x = fx["x"].cuda().requires_grad_(True)
out, aux, _, _ = layer(x)
((out.float() * fx["dy"].cuda().float()).sum() + aux.float()).backward()
with torch.no_grad():
    gidx = layer.topk_expert(layer.gate_logits(fx["x"].cuda()))[1]
    aidx = layer.competition_policy(fx["x"].cuda())[1]
g_gate, g_sparse, g_dense = oracle_dx_streams(fx, gidx.reshape(-1, gidx.shape[-1]), aidx.reshape(-1, aidx.shape[-1]))
print("order:", [n for n, _ in rec])
dense_hip = [g for n, g in rec if n == "DenseFFN"]          # run order: expert E-1 .. 0 ?
for j, g in enumerate(dense_hip):
    errs = [rel_l2(g.cpu().reshape(g_dense[0].shape), gd) for gd in g_dense]
    print(f"DenseFFN call {j}: closest oracle dense expert {min(range(len(errs)), key=errs.__getitem__)} err {min(errs):.3e}")
gs = [g for n, g in rec if n == "MoEFFNModules"][0]
gg = [g for n, g in rec if n == "GateSelect"][0]
print("sparse stream err", rel_l2(gs.cpu().reshape(g_sparse.shape), g_sparse), "gate stream err", rel_l2(gg.cpu().reshape(g_gate.shape), g_gate))
acc = None
for n, g in rec:
    acc = g if acc is None else acc + g
print("HIP streams summed in run order vs x.grad:", rel_l2(acc.reshape(x.shape), x.grad), " vs golden:", rel_l2(acc.cpu().reshape(x.shape), fx["x_grad"]))
print("x.grad vs golden", rel_l2(x.grad.cpu(), fx["x_grad"]))
