#!/usr/bin/env python3
"""Golden vectors for the pretrain stack's block around the layer by RUNNING THE REFERENCE's
RelativeMoeTransformerEncoderLayer (moe_pretrain_model/layers/transformer/relative_moe_transformer.py).

Build container only (needs /root/reference; imported, never copied).  Same stub packages as make_golden_pretrain.py (the pretrain
package does not import as shipped), plus: `layers.moe_layer` (absent upstream MoE class the transformer file
imports by name only), `wandb` and `framework.visualize.plot.CustomPlot` (imported by full_moe_relative_attention.py, never used
here: moe_attention=False) as empty stand-ins, and
the reference's own Triton cvmm kernels run by the Triton interpreter (tests/golden/ref_env.py).

The whole pre-LN transformer layer (rope attention half included) runs on CPU, in fp32 and under the CUDA bf16 autocast policy on an
fp32 residual stream (what simple_task.py:295 does on the GPU; ref_env.cuda_autocast_bf16).  Recorded is the MoE half: the tensor entering `norm2` (forward pre-hook, with
its gradient), norm2 / pkm parameters, the layer output and every gradient.  The attention half only hands the MoE half a
realistic non-leaf input.

Usage:  python tests/golden/make_golden_pretrain_block.py   (writes tests/golden/pretrain_block_*.pt)
"""
import importlib
import os
import sys
import tempfile
import types

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_pretrain as G  # noqa: E402


def _import_layer():
    G._import_reference()
    if "layers.transformer.relative_moe_transformer" in sys.modules:
        return sys.modules["layers.transformer.relative_moe_transformer"].RelativeMoeTransformerEncoderLayer
    ml = types.ModuleType("layers.moe_layer")
    ml.MoE = type("MoE", (), {})
    sys.modules["layers.moe_layer"] = ml
    sys.modules.setdefault("wandb", types.ModuleType("wandb"))
    lay = sys.modules["layers"]
    lay.moe_layer = ml
    cv = sys.modules["layers.cvmm"]
    for k in ("CVMMSel", "cvmm_prepare_sel2", "cvmm_prepare_sel"):
        if hasattr(cv, k):
            setattr(lay, k, getattr(cv, k))
    lm = sys.modules["layers.moe"]
    lm.get_moe = sys.modules["layers.moe.register"].get_moe
    lm.MoE = sys.modules["layers.moe.moe"].MoE
    fw = sys.modules["framework"]
    fwl = sys.modules["framework.layers"]
    fw.layers = fwl
    # pulled in by the MoE-attention module the transformer file imports (never instantiated here: moe_attention=False)
    lv = G._load("framework.layers.layer_with_visualization", os.path.join(G.REF, "framework", "layers", "layer_with_visualization.py"))
    fwl.LayerWithVisualization = lv.LayerWithVisualization
    vis = G._pkg("framework.visualize", os.path.join(G.REF, "framework", "visualize"))
    plot = types.ModuleType("framework.visualize.plot")
    plot.CustomPlot = type("CustomPlot", (), {})
    sys.modules["framework.visualize.plot"] = plot
    vis.plot = plot
    fw.visualize = vis
    G._pkg("layers.transformer", os.path.join(G.REF, "layers", "transformer"))
    mod = importlib.import_module("layers.transformer.relative_moe_transformer")
    return mod.RelativeMoeTransformerEncoderLayer


def run_case(name, moe_name, bf16, competition=False, B=2, N=48, D=64, E=8, F_=32, K=2, seed=0, args_kw=None):
    Layer = _import_layer()
    args = G.make_args(moe_name=moe_name, **(args_kw or {}))
    torch.manual_seed(seed + 21)
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp())       # set_total_steps appends to ./file_path.txt
    try:
        layer = Layer(D, 4, E, F_, n_layers=2, dropout=0.0, activation=F.relu, n_heads=K, preln=True, log_interval=None, args=args)
        layer.train()
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for p, std in ((layer.norm2.weight, None), (layer.norm2.bias, None)):
                p.copy_((1.0 if p is layer.norm2.weight else 0.0) + 0.1 * torch.randn(p.shape, generator=g))
        kw = {}
        fx = {"meta": dict(name=name, moe_name=moe_name, bf16=bf16, B=B, N=N, D=D, E=E, F=F_, K=K, competition=competition,
                           args=vars(args), cvmm=G.ref_env.CVMM_META)}
        upcasts = {}
        if moe_name == "competesmoe":
            torch.manual_seed(1234)
            layer.pkm.prob_flips_final = {}
            layer.pkm.set_total_steps(id_layer=0)
            if not competition:
                layer.pkm.prob_flips_final[0] = torch.zeros_like(layer.pkm.prob_flips_final[0])
            layer.pkm.set_current_steps(3)
            fx["prob_flips"] = layer.pkm.prob_flips_final[0].clone()
        kw["id_layer"] = 0
        cap = {}

        def grab(mod, inp):
            cap["mid"] = inp[0]
            inp[0].retain_grad()

        h = layer.norm2.register_forward_pre_hook(grab)
        x = torch.randn(B, N, D, generator=g)
        dy = torch.randn(B, N, D, generator=g)
        xg = x.clone().requires_grad_(True)
        layer.pkm.regularization_present = True
        # the reference's own top-k index results (gate first, affinity second on a competition step): torch.topk's choice among exactly
        # tied bf16 scores is unspecified, so the fixture records it (as tests/golden/make_golden_pretrain.py does)
        topk_idx = []
        orig_topk = torch.topk

        def spy_topk(*a_, **k_):
            r = orig_topk(*a_, **k_)
            topk_idx.append(r[1].detach().clone())
            return r
        torch.topk = spy_topk
        try:
            with G.amp(bf16, upcasts):
                out = layer(xg, None, **kw)
                reg = layer.pkm.get_reg_loss()
        finally:
            torch.topk = orig_topk
        fx["selected_experts"] = topk_idx[0]
        if competition and len(topk_idx) > 1:
            fx["aff_selected"] = topk_idx[-1]
        fx["meta"]["autocast_fp32_upcasts"] = dict(upcasts)
        h.remove()
        loss = (out.float() * dy).sum() + sum(v.float() for v in reg.values())
        loss.backward()
        mid = cap["mid"]
        fx["mid"], fx["mid_grad"] = mid.detach().clone(), mid.grad.clone()
        fx["dy"] = dy.clone()
        fx["output"] = out.detach().clone()
        fx["reg_loss"] = {k: v.detach().clone() for k, v in reg.items()}
        fx["norm2"] = {k: v.detach().clone() for k, v in layer.norm2.state_dict().items()}
        fx["norm2_grads"] = {k: p.grad.clone() for k, p in layer.norm2.named_parameters()}
        fx["state"] = {k: v.detach().clone() for k, v in layer.pkm.state_dict().items()}
        fx["grads"] = {k: (p.grad.clone() if p.grad is not None else None) for k, p in layer.pkm.named_parameters()}
        fx["eps"] = layer.norm2.eps
    finally:
        os.chdir(cwd)
    path = os.path.join(HERE, f"pretrain_block_{name}.pt")
    torch.save(fx, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB | out", tuple(out.shape), out.dtype, "mid", mid.dtype,
          {k: round(float(v), 6) for k, v in fx["reg_loss"].items()})


def main():
    torch.set_num_threads(4)
    for bf16, tag in ((False, "fp32"), (True, "bf16")):
        run_case(f"smoe_{tag}", "smoe", bf16)
        run_case(f"competesmoe_router_{tag}", "competesmoe", bf16, competition=False)
        run_case(f"competesmoe_comp_{tag}", "competesmoe", bf16, competition=True)
        run_case(f"deepseekv3_{tag}", "deepseekv3", bf16, K=3)


if __name__ == "__main__":
    main()
