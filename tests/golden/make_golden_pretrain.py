#!/usr/bin/env python3
"""Generate golden vectors for the LM-pretrain-stack MoE layers by RUNNING THE REFERENCE classes AND ITS TRITON KERNELS.

Build-container only (needs /root/reference).  tests/golden/ref_env.py sets up the environment: stub packages so the pretrain
package imports, the reference's own `cvmm()` / `CVMM` autograd function / `cvmm_kernel` /
`cvmm_backward_kernel3` executed by the Triton interpreter on CPU (one configuration of each kernel's own autotune list, no
benchmarking), and -- for the bf16 cases -- the CUDA autocast policy the reference trains under (simple_task.py:295)
reproduced on CPU tensors.  Gating, top-k, index preparation, both cvmm calls and their backward, losses, schedule and mixins
are all the reference's own code; nothing on the path is restated here.  meta["cvmm"] / meta["autocast_fp32_upcasts"] record
how each fixture was made.

Usage:  python tests/golden/make_golden_pretrain.py   (writes tests/golden/pretrain_*.pt)
"""
import contextlib
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_env  # noqa: E402  (sets TRITON_INTERPRET=1 before triton is imported)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

REF = ref_env.REF
_load, _pkg = ref_env._load, ref_env._pkg
_import_reference = ref_env.import_reference


def amp(bf16, log):
    return ref_env.cuda_autocast_bf16(log) if bf16 else contextlib.nullcontext()


def make_args(**kw):
    base = dict(
        moe_name="smoe", stop_after=10, warm_up=0.0, rate_flip=1.0, max_compete_in_iter=8,
        balance_loss_coef=0.01, balance_loss_coef_comp=0.02, router_loss_coef=0.03, router_theta=0.5,
        in_topk=False, hybrid=False, tribrid=False, balance_affinity=False,
        is_cosine=False, is_norm_weight=False, norm_sigmoid=False, scale_weight=1.0, test_only=False,
    )
    base.update(kw)
    return types.SimpleNamespace(**base)


def run_case(name, moe_name, bf16, *, B=2, N=64, D=64, E=8, F_=32, K=2, competition=False,
             args_kw=None, bias=False, seed=0, full=True):
    get_moe = _import_reference()
    args = make_args(moe_name=moe_name, **(args_kw or {}))
    torch.manual_seed(seed)
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)  # set_total_steps appends to ./file_path.txt (competesmoe.py:218-221)
    try:
        layer = get_moe(moe_name)(D, E, F_, n_heads=K, activation=F.relu, bias=bias, log_interval=None, args=args)
        layer.train()
        if hasattr(layer, "expert_embeddings"):      # smoe_perturbed leaves it torch.empty (smoe_perturbed.py:100-104)
            with torch.no_grad():
                layer.expert_embeddings.copy_(torch.randn(layer.expert_embeddings.shape, generator=torch.Generator().manual_seed(seed + 9)))
        if bias:
            g0 = torch.Generator().manual_seed(seed + 5)
            for p in (layer.bias, layer.o_bias, getattr(layer, "bias_shared", None)):
                if p is not None:
                    p.data = torch.randn(p.shape, generator=g0) * 0.1
        fx = {"meta": dict(name=name, moe_name=moe_name, bf16=bf16, B=B, N=N, D=D, E=E, F=F_, K=K, bias=bias,
                           competition=competition, args=vars(args),
                           cvmm=ref_env.CVMM_META)}
        upcasts = {}
        kw = {}
        if moe_name == "competesmoe":
            torch.manual_seed(1234)
            layer.prob_flips_final = {}
            pf = layer.set_total_steps(id_layer=0)
            if not competition:
                layer.prob_flips_final[0] = torch.zeros_like(layer.prob_flips_final[0])
            layer.set_current_steps(3)
            fx["prob_flips"] = layer.prob_flips_final[0].clone()
            kw["id_layer"] = 0
        g = torch.Generator().manual_seed(seed + 1)
        x = torch.randn(B, N, D, generator=g)
        dy = torch.randn(B, N, D, generator=g)
        fx["state"] = {k: v.clone() for k, v in layer.state_dict().items()}
        if full:
            fx["x"], fx["dy"] = x.clone(), dy.clone()
        else:
            fx["x_seed"] = seed + 1

        xg = x.clone().requires_grad_(True)
        layer.regularization_present = True
        # what the reference's forward selected: the index result of every torch.topk call it makes (gate top-k first; on a
        # competition step the affinity top-k second) -- torch.topk's choice among exactly tied bf16 scores is unspecified, so the
        # fixture records it (tests force the oracle to these rows to pin its gradients where a tie exists)
        topk_idx = []
        orig_topk = torch.topk

        def spy_topk(*a_, **k_):
            r = orig_topk(*a_, **k_)
            topk_idx.append(r[1].detach().clone())
            return r
        torch.topk = spy_topk
        try:
            with amp(bf16, upcasts):
                out = layer(xg, **kw)
                reg = layer.get_reg_loss()
        finally:
            torch.topk = orig_topk
        fx["selected_experts"] = topk_idx[0]
        fx["output"] = out.detach().clone()
        fx["reg_loss"] = {k: v.detach().clone() for k, v in reg.items()}
        loss = (out.float() * dy).sum() + sum(v.float() for v in reg.values())
        loss.backward()
        if full:
            fx["x_grad"] = xg.grad.clone()
            fx["grads"] = {k: (p.grad.clone() if p.grad is not None else None) for k, p in layer.named_parameters()}
        else:
            fx["x_grad_sum"] = xg.grad.double().sum()
            fx["x_grad_norm"] = xg.grad.double().norm()
            fx["grad_norms"] = {k: p.grad.double().norm() for k, p in layer.named_parameters() if p.grad is not None}
        with torch.no_grad(), amp(bf16, upcasts):
            gl = layer.compute_gate(x)
            fx["gate_logits"] = gl.clone()
            if competition:
                aw, aidx, asm, aff, _ = layer.competition_policy_mlp_faster(x)
                fx["aff_weights"], fx["aff_selected"] = aw.clone(), aidx.clone()
                fx["aff_softmax"], fx["aff_scores"] = asm.clone(), aff.clone()
        fx["meta"]["autocast_fp32_upcasts"] = dict(upcasts)
    finally:
        os.chdir(cwd)
    path = os.path.join(HERE, f"pretrain_{name}.pt")
    torch.save(fx, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB | out", tuple(out.shape),
          {k: round(float(v), 6) for k, v in fx["reg_loss"].items()})


def cvmm_index_case():
    """Pin cvmm_prepare_sel2 index semantics (reference function, integer work)."""
    _import_reference()
    cv = sys.modules["layers.cvmm"]
    g = torch.Generator().manual_seed(3)
    sel = torch.randint(0, 8, (2, 16, 2), generator=g, dtype=torch.int32)
    s = cv.cvmm_prepare_sel2(sel)
    fx = {"sel": sel, "sorted": s.sel.clone(), "sel_index": s.sel_index.clone(), "out_index": s.out_index.clone()}
    torch.save(fx, os.path.join(HERE, "pretrain_cvmm_sel.pt"))
    print("wrote pretrain_cvmm_sel.pt")


def cvmm_kernel_case(bf16):
    """The reference's `cvmm()` itself (both Triton kernels + the CVMM autograd function) on random operands with the two-call
    protocol of the MoE layers (smoe.py:237-248): scores = relu(cvmm(x, sel, keys)); out = cvmm(scores, sel', values) with
    reduction weights.  Outputs and the gradients of x, keys, values, w."""
    _import_reference()
    cv = sys.modules["layers.cvmm"]
    T, K, E, D, Fh = 300, 2, 8, 32, 48
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, T // 2, D, generator=g).requires_grad_(True)
    keys = (torch.randn(E, D, Fh, generator=g) / 6).requires_grad_(True)
    values = (torch.randn(E, Fh, D, generator=g) / 6).requires_grad_(True)
    idx = torch.rand(2, T // 2, E, generator=g).topk(K, -1).indices
    w = torch.rand(2, T // 2, K, generator=g).requires_grad_(True)
    dy = torch.randn(2, T // 2, D, generator=g)
    with amp(bf16, {}):
        sel = cv.cvmm_prepare_sel2(idx.int())
        scores = torch.relu(cv.cvmm(x, sel, keys))
        sel2 = sel.clone()
        sel2.reduction_weight = w
        sel2.sel_index = sel2.out_index
        sel2.out_index = None
        out = cv.cvmm(scores, sel2, values)
    (out.float() * dy).sum().backward()
    fx = {"meta": dict(bf16=bf16, T=T, K=K, E=E, D=D, F=Fh, cvmm=ref_env.CVMM_META),
          "x": x.detach().clone(), "keys": keys.detach().clone(), "values": values.detach().clone(), "idx": idx.clone(),
          "w": w.detach().clone(), "dy": dy, "scores": scores.detach().clone(), "output": out.detach().clone(),
          "grads": {"x": x.grad.clone(), "keys": keys.grad.clone(), "values": values.grad.clone(), "w": w.grad.clone()}}
    tag = "bf16" if bf16 else "fp32"
    torch.save(fx, os.path.join(HERE, f"pretrain_cvmm_kernels_{tag}.pt"))
    print(f"wrote pretrain_cvmm_kernels_{tag}.pt", out.dtype, scores.dtype)


def attention_projection_case(bf16):
    """MoE attention projection: the layer exactly as FullMoeRelativeAttentionCore.create_param_block builds
    it (full_moe_relative_attention.py:267-300: n_experts = experts per head x heads, expert_size = 1, is_att = True), then the two
    calls the attention makes: sel = att_forward(x, n_copies = heads, n_experts = E) (:375) and compute_moe(x, sel) (:383-388)."""
    get_moe = _import_reference()
    heads, E, K, Din, Dout, B, N = 4, 4, 2, 32, 16, 2, 24
    args = make_args(moe_name="smoe_perturbed")
    torch.manual_seed(31)
    std = 0.3
    layer = get_moe("smoe_perturbed")(n_experts=E * heads, dmodel=Din, out_dmodel=Dout * heads, n_heads=heads, topk=K, expert_size=1,
                                      args=args, is_att=True, std=std, inp_expert=Din, out_expert=Dout, selection_dropout=0.0,
                                      expert_dropout=0.0, std_gate=std, std_expert=std)
    layer.train()
    g = torch.Generator().manual_seed(32)
    with torch.no_grad():
        layer.expert_embeddings.copy_(torch.randn(layer.expert_embeddings.shape, generator=g))
    x = torch.randn(B, N, Din, generator=g)
    dy = torch.randn(B, N, heads, Dout, generator=g)
    fx = {"meta": dict(bf16=bf16, heads=heads, E=E, K=K, Din=Din, Dout=Dout, B=B, N=N, std=std, args=vars(args),
                       cvmm=ref_env.CVMM_META),
          "state": {k: v.clone() for k, v in layer.state_dict().items()}, "x": x.clone(), "dy": dy.clone()}
    xg = x.clone().requires_grad_(True)
    up = {}
    with amp(bf16, up):
        sel = layer.att_forward(xg, n_copies=heads, n_experts=E)
        out = layer.compute_moe(xg, sel)
    (out.float() * dy).sum().backward()
    fx["meta"]["autocast_fp32_upcasts"] = up
    fx["sel_val"], fx["sel_index"], fx["gate_logits"] = sel.sel_val.detach().clone(), sel.raw_sel_index.clone(), sel.raw_sel.detach().clone()
    fx["output"], fx["x_grad"] = out.detach().clone(), xg.grad.clone()
    fx["grads"] = {k: (p.grad.clone() if p.grad is not None else None) for k, p in layer.named_parameters()}
    tag = "bf16" if bf16 else "fp32"
    torch.save(fx, os.path.join(HERE, f"pretrain_att_proj_{tag}.pt"))
    print(f"wrote pretrain_att_proj_{tag}.pt", tuple(out.shape), out.dtype, {k: (None if v is None else tuple(v.shape)) for k, v in fx["grads"].items()})


def schedule_case():
    get_moe = _import_reference()
    args = make_args(moe_name="competesmoe", stop_after=40, warm_up=0.25, rate_flip=0.6, max_compete_in_iter=2)
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp())
    try:
        layers = [get_moe("competesmoe")(16, 4, 8, n_heads=2, activation=F.relu, log_interval=None, args=args)
                  for _ in range(4)]
        torch.manual_seed(7)
        prev = {}
        for i, l in enumerate(layers):   # transformer_lm_mixin.py:259-267 protocol
            l.prob_flips_final = prev
            prev = l.set_total_steps(id_layer=i)
    finally:
        os.chdir(cwd)
    fx = {"meta": dict(seed=7, args=vars(args)), "prob_flips": {int(k): v.clone() for k, v in prev.items()},
          "step_warm": layers[0].step_warm, "flip_steps": layers[0].flip_steps}
    torch.save(fx, os.path.join(HERE, "pretrain_schedule.pt"))
    print("wrote pretrain_schedule.pt")


def main():
    torch.set_num_threads(4)
    for bf16, tag in ((False, "fp32"), (True, "bf16")):
        run_case(f"smoe_{tag}", "smoe", bf16)
        run_case(f"smoe_bias_{tag}", "smoe", bf16, bias=True)
        run_case(f"competesmoe_router_{tag}", "competesmoe", bf16, competition=False)
        run_case(f"competesmoe_comp_{tag}", "competesmoe", bf16, competition=True)
        run_case(f"competesmoe_comp_hybrid_{tag}", "competesmoe", bf16, competition=True,
                 args_kw=dict(hybrid=True, balance_affinity=True))
        run_case(f"deepseekv2_{tag}", "deepseekv2", bf16, K=3)
        run_case(f"deepseekv3_{tag}", "deepseekv3", bf16, K=3)
    # option flags of the pretrain CompeteSMoE (competesmoe.py:435-464, 546-593), fp32
    run_case("competesmoe_cosine_fp32", "competesmoe", False, competition=False, args_kw=dict(is_cosine=True))
    run_case("competesmoe_normweight_fp32", "competesmoe", False, competition=False, args_kw=dict(is_norm_weight=True))
    run_case("competesmoe_normsigmoid_fp32", "competesmoe", False, competition=False, args_kw=dict(norm_sigmoid=True, scale_weight=2.0))
    run_case("competesmoe_comp_intopk_fp32", "competesmoe", False, competition=True, args_kw=dict(in_topk=True))
    run_case("competesmoe_comp_tribrid_fp32", "competesmoe", False, competition=True, args_kw=dict(tribrid=True))
    # BASELINE config 1: D=256, E=8, K=2, F=128, T=1024 as [4,256] -- checksums only
    run_case("config1_smoe_fp32", "smoe", False, B=4, N=256, D=256, E=8, F_=128, K=2, full=False)
    for bf16, tag in ((False, "fp32"), (True, "bf16")):
        run_case(f"smoe_perturbed_{tag}", "smoe_perturbed", bf16)
        attention_projection_case(bf16)
    cvmm_index_case()
    cvmm_kernel_case(False)
    cvmm_kernel_case(True)
    schedule_case()


if __name__ == "__main__":
    main()
