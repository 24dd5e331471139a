"""GPU parity of the block around the layer: LayerNorm(+gate) / LayerNorm backward / residual-combine kernels
against torch fp32 references, and MoEBlock against the golden vectors captured from the reference's SiglipEncoderMoELayer.
Tolerances: fp32 1e-5, bf16 2e-3 relative L2 (written per assert)."""
import types

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from tests.golden_util import load, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"

if torch.cuda.is_available():
    from competesmoe_amd import ops, _lib as L
    from competesmoe_amd.moe import get_moe, MoEBlock


def tol(dt):
    return 1e-5 if dt == torch.float32 else 2e-3


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("T,D,E", [(37, 64, 8), (128, 512, 64), (1000, 4096, 64), (16, 200, 5), (70, 1152, 4), (5, 2048, 70)])
def test_layernorm_gate_forward(dt, T, D, E):
    if dt == torch.float32 and D % 4 or dt == torch.bfloat16 and D % 8:
        pytest.skip("D not a multiple of the 16-byte chunk")
    g = torch.Generator(device=DEV).manual_seed(T + D)
    x = (torch.randn(T, D, device=DEV, generator=g) * 1.7 + 0.3).to(dt)
    gamma = (1 + 0.1 * torch.randn(D, device=DEV, generator=g)).to(dt)
    beta = (0.1 * torch.randn(D, device=DEV, generator=g)).to(dt)
    wg = (torch.randn(E, D, device=DEV, generator=g) * 0.05).to(dt)
    xn, mean, rstd, logits = ops.layernorm_gate(x, gamma, beta, 1e-6, wg)
    ref = F.layer_norm(x.float(), (D,), gamma.float(), beta.float(), 1e-6)
    assert rel_l2(xn, ref.to(dt)) <= tol(dt)
    assert rel_l2(mean, x.float().mean(-1)) <= 1e-5 and rel_l2(rstd, 1 / torch.sqrt(x.float().var(-1, unbiased=False) + 1e-6)) <= 1e-5
    # the gate sees the ROUNDED xn: compare with an fp32 product of the kernel's own xn
    assert rel_l2(logits, (xn.float() @ wg.float().t()).to(dt)) <= tol(dt)
    # without gate / without affine parameters
    xn2, _, _, lg2 = ops.layernorm_gate(x, None, None, 1e-5, None)
    assert lg2 is None and rel_l2(xn2, F.layer_norm(x.float(), (D,), None, None, 1e-5).to(dt)) <= tol(dt)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("T,D", [(37, 64), (513, 512), (3000, 4096), (9, 1152), (2100, 2048)])
def test_layernorm_backward(dt, T, D):
    g = torch.Generator(device=DEV).manual_seed(T * 7 + D)
    x = (torch.randn(T, D, device=DEV, generator=g) * 1.3).to(dt)
    gamma = (1 + 0.1 * torch.randn(D, device=DEV, generator=g)).to(dt)
    beta = (0.1 * torch.randn(D, device=DEV, generator=g)).to(dt)
    d1 = torch.randn(T, D, device=DEV, generator=g).to(dt)
    d2 = torch.randn(T, D, device=DEV, generator=g).to(dt)
    res = torch.randn(T, D, device=DEV, generator=g).to(dt)
    _, mean, rstd, _ = ops.layernorm_gate(x, gamma, beta, 1e-6, None)
    xr = x.float().requires_grad_(True)
    gr = gamma.float().requires_grad_(True)
    br = beta.float().requires_grad_(True)
    dsum = (d1.float() + d2.float()).to(dt).float()                 # the two gradient streams are summed in x.dtype
    F.layer_norm(xr, (D,), gr, br, 1e-6).backward(dsum)
    dx, dgam, dbet = ops.layernorm_bwd(d1, x, gamma, mean, rstd, add=res, dxn2=d2)
    ref_dx = (xr.grad.to(dt).float() + res.float()).to(dt)
    assert rel_l2(dx, ref_dx) <= 2 * tol(dt)
    assert rel_l2(dgam, gr.grad) <= (1e-4 if dt == torch.float32 else 4e-3)      # xhat is recomputed from the rounded x
    assert rel_l2(dbet, br.grad) <= 1e-5
    dx1, _, _ = ops.layernorm_bwd(dsum.to(dt), x, gamma, mean, rstd, want_affine_grads=False)
    assert rel_l2(dx1, xr.grad.to(dt)) <= 2 * tol(dt)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_combine_residual_rounding(dt):
    T, K, D, E = 300, 2, 256, 8
    g = torch.Generator(device=DEV).manual_seed(5)
    idx = torch.rand(T, E, device=DEV, generator=g).topk(K, -1).indices.int()
    w = torch.rand(T, K, device=DEV, generator=g)
    y = torch.randn(T * K, D, device=DEV, generator=g).to(dt)
    res = torch.randn(T, D, device=DEV, generator=g).to(dt)
    bins = ops.bin_tokens(idx, E)
    plain = ops.combine(y, bins, idx, w, L.COMBINE_SEQ, T)
    fused = ops.combine(y, bins, idx, w, L.COMBINE_SEQ, T, residual=res)
    assert torch.equal(fused, plain + res)            # bit-exact: one more x.dtype addition after the rounded result


def build_block(fx, dt):
    m = fx["meta"]
    args = types.SimpleNamespace(**m["args"])
    act = nn.GELU(approximate="tanh")
    experts = nn.ModuleList()
    for _ in range(m["E"]):
        e = nn.Module()
        e.fc1, e.activation_fn, e.fc2 = nn.Linear(m["D"], m["F"]), act, nn.Linear(m["F"], m["D"])
        experts.append(e)
    layer = get_moe(m["moe_name"])(m["D"], m["D"], m["E"], m["K"], experts, args)
    missing, unexpected = layer.load_state_dict({k: v for k, v in fx["moe_state"].items() if k != "prob_flips"}, strict=False)
    assert not unexpected and set(missing) <= {"prob_flips"}
    ln = nn.LayerNorm(m["D"], eps=m["eps"])
    ln.load_state_dict(fx["ln_state"])
    blk = MoEBlock(ln, layer).to(dt).to(DEV).train()
    if "competesmoe" in m["moe_name"]:
        layer.set_total_steps(10, 0, {})
        layer.prob_flips = fx["prob_flips"].to(DEV)
        layer._flips_host = None
        layer.set_current_steps(3)
    return blk


@pytest.mark.parametrize("tag", ["fp32", "bf16"])
@pytest.mark.parametrize("case", ["smoe", "competesmoe_router", "competesmoe_comp"])
def test_block_matches_reference_golden(case, tag):
    fx = load(f"block_{case}_{tag}")
    dt = torch.float32 if tag == "fp32" else torch.bfloat16
    m = fx["meta"]
    blk = build_block(fx, dt)
    x = fx["x_mid"].to(DEV).requires_grad_(True)
    out, aux, _ids, infor = blk(x)
    ((out.float() * fx["dy"].to(DEV).float()).sum() + aux.float()).backward()
    r = tol(dt)
    # rows whose selection differs from the reference's (near-ties in bf16 logits / affinities) are compared separately
    with torch.no_grad():
        xn, _, _, lg = ops.layernorm_gate(fx["x_mid"].to(DEV).reshape(-1, m["D"]), blk.layer_norm2.weight, blk.layer_norm2.bias,
                                          m["eps"], blk.moelayer.gate.weight)
        assert rel_l2(xn.cpu(), fx["xn"].reshape(-1, m["D"])) <= r
        assert rel_l2(lg.cpu(), fx["gate_logits"].reshape(-1, m["E"])) <= 2 * r
        if m["competition"]:
            sel = blk.moelayer.competition_policy(xn.view(m["B"], m["N"], m["D"]))[1].cpu().long()
            gold = fx["aff_selected"].long()
        else:
            sel = blk.moelayer.topk_expert(lg.view(m["B"], m["N"], m["E"]))[1].cpu().long()
            gold = fx["selected_experts"].long()
    same = (sel.sort(-1).values == gold.sort(-1).values).all(-1).reshape(-1)
    # bf16 competition: the affinities are bf16 means of softplus (~3 significant digits) -> near-ties between experts are
    # common and resolve with the accumulation order; the same slack as tests/test_llava_modules_gpu.py
    need = 0.99 if tag == "fp32" else (0.90 if m["competition"] else 0.99)     # observed: 88 of 96 rows / all rows
    assert same.float().mean() >= need
    o, go = out.detach().cpu().reshape(-1, m["D"]), fx["output"].reshape(-1, m["D"])
    assert rel_l2(o[same], go[same]) <= r
    # bf16 gradients against the fixture, observed: dx 1.9e-3, LayerNorm weight / bias 6.5e-3 / 4.8e-3 (the reference's CPU LayerNorm
    # backward, see below), gate the reference's bits, experts 1.8e-4 (1.9e-3 on the competition step)
    gdx, gln, ggate, gexp = (4 * r, 8 * r, 8 * r + 1e-4, 8 * r) if tag == "fp32" else (4e-3, 1e-2, 1e-3, 4e-3)
    ref_aux, ref_dx, ref_ln, ref_moe = fx["aux_loss"], fx["x_mid_grad"], fx["ln_grads"], fx["moe_grads"]
    if not bool(same.all()):
        # rows routed unlike the reference's run (exact ties of its bf16 affinities): gradients against the oracle, pinned by
        # tests/test_block_oracle_golden.py, evaluated with the KERNEL's indices -- no fixture's backward goes unchecked
        assert m["competition"] and tag == "bf16", "tie rows are expected on the bf16 competition step only"
        from tests.test_block_oracle_golden import oracle_block, experts_of
        xo = fx["x_mid"].clone().requires_grad_(True)
        lnw, lnb = (fx["ln_state"][k].clone().requires_grad_(True) for k in ("weight", "bias"))
        wgo = fx["moe_state"]["gate.weight"].clone().requires_grad_(True)
        ex = experts_of(fx)
        o_out, o_aux, _, _, _ = oracle_block(fx, xo, lnw, lnb, wgo, ex, aff_idx=sel.view(m["B"], m["N"], -1))
        ((o_out.float() * fx["dy"].float()).sum() + o_aux.float()).backward()
        assert rel_l2(o, o_out.detach().reshape(-1, m["D"])) <= r
        ref_aux, ref_dx, ref_ln = o_aux.detach(), xo.grad, {"weight": lnw.grad, "bias": lnb.grad}
        ref_moe = {"gate.weight": wgo.grad}
        for i, ts in enumerate(ex):
            for t, k in zip(ts, ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")):
                ref_moe[f"experts.{i}.{k}"] = t.grad
    if tag == "bf16":
        # What the loose LayerNorm / dx bounds above are made of: the reference's CPU bf16 LayerNorm BACKWARD is itself inexact -- from
        # the reference's own dxn (the pinned oracle's, bit-equal to the fixture everywhere else), LayerNorm backward with fp32
        # accumulation (what the CUDA kernel does, and this one) differs from the fixture by exactly the numbers observed here (dx
        # 1.85e-3, d gamma 6.5e-3, d beta 4.7e-3: 41 of its 64 d beta elements are not the correctly rounded column sums).  So the
        # sharp check is against THAT: the oracle's dxn through an fp32 LayerNorm backward, rounded once.
        from tests.test_block_oracle_golden import oracle_block, experts_of
        xo = fx["x_mid"].clone().requires_grad_(True)
        lnw, lnb = (fx["ln_state"][k].clone().requires_grad_(True) for k in ("weight", "bias"))
        o_out, o_aux, _, _, o_xn = oracle_block(fx, xo, lnw, lnb, fx["moe_state"]["gate.weight"].clone().requires_grad_(True), experts_of(fx),
                                               aff_idx=None if bool(same.all()) else sel.view(m["B"], m["N"], -1))
        o_xn.retain_grad()
        ((o_out.float() * fx["dy"].float()).sum() + o_aux.float()).backward()
        xf = fx["x_mid"].float().requires_grad_(True)
        gf, bf_ = (fx["ln_state"][k].float().requires_grad_(True) for k in ("weight", "bias"))
        F.layer_norm(xf, (m["D"],), gf, bf_, m["eps"]).backward(o_xn.grad.float())
        dx_exact = (xf.grad.bfloat16().float() + fx["dy"].float()).bfloat16()
        sharp = {"dx": rel_l2(x.grad.cpu(), dx_exact), "ln.weight": rel_l2(blk.layer_norm2.weight.grad.cpu(), gf.grad.bfloat16()),
                 "ln.bias": rel_l2(blk.layer_norm2.bias.grad.cpu(), bf_.grad.bfloat16())}
        print("block", case, tag, "against fp32-accumulated LayerNorm backward of the oracle's dxn:", {k: f"{v:.2e}" for k, v in sharp.items()})
        # observed: smoe / router step 0 (the bits of exact arithmetic on the reference's dxn); competition step 8e-4 / 1.9e-3 / 1.2e-3
        # (its dxn carries the E dense streams discussed in tests/test_llava_modules_gpu.py)
        assert all(v <= (3e-3 if m["competition"] else 1e-6) for v in sharp.values()), sharp
    assert abs(float(aux.detach()) - float(ref_aux)) <= 4 * r * max(1.0, abs(float(ref_aux)))
    errs = {"dx": rel_l2(x.grad.cpu(), ref_dx), "ln.weight": rel_l2(blk.layer_norm2.weight.grad.cpu(), ref_ln["weight"]),
            "ln.bias": rel_l2(blk.layer_norm2.bias.grad.cpu(), ref_ln["bias"]),
            "gate": rel_l2(blk.moelayer.gate.weight.grad.cpu(), ref_moe["gate.weight"])}
    params = dict(blk.moelayer.named_parameters())
    errs["experts"] = max(rel_l2(params[k].grad.cpu(), g) for k, g in ref_moe.items() if g is not None and k.startswith("experts."))
    print("block", case, tag, "rows routed alike", float(same.float().mean()), {k: f"{v:.2e}" for k, v in errs.items()})
    assert errs["dx"] <= gdx and errs["ln.weight"] <= gln and errs["ln.bias"] <= gln and errs["gate"] <= ggate and errs["experts"] <= gexp, errs
    # residual must have been taken by the combine epilogue, not by a separate add
    assert blk.moelayer._residual is None and blk.moelayer._pre_logits is None


def test_block_equals_unfused_composition():
    """MoEBlock == layer(LayerNorm(x)) + x composed from the unfused pieces (same kernels otherwise): bit-exact in fp32 forward."""
    D, F_, E, K = 128, 192, 8, 2
    torch.manual_seed(0)
    args = types.SimpleNamespace(balance_loss_coef=0.01, router_z_loss_coef=0.001)
    experts = nn.ModuleList([nn.Sequential(nn.Linear(D, F_), nn.GELU(), nn.Linear(F_, D)) for _ in range(E)])
    layer = get_moe("smoe")(D, D, E, K, experts, args).to(DEV)
    ln = nn.LayerNorm(D).to(DEV)
    blk = MoEBlock(ln, layer)
    x = torch.randn(3, 50, D, device=DEV)
    a = blk(x)[0]
    xn = ops.layernorm_gate(x.reshape(-1, D), ln.weight, ln.bias, ln.eps, None)[0].view_as(x)
    b = x + layer(xn)[0]
    assert rel_l2(a, b) <= 1e-6
    share = get_moe("smoe_share")(D, D, E, K, nn.Sequential(nn.Linear(D, F_), nn.GELU(), nn.Linear(F_, D)), args).to(DEV)
    blk2 = MoEBlock(ln, share)                          # not a single combine: residual added outside
    assert rel_l2(blk2(x)[0], x + share(xn)[0]) <= 1e-6
