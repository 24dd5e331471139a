"""GPU parity of the LLaVA-stack layers (HIP path) against the golden vectors captured from the reference.

Tolerances (written here, as the task statement asks): fp32 outputs/grads <= 1e-5 (max error relative to max |ref|);
bf16 <= 1e-3 relative L2 AND <= 2 bf16 ulp of max|ref| elementwise -- the HIP path reproduces the reference's rounding
sequence, so only fp32 accumulation-order noise flips a few roundings; router indices bit-exact on rows without a tie."""
import types

import pytest
import torch
import torch.nn as nn

from tests.golden_util import load, args_of, rel_l2, max_rel, unpack_experts, expert_keys, ACT_OF_KIND

pytestmark = pytest.mark.gpu
DEV = "cuda"

if torch.cuda.is_available():
    from competesmoe_amd.moe import get_moe


class TanhMLP(nn.Module):
    """fc1 / activation_fn / fc2 expert (the shape of SiglipMLP, siglip_smoe.py:85-97)."""
    def __init__(self, D, F):
        super().__init__()
        self.activation_fn = nn.GELU(approximate="tanh")
        self.fc1 = nn.Linear(D, F)
        self.fc2 = nn.Linear(F, D)

    def forward(self, x):
        return self.fc2(self.activation_fn(self.fc1(x)))


def build_layer(fx):
    m = fx["meta"]
    args = args_of(fx)
    D, F, Dout, E, K = m["D"], m["F"], m["Dout"], m["E"], m["K"]
    if m["expert_kind"] == "seq_gelu":
        mk = lambda: nn.Sequential(nn.Linear(D, F), nn.GELU(), nn.Linear(F, Dout))
    else:
        mk = lambda: TanhMLP(D, F)
    cls = get_moe(m["moe_name"])
    if m["moe_name"] in ("smoe_share", "deepseekv3"):
        layer = cls(D, Dout, E, K, mk(), args)
    else:
        layer = cls(D, Dout, E, K, nn.ModuleList([mk() for _ in range(E)]), args)
    state = dict(fx["state"])
    flips = state.pop("prob_flips", None)
    missing, unexpected = layer.load_state_dict(state, strict=False)
    assert not unexpected and set(missing) <= {"prob_flips"}, (missing, unexpected)
    dt = torch.float32 if m["dtype"] == "float32" else torch.bfloat16
    layer = layer.to(DEV).to(dt).train()
    if flips is not None:
        layer.total_steps, layer.step_warm, layer.flip_steps = 10, 0, 10
        layer.prob_flips = fx["prob_flips"].to(DEV)
        layer.set_current_steps(3)
    return layer, dt


def oracle_grads(fx, gate_idx, aff_idx):
    """Gradients of the pinned CPU oracle (tests/test_oracle_golden.py: with the REFERENCE's indices it reproduces every fixture's
    gradients) evaluated with the KERNEL's indices -- the reference for fixtures with rows where the reference's own scores tie
    exactly, torch.topk's pick among equal values being unspecified and the kernel's rule being the lowest index (VERDICT r2
    item 1a).  Returns (x_grad, {parameter name: grad})."""
    from oracle import moe_oracle as O
    m, args = fx["meta"], args_of(fx)
    experts = unpack_experts(fx, requires_grad=True)
    wg = fx["state"]["gate.weight"].clone().requires_grad_(True)
    x = fx["x"].clone().requires_grad_(True)
    act = ACT_OF_KIND[m["expert_kind"]]
    gi = gate_idx.cpu().long().view(*x.shape[:-1], -1)
    if m["moe_name"] in ("smoe_share", "deepseekv3"):
        out, aux, _, _ = O.llava_shared_forward(x, wg, experts, act, m["K"], args, m["moe_name"], out_dim=m["Dout"], forced_idx=gi)
    elif m["moe_name"] == "competesmoe":
        ai = None if aff_idx is None else aff_idx.cpu().long().view(*x.shape[:-1], -1)
        out, aux, _, _ = O.llava_competesmoe_forward(x, wg, experts, act, m["K"], args, m["competition"], out_dim=m["Dout"],
                                                     forced_idx=gi, forced_aff_idx=ai)
    else:
        out, aux, _, _ = O.llava_smoe_forward(x, wg, experts, act, m["K"], args, out_dim=m["Dout"], forced_idx=gi)
    ((out.float() * fx["dy"].float()).sum() + aux.float()).backward()
    grads = {"gate.weight": wg.grad}
    for i, ts in enumerate(experts):
        for k, t in zip(expert_keys(m["expert_kind"], i), ts):
            grads[k] = t.grad
    return x.grad, grads


def oracle_dx_streams(fx, gate_idx, aff_idx):
    """The gradient streams that meet in x, each from its own leaf copy of x through the pinned oracle: `gate` (router + its losses),
    `sparse` (the K selected experts of compute_moe) and one `dense[e]` per always-on / competing expert.  bf16 sums depend on the
    association; the autograd engine adds the streams in the order their consumers' backward nodes run (later-created first).
    Returns (gate, sparse, [dense...])."""
    import torch.nn.functional as F
    from oracle import moe_oracle as O
    m, args = fx["meta"], args_of(fx)
    experts = unpack_experts(fx)
    wg = fx["state"]["gate.weight"].clone()
    act = ACT_OF_KIND[m["expert_kind"]]
    leaf = lambda: fx["x"].clone().requires_grad_(True)
    xg, xr = leaf(), leaf()
    B, N, _ = xg.shape
    K = m["K"]
    lg = O.gate_logits(xg, wg)
    gsm = F.softmax(lg, dim=-1, dtype=torch.float32)
    gi = gate_idx.cpu().long().view(B, N, -1)
    gw = torch.gather(gsm, -1, gi)
    gw = gw / torch.sum(gw, dim=-1, keepdim=True).to(xg.dtype)
    dense = []
    if m["moe_name"] in ("smoe_share", "deepseekv3"):
        Er = wg.shape[0]
        routed = O.compute_moe(xr, gi, gw, experts[:Er], act, m["Dout"])
        dense.append(leaf())
        shared = O.expert_ffn(dense[0], *experts[Er], act)
        out = torch.zeros_like(routed) + ((shared * 0.5 + routed * 0.5) if m["moe_name"] == "smoe_share" else (shared + routed))
        aux = O.combine_loss(gi, gsm, lg, Er, args.balance_loss_coef, args.router_z_loss_coef)[0]
    else:
        E = wg.shape[0]
        dense = [leaf() for _ in range(E)]
        outs = [O.expert_ffn(dense[i], *experts[i], act) for i in range(E)]
        aff = torch.stack([torch.mean(F.softplus(o), dim=-1) for o in outs], dim=-1).to(xg.dtype)
        asm = F.softmax(aff, dim=-1, dtype=torch.float32)
        ai = aff_idx.cpu().long().view(B, N, K)
        aw = torch.gather(aff, -1, ai)
        aw = aw / torch.sum(aw, dim=-1, keepdim=True).to(xg.dtype)
        allo = torch.stack(outs, dim=2)
        topk_out = torch.gather(allo, 2, ai.unsqueeze(-1).expand(B, N, K, allo.size(-1)))
        rl = O.router_loss(gsm, asm.detach())
        if getattr(args, "hybrid", False):
            rl = rl + O.router_loss(torch.gather(gsm, -1, ai), torch.gather(asm, -1, ai).detach()) * args.router_theta
        aux = (rl * args.router_loss_coef + O.experts_diversity_loss(topk_out) * args.diversity_loss_coef
               + O.balanceloss(ai, asm, E) * args.bal_comp_loss_coef)
        out = O.compute_moe(xr, ai, aw, experts, act, m["Dout"])
    ((out.float() * fx["dy"].float()).sum() + aux.float()).backward()
    return xg.grad, xr.grad, [d.grad for d in dense]


def tols(dt):
    return (1e-5, 1e-5) if dt == torch.float32 else (1e-3, 2 * 2 ** -8)


CASES = ["smoe", "smoe_siglip", "smoe_proj", "competesmoe_router", "competesmoe_comp", "competesmoe_comp_hybrid",
         "smoe_share", "deepseekv3"]


@pytest.mark.parametrize("tag", ["fp32", "bf16"])
@pytest.mark.parametrize("case", CASES)
def test_layer_matches_reference_golden(case, tag):
    fx = load(f"llava_{case}_{tag}")
    layer, dt = build_layer(fx)
    rl, mr = tols(dt)
    x = fx["x"].to(DEV).requires_grad_(True)
    dy = fx["dy"].to(DEV)

    # --- router stage: logits / softmax / indices / weights
    with torch.no_grad():
        lg = layer.gate_logits(fx["x"].to(DEV))
        w, idx, sm = layer.topk_expert(lg)
    glg = fx["gate_logits"].to(DEV)
    assert max_rel(lg, glg) <= (1e-5 if dt == torch.float32 else 2 ** -7)
    # indices: bit-exact given identical logits; when the HIP logits differ from the CPU ones by an ulp, compare on the
    # reference's logits instead (the selection kernel itself must be exact)
    from competesmoe_amd import ops, _lib as L
    K = idx.shape[-1]
    sm2, idx2, w2 = ops.router_select(glg.reshape(-1, glg.shape[-1]).contiguous(), K, L.SEL_SOFTMAX, dt == torch.bfloat16)
    gi = fx["selected_experts"].to(DEV).reshape(-1, K)
    mism = (idx2.long() != gi).any(-1)
    if mism.any():   # only exact ties (torch.topk's CPU tie order is unspecified)
        gsm = fx["gate_softmax"].to(DEV).reshape(-1, glg.shape[-1])
        a = torch.gather(gsm, -1, idx2.long())[mism].sort(-1).values
        b = torch.gather(gsm, -1, gi)[mism].sort(-1).values
        assert torch.equal(a, b)
        assert int(mism.sum()) <= 4
    ok = ~mism
    assert torch.allclose(sm2, fx["gate_softmax"].to(DEV).reshape(sm2.shape), rtol=3e-6, atol=1e-8)
    assert torch.allclose(w2[ok], fx["weights"].to(DEV).reshape(-1, K)[ok], rtol=2e-6, atol=1e-8)

    # --- full forward / backward
    out, aux, none, infor = layer(x)
    assert none is None and out.dtype == dt and out.shape == fx["output"].shape
    rows_ok = torch.ones(out.shape[0] * out.shape[1], dtype=torch.bool, device=DEV)
    with torch.no_grad():
        # rows whose routing differs from the reference (ulp-level logit differences at a near-tie) are excluded
        # from the elementwise output check and counted
        live_idx = layer.topk_expert(layer.gate_logits(fx["x"].to(DEV)))[1].reshape(-1, K)
        if not fx["meta"]["competition"]:
            rows_ok = (live_idx.long() == gi).all(-1)
        else:
            # competition: routing = top-K of the mean-softplus affinities, computed in x.dtype.  The kernel's affinities are
            # bit-identical to the reference's; bf16 affinities are 8-bit values with exact ties, and torch.topk's choice among equal
            # values is unspecified (the kernel takes the lowest index) -- so a row may differ ONLY where the reference's own scores
            # tie, and must then select equal values.  Observed: 6 of 128 rows differ as sets, all exact ties
            # (profiles/r02/parity_report.txt); such rows are excluded from the elementwise output check.
            aw, aidx, asm, aff, _ = layer.competition_policy(fx["x"].to(DEV))
            ga = fx["aff_scores"].to(DEV).reshape(-1, aff.shape[-1])
            if dt == torch.float32:
                assert max_rel(aff.reshape(ga.shape), ga) <= 1e-5
            else:
                assert torch.equal(aff.reshape(ga.shape), ga.to(aff.dtype)), "bf16 affinities must be bit-identical to the reference's"
            ga = ga.float()
            gai = fx["aff_selected"].to(DEV).reshape(-1, K)
            rows_ok = (aidx.reshape(-1, K).long() == gai).all(-1)
            bad = ~rows_ok
            if dt == torch.float32:
                assert not bad.any()
            elif bad.any():
                a = torch.gather(ga, -1, aidx.reshape(-1, K).long())[bad].sort(-1).values
                b = torch.gather(ga, -1, gai)[bad].sort(-1).values
                assert torch.equal(a, b), "a row routed differently from the reference without an exact tie in its scores"
                print(f"bf16 competition rows whose top-K differs from the reference's (exact ties only): {float(bad.float().mean()):.4f}")
                # observed on both bf16 competition fixtures: 9 of 128 rows = 0.070 differ in ORDER or set (6 of them as sets,
                # profiles/r02/parity_report.txt; the other 3 swap two tied winners); one row of margin
                assert bad.float().mean() <= 0.08
    if not fx["meta"]["competition"]:
        assert int((~rows_ok).sum()) <= 2
    o = out.detach().reshape(-1, out.shape[-1])[rows_ok]
    g = fx["output"].to(DEV).reshape(-1, out.shape[-1])[rows_ok]
    assert rel_l2(o, g) <= rl, rel_l2(o, g)
    assert max_rel(o, g) <= mr, max_rel(o, g)
    # scalars: when a few bf16-competition rows route differently (near-ties, see above) the losses that average over
    # the selected experts (diversity, balance) move with them -> tolerance widened by the differing-row fraction
    slack = float((~rows_ok).float().mean()) if fx["meta"]["competition"] else 0.0
    assert abs(float(aux) - float(fx["aux_loss"])) <= ((2e-5 if dt == torch.float32 else 2e-3) + slack) * max(1.0, abs(float(fx["aux_loss"])))
    assert set(infor) == set(fx["infor_aux"])
    for k, v in fx["infor_aux"].items():
        assert abs(float(infor[k]) - float(v)) <= ((2e-5 if dt == torch.float32 else 5e-3) + 2 * slack) * max(1e-2, abs(float(v))), k

    ((out.float() * dy.float()).sum() + aux.float()).backward()
    ref_xg, ref_g = fx["x_grad"], fx["grads"]
    if not bool(rows_ok.all()):
        # rows routed unlike the reference's run (exact ties of its own scores, checked above): the gradients are compared with
        # the pinned oracle's under the KERNEL's indices, so no fixture's backward goes unchecked
        ref_xg, ref_g = oracle_grads(fx, live_idx, aidx if fx["meta"]["competition"] else None)
        # ... and, on the rows routed alike, still with the reference's own dx (the tied rows reach the others only through the
        # batch-level auxiliary losses, whose coefficients are 1e-2: observed <= 2e-3)
        xa = x.grad.reshape(-1, x.shape[-1])[rows_ok]
        xb = fx["x_grad"].to(DEV).reshape(-1, x.shape[-1])[rows_ok]
        assert rel_l2(xa, xb) <= 1e-2, rel_l2(xa, xb)
    if True:
        # bf16 gradients (profiles/r02/parity_report.txt): expert parameters are the reference's bits, the gate's and x's agree to
        # 1.4e-4 / 2.3e-5 now that the renormalisation's denominator gradient is rounded where autograd rounds it (the K-sum is a
        # bf16 tensor, smoe.py:44).  The shared-expert layers add one more bf16 gradient stream into x, summed in the engine's
        # order: dx 2.8e-3 there.
        gl = 4 * rl if dt == torch.float32 else 5e-4
        many = fx["meta"]["moe_name"] in ("smoe_share", "deepseekv3") or fx["meta"]["competition"]
        gx = gl
        if dt == torch.bfloat16 and many:
            # More than two bf16 gradient streams meet in x here (gate, the K routed experts, always-on / competing experts), and a
            # bf16 sum depends on its association.  The reference's autograd adds them ONE EXPERT AT A TIME in x.dtype, last-created
            # node first: measured on the shared-expert fixtures (CPU, the pinned oracle, every consumer of x given its own leaf) the
            # golden dx equals the chain `shared, routed experts E-1 .. 0, gate` BIT FOR BIT and is 2.8e-3 .. 3.4e-3 away from every
            # other order, including "(shared + round(sum of the routed)) + gate", which is what two autograd nodes compute.  Since
            # round 3 the always-on expert's dx enters the routed step's gather-sum as the chain's first addend
            # (functional.DxHandoff, csmoe_dispatch_rows_bwd `idx` / `pre`): observed 0 -- the reference's bits (2.8e-3 before, under a 4e-3 bound).
            gx = 2e-5 if not fx["meta"]["competition"] else 3e-4
            g_gate, g_sparse, g_dense = (oracle_dx_streams(fx, live_idx, aidx if fx["meta"]["competition"] else None))
            if fx["meta"]["competition"]:
                # The reference's own dx IS sparse + dense E-1 .. 0 + gate added in that order (bit for bit on CPU), and this path's
                # backward nodes run in the same order (tools/grad_order_probe.py, tools/grad_stream_probe.py).  Until round 3 the E
                # dense streams were 2e-3 .. 9e-3 off (dx 1.0e-3, expert weights up to 1.1e-3): the gradient of the bf16 affinity
                # weights was not formed as autograd forms it.  Two rounding sequences closed it (tools/comp_grad_probe.py: d w and
                # d affinity now the oracle's bits): `weights * out_exp` is a bf16 product, so d w = round(sum round(grad * out))
                # (csmoe_combine_bwd round_products), and `w / w.sum()` backpropagates through -grad * ((self / other) / other) with
                # every op rounded (router_select_bwd, SEL_RAW).  Observed now: dx 3.4e-5, streams 2.2e-5, parameters <= 1.5e-4.
                acc = g_sparse
                for gd in reversed(g_dense):
                    acc = acc + gd
                err_streams = rel_l2(x.grad, (acc + g_gate).to(DEV))
                print("competition dx against the oracle's streams added sparse, dense E-1..0, gate:", err_streams)
                assert err_streams <= 2e-4, err_streams
            else:
                print("shared-expert dx against the reference:", rel_l2(x.grad, ref_xg.to(DEV)))
        assert rel_l2(x.grad, ref_xg.to(DEV)) <= gx, rel_l2(x.grad, ref_xg.to(DEV))
        for name, p in layer.named_parameters():
            gg = ref_g.get(name)
            if gg is None:
                continue
            assert p.grad is not None, name
            # competition steps: every expert parameter receives TWO bf16 gradients (the dense pass and the sparse recompute);
            # observed <= 1.5e-4 (1.1e-3 before round 3's two rounding sequences, see above)
            gp = 5e-4 if (dt == torch.bfloat16 and fx["meta"]["competition"]) else gl
            assert rel_l2(p.grad, gg.to(DEV)) <= gp, (name, rel_l2(p.grad, gg.to(DEV)))

    # --- no-grad forward: aux is zero, same output
    with torch.no_grad():
        out_e, aux_e, _, infor_e = layer(fx["x"].to(DEV))
    if fx["meta"]["moe_name"] != "deepseekv3":
        assert float(aux_e) == 0.0 and infor_e == {}
    with torch.no_grad():
        ok_e = (layer.topk_expert(layer.gate_logits(fx["x"].to(DEV)))[1].reshape(-1, K).long() == gi).all(-1)
    oe = out_e.reshape(-1, out.shape[-1])[ok_e]
    ge = fx["output_nograd"].to(DEV).reshape(-1, out.shape[-1])[ok_e]
    assert int((~ok_e).sum()) <= 2 and rel_l2(oe, ge) <= max(rl, 1e-5)


def test_competition_with_sigmoid_normalised_scores_matches_golden():
    """args.norm_sigmoid: top-k and weights from sigmoid(affinity) (competesmoe.py:249-251), golden from the reference class."""
    test_layer_matches_reference_golden("competesmoe_comp_normsigmoid", "fp32")


def test_registry_and_errors():
    from competesmoe_amd.moe import MOE_REGISTRY, register_moe
    assert {"smoe", "competesmoe", "smoe_share", "deepseekv3"} <= set(MOE_REGISTRY)
    with pytest.raises(ValueError):
        get_moe("nope")
    with pytest.raises(ValueError):
        get_moe("competesmoe")(8, 8, 2, 1, None, types.SimpleNamespace())     # missing rate_flip / warm_up
    layer = get_moe("competesmoe")(8, 8, 2, 1, None, types.SimpleNamespace(rate_flip=0.5, warm_up=1.0, max_compete_in_iter=1))
    with pytest.raises(ValueError):
        layer.set_total_steps(10, 0, {})        # flip_steps <= 0
    with pytest.raises(AssertionError):
        layer.set_total_steps(10, None, {})


def test_schedule_matches_reference_golden():
    fx = load("llava_schedule")
    a = types.SimpleNamespace(**fx["meta"]["args"])
    cls = get_moe("competesmoe")
    layers = [cls(16, 16, 4, 2, None, a) for _ in range(4)]    # CPU modules: schedule is host logic
    # the reference draws from the CUDA generator when a GPU is visible; goldens were drawn on CPU -> compare structure
    torch.manual_seed(fx["meta"]["seed"])
    final = {}
    for i, l in enumerate(layers):
        final = l.set_total_steps(fx["meta"]["total_steps"], i, final)
    freq = sum(v.int().cpu() for v in final.values())
    assert int(freq.max()) <= a.max_compete_in_iter
    assert layers[0].step_warm == fx["step_warm"] and layers[0].flip_steps == fx["flip_steps"]
    assert layers[2].prob_flips.shape == (fx["flip_steps"],)


@pytest.mark.parametrize("reentrant", [False, True])
@pytest.mark.parametrize("case", ["smoe", "competesmoe_comp", "smoe_share"])
def test_layer_under_gradient_checkpointing(case, reentrant):
    """The reference trains with gradient checkpointing (scripts' `--gradient_checkpointing True`): the forward is executed twice
    per step and routing must be a pure function of (x, parameters, current step).  Outputs and every gradient of a
    checkpointed call are bit-identical to a plain call."""
    from torch.utils.checkpoint import checkpoint
    fx = load(f"llava_{case}_bf16")
    layer, dt = build_layer(fx)
    dy = fx["dy"].to(DEV)
    res = []
    for use_ckpt in (False, True):
        for p in layer.parameters():
            p.grad = None
        x = fx["x"].to(DEV).requires_grad_(True)

        def run(t):
            out, aux, _, _ = layer(t)
            return out, aux

        out, aux = checkpoint(run, x, use_reentrant=reentrant) if use_ckpt else run(x)
        torch.autograd.backward([out, aux.float()], [dy, torch.ones((), device=DEV)])
        res.append((out.detach().clone(), float(aux), x.grad.clone(), {n: p.grad.clone() for n, p in layer.named_parameters() if p.grad is not None}))
    (o1, a1, g1, p1), (o2, a2, g2, p2) = res
    assert torch.equal(o1, o2) and a1 == a2 and torch.equal(g1, g2)
    assert set(p1) == set(p2)
    for n in p1:
        assert torch.equal(p1[n], p2[n]), n


@pytest.mark.parametrize("case", ["smoe", "smoe_siglip", "smoe_proj"])
def test_sparse_step_as_one_autograd_node_is_bit_identical_to_the_two_node_form(case, monkeypatch):
    """`smoe` runs gate + selection + dispatch / FFN / combine as ONE autograd node (SparseMoEModules), so that the gate-path and the
    expert-path gradient of x meet inside the backward's gather-sum (csmoe_dispatch_rows_bwd's `add`) instead of in an elementwise
    add; CSMOE_FUSED_STEP=0 keeps GateSelect + MoEFFNModules as two nodes.  Outputs, losses and every gradient: the same bits."""
    from competesmoe_amd import functional as Fn
    fx = load(f"llava_{case}_bf16")
    res = []
    for flag in ("1", "0"):
        monkeypatch.setenv("CSMOE_FUSED_STEP", flag)
        calls = []
        orig = Fn.SparseMoEModules.forward

        def spy(ctx, *a):
            calls.append(1)
            return orig(ctx, *a)
        monkeypatch.setattr(Fn.SparseMoEModules, "forward", staticmethod(spy))
        layer, dt = build_layer(fx)
        x = fx["x"].to(DEV).requires_grad_(True)
        out, aux, _, info = layer(x)
        ((out.float() * fx["dy"].to(DEV).float()).sum() + aux.float()).backward()
        assert bool(calls) == (flag == "1")
        res.append((out.detach().clone(), float(aux), x.grad.clone(), {n: p.grad.clone() for n, p in layer.named_parameters() if p.grad is not None}))
        monkeypatch.setattr(Fn.SparseMoEModules, "forward", staticmethod(orig))
    (o1, a1, g1, p1), (o0, a0, g0, p0) = res
    assert torch.equal(o1, o0) and a1 == a0 and torch.equal(g1, g0) and set(p1) == set(p0)
    for n in p1:
        assert torch.equal(p1[n], p0[n]), n
    # the gate may be frozen (no d w needed through it): x still gets the expert path, the gate gets nothing
    monkeypatch.setenv("CSMOE_FUSED_STEP", "1")
    layer, dt = build_layer(fx)
    layer.gate.weight.requires_grad_(False)
    x = fx["x"].to(DEV).requires_grad_(True)
    out, aux, _, _ = layer(x)
    (out.float() * fx["dy"].to(DEV).float()).sum().backward()
    assert layer.gate.weight.grad is None and torch.isfinite(x.grad.float()).all()
