"""CPU: the oracle (oracle/moe_oracle.py) must reproduce the golden vectors captured from the reference.

Tolerances: fp32 <= 1e-5 (relative to max |ref|), bf16 <= 1e-3 relative L2 (and <= 2 bf16 ulp max);
router indices bit-exact on every row without an exact tie at the top-k boundary (tie rows are
counted and must select equal VALUES -- torch.topk on CPU has no stable tie rule).
"""
import pytest
import torch

from oracle import moe_oracle as O
from tests.golden_util import load, unpack_experts, args_of, rel_l2, max_rel, ACT_OF_KIND, expert_keys

LLAVA_SMOE = ["smoe", "smoe_siglip", "smoe_proj"]
TAGS = ["fp32", "bf16"]


def tol(tag):
    return (1e-5, 1e-5) if tag == "fp32" else (1e-3, 1.6e-2)   # (rel_l2, max_rel)


def check_idx(idx, gold_idx, values):
    """bit-exact unless the row has a tie: then the selected values must be equal."""
    mism = (idx != gold_idx).any(-1)
    n = int(mism.sum())
    if n:
        a = torch.gather(values, -1, idx)[mism]
        b = torch.gather(values, -1, gold_idx)[mism]
        assert torch.equal(a.sort(-1).values, b.sort(-1).values), "index mismatch that is not a tie"
    return n


def grads_close(fx, named, tag):
    r, m = tol(tag)
    for k, p in named.items():
        g = fx["grads"][k]
        if g is None:
            continue
        assert p.grad is not None, k
        assert rel_l2(p.grad, g) <= 4 * r, (k, rel_l2(p.grad, g))


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("case", LLAVA_SMOE)
def test_llava_smoe(case, tag):
    fx = load(f"llava_{case}_{tag}")
    meta, args = fx["meta"], args_of(fx)
    experts = unpack_experts(fx, requires_grad=True)
    wg = fx["state"]["gate.weight"].clone().requires_grad_(True)
    x = fx["x"].clone().requires_grad_(True)
    act = ACT_OF_KIND[meta["expert_kind"]]
    # stage parity
    with torch.no_grad():
        lg = O.gate_logits(fx["x"], wg)
        w, idx, sm = O.router_topk(lg, meta["K"], fx["x"].dtype)
    assert torch.equal(lg, fx["gate_logits"])
    assert torch.allclose(sm, fx["gate_softmax"], rtol=1e-6, atol=1e-7)
    nties = check_idx(idx, fx["selected_experts"], sm)
    assert nties <= 4
    out, aux, infor, st = O.llava_smoe_forward(x, wg, experts, act, meta["K"], args, out_dim=meta["Dout"],
                                               forced_idx=fx["selected_experts"])
    r, m = tol(tag)
    assert torch.allclose(st["weights"], fx["weights"], rtol=1e-6, atol=1e-7)
    assert rel_l2(out, fx["output"]) <= r and max_rel(out, fx["output"]) <= m
    assert abs(float(aux) - float(fx["aux_loss"])) <= 1e-5 * max(1.0, abs(float(fx["aux_loss"]))) * (1 if tag == "fp32" else 100)
    for k, v in fx["infor_aux"].items():
        assert abs(float(infor[k]) - float(v)) <= 2e-3 * max(1e-3, abs(float(v)))
    ((out.float() * fx["dy"].float()).sum() + aux.float()).backward()
    assert rel_l2(x.grad, fx["x_grad"]) <= 4 * r
    named = {"gate.weight": wg}
    for i, ts in enumerate(experts):
        for k, t in zip(expert_keys(meta["expert_kind"], i), ts):
            named[k] = t
    grads_close(fx, named, tag)


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("case,competing", [("competesmoe_router", False), ("competesmoe_comp", True),
                                            ("competesmoe_comp_hybrid", True)])
def test_llava_competesmoe(case, competing, tag):
    fx = load(f"llava_{case}_{tag}")
    meta, args = fx["meta"], args_of(fx)
    experts = unpack_experts(fx, requires_grad=True)
    wg = fx["state"]["gate.weight"].clone().requires_grad_(True)
    x = fx["x"].clone().requires_grad_(True)
    r, m = tol(tag)
    if competing:
        with torch.no_grad():
            aw, aidx, asm, aff, topk_out = O.competition_policy(fx["x"], unpack_experts(fx), "gelu", meta["K"])
        assert rel_l2(aff, fx["aff_scores"]) <= r
        score = aff
        nties = check_idx(aidx, fx["aff_selected"], fx["aff_scores"]) if tag == "fp32" else 0
        assert nties == 0
    out, aux, infor, st = O.llava_competesmoe_forward(
        x, wg, experts, "gelu", meta["K"], args, competing, forced_idx=fx["selected_experts"],
        forced_aff_idx=fx.get("aff_selected"))
    assert rel_l2(out, fx["output"]) <= r and max_rel(out, fx["output"]) <= m
    assert abs(float(aux) - float(fx["aux_loss"])) <= (2e-5 if tag == "fp32" else 2e-3) * max(1.0, abs(float(fx["aux_loss"])))
    assert set(infor) == set(fx["infor_aux"])
    for k, v in fx["infor_aux"].items():
        assert abs(float(infor[k]) - float(v)) <= (1e-5 if tag == "fp32" else 5e-3) * max(1e-2, abs(float(v))), k
    ((out.float() * fx["dy"].float()).sum() + aux.float()).backward()
    assert rel_l2(x.grad, fx["x_grad"]) <= 4 * r
    named = {"gate.weight": wg}
    for i, ts in enumerate(experts):
        for k, t in zip(expert_keys(meta["expert_kind"], i), ts):
            named[k] = t
    grads_close(fx, named, tag)


def test_llava_competesmoe_sigmoid_normalised_scores():
    """args.norm_sigmoid (competesmoe.py:249-251): golden from the reference class, fp32."""
    test_llava_competesmoe("competesmoe_comp_normsigmoid", True, "fp32")


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("mode", ["smoe_share", "deepseekv3"])
def test_llava_shared(mode, tag):
    fx = load(f"llava_{mode}_{tag}")
    meta, args = fx["meta"], args_of(fx)
    experts = unpack_experts(fx, requires_grad=True)
    wg = fx["state"]["gate.weight"].clone().requires_grad_(True)
    x = fx["x"].clone().requires_grad_(True)
    out, aux, infor, st = O.llava_shared_forward(x, wg, experts, "gelu", meta["K"], args, mode,
                                                 forced_idx=fx["selected_experts"])
    r, m = tol(tag)
    assert rel_l2(out, fx["output"]) <= r and max_rel(out, fx["output"]) <= m
    assert abs(float(aux) - float(fx["aux_loss"])) <= (1e-5 if tag == "fp32" else 2e-3) * max(1.0, abs(float(fx["aux_loss"])))
    ((out.float() * fx["dy"].float()).sum() + aux.float()).backward()
    assert rel_l2(x.grad, fx["x_grad"]) <= 4 * r


def test_schedule_llava_and_pretrain():
    for stack in ("llava", "pretrain"):
        fx = load(f"{stack}_schedule")
        a = fx["meta"]["args"]
        torch.manual_seed(fx["meta"]["seed"])
        prev = {}
        for i in sorted(fx["prob_flips"]):
            cur = O.make_prob_flips(fx["flip_steps"], a["rate_flip"], a["max_compete_in_iter"], prev)
            prev[i] = cur
            assert torch.equal(cur, fx["prob_flips"][i]), (stack, i)
        freq = sum(v.int() for v in prev.values())
        assert int(freq.max()) <= a["max_compete_in_iter"]


def test_bin_tokens_matches_cvmm_prepare_sel2():
    fx = load("pretrain_cvmm_sel")
    sel = fx["sel"]
    K = sel.shape[-1]
    counts, offsets, perm = O.bin_tokens(sel, 8)
    # sorted expert ids identical; within an expert the reference's sort is unstable, so compare as sets
    assert torch.equal(sel.flatten()[perm].int(), fx["sorted"].flatten())
    assert torch.equal(torch.sort(perm).values, torch.arange(perm.numel()))
    for e in range(8):
        a = set(perm[offsets[e]:offsets[e + 1]].tolist())
        b = set(fx["out_index"][offsets[e]:offsets[e + 1]].tolist())
        assert a == b
    assert torch.equal(fx["sel_index"], fx["out_index"] // K)


PRE = ["smoe", "smoe_bias", "competesmoe_router"]


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("case", PRE)
def test_pretrain_smoe(case, tag):
    fx = load(f"pretrain_{case}_{tag}")
    meta = fx["meta"]
    st = fx["state"]
    op = torch.bfloat16 if meta["bf16"] else torch.float32
    x = fx["x"].clone().requires_grad_(True)
    params = {k: st[k].clone().requires_grad_(True) for k in ("w_gate", "keys", "values")}
    bias = st["bias"].clone().requires_grad_(True) if "bias" in st else None
    o_bias = st["o_bias"].clone().requires_grad_(True) if "o_bias" in st else None
    xx = x.to(op) if meta["bf16"] else x
    lg = O.gate_logits(xx, params["w_gate"].to(xx.dtype))
    assert rel_l2(lg, fx["gate_logits"]) <= 1e-6
    w, idx, sm = O.router_topk(lg, meta["K"], x.dtype)
    out = O.pretrain_ffn(x, idx, w, params["keys"], params["values"], "relu", op, bias=bias, o_bias=o_bias)
    reg = O.entropy_balance(lg) * meta["args"]["balance_loss_coef"]
    # the fixtures come from the reference's own Triton kernels and CVMM autograd function; the restatement keeps their rounding
    # points, so bf16 agrees to a few last-place differences of the fp32 accumulation order (<= 1e-4), not just to bf16 noise
    r = 1e-5 if tag == "fp32" else 1e-4
    assert rel_l2(out, fx["output"]) <= r, rel_l2(out, fx["output"])
    assert abs(float(reg) - float(fx["reg_loss"]["mlp_ebalance"])) <= 1e-6
    ((out.float() * fx["dy"]).sum() + reg.float()).backward()
    assert rel_l2(x.grad, fx["x_grad"]) <= r
    for k, p in params.items():
        assert rel_l2(p.grad, fx["grads"][k]) <= r, k
    if bias is not None:
        assert rel_l2(bias.grad, fx["grads"]["bias"]) <= r
        assert rel_l2(o_bias.grad, fx["grads"]["o_bias"]) <= r


@pytest.mark.parametrize("tag", TAGS)
def test_pretrain_competition(tag):
    fx = load(f"pretrain_competesmoe_comp_{tag}")
    meta, st, a_ = fx["meta"], fx["state"], fx["meta"]["args"]
    op = torch.bfloat16 if meta["bf16"] else torch.float32
    x = fx["x"].clone().requires_grad_(True)
    keys, values, wg = (st[k].clone().requires_grad_(True) for k in ("keys", "values", "w_gate"))

    opd = op if meta["bf16"] else None
    xx = x.to(op) if meta["bf16"] else x
    lg = O.gate_logits(xx, wg.to(xx.dtype))
    gsm = torch.softmax(lg, -1, dtype=torch.float32)
    aw, aidx, asm, aff, topk_out = O.pretrain_dense_affinity(x, keys, values, "relu", meta["K"], x.dtype, op_dtype=opd)
    # under CUDA autocast the affinities are fp32 (softplus is an fp32-policy op): scores and indices must match the reference's
    assert aff.dtype == fx["aff_scores"].dtype == torch.float32
    assert rel_l2(aff, fx["aff_scores"]) <= (1e-5 if tag == "fp32" else 2e-4), rel_l2(aff, fx["aff_scores"])
    gi = fx["aff_selected"]
    mism = (aidx != gi).any(-1)
    if tag == "fp32":
        assert int(mism.sum()) == 0
    elif mism.any():       # fp32 scores from bf16 GEMM outputs whose accumulation order differs: near-ties only
        a = torch.gather(fx["aff_scores"], -1, aidx)[mism]
        b = torch.gather(fx["aff_scores"], -1, gi)[mism]
        assert float(mism.float().mean()) <= 0.02
        assert (a.sort(-1).values - b.sort(-1).values).abs().max() <= 2e-4 * b.abs().max()
    assert torch.allclose(torch.gather(aff, -1, gi) / torch.gather(aff, -1, gi).sum(-1, keepdim=True), fx["aff_weights"],
                          rtol=(1e-5 if tag == "fp32" else 2e-3), atol=1e-6)
    aw = torch.gather(aff, -1, gi)
    aw = aw / torch.sum(aw, dim=-1, keepdim=True).to(x.dtype)
    B, N, _ = x.shape
    topk_out = O.pretrain_dense_affinity(x, keys, values, "relu", meta["K"], x.dtype, op_dtype=opd)[4] if not mism.any() else None
    if topk_out is None:
        kk, vv = keys.to(op), values.to(op)
        eo = torch.matmul(O.ACTS["relu"](torch.matmul(xx.reshape(-1, xx.shape[-1]), kk)), vv).transpose(1, 0)
        eo = eo.reshape(B, N, *eo.shape[1:])
        topk_out = torch.gather(eo, 2, gi.unsqueeze(-1).expand(B, N, meta["K"], eo.size(-1)))
    out = O.pretrain_ffn(x, gi, aw, keys, values, "relu", op)
    div = O.pretrain_diversity_loss(topk_out, opd) * a_["balance_loss_coef_comp"] / 2
    rl = O.router_loss(gsm, asm.detach()) * a_["router_loss_coef"]
    r = 1e-5 if tag == "fp32" else 4e-3
    print("pretrain_competition", tag, "out", rel_l2(out, fx["output"]), "div", float(div), float(fx["reg_loss"]["mlp_comp_diver_loss"]),
          "rl", float(rl), float(fx["reg_loss"]["mlp_router_loss"]), "mism", int(mism.sum()))
    assert rel_l2(out, fx["output"]) <= r, rel_l2(out, fx["output"])
    assert abs(float(div) - float(fx["reg_loss"]["mlp_comp_diver_loss"])) <= (1e-7 if tag == "fp32" else 2e-6)
    assert abs(float(rl) - float(fx["reg_loss"]["mlp_router_loss"])) <= (1e-7 if tag == "fp32" else 1e-6)


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("mode", ["deepseekv2", "deepseekv3"])
def test_pretrain_deepseek(mode, tag):
    fx = load(f"pretrain_{mode}_{tag}")
    meta, st = fx["meta"], fx["state"]
    op = torch.bfloat16 if meta["bf16"] else torch.float32
    x = fx["x"].clone().requires_grad_(True)
    ps = {k: st[k].clone().requires_grad_(True) for k in ("w_gate", "keys", "values", "keys_shared", "values_shared")}
    out, lg = O.pretrain_deepseek_forward(x, ps["w_gate"], ps["keys"], ps["values"], ps["keys_shared"], ps["values_shared"],
                                          meta["K"], mode, op, x.dtype)
    reg = O.entropy_balance(lg) * meta["args"]["balance_loss_coef"]
    if tag == "fp32":
        assert rel_l2(out, fx["output"]) <= 1e-5
        assert abs(float(reg) - float(fx["reg_loss"]["mlp_ebalance"])) <= 1e-6
        ((out.float() * fx["dy"]).sum() + reg.float()).backward()
        assert rel_l2(x.grad, fx["x_grad"]) <= 4e-5
        for k, p in ps.items():
            assert rel_l2(p.grad, fx["grads"][k]) <= 4e-5, k
    else:
        # bf16 logits / sigmoids have exact ties: torch.topk's choice among equal values is unspecified, the restatement takes the
        # lowest index.  Such rows are rare outliers; every other row agrees to accumulation-order noise.
        o2, g2 = out.detach().reshape(-1, out.shape[-1]).double(), fx["output"].reshape(-1, out.shape[-1]).double()
        row_err = (o2 - g2).norm(dim=-1) / (g2.norm(dim=-1) + 1e-12)
        bad = row_err > 5e-2
        assert bad.float().mean() <= 0.01
        assert rel_l2(o2[~bad], g2[~bad]) <= 1e-4
        assert abs(float(reg) - float(fx["reg_loss"]["mlp_ebalance"])) <= 1e-6
        # ... and with the indices the reference's own torch.topk returned (fx["selected_experts"], recorded by the generator) the
        # whole output and EVERY gradient agree -- this is what pins the restatement's backward on the fixture with a tie row, and
        # what lets the GPU test use it (under the kernel's indices) as the reference for that row
        gi = fx["selected_experts"].long()
        assert gi.shape[-1] == meta["K"]
        x2 = fx["x"].clone().requires_grad_(True)
        ps2 = {k: st[k].clone().requires_grad_(True) for k in ps}
        out2, lg2 = O.pretrain_deepseek_forward(x2, ps2["w_gate"], ps2["keys"], ps2["values"], ps2["keys_shared"], ps2["values_shared"],
                                                meta["K"], mode, op, x2.dtype, forced_idx=gi)
        assert rel_l2(out2, fx["output"]) <= 1e-4, rel_l2(out2, fx["output"])
        reg2 = O.entropy_balance(lg2) * meta["args"]["balance_loss_coef"]
        ((out2.float() * fx["dy"]).sum() + reg2.float()).backward()
        assert rel_l2(x2.grad, fx["x_grad"]) <= 2e-4, rel_l2(x2.grad, fx["x_grad"])
        for k, p in ps2.items():
            assert rel_l2(p.grad, fx["grads"][k]) <= (1e-3 if k == "w_gate" else 1e-4), (k, rel_l2(p.grad, fx["grads"][k]))


@pytest.mark.parametrize("case", ["competesmoe_cosine", "competesmoe_normweight", "competesmoe_normsigmoid"])
def test_pretrain_gate_option_flags(case):
    """Cosine / weight-normalised gate (pretrain competesmoe.py:457-461) and sigmoid-normalised weights (:476-481), router branch,
    fp32 goldens from the reference class."""
    import torch.nn.functional as F
    fx = load(f"pretrain_{case}_fp32")
    meta, st, a_ = fx["meta"], fx["state"], fx["meta"]["args"]
    x = fx["x"].clone().requires_grad_(True)
    ps = {k: st[k].clone().requires_grad_(True) for k in ("w_gate", "keys", "values")}
    wgt = ps["w_gate"]
    if a_["is_cosine"]:
        lg = F.linear(F.normalize(x, p=2.0, dim=-1), F.normalize(wgt, p=2.0, dim=-1))
    elif a_["is_norm_weight"]:
        lg = F.linear(x, F.normalize(wgt, p=2.0, dim=-1))
    else:
        lg = O.gate_logits(x, wgt)
    assert rel_l2(lg, fx["gate_logits"]) <= 1e-6
    if a_["norm_sigmoid"]:
        w, idx = torch.topk(lg, meta["K"])
        w = torch.sigmoid(w / a_["scale_weight"])
        w = w / torch.sum(w, dim=-1, keepdim=True).to(x.dtype)
    else:
        w, idx, _ = O.router_topk(lg, meta["K"], x.dtype)
    out = O.pretrain_ffn(x, idx, w, ps["keys"], ps["values"], "relu", torch.float32)
    reg = O.entropy_balance(lg) * a_["balance_loss_coef"]
    assert rel_l2(out, fx["output"]) <= 1e-5
    assert abs(float(reg) - float(fx["reg_loss"]["mlp_ebalance"])) <= 1e-6
    ((out * fx["dy"]).sum() + reg).backward()
    assert rel_l2(x.grad, fx["x_grad"]) <= 4e-5
    for k, p in ps.items():
        assert rel_l2(p.grad, fx["grads"][k]) <= 4e-5, k


@pytest.mark.parametrize("case", ["competesmoe_comp_intopk", "competesmoe_comp_tribrid"])
def test_pretrain_router_loss_variants(case):
    """in_topk / tribrid router losses of the competition branch (pretrain competesmoe.py:546-593), fp32 goldens."""
    fx = load(f"pretrain_{case}_fp32")
    meta, st, a_ = fx["meta"], fx["state"], fx["meta"]["args"]
    x = fx["x"]
    keys, values, wg = (st[k] for k in ("keys", "values", "w_gate"))
    lg = O.gate_logits(x, wg)
    gw, gidx, gsm = O.router_topk(lg, meta["K"], x.dtype)
    aw, aidx, asm, aff, topk_out = O.pretrain_dense_affinity(x, keys, values, "relu", meta["K"], x.dtype)
    assert torch.equal(aidx, fx["aff_selected"])
    if a_["in_topk"]:
        rl = O.router_loss(torch.gather(gsm, -1, aidx), torch.gather(asm, -1, aidx))
    else:   # tribrid (without hybrid): full + theta * (affinity top-k + gate top-k)
        rl = O.router_loss(gsm, asm) + O.router_loss(torch.gather(gsm, -1, aidx), torch.gather(asm, -1, aidx)) * a_["router_theta"]
        rl = rl + O.router_loss(torch.gather(gsm, -1, gidx), torch.gather(asm, -1, gidx)) * a_["router_theta"]
    assert abs(float(rl * a_["router_loss_coef"]) - float(fx["reg_loss"]["mlp_router_loss"])) <= 1e-7
    out = O.pretrain_ffn(x, aidx, aw, keys, values, "relu", torch.float32)
    assert rel_l2(out, fx["output"]) <= 1e-5


@pytest.mark.parametrize("tag", TAGS)
def test_cvmm_restatement_against_the_reference_kernels(tag):
    """oracle.cvmm_ref (forward and CVMM.backward's rounding points) against the reference's own `cvmm()` run by the Triton
    interpreter on random operands, two-call protocol (tests/golden/make_golden_pretrain.py::cvmm_kernel_case)."""
    fx = load(f"pretrain_cvmm_kernels_{tag}")
    op = torch.bfloat16 if fx["meta"]["bf16"] else torch.float32
    K, E = fx["meta"]["K"], fx["meta"]["E"]
    x, keys, values, w = (fx[k].clone().requires_grad_(True) for k in ("x", "keys", "values", "w"))
    idx = fx["idx"]
    _, _, perm = O.bin_tokens(idx, E)
    ssel = idx.flatten()[perm]
    scores = torch.relu(O.cvmm_ref(x, ssel, perm // K, perm, keys, op).view(*idx.shape, -1))
    out = O.cvmm_ref(scores, ssel, perm, None, values, op, reduction_weight=w)
    r = 1e-5 if tag == "fp32" else 1e-4
    assert rel_l2(scores, fx["scores"]) <= r and rel_l2(out, fx["output"]) <= r
    (out.float() * fx["dy"]).sum().backward()
    for name, t in (("x", x), ("keys", keys), ("values", values), ("w", w)):
        assert rel_l2(t.grad, fx["grads"][name]) <= r, (name, rel_l2(t.grad, fx["grads"][name]))


@pytest.mark.parametrize("tag", TAGS)
def test_pretrain_smoe_perturbed(tag):
    """smoe_perturbed FFN form (smoe_perturbed.py:162-197): cosine gate over renormalised expert embeddings, softmax(gate / T),
    top-k of the softmax values re-normalised by a softmax; goldens from the reference class."""
    fx = load(f"pretrain_smoe_perturbed_{tag}")
    meta, st = fx["meta"], fx["state"]
    op = torch.bfloat16 if meta["bf16"] else torch.float32
    x = fx["x"].clone().requires_grad_(True)
    keys, values, esel = (st[k].clone().requires_grad_(True) for k in ("keys", "values", "expert_sel"))
    emb = O.renorm_embeddings(st["expert_embeddings"]).clone().requires_grad_(True)
    lg = O.perturbed_gate(x, esel, emb, op if meta["bf16"] else None)
    sm = torch.softmax((lg / 0.3).float(), -1).to(x.dtype)
    w, idx = O.top_softmax(sm, meta["K"])
    out = O.pretrain_ffn(x, idx, w, keys, values, "relu", op)
    reg = O.entropy_balance(lg) * meta["args"]["balance_loss_coef"]
    o2, g2 = out.detach().reshape(-1, out.shape[-1]).double(), fx["output"].reshape(-1, out.shape[-1]).double()
    bad = (o2 - g2).norm(dim=-1) / (g2.norm(dim=-1) + 1e-12) > 5e-2      # exact ties of bf16 logits (torch.topk's pick is unspecified)
    assert float(bad.float().mean()) <= (0.0 if tag == "fp32" else 0.03)
    r = 1e-5 if tag == "fp32" else 2e-4
    assert rel_l2(o2[~bad], g2[~bad]) <= r
    assert abs(float(reg) - float(fx["reg_loss"]["mlp_ebalance"])) <= 1e-6
    # gradients with the indices the reference's own torch.topk returned (fx["selected_experts"]): pins the restatement's backward
    # on the bf16 fixture too, where exact ties of the bf16 softmax values make the selection itself ambiguous
    gi = fx["selected_experts"].long()
    w = torch.softmax(torch.gather(sm, -1, gi), dim=-1)
    out = O.pretrain_ffn(x, gi, w, keys, values, "relu", op)
    assert rel_l2(out, fx["output"]) <= r, rel_l2(out, fx["output"])
    ((out.float() * fx["dy"]).sum() + reg.float()).backward()
    assert rel_l2(x.grad, fx["x_grad"]) <= 4 * r, rel_l2(x.grad, fx["x_grad"])
    for k, p in (("keys", keys), ("values", values), ("expert_sel", esel), ("expert_embeddings", emb)):
        assert rel_l2(p.grad, fx["grads"][k]) <= 4 * r, (k, rel_l2(p.grad, fx["grads"][k]))


@pytest.mark.parametrize("tag", TAGS)
def test_moe_attention_projection(tag):
    """att_forward + compute_moe of the layer built as FullMoeRelativeAttentionCore.create_param_block
    builds it (is_att = True), golden from the reference running its own cvmm kernels."""
    fx = load(f"pretrain_att_proj_{tag}")
    meta, st = fx["meta"], fx["state"]
    op = torch.bfloat16 if meta["bf16"] else torch.float32
    x = fx["x"].clone().requires_grad_(True)
    experts, esel = (st[k].clone().requires_grad_(True) for k in ("experts", "expert_sel"))
    emb = O.renorm_embeddings(st["expert_embeddings"]).clone().requires_grad_(True)
    out, val, index, logits = O.attention_projection(x, esel, emb, experts, meta["heads"], meta["E"], meta["K"], op)
    same = (index.sort(-1).values == fx["sel_index"].sort(-1).values).all(-1)          # [B, N, heads]
    assert float((~same).float().mean()) <= (0.0 if tag == "fp32" else 0.03)
    out, val, index, logits = O.attention_projection(x, esel, emb, experts, meta["heads"], meta["E"], meta["K"], op,
                                                     forced_index=fx["sel_index"])
    r = 1e-5 if tag == "fp32" else 2e-4
    assert rel_l2(logits, fx["gate_logits"]) <= r
    assert rel_l2(val, fx["sel_val"]) <= r
    assert out.shape == fx["output"].shape and out.dtype == fx["output"].dtype
    assert rel_l2(out, fx["output"]) <= r, rel_l2(out, fx["output"])
    (out.float() * fx["dy"]).sum().backward()
    assert rel_l2(x.grad, fx["x_grad"]) <= 4 * r
    for k, p in (("experts", experts), ("expert_sel", esel), ("expert_embeddings", emb)):
        assert rel_l2(p.grad, fx["grads"][k]) <= 4 * r, (k, rel_l2(p.grad, fx["grads"][k]))
