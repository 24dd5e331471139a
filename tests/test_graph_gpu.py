"""hipGraph capture of a whole fwd+bwd step (competesmoe_amd/graphs.py): a captured and replayed step equals the eager step bit
for bit -- outputs, aux loss, input gradient and every parameter gradient -- also on inputs that route differently from the
ones seen at capture time.  ONE capture per layer kind; the test is not looped (find the cause, fix, test once)."""
import pytest
import torch

from tests.golden_util import load

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _eager(layer, x, dy, kw=None):
    for p in layer.parameters():
        p.grad = None
    xg = x.detach().clone().requires_grad_(True)
    res = layer(xg, **(kw or {}))
    out, aux = (res[0], res[1]) if isinstance(res, tuple) else (res, None)
    loss = (out.float() * dy.float()).sum() + (aux.float() if aux is not None else 0.0)
    loss.backward()
    return (out.detach().clone(), None if aux is None else aux.detach().clone(), xg.grad.clone(),
            {n: p.grad.clone() for n, p in layer.named_parameters() if p.grad is not None})


def _check(step, layer, x, dy, ref):
    step(x)
    torch.cuda.synchronize()
    out, aux, xgrad, pgrads = ref
    loss, g_out, g_aux = step.outputs
    assert torch.equal(g_out.detach(), out)
    if aux is not None:
        assert torch.equal(g_aux.detach(), aux)
    assert torch.equal(step.static_inputs[0].grad, xgrad)
    got = {n: p.grad for n, p in layer.named_parameters() if p.grad is not None}
    assert set(got) == set(pgrads)
    for n in pgrads:
        assert torch.equal(got[n], pgrads[n]), n


def test_llava_smoe_step_captured_and_replayed_equals_eager():
    import tests.test_llava_modules_gpu as TL
    from competesmoe_amd.graphs import GraphedStep
    fx = load("llava_smoe_bf16")
    layer, dt = TL.build_layer(fx)
    x = fx["x"].to(DEV)
    dy = fx["dy"].to(DEV)
    g = torch.Generator().manual_seed(77)
    x2 = (torch.randn(x.shape, generator=g) * 1.5).to(dt).to(DEV)       # routes differently from x
    ref1 = _eager(layer, x, dy)
    ref2 = _eager(layer, x2, dy)
    assert not torch.equal(ref1[0], ref2[0])

    def fn(xs):
        out, aux, _, _ = layer(xs)
        return (out.float() * dy.float()).sum() + aux.float(), out, aux

    step = GraphedStep(fn, [x.clone().requires_grad_(True)], list(layer.parameters()))
    _check(step, layer, x, dy, ref1)
    _check(step, layer, x2, dy, ref2)
    _check(step, layer, x, dy, ref1)


def test_pretrain_smoe_step_under_autocast_captured_and_replayed_equals_eager():
    """The pretrain stack's step: fp32 master weights, bf16 autocast (operand casts, the side-stream cast of `values`, fp32
    weight-gradient outputs) inside the capture."""
    import tests.test_pretrain_modules_gpu as TP
    from competesmoe_amd.graphs import GraphedStep
    fx = load("pretrain_smoe_bf16")
    layer, kw = TP.build(fx)
    x = fx["x"].to(DEV)
    dy = fx["dy"].to(DEV)

    def run(xs):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = layer(xs, **kw)
            reg = layer.get_reg_loss()
        return out, sum(v.float() for v in reg.values())

    def eager(xin):
        for p in layer.parameters():
            p.grad = None
        xg = xin.detach().clone().requires_grad_(True)
        out, reg = run(xg)
        ((out.float() * dy).sum() + reg).backward()
        return out.detach().clone(), reg.detach().clone(), xg.grad.clone(), {n: p.grad.clone() for n, p in layer.named_parameters() if p.grad is not None}

    g = torch.Generator().manual_seed(5)
    x2 = torch.randn(x.shape, generator=g).to(DEV) * 2
    ref1, ref2 = eager(x), eager(x2)

    def fn(xs):
        out, reg = run(xs)
        return (out.float() * dy).sum() + reg, out, reg

    step = GraphedStep(fn, [x.clone().requires_grad_(True)], list(layer.parameters()))
    _check(step, layer, x, dy, ref1)
    _check(step, layer, x2, dy, ref2)
