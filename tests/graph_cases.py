"""Bodies of the hipGraph capture tests (tests/test_graph_gpu.py runs each in a child process: a capture that goes wrong dies
inside the HIP runtime with a segfault, which must not take the whole pytest session down)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from tests.golden_util import load  # noqa: E402

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


def llava_smoe():
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


def pretrain_smoe():
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


def refuses_stale_graph():
    """The failure mode found by bisection: an eager step on the default stream whose outputs are still referenced leaves the
    parameters' AccumulateGrad nodes on that stream; GraphedStep must refuse (a capture would crash the HIP runtime)."""
    import tests.test_llava_modules_gpu as TL
    from competesmoe_amd.graphs import GraphedStep
    fx = load("llava_smoe_bf16")
    layer, dt = TL.build_layer(fx)
    x = fx["x"].to(DEV).requires_grad_(True)
    out, aux, _, _ = layer(x)
    held = out.float().sum() + aux.float()          # a live loss: keeps the graph (and the AccumulateGrad nodes) of this step
    held.backward(retain_graph=True)

    def fn(xs):
        o, a, _, _ = layer(xs)
        return o.float().sum() + a.float(), o, a

    try:
        GraphedStep(fn, [fx["x"].to(DEV).requires_grad_(True)], list(layer.parameters()))
    except RuntimeError as e:
        assert "AccumulateGrad" in str(e)
    else:
        raise AssertionError("the stale autograd graph was not detected")
    del held, out, aux
    GraphedStep(fn, [fx["x"].to(DEV).requires_grad_(True)], list(layer.parameters()))      # and is fine once they are dropped


if __name__ == "__main__":
    {"llava_smoe": llava_smoe, "pretrain_smoe": pretrain_smoe, "refuses_stale_graph": refuses_stale_graph}[sys.argv[1]]()
    torch.cuda.synchronize()
    print("GRAPH-CASE-OK")
