"""GPU edge cases of the layer against the CPU oracle: empty batch, a single expert, K == E, every token routed to ONE expert
(63 empty bins + one ragged maximum-size bin), sizes that are not multiples of any tile, non-contiguous input."""
import os
import types

import pytest
import torch
import torch.nn as nn

from tests.golden_util import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"

if torch.cuda.is_available():
    from competesmoe_amd.moe import get_moe
    from oracle import moe_oracle as O

ARGS = types.SimpleNamespace(balance_loss_coef=0.01, router_z_loss_coef=0.001)


def make(D, F, E, K, dt, seed=0, bias=True):
    torch.manual_seed(seed)
    experts = nn.ModuleList([nn.Sequential(nn.Linear(D, F, bias=bias), nn.GELU(), nn.Linear(F, D, bias=bias)) for _ in range(E)])
    return get_moe("smoe")(D, D, E, K, experts, ARGS).to(dt)


def oracle_run(layer, x, dy, K):
    ex = [tuple(None if p is None else p.detach().clone().requires_grad_(True) for p in (m[0].weight, m[0].bias, m[2].weight, m[2].bias))
          for m in layer.experts]
    wg = layer.gate.weight.detach().clone().requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    out, aux, _, st = O.llava_smoe_forward(xo, wg, ex, "gelu", K, ARGS)
    ((out.float() * dy.float()).sum() + aux.float()).backward()
    return out, aux, xo.grad, wg.grad, ex, st


def run_both(D, F, E, K, B, N, dt, gate_bias_row=None, bias=True, strided=False):
    layer = make(D, F, E, K, dt, bias=bias)
    if gate_bias_row is not None:
        with torch.no_grad():
            layer.gate.weight.zero_()
            layer.gate.weight[gate_bias_row] = 1.0     # with x > 0 every token prefers this expert
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, N, D, generator=g).to(dt)
    if gate_bias_row is not None:
        x = x.abs() + 0.1
    dy = torch.randn(B, N, D, generator=g).to(dt)
    o_out, o_aux, o_dx, o_dwg, ex, st = oracle_run(layer, x, dy, K)
    layer = layer.to(DEV)
    if strided:
        big = torch.zeros(B, N, 2 * D, dtype=dt, device=DEV)
        big[..., :D] = x.to(DEV)
        xg = big[..., :D].detach().requires_grad_(True)       # non-contiguous rows
    else:
        xg = x.to(DEV).requires_grad_(True)
    out, aux, _, _ = layer(xg)
    ((out.float() * dy.to(DEV).float()).sum() + aux.float()).backward()
    with torch.no_grad():
        idx = layer.topk_expert(layer.gate_logits(x.to(DEV)))[1].cpu().long()
    same = (idx == st["selected_experts"]).all(-1).reshape(-1) if idx.numel() else torch.ones(0, dtype=torch.bool)
    tol = 2e-5 if dt == torch.float32 else 2e-3
    if same.numel():
        assert same.float().mean() >= 0.97
        a = out.detach().cpu().reshape(-1, D)[same]
        b = o_out.detach().reshape(-1, D)[same]
        assert rel_l2(a, b) <= tol, rel_l2(a, b)
        if bool(same.all()):
            assert rel_l2(xg.grad.cpu(), o_dx) <= 4 * tol
            assert rel_l2(layer.gate.weight.grad.cpu(), o_dwg) <= 8 * tol + 1e-4
            for e in range(E):
                m = layer.experts[e]
                for p, q in zip((m[0].weight, m[0].bias, m[2].weight, m[2].bias), ex[e]):
                    if p is None:
                        continue
                    if q.grad is None or float(q.grad.abs().max()) == 0.0:
                        assert p.grad is None or float(p.grad.abs().max()) == 0.0      # empty expert -> exact zeros
                    else:
                        assert rel_l2(p.grad.cpu(), q.grad) <= 8 * tol, e
    assert abs(float(aux.detach()) - float(o_aux.detach())) <= 1e-3 * max(1.0, abs(float(o_aux.detach())))
    return out


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_single_expert_and_k_equals_e(dt):
    run_both(32, 48, 1, 1, 2, 17, dt)
    run_both(32, 48, 4, 4, 2, 33, dt)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_all_tokens_to_one_expert(dt):
    run_both(64, 96, 8, 1, 3, 171, dt, gate_bias_row=5)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_odd_sizes_no_bias_and_strided_input(dt):
    run_both(40, 72, 6, 2, 1, 257, dt, bias=False)            # D, F multiples of 8 but of no tile; E not a power of two
    run_both(36, 52, 5, 3, 2, 19, dt)                          # D % 8 != 0 -> generic kernels on the bf16 path too
    run_both(64, 128, 8, 2, 2, 64, dt, strided=True)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_empty_batch(dt):
    layer = make(32, 64, 4, 2, dt).to(DEV)
    x = torch.zeros(2, 0, 32, dtype=dt, device=DEV, requires_grad=True)
    out, aux, none, infor = layer(x)
    assert out.shape == (2, 0, 32) and none is None
    out.sum().backward()                                       # nothing to route: all parameter grads are zero / absent
    for p in layer.experts.parameters():
        assert p.grad is None or float(p.grad.abs().sum()) == 0.0


class QuickGELU(nn.Module):                     # what transformers' ACT2FN["quick_gelu"] computes (CLIP towers)
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class CLIPMLPLike(nn.Module):
    def __init__(self, D, F):
        super().__init__()
        self.activation_fn = QuickGELU()
        self.fc1, self.fc2 = nn.Linear(D, F), nn.Linear(F, D)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_quick_gelu_experts_and_unknown_activations(dt):
    """CLIP-style experts (fc1 / quick-GELU / fc2) against the oracle; activations are recognised by what they compute, so an
    nn.Tanh or a LeakyReLU is refused instead of being taken for a GELU / ReLU by its name."""
    D, F, E, K = 64, 96, 4, 2
    torch.manual_seed(0)
    experts = nn.ModuleList([CLIPMLPLike(D, F) for _ in range(E)])
    layer = get_moe("smoe")(D, D, E, K, experts, ARGS).to(dt)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 40, D, generator=g).to(dt)
    ex = [tuple(p.detach().clone() for p in (m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias)) for m in layer.experts]
    with torch.no_grad():
        o_out, _, _, st = O.llava_smoe_forward(x, layer.gate.weight.detach().clone(), ex, "quick_gelu", K, ARGS)
    layer = layer.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    out = layer(xg)[0]
    out.float().sum().backward()
    with torch.no_grad():
        idx = layer.topk_expert(layer.gate_logits(x.to(DEV)))[1].cpu().long()
    same = (idx == st["selected_experts"]).all(-1).reshape(-1)          # bf16 logits: near-ties may route a row differently
    assert same.float().mean() >= 0.95
    assert rel_l2(out.detach().cpu().reshape(-1, D)[same], o_out.reshape(-1, D)[same]) <= (2e-5 if dt == torch.float32 else 2e-3)
    assert torch.isfinite(xg.grad).all() and float(layer.experts[0].fc1.weight.grad.abs().sum()) > 0
    for bad in (nn.Tanh(), nn.LeakyReLU(0.1), nn.ReLU6()):
        e2 = nn.ModuleList([nn.Sequential(nn.Linear(D, F), bad, nn.Linear(F, D)) for _ in range(E)])
        with pytest.raises(NotImplementedError):
            get_moe("smoe")(D, D, E, K, e2, ARGS).to(DEV)(x.float().to(DEV))


def test_package_imported_before_torch_uses_torchs_hip_runtime():
    """A fresh interpreter that imports the package FIRST (what build() followed by smoke() in one process does): the library
    must bind to the HIP runtime PyTorch ships, not bring a second one (launches then fail with "no ROCm-capable device")."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import competesmoe_amd\n"
            "import torch\n"
            "from competesmoe_amd import ops\n"
            "x = torch.randn(64, 32, device='cuda')\n"
            "w = torch.randn(8, 32, device='cuda')\n"
            "lg = ops.gate_logits(x, w)\n"
            "torch.cuda.synchronize()\n"
            "assert torch.allclose(lg, x @ w.t(), atol=1e-4)\n"
            "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-1500:]


def test_split_k_grouped_weight_gradient_matches_the_single_pass():
    """Few experts x few output tiles (the reference's E = 4 LLaVA layers): every expert's rows are cut into chunks handled as
    pseudo-experts (fp32 partials) and summed.  Against the unsplit launch and an fp64 reference, ragged and empty experts."""
    from competesmoe_amd import ops, functional as Fn
    torch.manual_seed(3)
    E, n, Na, Nb = 4, 9000, 520, 300
    counts = torch.tensor([4100, 0, 2900, 2000])
    off = torch.zeros(E + 1, dtype=torch.int32)
    off[1:] = counts.cumsum(0)
    bins = ops.Bins(counts.int().cuda(), off.cuda(), None, None, n, E, 1)
    a = torch.randn(n, Na, device="cuda").bfloat16()
    b = torch.randn(n, Nb, device="cuda").bfloat16()
    split = Fn._grouped_wgrad(a, b, bins, E, torch.bfloat16)
    assert bins._chunks, "the split path was expected for this shape"
    co = next(iter(bins._chunks.values())).cpu()
    assert int(co[0]) == 0 and int(co[-1]) == n and bool((co[1:] >= co[:-1]).all())
    old, Fn._WGRAD_SPLIT = Fn._WGRAD_SPLIT, False
    try:
        single = Fn._grouped_wgrad(a, b, bins, E, torch.bfloat16)
    finally:
        Fn._WGRAD_SPLIT = old
    ref = torch.stack([a[int(off[e]):int(off[e + 1])].double().T @ b[int(off[e]):int(off[e + 1])].double() for e in range(E)])
    scale = float(ref.abs().max())
    assert float((split.double() - ref).abs().max()) <= 2 ** -7 * scale
    assert float((single.double() - ref).abs().max()) <= 2 ** -7 * scale
    assert float(split[1].abs().max()) == 0.0          # the empty expert
