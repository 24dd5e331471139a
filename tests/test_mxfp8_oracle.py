"""CPU: properties of the MXFP8 emulation in oracle/mxfp8.py (the checker of the fp8 kernels).  The reference has no fp8, so the
pin is the published format itself (OCP Microscaling v1.0): known element encodings, the shared-exponent rule, the error bound of
a 3-bit mantissa, idempotence and linearity in the scale."""
import torch

from oracle import mxfp8 as MX


def test_known_e4m3_encodings():
    v = torch.tensor([0.0, 1.0, -1.0, 448.0, -448.0, 0.015625, 0.001953125, 1.5, 240.0, 1e6])
    assert MX.to_e4m3_bytes(v).tolist() == [0x00, 0x38, 0xB8, 0x7E, 0xFE, 0x08, 0x01, 0x3C, 0x77, 0x7E]
    assert torch.equal(MX.from_e4m3_bytes(MX.to_e4m3_bytes(v[:9])), v[:9])


def test_shared_exponent_and_error_bound():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 256, generator=g) * torch.exp(torch.randn(64, 1, generator=g) * 4)
    q, s = MX.quantize(x)
    amax = x.reshape(64, 8, 32).abs().amax(-1)
    assert torch.equal(s.int() - 127, torch.floor(torch.log2(amax)).int() - 8)        # 2^(floor(log2 amax) - emax_elem)
    d = MX.dequantize(q, s)
    scale = torch.pow(2.0, s.double() - 127).repeat_interleave(32, -1)
    mag = x.double().abs()
    # 3 mantissa bits: half an ulp = 2^-4 relative for normal elements (>= 2^-6 * scale), 2^-10 * scale below; the block maximum
    # may exceed 448 * scale (up to 512 * scale) and saturates: at most 12.5 %
    bound = torch.where(mag >= 448 * scale, mag * 0.125 + 1e-30, torch.maximum(mag * 2.0 ** -4, scale * 2.0 ** -10))
    assert bool(((d - x.double()).abs() <= bound * (1 + 1e-12)).all())
    q2, s2 = MX.quantize(d.float())
    assert torch.equal(q2, q) and torch.equal(s2, s)                                 # idempotent
    q4, s4 = MX.quantize(x * 4)
    assert torch.equal(q4, q) and torch.equal(s4.int(), s.int() + 2)                 # a power-of-two factor moves only the scales


def test_zero_blocks_and_grouped_matmul():
    x = torch.zeros(2, 64)
    x[1, 40] = 3.0
    q, s = MX.quantize(x)
    assert s[0].tolist() == [0, 0] and int(q[0].sum()) == 0 and float(MX.dequantize(q, s)[1, 40]) == 3.0
    g = torch.Generator().manual_seed(1)
    A, B = torch.randn(10, 64, generator=g), torch.randn(2, 6, 64, generator=g)
    Aq, As = MX.quantize(A)
    Bq, Bs = MX.quantize(B)
    off = torch.tensor([0, 4, 10])
    out = MX.grouped_matmul(Aq, As, Bq, Bs, off)
    ref = torch.cat([A[:4] @ B[0].t(), A[4:] @ B[1].t()])
    assert float((out - ref.double()).norm() / ref.double().norm()) <= 5e-2


def test_mx_ffn_tracks_the_exact_ffn_within_the_format_error_and_is_exact_on_exact_data():
    """ffn_forward_backward (what tests/test_fp8_gpu.py checks the fp8 layer against): on random data every output and gradient is
    within the format's error of the fp64 FFN (3.6 % per quantised operand); on data every quantiser represents
    exactly (small integers, ReLU) it IS the fp64 FFN."""
    torch.manual_seed(0)
    T, D, Fh, E, K = 64, 64, 96, 4, 2
    idx = torch.stack([torch.randperm(E)[:K] for _ in range(T)])

    def exact(x, w, keys, values, dout, act):
        xd, kd, vd = x.double().requires_grad_(True), keys.double().requires_grad_(True), values.double().requires_grad_(True)
        wd = w.bfloat16().double().requires_grad_(True)
        out = torch.zeros(T, D, dtype=torch.float64)
        for k in range(K):
            for e in range(E):
                m = (idx[:, k] == e).nonzero().squeeze(-1)
                if m.numel():
                    out = out.index_add(0, m, wd[m, k].unsqueeze(-1) * (act(xd[m] @ kd[e]) @ vd[e]))
        out.backward(dout.double())
        return {"out": out.detach(), "dx": xd.grad, "dw": wd.grad, "gk": kd.grad, "gv": vd.grad}

    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    x, w, dout = torch.randn(T, D).bfloat16(), torch.rand(T, K), torch.randn(T, D).bfloat16()
    keys, values = torch.randn(E, D, Fh) * 0.1, torch.randn(E, Fh, D) * 0.1
    r = MX.ffn_forward_backward(x, idx, w, keys, values, "gelu", dout)
    ex = exact(x, w, keys, values, dout, torch.nn.functional.gelu)
    for k in ex:
        assert 1e-3 < rel(r[k], ex[k]) <= 0.1, (k, rel(r[k], ex[k]))
    xi = torch.randint(-2, 3, (T, D)).float().bfloat16()
    ki, vi = torch.randint(-1, 2, (E, D, Fh)).float(), torch.randint(-1, 2, (E, Fh, D)).float()
    ki[:, :, ::2], vi[:, ::3] = 0, 0                       # keeps |h|, |y| <= 256: every intermediate is a bf16 AND an e4m3-block value
    xi[:, 8:], vi[:, 40:] = 0, 0
    wi, di = torch.ones(T, K), torch.randint(-1, 2, (T, D)).float().bfloat16()
    r = MX.ffn_forward_backward(xi, idx, wi, ki, vi, "relu", di)
    ex = exact(xi, wi, ki, vi, di, torch.relu)
    assert float(ex["out"].abs().max()) > 0
    for k in ("out", "dw", "gv"):
        assert rel(r[k], ex[k]) == 0.0, (k, rel(r[k], ex[k]))
