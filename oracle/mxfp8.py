"""CPU ORACLE for the MXFP8 path -- test infrastructure, NOT product code (same rules as oracle/moe_oracle.py).

The reference has no fp8 (BASELINE config 5: "no fp8 and no 128+2 config exist in the reference"), so there
is nothing upstream to restate or to pin against: PARITY UNPINNED by the reference.  What this file restates is the published
format -- OCP Microscaling Formats (MX) v1.0: blocks of 32 elements sharing one E8M0 scale 2^(s - 127), elements OCP FP8 E4M3
(`torch.float8_e4m3fn`); shared exponent = floor(log2(max |x|)) - emax_elem with emax_elem = 8; elements =
round-to-nearest-even(x / scale) saturated at +-448 -- in plain torch, as the checker of competesmoe_amd/csrc/fp8_quant.hip and
gemm_fp8.hip."""
import torch


def to_e4m3_bytes(v: torch.Tensor) -> torch.Tensor:
    """float -> OCP e4m3 bytes (round to nearest even, saturating)."""
    return v.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)


def from_e4m3_bytes(q: torch.Tensor) -> torch.Tensor:
    return q.view(torch.float8_e4m3fn).float()


def quantize(x: torch.Tensor):
    """x [..., C] fp32 (C % 32 == 0) -> (q uint8 [..., C], s uint8 [..., C/32]), blocks along the last dim."""
    shp = x.shape
    xb = x.float().reshape(*shp[:-1], shp[-1] // 32, 32)
    amax = xb.abs().amax(-1)
    bits = amax.contiguous().view(torch.int32)
    be = (bits >> 23) & 0xFF                                   # floor(log2(amax)) + 127 for normal amax, 0 for zero / subnormal
    sb = (be - 8).clamp(0, 254)
    inv = torch.pow(2.0, (127 - sb).double()).float()          # 2^-(sb - 127), exact
    q = to_e4m3_bytes(xb * inv.unsqueeze(-1))
    return q.reshape(shp), sb.to(torch.uint8)


def dequantize(q: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    shp = q.shape
    v = from_e4m3_bytes(q).reshape(*shp[:-1], shp[-1] // 32, 32).double()
    return (v * torch.pow(2.0, s.double() - 127).unsqueeze(-1)).reshape(shp)


def grouped_matmul(Aq, As, Bq, Bs, offsets) -> torch.Tensor:
    """fp64 C[m] = sum_k dq(A)[m,k] dq(B_e)[n,k] for the rows of expert e."""
    A = dequantize(Aq, As)
    out = torch.zeros(Aq.shape[0], Bq.shape[1], dtype=torch.float64)
    for e in range(Bq.shape[0]):
        r0, r1 = int(offsets[e]), int(offsets[e + 1])
        if r1 > r0:
            out[r0:r1] = A[r0:r1] @ dequantize(Bq[e], Bs[e]).t()
    return out
