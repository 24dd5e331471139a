"""Thin tensor-level wrappers over the C ABI: allocate outputs with torch, pass raw device pointers and the
current HIP stream.  No math happens here."""
from __future__ import annotations

import os
from typing import Optional, Sequence

import torch

from . import _lib as L

lib = L.lib

# ---- optional per-launch HIP-event timing (bench.py): events are recorded on the launch stream around every C call ----
_PROFILE = None   # None, or dict name -> list[(start_event, end_event, work)]


def profile_start():
    global _PROFILE
    _PROFILE = {}


def profile_stop():
    """Returns {name: {"calls": n, "ms": mean launch duration, "work": mean algorithmic work units}} (synchronises)."""
    global _PROFILE
    rec, _PROFILE = _PROFILE, None
    torch.cuda.synchronize()
    out = {}
    for name, evs in (rec or {}).items():
        ms = [a.elapsed_time(b) for a, b, _ in evs]
        out[name] = {"calls": len(ms), "ms": sum(ms) / len(ms), "work": sum(w for _, _, w in evs) / len(evs)}
    return out


class _timed:
    __slots__ = ("name", "work", "a")

    def __init__(self, name, work=0.0):
        self.name, self.work, self.a = name, work, None

    def __enter__(self):
        if _PROFILE is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.a is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            _PROFILE.setdefault(self.name, []).append((self.a, b, self.work))
        return False


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.F32
    if t.dtype == torch.bfloat16:
        return L.BF16
    raise ValueError(f"csmoe: unsupported dtype {t.dtype} (float32 / bfloat16 only)")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise ValueError("csmoe: tensors must live on the GPU (the HIP path has no CPU fallback)")


def _capturing() -> bool:
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def ptr_array(tensors: Sequence[Optional[torch.Tensor]], device) -> torch.Tensor:
    """Device array of raw pointers (int64) -- the per-expert weight table handed to the grouped GEMM.  A pageable host-to-device
    copy: not allowed inside a hipGraph capture (competesmoe_amd/graphs.py) -- the layers cache their tables by parameter address,
    so one eager step before the capture builds them."""
    if _capturing():
        raise RuntimeError("competesmoe_amd: a per-expert pointer table would have to be built inside a graph capture "
                           "(host-to-device copy); run one eager step with the same parameters first (graphs.GraphedStep does)")
    return torch.tensor([0 if t is None else t.data_ptr() for t in tensors], dtype=torch.int64, device=device)


# ------------------------------------------------------------------------------------------------ router
def gate_logits(x2: torch.Tensor, w_gate: torch.Tensor) -> torch.Tensor:
    _need_cuda(x2, w_gate)
    T, D = x2.shape
    E = w_gate.shape[0]
    out = torch.empty(T, E, dtype=x2.dtype, device=x2.device)
    with _timed("gate_logits", T * D * x2.element_size()):
        L.check(lib.csmoe_gate_logits(x2.data_ptr(), w_gate.data_ptr(), out.data_ptr(), T, D, E, _dt(x2), _stream()), "gate_logits")
    return out


# The block histogram of the last one-pass routing, handed to the bin_tokens call that sorts THOSE ids: (idx, its version counter,
# histogram, ids per block).  Holding `idx` keeps its storage alive, so an equal data_ptr can only be a view of it, and the version
# counter shows any in-place edit since; anything else falls back to the counting pass.
_ROUTE_HIST = None


def _route_hist_for(idx: torch.Tensor, E: int):
    h = _ROUTE_HIST
    if h is None:
        return None
    ref, ver, hist, chunk = h
    if (idx.data_ptr() == ref.data_ptr() and idx.numel() == ref.numel() and idx._version == ver and ref._version == ver
            and hist.shape[1] == E and idx.is_contiguous()):
        return hist, chunk
    return None


def gate_select_ok(x2: torch.Tensor, w_gate: torch.Tensor, K: int) -> bool:
    """Does the one-pass router (csmoe_gate_select) take this shape?  CSMOE_FUSED_ROUTER=0 turns it off (A/B runs)."""
    if os.environ.get("CSMOE_FUSED_ROUTER", "1") == "0" or not x2.is_cuda:
        return False
    T, D = x2.shape
    return bool(lib.csmoe_gate_select_ok(T, D, w_gate.shape[0], K, _dt(x2))) and x2.data_ptr() % 16 == 0 and w_gate.data_ptr() % 16 == 0


def gate_select(x2: torch.Tensor, w_gate: torch.Tensor, K: int, mode: int, round_sum_bf16: bool, want_softmax: bool = True,
                param: float = 1.0, want_hist: bool = True):
    """(logits [T,E], softmax fp32 | None, idx [T,K] int32, w [T,K] fp32) in one launch that reads x2 once.  The launch also counts
    the ids per block of rows; the bin_tokens call that sorts this `idx` (unmodified) picks the histogram up and skips its counting
    pass."""
    _need_cuda(x2, w_gate)
    T, D = x2.shape
    E = w_gate.shape[0]
    dev = x2.device
    logits = torch.empty(T, E, dtype=x2.dtype, device=dev)
    sm = torch.empty(T, E, dtype=torch.float32, device=dev) if want_softmax else None
    idx = torch.empty(T, K, dtype=torch.int32, device=dev)
    w = torch.empty(T, K, dtype=torch.float32, device=dev)
    rows = int(lib.csmoe_gate_select_rows())
    hist = torch.empty((T + rows - 1) // rows, E, dtype=torch.int32, device=dev) if want_hist and T > 0 else None
    with _timed("gate_select", T * D * x2.element_size()):
        L.check(lib.csmoe_gate_select(x2.data_ptr(), w_gate.data_ptr(), T, D, E, K, mode, int(round_sum_bf16), float(param), _dt(x2),
                                      logits.data_ptr(), _ptr(sm), idx.data_ptr(), w.data_ptr(), _ptr(hist), _stream()), "gate_select")
    global _ROUTE_HIST
    _ROUTE_HIST = (idx, idx._version, hist, rows * K) if hist is not None else None
    return logits, sm, idx, w


def router_select(scores: torch.Tensor, K: int, mode: int, round_sum_bf16: bool, want_softmax: bool = True, param: float = 1.0):
    _need_cuda(scores)
    T, E = scores.shape
    sm = torch.empty(T, E, dtype=torch.float32, device=scores.device) if want_softmax else None
    idx = torch.empty(T, K, dtype=torch.int32, device=scores.device)
    w = torch.empty(T, K, dtype=torch.float32, device=scores.device)
    L.check(lib.csmoe_router_select(scores.data_ptr(), _dt(scores), T, E, K, mode, int(round_sum_bf16), float(param), _ptr(sm),
                                    idx.data_ptr(), w.data_ptr(), _stream()), "router_select")
    return sm, idx, w


def router_select_bwd(scores, K, mode, round_sum_bf16, sm, idx, w, dw, dsm, param: float = 1.0):
    T, E = scores.shape
    out = torch.empty_like(scores)
    L.check(lib.csmoe_router_select_bwd(scores.data_ptr(), _dt(scores), T, E, K, mode, int(round_sum_bf16), float(param), _ptr(sm),
                                        idx.data_ptr(), w.data_ptr(), _ptr(dw), _ptr(dsm), out.data_ptr(), _stream()),
            "router_select_bwd")
    return out


def router_aux(logits: Optional[torch.Tensor], sm: torch.Tensor, idx: torch.Tensor):
    """(out2 [2] fp32 = balance, z; dens [B,E] fp32; lse [B*N] fp32 | None) of csmoe_router_aux for sm [B,N,E] fp32, idx [B,N,K]
    int32 and (optionally) the logits [B,N,E] whose softmax sm is."""
    _need_cuda(sm, idx, logits)
    B, N, E = sm.shape
    K = idx.shape[-1]
    dev = sm.device
    out2 = torch.empty(2, dtype=torch.float32, device=dev)
    dens = torch.empty(B, E, dtype=torch.float32, device=dev)
    lse = torch.empty(B * N, dtype=torch.float32, device=dev) if logits is not None else None
    ws = torch.empty(max(1, int(lib.csmoe_router_aux_workspace_floats(B, N, E))), dtype=torch.float32, device=dev)
    dt = _dt(logits) if logits is not None else L.F32
    with _timed("router_aux", B * N * E * 4):
        L.check(lib.csmoe_router_aux(_ptr(logits), sm.data_ptr(), idx.data_ptr(), _ptr(lse), ws.data_ptr(), dens.data_ptr(),
                                     out2.data_ptr(), B, N, E, K, dt, _stream()), "router_aux")
    return out2, dens, lse


def router_aux_bwd(sm: torch.Tensor, dens: torch.Tensor, lse: Optional[torch.Tensor], g_bal: Optional[torch.Tensor],
                   g_z: Optional[torch.Tensor], logits_dtype):
    """(dsoftmax fp32 | None, dlogits logits_dtype | None) of csmoe_router_aux_bwd; g_bal / g_z are 1-element fp32 device tensors."""
    B, N, E = sm.shape
    dsm = torch.empty_like(sm) if g_bal is not None else None
    dlogits = torch.empty(B, N, E, dtype=logits_dtype, device=sm.device) if (g_z is not None and lse is not None) else None
    if dsm is None and dlogits is None:
        return None, None
    dt = L.BF16 if logits_dtype == torch.bfloat16 else L.F32
    L.check(lib.csmoe_router_aux_bwd(sm.data_ptr(), dens.data_ptr(), _ptr(lse), _ptr(g_bal), _ptr(g_z), _ptr(dsm), _ptr(dlogits),
                                     B, N, E, dt, _stream()), "router_aux_bwd")
    return dsm, dlogits


# ------------------------------------------------------------------------------------------------ binning
class Bins:
    """Binned row space of one routing decision."""
    __slots__ = ("counts", "offsets", "perm", "slot_of", "n", "E", "K", "_xcd_order", "_chunks")

    def __init__(self, counts, offsets, perm, slot_of, n, E, K):
        self.counts, self.offsets, self.perm, self.slot_of = counts, offsets, perm, slot_of
        self.n, self.E, self.K = n, E, K
        self._xcd_order = None
        self._chunks = {}

    def chunk_offsets(self, P: int) -> torch.Tensor:
        """offsets [E*P + 1] of every expert's rows cut into P chunks of whole 64-row K-tiles (the last takes the rest): the
        pseudo-experts of a split-K weight gradient.  Computed once per routing decision."""
        co = self._chunks.get(P)
        if co is None:
            co = self._chunks[P] = chunk_offsets(self.offsets, self.E, P, 64)
        return co

    @property
    def xcd_order(self) -> torch.Tensor:
        """Experts dealt to the XCDs by row count (csmoe_expert_order), computed once per routing decision."""
        if self._xcd_order is None:
            self._xcd_order = expert_order(self.offsets, self.E)
        return self._xcd_order


_STRIDE_TABLES = {}
_PTR_TABLES = {}     # (base address, n, stride) -> device table; content is a pure function of the key


def ptr_table(buf: torch.Tensor, n: int, stride_bytes: int) -> torch.Tensor:
    """Device table of n pointers buf.data_ptr() + i * stride_bytes.  The i * stride part is cached per (n, stride, device), so a
    table costs ONE tiny add kernel per call instead of arange + mul + add (the backward builds four of them per step, and at the
    reference's small LLaVA shapes a step is bound by the number of launches)."""
    base = buf.data_ptr()
    full = (base, n, stride_bytes, buf.device)
    cap = _capturing()
    tab = None if cap else _PTR_TABLES.get(full)   # the caching allocator hands the same addresses out step after step: usually a hit
    if tab is not None:
        return tab
    key = (n, stride_bytes, buf.device)
    offs = _STRIDE_TABLES.get(key)
    if offs is None:
        if cap:
            raise RuntimeError("competesmoe_amd: a stride table is missing inside a graph capture; run one eager step first")
        offs = _STRIDE_TABLES[key] = torch.arange(n, device=buf.device, dtype=torch.int64) * stride_bytes
    if cap:
        # inside a capture the add is a graph node (recomputed by every replay); it must not enter the address-keyed cache: kernels
        # do not run while capturing, and the graph's private pool re-uses addresses the eager allocator also hands out
        return offs + base
    if len(_PTR_TABLES) >= 512:
        _PTR_TABLES.clear()
    tab = _PTR_TABLES[full] = offs + base
    return tab


_ARANGES = {}


def cached_arange(n: int, device, dtype=torch.int64) -> torch.Tensor:
    key = (n, torch.device(device), dtype)
    t = _ARANGES.get(key)
    if t is None:
        t = _ARANGES[key] = torch.arange(n, device=device, dtype=dtype)
    return t


def chunk_offsets(offsets: torch.Tensor, E: int, P: int, align: int = 1) -> torch.Tensor:
    """[E*P + 1] int32 offsets of every expert's row range cut into P chunks (csmoe_chunk_offsets): one launch."""
    out = torch.empty(E * P + 1, dtype=torch.int32, device=offsets.device)
    L.check(lib.csmoe_chunk_offsets(offsets.data_ptr(), E, P, align, out.data_ptr(), _stream()), "chunk_offsets")
    return out


_SEG_OFFSETS = {}


def sum_partials(part: torch.Tensor, E: int, P: int, out_dtype) -> torch.Tensor:
    """part [E*P, N] fp32 partial rows -> [E, N] in out_dtype: sum over the P partials of every expert and the cast in ONE launch of
    the grouped column-sum kernel (torch: a reduce kernel and a cast)."""
    N = part.shape[1]
    key = (E, P, part.device)
    seg = _SEG_OFFSETS.get(key)
    if seg is None:
        seg = _SEG_OFFSETS[key] = (torch.arange(E + 1, dtype=torch.int64) * P).int().to(part.device)
    out = torch.empty(E, N, dtype=out_dtype, device=part.device)
    grouped_colsum(part, seg, E, out, ptr_table(out, E, N * out.element_size()))
    return out


def bin_tokens(idx: torch.Tensor, E: int, hist=None) -> Bins:
    """`hist` = (block histogram [nb, E] int32, ids per block) from gate_select: scan + scatter only."""
    _need_cuda(idx)
    if idx.dtype != torch.int32:
        raise ValueError("csmoe: expert indices must be int32")
    idx = idx.contiguous()
    n = idx.numel()
    K = idx.shape[-1]
    dev = idx.device
    counts = torch.empty(E, dtype=torch.int32, device=dev)
    offsets = torch.empty(E + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    slot_of = torch.empty(n, dtype=torch.int32, device=dev)
    if hist is None:
        hist = _route_hist_for(idx, E)
    if hist is not None and n > 0:
        bh, chunk = hist
        if bh.shape != ((n + chunk - 1) // chunk, E) or bh.dtype != torch.int32:
            raise ValueError("csmoe: block histogram does not match the ids it was built from")
        base = torch.empty_like(bh)
        L.check(lib.csmoe_bin_tokens_hist(idx.data_ptr(), n, E, chunk, bh.data_ptr(), base.data_ptr(), counts.data_ptr(),
                                          offsets.data_ptr(), perm.data_ptr(), slot_of.data_ptr(), _stream()), "bin_tokens_hist")
        return Bins(counts, offsets, perm, slot_of, n, E, K)
    ws = torch.empty(max(1, lib.csmoe_bin_workspace_bytes(n, E)), dtype=torch.uint8, device=dev)
    L.check(lib.csmoe_bin_tokens(idx.data_ptr(), n, E, counts.data_ptr(), offsets.data_ptr(), perm.data_ptr(),
                                 slot_of.data_ptr(), ws.data_ptr(), _stream()), "bin_tokens")
    return Bins(counts, offsets, perm, slot_of, n, E, K)


# ------------------------------------------------------------------------------------------------ dispatch / combine
def dispatch_rows(x2: torch.Tensor, bins: Bins) -> torch.Tensor:
    T, D = x2.shape
    xs = torch.empty(bins.n, D, dtype=x2.dtype, device=x2.device)
    with _timed("dispatch_rows", (T + bins.n) * D * x2.element_size()):     # read D, write K*D per token
        L.check(lib.csmoe_dispatch_rows(x2.data_ptr(), bins.perm.data_ptr(), bins.K, xs.data_ptr(), bins.n, D, _dt(x2), _stream()),
                "dispatch_rows")
    return xs


def dispatch_tokens(x2: torch.Tensor, bins: Bins) -> torch.Tensor:
    """Token-major dispatch: same xs as dispatch_rows, x fetched from HBM once per token."""
    T, D = x2.shape
    xs = torch.empty(bins.n, D, dtype=x2.dtype, device=x2.device)
    with _timed("dispatch_rows", (T + bins.n) * D * x2.element_size()):
        L.check(lib.csmoe_dispatch_tokens(x2.data_ptr(), bins.slot_of.data_ptr(), bins.K, xs.data_ptr(), T, D, _dt(x2), _stream()),
                "dispatch_tokens")
    return xs


def dispatch_rows_bwd(dxs: torch.Tensor, bins: Bins, T: int, add: Optional[torch.Tensor] = None, idx: Optional[torch.Tensor] = None,
                      pre: Optional[torch.Tensor] = None, out_f32: bool = False) -> torch.Tensor:
    """dx[t] = round(round(sum_k dxs[slot]) + add[t]); with `idx` [T, K] (int32) the sequential form of the LLaVA stack's autograd:
    (((pre[t] + dxs[slot of the highest expert]) + ...) + dxs[slot of the lowest]) + add[t], rounded after every add (csmoe.h).
    `out_f32` (bf16 rows, D % 8 == 0): dx fp32 = float(round_bf16(sum_k dxs[slot])) + float(add[t]) (csmoe_dispatch_rows_bwd_mixed)."""
    D = dxs.shape[1]
    if out_f32:
        if dxs.dtype != torch.bfloat16 or idx is not None or pre is not None:
            raise ValueError("dispatch_rows_bwd(out_f32=True): bf16 rows, no sequential form")
        dx = torch.empty(T, D, dtype=torch.float32, device=dxs.device)
        with _timed("dispatch_rows_bwd", (T * (2 + (add is not None)) + bins.n) * D * 2):
            L.check(lib.csmoe_dispatch_rows_bwd_mixed(dxs.data_ptr(), bins.slot_of.data_ptr(), bins.K, _ptr(add), dx.data_ptr(), T, D,
                                                      _stream()), "dispatch_rows_bwd_mixed")
        return dx
    dx = torch.empty(T, D, dtype=dxs.dtype, device=dxs.device)
    streams = 1 + (add is not None) + (pre is not None)
    with _timed("dispatch_rows_bwd", (T * streams + bins.n) * D * dxs.element_size()):     # `add` / `pre` are [T, D] reads
        L.check(lib.csmoe_dispatch_rows_bwd(dxs.data_ptr(), bins.slot_of.data_ptr(), bins.K, _ptr(add), dx.data_ptr(), T, D,
                                            _dt(dxs), _ptr(idx), _ptr(pre), _stream()), "dispatch_rows_bwd")
    return dx


_IDENT = {}


def scale_rows(rows: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """round(w[m] * rows[m, :]) in rows.dtype, the product in fp32 (csmoe_combine_bwd over the identity map with K = 1): what
    `(rows.float() * w[:, None]).to(rows.dtype)` computes in three passes."""
    n = rows.shape[0]
    if n == 0:
        return rows
    ident = _IDENT.get(rows.device)
    if ident is None or ident.numel() < n:          # one growing index vector per device: row counts differ from step to step
        ident = _IDENT[rows.device] = torch.arange(max(n, 2 * (0 if ident is None else ident.numel())), dtype=torch.int32, device=rows.device)
    ar = ident[:n]
    dy, _ = combine_bwd(rows.contiguous(), None, Bins(None, None, ar, ar, n, 1, 1), w.reshape(-1, 1).float().contiguous(), want_dw=False)
    return dy


def widen_sum(streams) -> torch.Tensor:
    """fp32 sum of up to three bf16 tensors of one shape, added in the order given (csmoe_widen_sum): the gradient autograd leaves in
    an fp32 tensor whose bf16 casts fed several ops."""
    streams = [s.contiguous() for s in streams]
    a = streams[0]
    if not 1 <= len(streams) <= 3 or any(s.dtype != torch.bfloat16 or s.shape != a.shape for s in streams):
        raise ValueError("widen_sum: one to three bf16 tensors of one shape")
    out = torch.empty(a.shape, dtype=torch.float32, device=a.device)
    p = [s.data_ptr() for s in streams] + [None] * (3 - len(streams))
    with _timed("widen_sum", a.numel() * (2 * len(streams) + 4)):
        L.check(lib.csmoe_widen_sum(p[0], p[1], p[2], out.data_ptr(), a.numel(), _stream()), "widen_sum")
    return out


def combine(y: torch.Tensor, bins: Bins, idx: torch.Tensor, w: torch.Tensor, mode: int, T: int,
            obias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    D = y.shape[1]
    if residual is not None and residual.dtype == torch.float32 and y.dtype == torch.bfloat16:
        # fp32 residual stream around bf16 activations (pretrain stack under autocast): fp32 output, no cast passes
        assert residual.shape == (T, D) and residual.is_contiguous() and obias is None
        out = torch.empty(T, D, dtype=torch.float32, device=y.device)
        with _timed("combine", (bins.n * 2 + T * 8) * D + bins.n * 4):
            L.check(lib.csmoe_combine_mixed(y.data_ptr(), bins.slot_of.data_ptr(), _ptr(idx), w.data_ptr(), residual.data_ptr(),
                                            out.data_ptr(), T, bins.K, D, mode, _stream()), "combine_mixed")
        return out
    out = torch.empty(T, D, dtype=y.dtype, device=y.device)
    if residual is not None:
        assert residual.shape == (T, D) and residual.dtype == y.dtype and residual.is_contiguous()
    nbytes = (T + bins.n + (T if residual is not None else 0)) * D * y.element_size() + bins.n * 4
    with _timed("combine", nbytes):
        L.check(lib.csmoe_combine(y.data_ptr(), bins.slot_of.data_ptr(), _ptr(idx), w.data_ptr(), _ptr(obias), _ptr(residual),
                                  out.data_ptr(), T, bins.K, D, _dt(y), mode, _stream()), "combine")
    return out


def layernorm_gate(x2: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float,
                   w_gate: Optional[torch.Tensor] = None):
    """xn, mean, rstd[, logits] = LayerNorm(x2) [and xn @ w_gate^T] -- csmoe_layernorm_gate."""
    T, D = x2.shape
    xn = torch.empty_like(x2)
    mean = torch.empty(T, dtype=torch.float32, device=x2.device)
    rstd = torch.empty(T, dtype=torch.float32, device=x2.device)
    logits, E = None, 0
    if w_gate is not None:
        E = w_gate.shape[0]
        assert w_gate.dtype == x2.dtype and w_gate.is_contiguous() and w_gate.shape[1] == D
        logits = torch.empty(T, E, dtype=x2.dtype, device=x2.device)
    with _timed("layernorm_gate", 2 * T * D * x2.element_size()):
        L.check(lib.csmoe_layernorm_gate(x2.data_ptr(), _ptr(gamma), _ptr(beta), float(eps), xn.data_ptr(), mean.data_ptr(),
                                         rstd.data_ptr(), T, D, _dt(x2), _ptr(w_gate), _ptr(logits), E, _stream()),
                "layernorm_gate")
    return xn, mean, rstd, logits


def layernorm_gate_mixed(x2: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float,
                         w_gate: Optional[torch.Tensor] = None):
    """fp32 x / gamma / beta -> bf16 xn (+ bf16 logits), fp32 mean / rstd -- csmoe_layernorm_gate_mixed."""
    T, D = x2.shape
    assert x2.dtype == torch.float32 and (gamma is None or gamma.dtype == torch.float32) and (beta is None or beta.dtype == torch.float32)
    xn = torch.empty(T, D, dtype=torch.bfloat16, device=x2.device)
    mean = torch.empty(T, dtype=torch.float32, device=x2.device)
    rstd = torch.empty(T, dtype=torch.float32, device=x2.device)
    logits, E = None, 0
    if w_gate is not None:
        E = w_gate.shape[0]
        assert w_gate.dtype == torch.bfloat16 and w_gate.is_contiguous() and w_gate.shape[1] == D
        logits = torch.empty(T, E, dtype=torch.bfloat16, device=x2.device)
    with _timed("layernorm_gate", T * D * 6):
        L.check(lib.csmoe_layernorm_gate_mixed(x2.data_ptr(), _ptr(gamma), _ptr(beta), float(eps), xn.data_ptr(), mean.data_ptr(),
                                               rstd.data_ptr(), T, D, _ptr(w_gate), _ptr(logits), E, _stream()), "layernorm_gate_mixed")
    return xn, mean, rstd, logits


def layernorm_bwd_mixed(dxn: torch.Tensor, x2: torch.Tensor, gamma: Optional[torch.Tensor], mean: torch.Tensor, rstd: torch.Tensor,
                        add: Optional[torch.Tensor] = None, want_affine_grads: bool = True, dxn2: Optional[torch.Tensor] = None):
    """fp32 dx [T,D] (+ fp32 add), dgamma / dbeta [D] fp32 from bf16 gradient stream(s) of xn and fp32 x."""
    T, D = x2.shape
    assert dxn.dtype == torch.bfloat16 and x2.dtype == torch.float32 and (add is None or add.dtype == torch.float32)
    assert dxn2 is None or dxn2.dtype == torch.bfloat16
    dx = torch.empty_like(x2)
    nb = int(lib.csmoe_layernorm_bwd_blocks(T))
    partial = torch.empty(nb, 2 * D, dtype=torch.float32, device=x2.device)
    with _timed("layernorm_bwd", (8 + 2 + 4 * (add is not None) + 2 * (dxn2 is not None)) * T * D):
        L.check(lib.csmoe_layernorm_bwd_mixed(dxn.data_ptr(), _ptr(dxn2), x2.data_ptr(), _ptr(gamma), mean.data_ptr(), rstd.data_ptr(),
                                              _ptr(add), dx.data_ptr(), partial.data_ptr(), T, D, _stream()), "layernorm_bwd_mixed")
    if not want_affine_grads:
        return dx, None, None
    sums = torch.empty(2 * D, dtype=torch.float32, device=x2.device)
    L.check(lib.csmoe_dense_colsum(partial.data_ptr(), 2 * D, nb, 2 * D, sums.data_ptr(), L.F32, L.F32, _stream()), "layernorm_bwd sums")
    return dx, sums[:D], sums[D:]


def layernorm_bwd(dxn: torch.Tensor, x2: torch.Tensor, gamma: Optional[torch.Tensor], mean: torch.Tensor, rstd: torch.Tensor,
                  add: Optional[torch.Tensor] = None, want_affine_grads: bool = True, dxn2: Optional[torch.Tensor] = None):
    """dx [T,D] (+ add), dgamma [D] fp32, dbeta [D] fp32 -- csmoe_layernorm_bwd + the column sums of its partial rows."""
    T, D = x2.shape
    dx = torch.empty_like(x2)
    nb = int(lib.csmoe_layernorm_bwd_blocks(T))
    partial = torch.empty(nb, 2 * D, dtype=torch.float32, device=x2.device)
    with _timed("layernorm_bwd", (3 + (add is not None) + (dxn2 is not None)) * T * D * x2.element_size()):
        L.check(lib.csmoe_layernorm_bwd(dxn.data_ptr(), _ptr(dxn2), x2.data_ptr(), _ptr(gamma), mean.data_ptr(), rstd.data_ptr(), _ptr(add),
                                        dx.data_ptr(), partial.data_ptr(), T, D, _dt(x2), _stream()), "layernorm_bwd")
    if not want_affine_grads:
        return dx, None, None
    sums = torch.empty(2 * D, dtype=torch.float32, device=x2.device)
    L.check(lib.csmoe_dense_colsum(partial.data_ptr(), 2 * D, nb, 2 * D, sums.data_ptr(), L.F32, L.F32, _stream()), "layernorm_bwd sums")
    return dx, sums[:D], sums[D:]


def combine_bwd(dout: torch.Tensor, y: Optional[torch.Tensor], bins: Bins, w: torch.Tensor, want_dw: bool = True, act_dtype=None,
                round_products: bool = False):
    """`act_dtype` = dtype of the expert rows when it differs from dout's (fp32 upstream gradient, bf16 rows).
    `round_products`: d w as autograd forms it when the weights are an x.dtype tensor (competition steps): every product
    dout * y rounded to the dtype, the sum rounded once."""
    T, D = dout.shape
    if act_dtype == torch.bfloat16 and dout.dtype == torch.float32:
        dy = torch.empty(bins.n, D, dtype=torch.bfloat16, device=dout.device)
        dw = torch.empty(T, bins.K, dtype=torch.float32, device=dout.device) if (want_dw and y is not None) else None
        with _timed("combine_bwd", (T * 4 + bins.n * 2 * (2 if y is not None else 1)) * D):
            L.check(lib.csmoe_combine_bwd_mixed(dout.data_ptr(), _ptr(y), bins.slot_of.data_ptr(), w.data_ptr(), dy.data_ptr(), _ptr(dw),
                                                T, bins.K, D, _stream()), "combine_bwd_mixed")
        return dy, dw
    dy = torch.empty(bins.n, D, dtype=dout.dtype, device=dout.device)
    dw = torch.empty(T, bins.K, dtype=torch.float32, device=dout.device) if (want_dw and y is not None) else None
    with _timed("combine_bwd", (T + bins.n * (2 if y is not None else 1)) * D * dout.element_size()):
        L.check(lib.csmoe_combine_bwd(dout.data_ptr(), _ptr(y), bins.perm.data_ptr(), bins.slot_of.data_ptr(), w.data_ptr(),
                                      dy.data_ptr(), _ptr(dw), T, bins.K, D, _dt(dout), int(round_products), _stream()), "combine_bwd")
    return dy, dw


# ------------------------------------------------------------------------------------------------ grouped GEMMs
def grouped_gemm(A: torch.Tensor, b_ptrs: torch.Tensor, b_layout: int, ldb: int, N: int, offsets: torch.Tensor, E: int,
                 bias_ptrs: Optional[torch.Tensor] = None, epilogue: int = L.EPI_PLAIN, act: int = L.ACT_NONE,
                 aux: Optional[torch.Tensor] = None, want_c2: bool = False, force_generic: bool = False, want_c: bool = True,
                 row_scale: Optional[torch.Tensor] = None, row_dot: Optional[torch.Tensor] = None, kernel: int = 0):
    """`want_c=False` (with want_c2 and EPI_BIAS_ACT): only the activated output is written and (None, C2) returned.
    `row_scale` (fp32 [M], with EPI_ACTGRAD_ROWSCALE): multiplies the rounded product row by row before the activation gradient;
    `row_dot` (fp32 [M, rowdot_cols(...)]): receives the partial sums of product * aux (see rowdot_cols / finish_row_dot).
    `kernel`: CSMOE_KERNEL_* selector (0 = the library chooses, 2 / 4 = the 8-wave / one-wave-per-SIMD 256x256 kernel: tests, A/B)."""
    M, Kd = A.shape
    Cm = torch.empty(M, N, dtype=A.dtype, device=A.device) if want_c else None
    C2 = torch.empty(M, N, dtype=A.dtype, device=A.device) if want_c2 else None
    if epilogue == L.EPI_ACTGRAD_ROWSCALE:
        if row_scale is None or row_scale.dtype != torch.float32 or row_scale.numel() != M or not row_scale.is_contiguous():
            raise ValueError("csmoe: EPI_ACTGRAD_ROWSCALE needs a contiguous fp32 row_scale [M]")
        C2 = row_scale                      # the interface passes the scales in the C2 slot
        if row_dot is not None:
            bias_ptrs = ptr_table(row_dot, E, 0)        # the interface passes the table in the bias slot (every expert the same)
    with _timed("grouped_gemm_" + ("nt" if b_layout == L.B_NK else "nn"), 2.0 * M * N * Kd):
        L.check(lib.csmoe_grouped_gemm(A.data_ptr(), A.stride(0), b_ptrs.data_ptr(), b_layout, ldb, _ptr(bias_ptrs),
                                       offsets.data_ptr(), E, M, N, Kd, _ptr(Cm), _ptr(C2), _ptr(aux), N, epilogue, act,
                                       _dt(A), 1 if force_generic else int(kernel), _stream()), "grouped_gemm")
    return (Cm, C2) if want_c2 else Cm


def rowdot_cols(M: int, N: int, Kd: int, lda: int, ldb: int, ldc: int, dtype) -> int:
    """Partial sums per row the EPI_ACTGRAD_ROWSCALE launch of this shape writes into a dot table (0: it would not write one)."""
    return int(lib.csmoe_grouped_gemm_rowdot_cols(M, N, Kd, lda, ldb, ldc, L.BF16 if dtype == torch.bfloat16 else L.F32))


def finish_row_dot(table: torch.Tensor) -> torch.Tensor:
    """Row sums of a dot table, columns in ascending order (csmoe_affinity_finish with D = 1): fp32 [M]."""
    M, nt = table.shape
    out = torch.empty(M, dtype=torch.float32, device=table.device)
    L.check(lib.csmoe_affinity_finish(table.data_ptr(), M, nt, 1, out.data_ptr(), 1, L.F32, _stream()), "affinity_finish")
    return out


def grouped_gemm_f32w(A: torch.Tensor, B32: torch.Tensor, offsets: torch.Tensor, copy: Optional[torch.Tensor] = None,
                      bias_ptrs: Optional[torch.Tensor] = None, epilogue: int = L.EPI_PLAIN, act: int = L.ACT_NONE,
                      aux: Optional[torch.Tensor] = None, want_c2: bool = False, want_c: bool = True):
    """csmoe_grouped_gemm_f32w: A [M, Kd] bf16, B32 [E, Kd, N] FP32 masters converted inside the tile fill; `copy` [E, Kd, N] bf16
    receives the converted weights of every expert that has rows."""
    M, Kd = A.shape
    E, _, N = B32.shape
    Cm = torch.empty(M, N, dtype=A.dtype, device=A.device) if want_c else None
    C2 = torch.empty(M, N, dtype=A.dtype, device=A.device) if want_c2 else None
    bp = ptr_table(B32, E, Kd * N * 4)
    cp = ptr_table(copy, E, Kd * N * 2) if copy is not None else None
    with _timed("grouped_gemm_nn", 2.0 * M * N * Kd):
        L.check(lib.csmoe_grouped_gemm_f32w(A.data_ptr(), A.stride(0), bp.data_ptr(), N, _ptr(cp), _ptr(bias_ptrs), offsets.data_ptr(), E, M,
                                            N, Kd, _ptr(Cm), _ptr(C2), _ptr(aux), N, epilogue, act, _stream()), "grouped_gemm_f32w")
    return (Cm, C2) if want_c2 else Cm


def f32w_ok(M: int, N: int, Kd: int) -> bool:
    """Shapes for which the fp32-master kernel is the right one (its 256 x 256 tiles need a big launch, as use_v2_rowspace)."""
    return N >= 256 and Kd >= 128 and M >= 2048 and N % 8 == 0 and Kd % 8 == 0 and Kd * N * 4 < (1 << 31)


def dense_gemm(A: torch.Tensor, B: torch.Tensor, b_layout: int, bias: Optional[torch.Tensor] = None,
               epilogue: int = L.EPI_PLAIN, act: int = L.ACT_NONE, aux: Optional[torch.Tensor] = None, want_c2: bool = False,
               force_generic: bool = False, want_c: bool = True, kernel: int = 0):
    M, Kd = A.shape
    N = B.shape[0] if b_layout == L.B_NK else B.shape[1]
    Cm = torch.empty(M, N, dtype=A.dtype, device=A.device) if want_c else None
    C2 = torch.empty(M, N, dtype=A.dtype, device=A.device) if want_c2 else None
    L.check(lib.csmoe_dense_gemm(A.data_ptr(), A.stride(0), B.data_ptr(), b_layout, B.stride(0), _ptr(bias), M, N, Kd,
                                 _ptr(Cm), _ptr(C2), _ptr(aux), N, epilogue, act, _dt(A), 1 if force_generic else int(kernel), _stream()),
            "dense_gemm")
    return (Cm, C2) if want_c2 else Cm


def affinity_ok(x2: torch.Tensor, d_ff: int, d_out: int) -> bool:
    """Can the competition pass of one dense expert run with the affinity epilogues (csmoe_dense_gemm, SOFTPLUS_ROWSUM / _GRAD)?"""
    T, D = x2.shape
    return (x2.is_cuda and x2.dtype == torch.bfloat16 and D % 8 == 0 and d_ff % 8 == 0 and d_out % 8 == 0
            and T * max(D, d_ff, d_out) * 2 < 2 ** 31 and max(D, d_ff, d_out) * max(D, d_ff, d_out) * 2 < 2 ** 31)


def _softplus_flags(rounded: bool) -> int:
    return (1 if SOFTPLUS_PRECISE else 0) | (2 if rounded else 0)


def dense_gemm_affinity(h: torch.Tensor, w2: torch.Tensor, b_layout: int, bias: Optional[torch.Tensor], aff_col: torch.Tensor,
                        rounded: bool) -> None:
    """aff_col[t] = mean_n softplus(round(h[t] @ w2 (+ bias))[n]) without storing the product: the second GEMM of a dense expert with
    the SOFTPLUS_ROWSUM epilogue + csmoe_affinity_finish.  `aff_col` is a (strided) column of the affinity matrix [T, E];
    `rounded`: every softplus value is rounded to bf16 before the mean (x.dtype tensor ops, LLaVA stack)."""
    M, Kd = h.shape
    N = w2.shape[0] if b_layout == L.B_NK else w2.shape[1]
    nt = (N + 255) // 256
    part = torch.empty(M, nt, dtype=torch.float32, device=h.device)
    with _timed("dense_gemm_affinity", 2.0 * M * N * Kd):
        L.check(lib.csmoe_dense_gemm(h.data_ptr(), h.stride(0), w2.data_ptr(), b_layout, w2.stride(0), _ptr(bias), M, N, Kd,
                                     part.data_ptr(), None, None, nt, L.EPI_SOFTPLUS_ROWSUM, _softplus_flags(rounded), _dt(h), 0,
                                     _stream()), "dense_gemm(SOFTPLUS_ROWSUM)")
    L.check(lib.csmoe_affinity_finish(part.data_ptr(), M, nt, N, aff_col.data_ptr(), aff_col.stride(0), _dt(aff_col), _stream()),
            "affinity_finish")


def dense_gemm_affinity_grad(h: torch.Tensor, w2: torch.Tensor, b_layout: int, bias: Optional[torch.Tensor],
                             daff: torch.Tensor, rounded: bool) -> torch.Tensor:
    """dy[t, n] = round(g[t] * sigmoid(round(h[t] @ w2 (+ bias))[n])), g = daff / N (rounded to bf16 first when `rounded`): the gradient
    of dense_gemm_affinity's mean with respect to the (recomputed, never stored) expert output; daff fp32 [T] contiguous."""
    M, Kd = h.shape
    N = w2.shape[0] if b_layout == L.B_NK else w2.shape[1]
    dy = torch.empty(M, N, dtype=h.dtype, device=h.device)
    with _timed("dense_gemm_affinity_grad", 2.0 * M * N * Kd):
        L.check(lib.csmoe_dense_gemm(h.data_ptr(), h.stride(0), w2.data_ptr(), b_layout, w2.stride(0), _ptr(bias), M, N, Kd,
                                     dy.data_ptr(), None, daff.data_ptr(), N, L.EPI_SOFTPLUS_GRAD, _softplus_flags(rounded),
                                     _dt(h), 0, _stream()), "dense_gemm(SOFTPLUS_GRAD)")
    return dy


def expert_order(offsets: torch.Tensor, E: int) -> torch.Tensor:
    order = torch.empty(8 * ((E + 7) // 8) + 1, dtype=torch.int32, device=offsets.device)
    L.check(lib.csmoe_expert_order(offsets.data_ptr(), E, order.data_ptr(), _stream()), "expert_order")
    return order


def grouped_wgrad(A: torch.Tensor, B: torch.Tensor, offsets: torch.Tensor, E: int, out: torch.Tensor,
                  out_ptrs: torch.Tensor, accumulate: bool = False, force_generic: bool = False, tag: str = "grouped_wgrad_tn",
                  xcd_order: Optional[torch.Tensor] = None):
    """out[e] = A_e^T @ B_e for every expert; `out` is [E, Na, Nb] (or any buffer the pointers index).  `xcd_order`
    (Bins.xcd_order) balances the persistent kernel under skewed routing."""
    M, Na = A.shape
    Nb = B.shape[1]
    with _timed(tag, 2.0 * M * Na * Nb):
        L.check(lib.csmoe_grouped_wgrad(A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), offsets.data_ptr(), E, M, Na, Nb,
                                        out_ptrs.data_ptr(), Nb, _dt(A), _dt(out), int(accumulate), int(force_generic),
                                        _ptr(xcd_order), _stream()),
                "grouped_wgrad")
    return out


def dense_wgrad(A: torch.Tensor, B: torch.Tensor, out_dtype=None, force_generic: bool = False) -> torch.Tensor:
    M, Na = A.shape
    Nb = B.shape[1]
    out = torch.empty(Na, Nb, dtype=out_dtype or A.dtype, device=A.device)
    with _timed("dense_wgrad_tn", 2.0 * M * Na * Nb):
        L.check(lib.csmoe_dense_wgrad(A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), M, Na, Nb, out.data_ptr(), Nb, _dt(A),
                                      _dt(out), 0, int(force_generic), _stream()), "dense_wgrad")
    return out


def grouped_colsum(G: torch.Tensor, offsets: torch.Tensor, E: int, out: torch.Tensor, out_ptrs: torch.Tensor):
    M, N = G.shape
    with _timed("grouped_colsum", M * N * G.element_size()):
        L.check(lib.csmoe_grouped_colsum(G.data_ptr(), G.stride(0), offsets.data_ptr(), E, N, out_ptrs.data_ptr(), _dt(G), _dt(out),
                                         _stream()), "grouped_colsum")
    return out


def dense_colsum(G: torch.Tensor, out_dtype=None) -> torch.Tensor:
    M, N = G.shape
    out = torch.empty(N, dtype=out_dtype or G.dtype, device=G.device)
    L.check(lib.csmoe_dense_colsum(G.data_ptr(), G.stride(0), M, N, out.data_ptr(), _dt(G), _dt(out), _stream()), "dense_colsum")
    return out


# ------------------------------------------------------------------------------------------------ affinity
# softplus of the affinity kernels: hardware exp / log (default) or expf / log1pf as torch evaluates them (=1).  A/B on every
# competition fixture (tools/parity_report.py, profiles/r02/parity_report.txt): the same rows route differently from the reference
# with either form (the bf16 near-ties of the LLaVA stack's x.dtype affinities), none with fp32 affinities -- so the fast form stays.
SOFTPLUS_PRECISE = os.environ.get("CSMOE_SOFTPLUS_PRECISE", "0") == "1"


def softplus_mean(y: torch.Tensor, aff_dtype=None) -> torch.Tensor:
    """aff_dtype: y.dtype (x.dtype tensor-op semantics) or torch.float32 around bf16 rows (CUDA-autocast semantics)."""
    R, D = y.shape
    aff = torch.empty(R, dtype=aff_dtype or y.dtype, device=y.device)
    L.check(lib.csmoe_softplus_mean(y.data_ptr(), aff.data_ptr(), R, D, _dt(y), _dt(aff), int(SOFTPLUS_PRECISE), _stream()), "softplus_mean")
    return aff


def softplus_mean_bwd(y: torch.Tensor, daff: torch.Tensor, dy_add: Optional[torch.Tensor] = None) -> torch.Tensor:
    R, D = y.shape
    dy = torch.empty_like(y)
    L.check(lib.csmoe_softplus_mean_bwd(y.data_ptr(), daff.data_ptr(), _ptr(dy_add), dy.data_ptr(), R, D, _dt(y), _dt(daff),
                                        int(SOFTPLUS_PRECISE), _stream()), "softplus_mean_bwd")
    return dy


def pair_cosine(y3: torch.Tensor) -> torch.Tensor:
    """tok_loss[T] (fp32) of csmoe_pair_cosine for y3 [T, K, D]."""
    T, K, D = y3.shape
    out = torch.empty(T, dtype=torch.float32, device=y3.device)
    with _timed("pair_cosine", T * K * D * y3.element_size()):
        L.check(lib.csmoe_pair_cosine(y3.data_ptr(), out.data_ptr(), T, K, D, _dt(y3), _stream()), "pair_cosine")
    return out


def pair_cosine_bwd(y3: torch.Tensor, gscale: torch.Tensor) -> torch.Tensor:
    T, K, D = y3.shape
    dy = torch.empty_like(y3)
    with _timed("pair_cosine_bwd", 2 * T * K * D * y3.element_size()):
        L.check(lib.csmoe_pair_cosine_bwd(y3.data_ptr(), gscale.data_ptr(), dy.data_ptr(), T, K, D, _dt(y3), _stream()), "pair_cosine_bwd")
    return dy


def gate_bwd_small_ok(D: int, E: int, dtype) -> bool:
    return bool(lib.csmoe_gate_bwd_small_ok(D, E, L.BF16 if dtype == torch.bfloat16 else L.F32))


def gate_bwd_dx(dlogits: torch.Tensor, w_gate: torch.Tensor) -> torch.Tensor:
    T, E = dlogits.shape
    D = w_gate.shape[1]
    dx = torch.empty(T, D, dtype=dlogits.dtype, device=dlogits.device)
    with _timed("gate_bwd_dx", T * D * dx.element_size()):
        L.check(lib.csmoe_gate_bwd_dx(dlogits.data_ptr(), w_gate.data_ptr(), dx.data_ptr(), T, D, E, _dt(dlogits), _stream()), "gate_bwd_dx")
    return dx


def gate_bwd_dw(dlogits: torch.Tensor, x2: torch.Tensor, out_dtype) -> torch.Tensor:
    T, E = dlogits.shape
    D = x2.shape[1]
    if T == 0:
        return torch.zeros(E, D, dtype=out_dtype, device=x2.device)
    nr = int(lib.csmoe_gate_bwd_dw_ranges(T, D, _dt(x2)))
    part = torch.empty(nr, E * D, dtype=torch.float32, device=x2.device)
    with _timed("gate_bwd_dw", T * D * x2.element_size()):
        L.check(lib.csmoe_gate_bwd_dw(dlogits.data_ptr(), x2.data_ptr(), part.data_ptr(), T, D, E, _dt(x2), nr, _stream()), "gate_bwd_dw")
    out = torch.empty(E * D, dtype=torch.float32, device=x2.device)
    L.check(lib.csmoe_dense_colsum(part.data_ptr(), E * D, nr, E * D, out.data_ptr(), L.F32, L.F32, _stream()), "gate_bwd_dw sum")
    return out.view(E, D).to(out_dtype)


# ------------------------------------------------------------------------------------------------ MXFP8 (BASELINE config 5)
def quantize_mxfp8(x: torch.Tensor, transpose: bool = False):
    """x [R, C] or [E, R, C] (bf16 / fp32, contiguous) -> (q uint8, s uint8): e4m3 elements + e8m0 scales per 32 elements along C
    (q [.., R, C], s [.., R, C/32]) or, transposed, along R (q [.., C, R], s [.., C, R/32]) -- csmoe_quantize_mxfp8."""
    _need_cuda(x)
    x = x.contiguous()
    lead = x.shape[:-2]
    R, C = x.shape[-2:]
    E = 1
    for d in lead:
        E *= d
    if transpose:
        q = torch.empty(*lead, C, R, dtype=torch.uint8, device=x.device)
        s = torch.empty(*lead, C, R // 32, dtype=torch.uint8, device=x.device)
    else:
        q = torch.empty(*lead, R, C, dtype=torch.uint8, device=x.device)
        s = torch.empty(*lead, R, C // 32, dtype=torch.uint8, device=x.device)
    ptrs = None if E == 1 else ptr_table(x, E, R * C * x.element_size())
    nbytes = x.numel() * (x.element_size() + 1)
    with _timed("quantize_mxfp8" + ("_t" if transpose else ""), nbytes):
        L.check(lib.csmoe_quantize_mxfp8(x.data_ptr() if E == 1 else None, _ptr(ptrs), E, C, R, C, _dt(x), int(transpose),
                                         q.data_ptr(), s.data_ptr(), _stream()), "quantize_mxfp8")
    return q, s


def quantize_mxfp8_both(x: torch.Tensor):
    """x [R, C] or [E, R, C] -> ((q, s), (qt, st)): both orientations of quantize_mxfp8 in one pass over x (weights: the forward
    and the backward product reduce over different dims).  Falls back to two passes when R or C is not a multiple of 32."""
    x = x.contiguous()
    R, C = x.shape[-2:]
    if R % 32 != 0 or C % 32 != 0:
        return quantize_mxfp8(x), quantize_mxfp8(x, transpose=True)
    lead = x.shape[:-2]
    E = 1
    for d in lead:
        E *= d
    dev = x.device
    q = torch.empty(*lead, R, C, dtype=torch.uint8, device=dev)
    s = torch.empty(*lead, R, C // 32, dtype=torch.uint8, device=dev)
    qt = torch.empty(*lead, C, R, dtype=torch.uint8, device=dev)
    st = torch.empty(*lead, C, R // 32, dtype=torch.uint8, device=dev)
    ptrs = None if E == 1 else ptr_table(x, E, R * C * x.element_size())
    with _timed("quantize_mxfp8_both", x.numel() * (x.element_size() + 2)):
        L.check(lib.csmoe_quantize_mxfp8_both(x.data_ptr() if E == 1 else None, _ptr(ptrs), E, C, R, C, _dt(x), q.data_ptr(),
                                              s.data_ptr(), qt.data_ptr(), st.data_ptr(), _stream()), "quantize_mxfp8_both")
    return (q, s), (qt, st)


def grouped_gemm_mxfp8(Aq: torch.Tensor, As: torch.Tensor, Bq: torch.Tensor, Bs: torch.Tensor, offsets: torch.Tensor,
                       bias_ptrs: Optional[torch.Tensor] = None, epilogue: int = L.EPI_PLAIN, act: int = L.ACT_NONE,
                       aux: Optional[torch.Tensor] = None, want_c2: bool = False, want_c: bool = True):
    """Row-space grouped GEMM on the block-scaled fp8 MFMA: Aq [M, Kd] / As [M, Kd/32], Bq [E, N, Kd] / Bs [E, N, Kd/32] (uint8),
    bf16 outputs [M, N] -- csmoe_grouped_gemm_mxfp8."""
    M, Kd = Aq.shape
    E, N, _ = Bq.shape
    Cm = torch.empty(M, N, dtype=torch.bfloat16, device=Aq.device) if want_c else None
    C2 = torch.empty(M, N, dtype=torch.bfloat16, device=Aq.device) if want_c2 else None
    bq = ptr_table(Bq, E, N * Kd)
    bs = ptr_table(Bs, E, N * (Kd // 32))
    with _timed("grouped_gemm_mxfp8", 2.0 * M * N * Kd):
        L.check(lib.csmoe_grouped_gemm_mxfp8(Aq.data_ptr(), Kd, As.data_ptr(), Kd // 32, bq.data_ptr(), bs.data_ptr(), Kd, Kd // 32,
                                             _ptr(bias_ptrs), offsets.data_ptr(), E, M, N, Kd, _ptr(Cm), _ptr(C2), _ptr(aux), N,
                                             epilogue, act, _stream()), "grouped_gemm_mxfp8")
    return (Cm, C2) if want_c2 else Cm


def dense_gemm_mxfp8(Aq: torch.Tensor, As: torch.Tensor, Bq: torch.Tensor, Bs: torch.Tensor, bias: Optional[torch.Tensor] = None,
                     epilogue: int = L.EPI_PLAIN, act: int = L.ACT_NONE, aux: Optional[torch.Tensor] = None, want_c2: bool = False,
                     want_c: bool = True):
    """Dense form: Bq [N, Kd], Bs [N, Kd/32]."""
    M, Kd = Aq.shape
    N = Bq.shape[0]
    Cm = torch.empty(M, N, dtype=torch.bfloat16, device=Aq.device) if want_c else None
    C2 = torch.empty(M, N, dtype=torch.bfloat16, device=Aq.device) if want_c2 else None
    with _timed("dense_gemm_mxfp8", 2.0 * M * N * Kd):
        L.check(lib.csmoe_dense_gemm_mxfp8(Aq.data_ptr(), Kd, As.data_ptr(), Kd // 32, Bq.data_ptr(), Bs.data_ptr(), Kd, Kd // 32,
                                           _ptr(bias), M, N, Kd, _ptr(Cm), _ptr(C2), _ptr(aux), N, epilogue, act, _stream()),
                "dense_gemm_mxfp8")
    return (Cm, C2) if want_c2 else Cm
