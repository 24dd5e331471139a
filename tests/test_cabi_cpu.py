"""CPU: the C-ABI library builds, loads, and exports every symbol include/csmoe.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "csmoe.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(csmoe_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from competesmoe_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 18
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in csmoe.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(syms)


def test_version_and_error_string():
    from competesmoe_amd import _lib
    assert _lib.lib.csmoe_version() >= 100
    assert isinstance(_lib.lib.csmoe_last_error(), bytes)


def test_argument_validation_without_gpu():
    """Validation runs before any launch, so bad arguments are reported on a CPU-only box too."""
    import pytest
    from competesmoe_amd import _lib
    rc = _lib.lib.csmoe_router_select(None, 7, 4, 8, 2, 0, 0, 1.0, None, None, None, None)
    assert rc == 1 and b"dtype" in _lib.lib.csmoe_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc, "router_select")
    rc = _lib.lib.csmoe_grouped_gemm(None, 8, None, 0, 8, None, None, 0, 4, 8, 8, None, None, None, 8, 0, 0, 1, 0, None)
    assert rc == 1


def test_library_has_no_undefined_symbols():
    """dlopen with RTLD_NOW: an internal symbol that is declared but not defined must fail HERE, not at the first call on the GPU
    box (lazy binding hides it on the build container)."""
    import ctypes
    from competesmoe_amd import _lib
    ctypes.CDLL(_lib.LIB_PATH, mode=ctypes.RTLD_GLOBAL | 2)      # 2 = RTLD_NOW


def test_round2_entries_validate_and_answer_without_gpu():
    """The entries added in round 2 report bad arguments / unsupported shapes before any launch, and the shape queries are pure host
    functions."""
    from competesmoe_amd import _lib
    L = _lib.lib
    bf16 = _lib.BF16
    assert L.csmoe_gate_select_rows() == 64
    assert L.csmoe_gate_select_ok(1000, 4096, 64, 2, bf16) == 1
    assert L.csmoe_gate_select_ok(1000, 4096, 65, 2, bf16) == 0          # more than 64 experts: the two-launch path
    assert L.csmoe_gate_select_ok(1000, 4100, 64, 2, bf16) == 0          # D not a multiple of 8
    assert L.csmoe_gate_select_ok(1000, 4096, 64, 2, 1 - bf16) == 0      # fp32: the two-launch path
    rc = L.csmoe_gate_select(None, None, 8, 64, 4, 9, 0, 0, 1.0, bf16, None, None, None, None, None, None)
    assert rc == 1 and b"gate_select" in L.csmoe_last_error()             # K > E
    rc = L.csmoe_bin_tokens_hist(None, 8, 4, 0, None, None, None, None, None, None, None)
    assert rc == 1 and b"chunk" in L.csmoe_last_error()
    rc = L.csmoe_affinity_finish(None, 4, 0, 8, None, 1, bf16, None)
    assert rc == 1 and b"affinity_finish" in L.csmoe_last_error()
    # SOFTPLUS_GRAD without its row scales
    rc = L.csmoe_dense_gemm(1 << 12, 8, 1 << 12, 0, 8, None, 4, 8, 8, 1 << 12, None, None, 8, _lib.EPI_SOFTPLUS_GRAD, 0, bf16, 0, None)
    assert rc == 1 and b"SOFTPLUS_GRAD" in L.csmoe_last_error()


def test_route_histogram_is_only_reused_for_the_ids_it_counted():
    """ops._route_hist_for: the block histogram of the one-pass router is handed to bin_tokens only for the very idx tensor it was
    built from (same storage, same version counter); a copy, a different shape or an in-place edit fall back to the counting pass."""
    import torch
    from competesmoe_amd import ops
    idx = torch.zeros(128, 2, dtype=torch.int32)
    hist = torch.zeros(2, 8, dtype=torch.int32)
    old = ops._ROUTE_HIST
    try:
        ops._ROUTE_HIST = (idx, idx._version, hist, 128)
        assert ops._route_hist_for(idx, 8) is not None
        assert ops._route_hist_for(idx.view(2, 64, 2).view(128, 2), 8) is not None       # a view of the same ids
        assert ops._route_hist_for(idx.clone(), 8) is None                                # other storage
        assert ops._route_hist_for(idx, 16) is None                                       # other expert count
        assert ops._route_hist_for(idx[:64], 8) is None                                   # other length
        idx[0, 0] = 3                                                                     # edited in place
        assert ops._route_hist_for(idx, 8) is None
    finally:
        ops._ROUTE_HIST = old
