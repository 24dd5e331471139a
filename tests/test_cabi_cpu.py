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
