"""CPU: the C-ABI library loads and exports exactly what include/llx.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="llx.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(llx_[a-z0-9_]+)\s*\(", text)))


def test_header_and_library_agree():
    from llx import _lib as L

    lib = L.load()  # raises loudly if the .so is missing
    declared = _declared()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/llx.h but not exported"
        assert name in L.SIGNATURES, f"{name} has no ctypes signature in llx/_lib.py"
    for name in L.SIGNATURES:
        assert name in declared, f"{name} bound in llx/_lib.py but missing from include/llx.h"
    debug = _declared("llx_debug.h")  # diagnostic probes live in their own header, outside the boundary
    assert debug and not set(debug) & set(declared) and all(n.startswith("llx_debug_") for n in debug)
    assert not any(n.startswith("llx_debug_") for n in declared)
    for name in debug:
        assert hasattr(lib, name) and name in L.DEBUG_SIGNATURES, name
    assert sorted(L.DEBUG_SIGNATURES) == debug
    assert sorted(f for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith(".h")) == ["llx.h", "llx_debug.h"]
    assert lib.llx_version() == 105
    assert isinstance(lib.llx_last_error_string(), (bytes, type(None)))


def test_host_only_entry_points():
    from llx import _lib as L

    lib = L.load()
    assert lib.llx_rmsnorm_bwd_workspace_bytes(4096, 4096) == 256 * 4096 * 4
    assert lib.llx_attn_flags_bytes(2, 4096) == 2 * 32 * 64
    assert lib.llx_ce_workspace_bytes(4096) == (4096 + 2) * 4
    assert lib.llx_skinny_tn_workspace_bytes(4096, 4096, 16) > 0
    # argument validation happens before any launch: a bad shape returns an error code and a message, no GPU needed
    rc = lib.llx_gemm_nt_bf16(ctypes.c_void_p(16), 100, ctypes.c_void_p(16), 100, ctypes.c_void_p(16), 64, 64, 64, 100, None, 0, None, 0, 0, 0, None, 0, None)
    assert rc == -1 and b"multiples of 64" in lib.llx_last_error_string()
    rc = lib.llx_attn_fwd(ctypes.c_void_p(16), 0, 0, ctypes.c_void_p(16), 0, 0, ctypes.c_void_p(16), 0, 0, ctypes.c_void_p(16), 0, 0, None, None, None, None, 1, 128, 4, 1, 64, 0.1, None)
    assert rc == -1 and b"head_dim" in lib.llx_last_error_string()
