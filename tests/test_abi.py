"""The C-ABI library loads and exports every symbol the public header declares (CPU only)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "ldm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ldm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_header_symbols():
    from src import _native as N
    assert os.path.exists(N.LIB_PATH), "build with __graft_entry__.build()"
    handle = ctypes.CDLL(N.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/ldm_hip.h but not exported"
    # the ctypes table covers exactly the declared functions
    assert sorted(N.SIGNATURES) == names


def test_version_and_error_strings():
    from src import _native as N
    assert N.lib().ldm_abi_version() == N.ABI_VERSION
    assert b"aligned" in N.lib().ldm_error_string(-2)
    assert N.lib().ldm_rmsnorm_bwd_ws_floats(100, 64) > 0
