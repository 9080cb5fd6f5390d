"""The C-ABI library loads and exports every symbol include/uw_hip.h declares (no GPU needed)."""
import os
import re

from whisper_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "uw_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(uw_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(L.SIGNATURES)


def test_library_exports_every_declared_symbol():
    lib = L.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.uw_abi_version() == 1


def test_errors_are_reported_not_thrown():
    lib = L.load()
    # argument validation happens before any GPU call
    rc = lib.uw_cost_matrix(None, 1, None, 1, 1, 1500, None, None)
    assert rc != 0 and b"null" in lib.uw_last_error()
    rc = lib.uw_dtw(None, None, 1, 1, 1500, None, None, None, None, None, None)
    assert rc != 0
    assert lib.uw_dtw_workspace_bytes(2, 448, 1500) == 2 * 448 * 94 * 4
