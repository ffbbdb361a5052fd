"""The C-ABI library loads without a GPU and exports every symbol include/shotvae_hip.h declares."""
import ctypes
import os
import re

from shot_vae_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "shotvae_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sv_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(L.LIB_PATH):
        L.build()
    lib = ctypes.CDLL(L.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 28
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert set(names) == set(L.EXPORTS), set(names) ^ set(L.EXPORTS)
    assert L.lib().sv_version() == 1


def test_struct_layout_matches_header():
    assert ctypes.sizeof(L.SvPhase) == 72
    assert ctypes.sizeof(L.SvGeom) == 72 + 4 * 72
    assert ctypes.sizeof(L.SvIgemmArgs) == 17 * 8
    assert ctypes.sizeof(L.SvBnBranch) == 48
