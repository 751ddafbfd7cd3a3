"""CPU: libmhx.so loads and exports every symbol include/mhx.h declares (no compute calls)."""
import ctypes
import os
import re

from megahit_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    with open(os.path.join(ROOT, "include", "mhx.h")) as f:
        src = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    names = set(re.findall(r"\b(mhx_[a-z0-9_]+)\s*\(", src))
    return {n for n in names if not n.endswith("_fn")}


def test_library_exports_all_declared_symbols():
    assert os.path.exists(lib.LIB_PATH), "libmhx.so not built (run __graft_entry__.build())"
    L = ctypes.CDLL(lib.LIB_PATH)
    decl = declared_symbols()
    assert len(decl) >= 25
    for name in sorted(decl):
        assert hasattr(L, name), name
    assert decl == set(lib.SYMBOLS), (decl ^ set(lib.SYMBOLS))


def test_binding_loads_without_gpu():
    L = lib.load()
    assert L.mhx_version().startswith(b"mhx")
