"""The C-ABI library must load on a CPU-only box and export every symbol include/pgx.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pgx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pgx_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_the_whole_abi():
    from pyprogressivex import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 35
    for sym in declared:
        assert hasattr(lib, sym), f"libpgx.so does not export {sym}"
    assert sorted(_lib.ABI_SYMBOLS) == declared, "pyprogressivex/_lib.py ABI list out of sync with include/pgx.h"
    assert lib.pgx_version() >= 100


def test_model_dims_table():
    from pyprogressivex import _lib
    lib = _lib.load()
    for mt in range(6):
        d, p = ctypes.c_int(), ctypes.c_int()
        assert lib.pgx_model_dims(mt, ctypes.byref(d), ctypes.byref(p)) == 0
        assert (d.value, p.value) == (_lib.POINT_DIM[mt], _lib.PARAM_DIM[mt])
    assert lib.pgx_model_dims(9, None, None) != 0


def test_product_fails_loudly_without_a_gpu():
    """No CPU fallback: on a box without a HIP device creating a context raises, it never routes to the oracle."""
    from pyprogressivex import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.PgxError):
        _lib.Context(0)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "progressive-x_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inl", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pgx_oracle" not in src and "pgxo_" not in src, f"{f} references the oracle"
