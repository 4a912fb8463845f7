"""CPU-only: the C-ABI library builds for gfx950, loads, and exports exactly what include/nrsc5hip.h
declares (no compute calls here -- there is no GPU in this container)."""
import ctypes
import os
import re

from nrsc5_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "nrsc5hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nrsc5hip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(engine.EXPORTED_SYMBOLS)


def test_hip_library_exports_every_declared_symbol(hip_lib):
    lib = ctypes.CDLL(hip_lib)
    for name in _declared():
        assert hasattr(lib, name), f"{name} missing from libnrsc5hip.so"


def test_record_layout_is_96_bytes():
    assert engine.RECORD_DTYPE.itemsize == 96
    assert engine.RECORD_DTYPE.fields["pids"][1] == 80


def test_missing_library_is_loud(tmp_path):
    import pytest
    with pytest.raises(engine.Nrsc5HipError):
        engine.load_library(str(tmp_path / "nope.so"))


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "nrsc5_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "from oracle" not in src and "import oracle" not in src and "liboracle" not in src, f
