"""CPU: libegomi.so builds for gfx950, loads, and exports every symbol include/egomi.h declares.
No compute calls (no GPU here)."""
import ctypes
import os
import re

from egoscaler_amd import build as B, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "egomi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(egomi_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_header_symbols():
    path = B.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    syms = declared_symbols()
    assert len(syms) >= 6
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in egomi.h but not exported: {missing}"
    assert lib.egomi_version() >= 100
    lib.egomi_strerror.restype = ctypes.c_char_p
    assert lib.egomi_strerror(-2).decode().startswith("shape")


def test_loader_fails_loudly_without_library(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.lib()
    except _lib.EgomiError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("missing library must raise")


def test_product_package_never_imports_oracle():
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "egoscaler_amd")):
        for f in files:
            if f.endswith(".py"):
                s = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M):
                    bad.append(f)
    assert not bad
