"""The C-ABI library loads and exports every symbol include/vo355.h declares (no compute)."""
import ctypes
import os
import re

import pytest

from openvo_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "vo355.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vo_[A-Za-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    _native.build_native()
    L = ctypes.CDLL(_native.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "libvo355.so does not export %s" % n
    assert sorted(_native.SYMBOLS) == names


def test_no_cpu_fallback():
    """Without a HIP device vo_create must fail loudly (the product never routes to the CPU)."""
    try:
        c = _native.Context(0, 128, 128, 16, 64)
    except _native.VoError as e:
        assert e.code in (-2, -1)
        return
    c.close()
    pytest.skip("a GPU is present on this machine")


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "openvo_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                for pat in (r"(from|import)\s+oracle", r"vo_ref_", r"libvo_oracle", r"oracle/", r"vo_oracle\.h"):
                    assert not re.search(pat, src), "%s reaches into the oracle (%s)" % (f, pat)
