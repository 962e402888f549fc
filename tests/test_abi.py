"""The C-ABI library loads on a machine without a GPU and exports exactly what include/gba.h
declares; no compute entry point is called here."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "gba.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(gba_[a-z_]+)\s*\(", hdr)))


def test_exports_match_header():
    from blasr_amd import _lib
    L = _lib.lib()
    names = _declared()
    assert set(names) == set(_lib.EXPORTS), (names, _lib.EXPORTS)
    for n in names:
        assert getattr(L, n) is not None


def test_version_and_strerror():
    from blasr_amd import _lib
    L = _lib.lib()
    assert L.gba_version() == 382
    assert L.gba_strerror(0) == b"ok"
    assert b"arena" in L.gba_strerror(_lib.E_NOSPACE)
    assert b"unknown" in L.gba_strerror(12345)


def test_struct_layouts():
    from blasr_amd import _lib
    assert _lib.PAIR_RESULT_DT.itemsize == 88
    assert _lib.BLOCK_DT.itemsize == 12
    assert C.sizeof(_lib.Rescorer) == (25 + 5) * 4
    assert C.sizeof(_lib.Scoring) == (25 + 14 + 1 + 2 * 30 + 1) * 4
    assert C.sizeof(_lib.Batch) == 8 + 9 * 8 + 3 * 8 + 8 + 8 + 8      # q_packed and its padding, q_profile
    assert C.sizeof(_lib.Results) == 72


def test_argument_errors_without_gpu():
    from blasr_amd import _lib
    L = _lib.lib()
    assert L.gba_create(None, None) == _lib.E_INVALID
    assert L.gba_set_scoring(None, None) == _lib.E_INVALID
    assert L.gba_submit(None, None, None) == _lib.E_INVALID
    assert L.gba_wait(None) == _lib.E_INVALID
    assert L.gba_get_timing(None, None) == _lib.E_INVALID
    L.gba_destroy(None)


def test_no_fallback_when_library_missing(monkeypatch):
    from blasr_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgba.so")
    try:
        _lib.lib()
        raise AssertionError("expected a loud failure")
    except RuntimeError as e:
        assert "not built" in str(e)


def test_product_package_does_not_import_oracle():
    pkg = os.path.join(ROOT, "blasr_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".c", ".cpp", ".h", ".hpp")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.lower().replace("# oracle-free", ""), os.path.join(dp, f)
