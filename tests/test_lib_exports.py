"""The C-ABI library builds for gfx950, loads, and exports every symbol include/xrft_hip.h declares.
No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(REPO, "include", "xrft_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(xrfthip_[a-z_0-9]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from xrft_amd import _lib

    assert declared_symbols() == sorted(_lib.EXPORTS)


def test_library_builds_loads_and_exports():
    import __graft_entry__ as g

    if not os.path.exists("/opt/rocm/bin/hipcc") and not os.path.exists(g.OUT):
        pytest.skip("no hipcc and no prebuilt library")
    if os.path.exists("/opt/rocm/bin/hipcc"):
        g.build(force=False)
    dll = ctypes.CDLL(g.OUT)
    for name in declared_symbols():
        assert hasattr(dll, name), name
    dll.xrfthip_version.restype = ctypes.c_int
    assert dll.xrfthip_version() == 106  # XRFTHIP_VERSION in include/xrft_hip.h
    dll.xrfthip_strerror.restype = ctypes.c_char_p
    assert b"unsupported" in dll.xrfthip_strerror(-2)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from xrft_amd import _lib

    monkeypatch.setattr(_lib, "_HERE", str(tmp_path))
    saved = dict(_lib._state)
    _lib._state.update(dll=None, path=None)
    try:
        with pytest.raises(_lib.XrftHipUnavailable):
            _lib.load()
    finally:
        _lib._state.update(saved)
