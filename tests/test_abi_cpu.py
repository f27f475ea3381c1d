"""CPU-side checks of the C-ABI boundary: the library builds for gfx950, loads without a GPU and exports every
symbol include/frhip.h declares (no compute calls here)."""
import ctypes
import os

import pytest


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from frhip import _abi
    protos = _abi.parse_header()
    assert len(protos) >= 35
    handle = ctypes.CDLL(_abi.LIB_PATH)
    missing = [n for n in protos if not hasattr(handle, n)]
    assert not missing, missing
    lib = _abi.lib()
    assert lib.frhip_abi_version() == 1
    assert lib.frhip_nt_block_m(64) == 256 and lib.frhip_nt_block_m(128) == 128
    assert lib.frhip_head_groups(122000) == 1908


def test_bad_arguments_are_reported_not_thrown():
    from frhip import _abi
    lib = _abi.lib()
    # channel count that is not a multiple of the K step: must return FRHIP_EINVAL before touching the GPU
    rc = lib.frhip_conv_fwd(0, None, None, None, None, 1, 8, 8, 3, 64, 3, 3, 1, 1, None)
    assert rc == -1
    assert b"unsupported shape" in lib.frhip_last_error()
    with pytest.raises(_abi.FrhipError):
        _abi.check(rc, "frhip_conv_fwd")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from frhip import _abi
    monkeypatch.setattr(_abi, "_LIB", None)
    monkeypatch.setattr(_abi, "LIB_PATH", os.path.join(tmp_path, "nope.so"))
    with pytest.raises(_abi.FrhipError):
        _abi.lib()
