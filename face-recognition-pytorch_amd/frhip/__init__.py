"""frhip -- host binding of the MI355X (gfx950) HIP kernels for the face-embedding training path.

Import never builds anything and never falls back: `frhip._abi.lib()` raises FrhipError when libfrhip.so is
absent.  Build with `python -m frhip.build` (or `__graft_entry__.build()`).
"""
from ._abi import DT_BF16, DT_F32, FrhipError, lib  # noqa: F401
