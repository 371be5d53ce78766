"""MI355X-native PacBio self-correction hot path (drop-in for `stride pbcorrect`'s FM-index path).

The product is the C-ABI shared library ``_build/liblrsc_hip.so`` (``include/lrsc.h``) built
from ``csrc/`` for gfx950; this package only binds it with ctypes for tests and bench.py.
There is no CPU fallback: importing :mod:`longreadselfcorrect_amd.capi` fails loudly when the
HIP library has not been built.
"""
from .capi import Lrsc, LrscError, lib_path  # noqa: F401

__all__ = ["Lrsc", "LrscError", "lib_path"]
