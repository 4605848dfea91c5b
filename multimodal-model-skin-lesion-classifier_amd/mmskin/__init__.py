"""mmskin: MI355X (gfx950) implementation of the MultimodalModel hot path.

Layout:
  csrc/      hand-written HIP kernels + the C ABI (built into libmmskin_hip.so)
  mmskin/    ctypes binding, autograd wrappers, parameter-holder modules, DP helper
  models/    drop-in modules with the reference's file / class names
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
