"""linux-fg hot path (upscale -> motion -> interpolate), MI355X-native.

Layout:
  csrc/    hand-written HIP kernels for gfx950 + the C-ABI (include/linuxfg_hip.h)
  host/    C++ mirror of the reference's Scaler / FrameManager / Frame surface, calling only the C-ABI
  capi.py  ctypes binding of the C-ABI, used by tests/ and bench.py
  synth.py synthetic frame source (stand-in for the reference's X11 capture)

Nothing in here imports, links or calls oracle/: a missing HIP library is a hard error.
"""
from . import synth  # noqa: F401

__all__ = ["synth", "capi"]


def __getattr__(name):
    if name == "capi":
        import importlib
        return importlib.import_module(".capi", __name__)
    raise AttributeError(name)
