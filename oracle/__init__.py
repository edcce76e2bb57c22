"""ctypes front end of the CPU oracle (oracle/lfg_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of lfg_oracle.c.  Only tests/,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this
package; the product package (linux-fg_amd/) never does.

PARITY UNPINNED: the reference has no golden vectors and cannot run here
(SURVEY.md F8/F9); the oracle is pinned by analytic known-answer tests only.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblfg_oracle.so")
_lib = None

_u8p = ctypes.POINTER(ctypes.c_uint8)
_f32p = ctypes.POINTER(ctypes.c_float)


def build(force: bool = False) -> str:
    """Compile liblfg_oracle.so with gcc if it is missing or older than its source."""
    src = os.path.join(_HERE, "lfg_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liblfg_oracle.so"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        i = ctypes.c_int
        L.lfg_oracle_scale.argtypes = [_u8p, i, i, _u8p, i, i, i, i, i, i, i]
        L.lfg_oracle_scale.restype = i
        L.lfg_oracle_motion.argtypes = [_u8p, _u8p, i, i, i, ctypes.c_float, _f32p, i, i, i, i, i]
        L.lfg_oracle_motion.restype = i
        L.lfg_oracle_interpolate.argtypes = [_u8p, _u8p, _f32p, i, i, ctypes.c_float, _u8p, i, i, i, i, i]
        L.lfg_oracle_interpolate.restype = i
        L.lfg_oracle_motion_ex.argtypes = [_u8p, _u8p, i, i, i, ctypes.c_float, _f32p, i, i, i, i, i, i]
        L.lfg_oracle_motion_ex.restype = i
        L.lfg_oracle_interpolate_ex.argtypes = [_u8p, _u8p, _f32p, i, i, ctypes.c_float, _u8p, i, i, i, i, i, i]
        L.lfg_oracle_interpolate_ex.restype = i
        L.lfg_oracle_lanczos_taps.argtypes = [i, i, i, _f32p]
        L.lfg_oracle_lanczos_taps.restype = i
        _lib = L
    return _lib


def default_threads() -> int:
    """Host threads for the oracle: the affinity mask, capped at 16 (a one-GPU box's CPU share)."""
    cap = int(os.environ.get("LFG_ORACLE_THREADS", "16"))
    return max(1, min(cap, len(os.sched_getaffinity(0))))


def _frame(a: np.ndarray, name: str) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError(f"{name}: expected an (H, W, 4) uint8 RGBA frame, got {a.shape}")
    return a


def _roi(roi, W, H):
    if roi is None:
        return 0, 0, W, H
    x0, y0, x1, y1 = (int(v) for v in roi)
    return x0, y0, x1, y1


def scale(frame: np.ndarray, out_w: int, out_h: int, roi=None, threads: int | None = None) -> np.ndarray:
    """shaders/scale.comp.  Returns an (out_h, out_w, 4) uint8 frame; pixels outside ``roi`` are 0."""
    f = _frame(frame, "frame")
    out = np.zeros((out_h, out_w, 4), np.uint8)
    x0, y0, x1, y1 = _roi(roi, out_w, out_h)
    rc = lib().lfg_oracle_scale(f.ctypes.data_as(_u8p), f.shape[1], f.shape[0],
                                out.ctypes.data_as(_u8p), out_w, out_h, x0, y0, x1, y1,
                                threads or default_threads())
    if rc != 0:
        raise ValueError("lfg_oracle_scale: bad arguments")
    return out


REFERENCE, INTENDED = 0, 1      # semantics: the shaders as written / the opt-in "intended" variant (not the parity contract)


def motion(prev: np.ndarray, curr: np.ndarray, block_size: int = 8, search_radius: float = 16.0,
           roi=None, threads: int | None = None, semantics: int = REFERENCE) -> np.ndarray:
    """shaders/motion.comp.  Returns (H, W, 2) float32 motion vectors (x, y); 0 outside ``roi``.
    ``semantics=INTENDED``: equal costs are broken towards the shortest vector (flat areas give (0,0))."""
    p, c = _frame(prev, "prev"), _frame(curr, "curr")
    if p.shape != c.shape:
        raise ValueError("prev and curr differ in size")
    H, W = p.shape[:2]
    mv = np.zeros((H, W, 2), np.float32)
    x0, y0, x1, y1 = _roi(roi, W, H)
    rc = lib().lfg_oracle_motion_ex(p.ctypes.data_as(_u8p), c.ctypes.data_as(_u8p), W, H, int(block_size),
                                    float(search_radius), mv.ctypes.data_as(_f32p), x0, y0, x1, y1,
                                    threads or default_threads(), int(semantics))
    if rc != 0:
        raise ValueError("lfg_oracle_motion: bad arguments")
    return mv


def interpolate(prev: np.ndarray, curr: np.ndarray, mv: np.ndarray, factor: float = 0.5,
                roi=None, threads: int | None = None, semantics: int = REFERENCE) -> np.ndarray:
    """shaders/interpolate.comp.  ``mv`` is (H, W, 2) float32 in whole pixels (literal semantics, F5).
    ``semantics=INTENDED``: the vector is divided by the image size before it is added to uv."""
    p, c = _frame(prev, "prev"), _frame(curr, "curr")
    if p.shape != c.shape:
        raise ValueError("prev and curr differ in size")
    H, W = p.shape[:2]
    m = np.ascontiguousarray(mv, dtype=np.float32)
    if m.shape != (H, W, 2):
        raise ValueError(f"mv: expected {(H, W, 2)}, got {m.shape}")
    out = np.zeros((H, W, 4), np.uint8)
    x0, y0, x1, y1 = _roi(roi, W, H)
    rc = lib().lfg_oracle_interpolate_ex(p.ctypes.data_as(_u8p), c.ctypes.data_as(_u8p), m.ctypes.data_as(_f32p),
                                         W, H, float(factor), out.ctypes.data_as(_u8p), x0, y0, x1, y1,
                                         threads or default_threads(), int(semantics))
    if rc != 0:
        raise ValueError("lfg_oracle_interpolate: bad arguments")
    return out


def lanczos_taps(p: int, in_size: int, out_size: int):
    """(start index, six raw weights) for output coordinate ``p`` on one axis (scale.comp:24-41)."""
    w = (ctypes.c_float * 6)()
    s = lib().lfg_oracle_lanczos_taps(int(p), int(in_size), int(out_size), w)
    return s, np.array(list(w), np.float32)
