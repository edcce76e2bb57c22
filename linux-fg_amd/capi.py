"""ctypes binding of the C-ABI in include/linuxfg_hip.h (liblinuxfg_hip.so).

This is the only way Python code (tests/, bench.py, __graft_entry__) reaches the HIP path.
There is no fallback: if the library is missing, does not export a declared symbol, or no GPU
is present, the failure is raised, never papered over.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LFG_LIB: a diagnostic build of the same library (tools/build_variant.sh); never a different implementation.
LIB_PATH = os.environ.get("LFG_LIB") or os.path.join(_HERE, "liblinuxfg_hip.so")

FORMAT_RGBA8 = 0
FORMAT_MV_S8X2 = 1
STAGE_SCALE, STAGE_MOTION, STAGE_INTERPOLATE = 0, 1, 2
MOTION_PREFILTERED, MOTION_EXACT_ONLY = 0, 1
SEMANTICS_REFERENCE, SEMANTICS_INTENDED = 0, 1
_BPP = {FORMAT_RGBA8: 4, FORMAT_MV_S8X2: 2}
COMM_ID_BYTES = 128
MAX_LANES = 4


class LfgError(RuntimeError):
    pass


class Frame(ctypes.Structure):
    """struct lfg_frame."""
    _fields_ = [("data", ctypes.c_void_p), ("width", ctypes.c_uint32), ("height", ctypes.c_uint32),
                ("pitch", ctypes.c_uint32), ("format", ctypes.c_uint32), ("owned", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32)]


_vp, _i, _u32, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_size_t
_FP = ctypes.POINTER(Frame)

# name -> (restype, argtypes): every entry point include/linuxfg_hip.h declares.
SIGNATURES = {
    "lfg_abi_version": (_i, []),
    "lfg_device_count": (_i, []),
    "lfg_context_create": (_i, [_i, ctypes.POINTER(_vp)]),
    "lfg_context_destroy": (None, [_vp]),
    "lfg_context_set_stream": (_i, [_vp, _vp]),
    "lfg_context_get_stream": (_vp, [_vp]),
    "lfg_context_device": (_i, [_vp]),
    "lfg_sync": (_i, [_vp]),
    "lfg_lanes": (_i, [_vp, _i]),
    "lfg_lane_count": (_i, [_vp]),
    "lfg_lane_current": (_i, [_vp]),
    "lfg_lane_select": (_i, [_vp, _i]),
    "lfg_lane_mark": (_i, [_vp]),
    "lfg_lane_wait": (_i, [_vp, _i]),
    "lfg_lane_sync": (_i, [_vp]),
    "lfg_last_error": (ctypes.c_char_p, [_vp]),
    "lfg_frame_create": (_i, [_vp, _u32, _u32, _u32, _FP]),
    "lfg_frame_destroy": (None, [_vp, _FP]),
    "lfg_frame_wrap": (_i, [_vp, _u32, _u32, _u32, _u32, _FP]),
    "lfg_frame_copy": (_i, [_vp, _FP, _FP]),
    "lfg_staging_create": (_i, [_vp, _sz, ctypes.POINTER(_vp)]),
    "lfg_staging_destroy": (None, [_vp, _vp]),
    "lfg_frame_upload": (_i, [_vp, _FP, _vp, _sz]),
    "lfg_frame_download": (_i, [_vp, _FP, _vp, _sz]),
    "lfg_ring_create": (_i, [_vp, _u32, _sz, ctypes.POINTER(_vp)]),
    "lfg_ring_destroy": (None, [_vp]),
    "lfg_ring_acquire": (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_u32)]),
    "lfg_ring_upload": (_i, [_vp, _u32, _FP]),
    "lfg_ring_download": (_i, [_vp, _u32, _FP]),
    "lfg_ring_wait": (_i, [_vp, _u32]),
    "lfg_ring_fence_slot": (_i, [_vp, _u32]),
    "lfg_scale": (_i, [_vp, _FP, _FP]),
    "lfg_motion": (_i, [_vp, _FP, _FP, _FP, _i, ctypes.c_float]),
    "lfg_set_motion_mode": (_i, [_vp, _i]),
    "lfg_motion_last_stats": (_i, [_vp, ctypes.POINTER(_u32), ctypes.POINTER(_u32), ctypes.POINTER(ctypes.c_double)]),
    "lfg_motion_open_segments": (_i, [_vp, ctypes.POINTER(_u32), ctypes.POINTER(_u32)]),
    "lfg_motion_lean_stats": (_i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(_u32), ctypes.POINTER(_u32)]),
    "lfg_motion_strip_stats": (_i, [_vp, ctypes.POINTER(_u32), ctypes.POINTER(_u32)]),
    "lfg_motion_prediction_stats": (_i, [_vp] + [ctypes.POINTER(ctypes.c_uint64)] * 4),
    "lfg_motion_workspace_size": (_i, [_vp, _u32, _u32, ctypes.POINTER(ctypes.c_uint64)]),
    "lfg_motion_plan": (_i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "lfg_set_semantics": (_i, [_vp, _i]),
    "lfg_interpolate": (_i, [_vp, _FP, _FP, _FP, _FP, ctypes.c_float]),
    "lfg_interpolate_frames": (_i, [_vp, _FP, _FP, _FP, ctypes.c_float]),
    "lfg_interpolate_multi": (_i, [_vp, _FP, _FP, _FP, ctypes.POINTER(_FP), ctypes.POINTER(ctypes.c_float), _u32]),
    "lfg_interpolate_frames_multi": (_i, [_vp, _FP, _FP, ctypes.POINTER(_FP), ctypes.POINTER(ctypes.c_float), _u32]),
    "lfg_interpolate_scale": (_i, [_vp, _FP, _FP, _FP, _FP, ctypes.c_float]),
    "lfg_set_fused_interpolate_scale": (_i, [_vp, _i]),
    "lfg_set_fused_motion_interpolate": (_i, [_vp, _i]),
    "lfg_mv_export_rgba32f": (_i, [_vp, _FP, _vp]),
    "lfg_selftest_sqrt": (_i, [_vp, _u32, _u32, ctypes.POINTER(ctypes.c_uint64)]),
    "lfg_comm_unique_id": (_i, [_vp]),
    "lfg_comm_init": (_i, [_vp, _i, _i, _vp]),
    "lfg_comm_rank": (_i, [_vp]),
    "lfg_comm_ranks": (_i, [_vp]),
    "lfg_broadcast_frame": (_i, [_vp, _FP, _i]),
    "lfg_comm_wait": (_i, [_vp]),
    "lfg_comm_destroy": (_i, [_vp]),
    "lfg_comm_sync": (_i, [_vp]),
    "lfg_comm_reserved_cus": (_i, [_vp]),
    "lfg_comm_cu_mask": (_i, [_vp, ctypes.POINTER(_u32), _i]),
    "lfg_comm_probe": (_i, [_vp, _i, _i, _i]),
    "lfg_broadcast_frame_lane": (_i, [_vp, _FP, _i]),
    "lfg_comm_probe_ms": (_i, [_vp, ctypes.POINTER(ctypes.c_float)]),
    "lfg_motion_last_variant": (_i, [_vp]),
    "lfg_diag_scale_2x_strip": (_i, [_u32, _u32, _u32, ctypes.POINTER(_u32), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "lfg_profile_enable": (_i, [_vp, _i]),
    "lfg_profile_reset": (_i, [_vp]),
    "lfg_profile_get": (_i, [_vp, _i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load liblinuxfg_hip.so and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LfgError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
        # HIP gives a process four hardware queues unless told otherwise, and streams beyond that share them: three lanes, a
        # communication stream and a framework's own stream are five -- two lanes on one queue run in turn (measured on the MI355X:
        # 3,684 -> 2,750 frames/s for every context of a process but its first; NOTES_r05.md section 7).  Read by the HIP runtime at
        # its first call, so this has effect only if nothing in the process has touched the GPU yet; an explicit setting wins.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


class Context:
    """One GPU, one stream (lfg_context).  Mirrors the reference's VulkanContext + FrameManager calls."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = load()
        h = _vp()
        rc = self.lib.lfg_context_create(device, ctypes.byref(h))
        if rc != 0:
            raise LfgError(f"lfg_context_create failed ({rc}): {self.lib.lfg_last_error(None).decode()}")
        self.h = h
        self._frames: list[Frame] = []
        if stream is not None:
            self.set_stream(stream)

    # -- plumbing
    def _check(self, rc: int, what: str):
        if rc != 0:
            raise LfgError(f"{what} failed ({rc}): {self.lib.lfg_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            for f in self._frames:
                self.lib.lfg_frame_destroy(self.h, ctypes.byref(f))
            self._frames.clear()
            self.lib.lfg_context_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream: int | None):
        self._check(self.lib.lfg_context_set_stream(self.h, _vp(stream or 0)), "lfg_context_set_stream")

    def sync(self):
        self._check(self.lib.lfg_sync(self.h), "lfg_sync")

    # lanes: several frames in flight on one GPU (include/linuxfg_hip.h)
    def lanes(self, count: int):
        self._check(self.lib.lfg_lanes(self.h, int(count)), "lfg_lanes")

    def lane_count(self) -> int:
        return int(self.lib.lfg_lane_count(self.h))

    def lane_current(self) -> int:
        return int(self.lib.lfg_lane_current(self.h))

    def lane_select(self, lane: int):
        self._check(self.lib.lfg_lane_select(self.h, int(lane)), "lfg_lane_select")

    def lane_mark(self):
        self._check(self.lib.lfg_lane_mark(self.h), "lfg_lane_mark")

    def lane_sync(self):
        """The host waits for the selected lane alone."""
        self._check(self.lib.lfg_lane_sync(self.h), "lfg_lane_sync")

    def lane_wait(self, other: int):
        self._check(self.lib.lfg_lane_wait(self.h, int(other)), "lfg_lane_wait")

    # -- frames
    def create_frame(self, width: int, height: int, fmt: int = FORMAT_RGBA8) -> Frame:
        f = Frame()
        self._check(self.lib.lfg_frame_create(self.h, width, height, fmt, ctypes.byref(f)), "lfg_frame_create")
        self._frames.append(f)
        return f

    def destroy_frame(self, f: Frame):
        self.lib.lfg_frame_destroy(self.h, ctypes.byref(f))
        self._frames = [g for g in self._frames if g is not f]

    @staticmethod
    def wrap(device_ptr: int, width: int, height: int, fmt: int = FORMAT_RGBA8, pitch: int | None = None) -> Frame:
        f = Frame()
        rc = load().lfg_frame_wrap(_vp(device_ptr), width, height, pitch or width * _BPP[fmt], fmt, ctypes.byref(f))
        if rc != 0:
            raise LfgError(f"lfg_frame_wrap failed ({rc})")
        return f

    def upload(self, f: Frame, host: np.ndarray):
        a = np.ascontiguousarray(host)
        self._check(self.lib.lfg_frame_upload(self.h, ctypes.byref(f), a.ctypes.data_as(_vp), a.nbytes), "lfg_frame_upload")
        self.sync()                      # `a` is pageable and may be a temporary

    def download(self, f: Frame) -> np.ndarray:
        if f.format == FORMAT_RGBA8:
            out = np.empty((f.height, f.width, 4), np.uint8)
        else:
            out = np.empty((f.height, f.width, 2), np.int8)
        self._check(self.lib.lfg_frame_download(self.h, ctypes.byref(f), out.ctypes.data_as(_vp), out.nbytes), "lfg_frame_download")
        self.sync()
        return out

    def frame_from(self, host: np.ndarray, fmt: int = FORMAT_RGBA8) -> Frame:
        f = self.create_frame(host.shape[1], host.shape[0], fmt)
        self.upload(f, host)
        return f

    def copy(self, src: Frame, dst: Frame):
        self._check(self.lib.lfg_frame_copy(self.h, ctypes.byref(src), ctypes.byref(dst)), "lfg_frame_copy")

    # -- staging (pinned host memory)
    def staging_create(self, nbytes: int) -> np.ndarray:
        """lfg_staging_create: ``nbytes`` of pinned host memory as a uint8 array (no copy); release it with
        staging_destroy(array)."""
        p = _vp()
        self._check(self.lib.lfg_staging_create(self.h, nbytes, ctypes.byref(p)), "lfg_staging_create")
        a = np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(p.value))
        self._staging = getattr(self, "_staging", {})
        self._staging[a.ctypes.data] = p.value
        return a

    def staging_destroy(self, a: np.ndarray):
        self.lib.lfg_staging_destroy(self.h, _vp(self._staging.pop(a.ctypes.data)))

    def upload_async(self, f: Frame, host: np.ndarray):
        """lfg_frame_upload without the wait: ``host`` must be pinned and stay alive until sync()."""
        self._check(self.lib.lfg_frame_upload(self.h, ctypes.byref(f), host.ctypes.data_as(_vp), host.nbytes), "lfg_frame_upload")

    def download_async(self, f: Frame, host: np.ndarray):
        self._check(self.lib.lfg_frame_download(self.h, ctypes.byref(f), host.ctypes.data_as(_vp), host.nbytes), "lfg_frame_download")

    def mv_export_rgba32f(self, mv: Frame) -> np.ndarray:
        """The reference's rgba32f motion-vector image, vec4(mv.x, mv.y, 0, 1) per pixel, as (H, W, 4) float32."""
        tmp = self.create_frame(mv.width * 4, mv.height)          # W*H*16 bytes of device memory
        self._check(self.lib.lfg_mv_export_rgba32f(self.h, ctypes.byref(mv), _vp(tmp.data)), "lfg_mv_export_rgba32f")
        raw = self.download(tmp)
        self.destroy_frame(tmp)
        return raw.reshape(mv.height, mv.width * 16).view(np.float32).reshape(mv.height, mv.width, 4)

    # -- stages (enqueue only)
    def scale(self, src: Frame, dst: Frame):
        self._check(self.lib.lfg_scale(self.h, ctypes.byref(src), ctypes.byref(dst)), "lfg_scale")

    def motion(self, prev: Frame, curr: Frame, mv: Frame, block_size: int = 8, search_radius: float = 16.0):
        self._check(self.lib.lfg_motion(self.h, ctypes.byref(prev), ctypes.byref(curr), ctypes.byref(mv),
                                        block_size, search_radius), "lfg_motion")

    def set_motion_mode(self, mode: int):
        """0 = prefiltered (default), 1 = exact kernel only; results are identical."""
        self._check(self.lib.lfg_set_motion_mode(self.h, mode), "lfg_set_motion_mode")

    def set_semantics(self, semantics: int):
        """0 = the shaders as written (parity contract), 1 = opt-in "intended" tie-break and motion-vector units."""
        self._check(self.lib.lfg_set_semantics(self.h, semantics), "lfg_set_semantics")

    def motion_last_stats(self):
        """(tiles, tiles that fell back to the exact kernel, mean candidates recorded per pixel elsewhere)."""
        t, f, m = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_double()
        self._check(self.lib.lfg_motion_last_stats(self.h, ctypes.byref(t), ctypes.byref(f), ctypes.byref(m)), "lfg_motion_last_stats")
        return t.value, f.value, m.value

    def motion_open_segments(self):
        """(segments the prefilter left to the resolve kernel, segments of the frame) of the last prefiltered lfg_motion."""
        a, b = ctypes.c_uint32(), ctypes.c_uint32()
        self._check(self.lib.lfg_motion_open_segments(self.h, ctypes.byref(a), ctypes.byref(b)), "lfg_motion_open_segments")
        return a.value, b.value

    def motion_lean_stats(self):
        """(the selected lane's last lfg_motion went through the lean kernel, tiles listed for it, tiles in which it left work)."""
        u, t, l = ctypes.c_int(), ctypes.c_uint32(), ctypes.c_uint32()
        self._check(self.lib.lfg_motion_lean_stats(self.h, ctypes.byref(u), ctypes.byref(t), ctypes.byref(l)), "lfg_motion_lean_stats")
        return bool(u.value), t.value, l.value

    def motion_strip_stats(self):
        """(pixel rows whose left or right band, pixel columns whose top or bottom band the strip kernel decided in the last lfg_motion)."""
        a, b = ctypes.c_uint32(), ctypes.c_uint32()
        self._check(self.lib.lfg_motion_strip_stats(self.h, ctypes.byref(a), ctypes.byref(b)), "lfg_motion_strip_stats")
        return a.value, b.value

    def motion_prediction_stats(self):
        """(calls whose verdict came back, of which launched on a wrong guess about: the lean kernel, the persistent grid, the
        second pass) since the context was created, all lanes together (lfg_motion_prediction_stats)."""
        v = [ctypes.c_uint64() for _ in range(4)]
        self._check(self.lib.lfg_motion_prediction_stats(self.h, *[ctypes.byref(x) for x in v]), "lfg_motion_prediction_stats")
        return tuple(int(x.value) for x in v)

    def motion_workspace_size(self, width: int, height: int) -> int:
        """Bytes the prefiltered motion path keeps for frames of this size (per lane)."""
        n = ctypes.c_uint64()
        self._check(self.lib.lfg_motion_workspace_size(self.h, width, height, ctypes.byref(n)), "lfg_motion_workspace_size")
        return n.value

    def motion_last_variant(self) -> int:
        """Which variant of the persistent kernel the context's last lfg_motion launched: 0 the default, 1 the one for moderate sensor noise."""
        return int(self.lib.lfg_motion_last_variant(self.h))

    def motion_plan(self):
        """(rim split, persistent workgroups) of the prefiltered motion path on this context (lfg_motion_plan)."""
        a, b = ctypes.c_int(), ctypes.c_int()
        self._check(self.lib.lfg_motion_plan(self.h, ctypes.byref(a), ctypes.byref(b)), "lfg_motion_plan")
        return a.value, b.value

    def interpolate(self, prev: Frame, curr: Frame, mv: Frame, out: Frame, factor: float = 0.5):
        self._check(self.lib.lfg_interpolate(self.h, ctypes.byref(prev), ctypes.byref(curr), ctypes.byref(mv),
                                             ctypes.byref(out), factor), "lfg_interpolate")

    def interpolate_frames(self, prev: Frame, curr: Frame, out: Frame, factor: float = 0.5):
        self._check(self.lib.lfg_interpolate_frames(self.h, ctypes.byref(prev), ctypes.byref(curr),
                                                    ctypes.byref(out), factor), "lfg_interpolate_frames")

    @staticmethod
    def _multi_args(outs, factors):
        n = len(outs)
        if n != len(factors):
            raise ValueError("one output frame per factor")
        return (_FP * n)(*[ctypes.pointer(o) for o in outs]), (ctypes.c_float * n)(*factors), n

    def interpolate_multi(self, prev: Frame, curr: Frame, mv: Frame, outs, factors):
        """One pass over prev / curr / mv, one generated frame per factor (lfg_interpolate_multi)."""
        po, pf, n = self._multi_args(outs, factors)
        self._check(self.lib.lfg_interpolate_multi(self.h, ctypes.byref(prev), ctypes.byref(curr), ctypes.byref(mv), po, pf, n),
                    "lfg_interpolate_multi")

    def interpolate_frames_multi(self, prev: Frame, curr: Frame, outs, factors):
        po, pf, n = self._multi_args(outs, factors)
        self._check(self.lib.lfg_interpolate_frames_multi(self.h, ctypes.byref(prev), ctypes.byref(curr), po, pf, n),
                    "lfg_interpolate_frames_multi")

    def set_fused_motion_interpolate(self, on: bool):
        """lfg_interpolate_frames in the north-star order: the motion kernels write the generated frame themselves."""
        self._check(self.lib.lfg_set_fused_motion_interpolate(self.h, int(on)), "lfg_set_fused_motion_interpolate")

    def set_fused_interpolate_scale(self, on: bool):
        self._check(self.lib.lfg_set_fused_interpolate_scale(self.h, int(on)), "lfg_set_fused_interpolate_scale")

    def interpolate_scale(self, prev: Frame, curr: Frame, mv: Frame, out: Frame, factor: float = 0.5):
        """interpolate at input resolution and upscale, one call (one kernel when out is exactly 2x the inputs)."""
        self._check(self.lib.lfg_interpolate_scale(self.h, ctypes.byref(prev), ctypes.byref(curr), ctypes.byref(mv),
                                                   ctypes.byref(out), factor), "lfg_interpolate_scale")

    def selftest_sqrt(self, lo_bits: int, hi_bits: int) -> int:
        n = ctypes.c_uint64()
        self._check(self.lib.lfg_selftest_sqrt(self.h, lo_bits, hi_bits, ctypes.byref(n)), "lfg_selftest_sqrt")
        return n.value

    # -- multi-GPU: the shared previous frame (RCCL behind the C-ABI)
    @staticmethod
    def comm_unique_id() -> bytes:
        """128 bytes made by ONE rank; hand them to the others (torch.distributed store, file, ...)."""
        buf = ctypes.create_string_buffer(COMM_ID_BYTES)
        rc = load().lfg_comm_unique_id(buf)
        if rc != 0:
            raise LfgError(f"lfg_comm_unique_id failed ({rc}): is librccl.so available?")
        return buf.raw

    def comm_init(self, nranks: int, rank: int, comm_id: bytes):
        if len(comm_id) != COMM_ID_BYTES:
            raise ValueError(f"communicator id must be {COMM_ID_BYTES} bytes")
        buf = ctypes.create_string_buffer(comm_id, COMM_ID_BYTES)
        self._check(self.lib.lfg_comm_init(self.h, nranks, rank, buf), "lfg_comm_init")

    def comm_ranks(self) -> int:
        """Ranks of this context's communicator (0 without one)."""
        return int(self.lib.lfg_comm_ranks(self.h))

    def comm_rank(self) -> int:
        return int(self.lib.lfg_comm_rank(self.h))

    def broadcast_frame(self, f: Frame, root: int = 0):
        self._check(self.lib.lfg_broadcast_frame(self.h, ctypes.byref(f), root), "lfg_broadcast_frame")

    def comm_wait(self):
        self._check(self.lib.lfg_comm_wait(self.h), "lfg_comm_wait")

    def comm_sync(self):
        """The host waits for every broadcast issued so far."""
        self._check(self.lib.lfg_comm_sync(self.h), "lfg_comm_sync")

    def comm_reserved_cus(self) -> int:
        """CUs the library's own streams leave to the communicator (0 without one)."""
        return int(self.lib.lfg_comm_reserved_cus(self.h))

    def comm_cu_mask(self, words: int = 8) -> list[int]:
        """The CU mask (32-bit words, bit i = CU i usable) a caller-supplied compute stream should be created with."""
        buf = (_u32 * words)()
        self._check(self.lib.lfg_comm_cu_mask(self.h, buf, words), "lfg_comm_cu_mask")
        return [int(x) for x in buf]

    def comm_probe(self, workgroups: int = 8, microseconds: int = 50, every_lane: bool = True):
        """Diagnostic: a kernel of RCCL's device kernel's footprint where a broadcast would run (csrc/comm_probe.hip)."""
        self._check(self.lib.lfg_comm_probe(self.h, workgroups, microseconds, int(every_lane)), "lfg_comm_probe")

    def comm_probe_ms(self) -> float:
        """Device milliseconds of the last probe from ready to done (waits for it)."""
        ms = ctypes.c_float()
        self._check(self.lib.lfg_comm_probe_ms(self.h, ctypes.byref(ms)), "lfg_comm_probe_ms")
        return float(ms.value)

    def broadcast_frame_lane(self, f: Frame, root: int = 0):
        """lfg_broadcast_frame ordered behind the selected lane only (the caller has ordered that lane behind the frame's readers)."""
        self._check(self.lib.lfg_broadcast_frame_lane(self.h, ctypes.byref(f), root), "lfg_broadcast_frame_lane")

    def comm_destroy(self):
        self._check(self.lib.lfg_comm_destroy(self.h), "lfg_comm_destroy")

    # -- measurement
    def profile_enable(self, on: bool = True):
        self._check(self.lib.lfg_profile_enable(self.h, int(on)), "lfg_profile_enable")

    def profile_reset(self):
        self._check(self.lib.lfg_profile_reset(self.h), "lfg_profile_reset")

    def profile_get(self, stage: int):
        ms, n = ctypes.c_double(), ctypes.c_uint64()
        self._check(self.lib.lfg_profile_get(self.h, stage, ctypes.byref(ms), ctypes.byref(n)), "lfg_profile_get")
        return ms.value, n.value
