"""The C-ABI boundary without a GPU: the library loads, exports every symbol include/linuxfg_hip.h
declares (and nothing is declared that the binding does not know), and refuses to work -- loudly --
when no HIP device is present.  No compute calls here.  CPU only."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "linuxfg_hip.h")


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as entry
    from linux_fg_amd import capi as c
    if not os.path.exists(c.LIB_PATH):
        entry.build()
    return c


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(lfg_[a-z0-9_]+)\s*\(", text))


def test_every_declared_symbol_is_exported_and_bound(capi):
    lib = capi.load()
    declared = declared_functions()
    assert declared, "no declarations parsed from the header"
    assert declared == set(capi.SIGNATURES), (declared ^ set(capi.SIGNATURES))
    for name in declared:
        assert getattr(lib, name) is not None


def test_no_torch_or_oracle_in_the_boundary(capi):
    """The shipped library must not depend on the oracle or on torch: plain C-ABI over HIP."""
    import subprocess
    out = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out and "torch" not in out
    syms = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in syms.splitlines() if " T " in l}
    assert {s for s in exported if s.startswith("lfg_")} >= set(capi.SIGNATURES)
    assert not [s for s in exported if "oracle" in s]


def test_abi_version_and_frame_layout(capi):
    lib = capi.load()
    assert lib.lfg_abi_version() == 1
    assert ctypes.sizeof(capi.Frame) == 32          # void* + 6 x uint32 on LP64
    f = capi.Frame()
    assert lib.lfg_frame_wrap(ctypes.c_void_p(0x1000), 8, 4, 32, capi.FORMAT_RGBA8, ctypes.byref(f)) == 0
    assert (f.width, f.height, f.pitch, f.owned) == (8, 4, 32, 0)
    assert lib.lfg_frame_wrap(ctypes.c_void_p(0x1000), 8, 4, 31, capi.FORMAT_RGBA8, ctypes.byref(f)) != 0   # pitch too small
    assert lib.lfg_frame_wrap(None, 8, 4, 32, capi.FORMAT_RGBA8, ctypes.byref(f)) != 0


def test_no_gpu_means_a_loud_failure_not_a_fallback(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = capi.load()
    assert lib.lfg_device_count() == 0
    h = ctypes.c_void_p()
    rc = lib.lfg_context_create(0, ctypes.byref(h))
    assert rc != 0 and not h.value
    msg = lib.lfg_last_error(None).decode()
    assert "no HIP device" in msg and "no CPU fallback" in msg
    with pytest.raises(capi.LfgError):
        capi.Context(0)


def test_null_arguments_are_rejected(capi):
    lib = capi.load()
    assert lib.lfg_sync(None) != 0
    assert lib.lfg_scale(None, None, None) != 0
    assert lib.lfg_motion(None, None, None, None, 8, 16.0) != 0
    assert lib.lfg_interpolate(None, None, None, None, None, 0.5) != 0
    # lanes (frames in flight): no context, no lanes
    assert lib.lfg_lanes(None, 2) != 0 and lib.lfg_lane_select(None, 0) != 0
    assert lib.lfg_lane_mark(None) != 0 and lib.lfg_lane_wait(None, 0) != 0
    assert lib.lfg_lane_count(None) == 0 and lib.lfg_lane_current(None) == -1
    assert lib.lfg_motion_workspace_size(None, 3840, 2160, None) != 0      # (a size query still needs a context: a device)
    lib.lfg_context_destroy(None)                   # NULL-safe, like the reference's Cleanup()
    lib.lfg_frame_destroy(None, None)
