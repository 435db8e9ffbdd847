"""CPU: the C-ABI library loads, exports every symbol include/rcf_hip.h declares, and rejects bad
arguments with an error code before touching the GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

import rcf_amd
from rcf_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "rcf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rcf_[a-z0-9_]+)\s*\(", src)))


@pytest.mark.parametrize("which", ["bf16", "f16"])
def test_header_symbols_exported_and_bound(which):
    """both builds of the library: librcf_hip.so (bf16 as the 16-bit storage type) and librcf_hip_f16.so (IEEE fp16)"""
    names = declared_functions()
    assert len(names) >= 30
    lib = _lib.load(which)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/rcf_hip.h but not exported: {missing}"
    unbound = [n for n in names if n not in _lib.PROTOS]
    assert not unbound, f"exported but without a ctypes prototype: {unbound}"
    assert not [n for n in _lib.PROTOS if n not in names], "ctypes prototypes for undeclared functions"
    assert lib.rcf_version().decode().startswith("rcf_hip") and "gfx950" in lib.rcf_version().decode()


def test_bad_arguments_return_codes_not_crashes():
    lib = _lib.load()
    sz = ctypes.sizeof(_lib.ConvShape)
    shape = lambda *dims, size=sz: _lib.ConvShape(*dims, None, None, None, None, None, None, None, None, 0, size)
    s = shape(1, 8, 8, 3, 8, 8, 8, 3, 3, 1, 1, 1, 3, 8)                  # Cin % 4 != 0
    assert lib.rcf_conv2d_fwd_f32(None, None, None, None, ctypes.byref(s), 0, 0.0, 0, None) == -1
    s = shape(1, 8, 8, 4, 7, 8, 8, 3, 3, 1, 1, 1, 4, 8)                  # wrong Ho
    assert lib.rcf_conv2d_dgrad_f32(None, None, None, ctypes.byref(s), 0, None, 0, None) == -1
    assert lib.rcf_bn_stats_f32(None, 10, 6, 6, None, None, 0, None) == -1
    assert lib.rcf_crf_soft(None, None, 8, 8, 1, 0., 0., 5., 60., 5., 5, None, None, None, None, 0, None) == -1
    assert lib.rcf_flow_warp_f32(None, None, None, 1, 3, 8, 8, 0, None) == -1
    assert lib.rcf_crf_workspace_bytes(854, 480, 1) > 100e6
    good = (16, 120, 214, 64, 120, 214, 64, 3, 3, 1, 1, 1, 64, 64)
    assert lib.rcf_conv2d_wgrad_workspace_bytes(ctypes.byref(shape(*good))) > 0
    # a caller compiled against another (older, shorter) rcf_conv_shape is refused, not read out of bounds
    assert lib.rcf_conv2d_wgrad_workspace_bytes(ctypes.byref(shape(*good, size=sz - 8))) == 0
    assert lib.rcf_conv2d_fwd_f32(None, None, None, None, ctypes.byref(shape(*good, size=0)), 0, 0.0, 0, None) == -1
    with pytest.raises(_lib.RcfHipError):
        _lib.call("rcf_copy2d_f32", None, 4, None, 4, 1, 4, 0, None)


def test_product_refuses_cpu_tensors():
    import torch
    from rcf_amd import ops
    with pytest.raises(_lib.RcfHipError):
        ops.conv2d_fwd(torch.zeros(1, 4, 4, 4), torch.zeros(4, 4, 1, 1))
    with pytest.raises(_lib.RcfHipError):
        ops.flow_warp(torch.zeros(1, 3, 4, 4), torch.zeros(1, 2, 4, 4))


def test_half_storage_switches_the_library_and_guards_the_types():
    """rcf_amd.ops.half_storage: the 16-bit storage type is a property of the library BUILD; a context picks the build, a tensor of
    the other 16-bit type is refused before any launch"""
    import torch
    from rcf_amd import ops
    assert _lib.ACTIVE == "bf16" and ops.half() == torch.bfloat16
    with ops.half_storage(torch.float16):
        assert _lib.ACTIVE == "f16" and ops.half() == torch.float16 and _lib.load() is _lib.load("f16")
        assert ops._dt(torch.zeros(1, dtype=torch.float16)) == _lib.BF16 and ops._dt(torch.zeros(1)) == _lib.F32
        with pytest.raises(_lib.RcfHipError, match="half_storage"):
            ops._dt(torch.zeros(1, dtype=torch.bfloat16))
        with ops.half_storage(torch.float32):                       # an fp32 model inside: leaves the choice alone
            assert _lib.ACTIVE == "f16"
    assert _lib.ACTIVE == "bf16" and _lib.load() is _lib.load("bf16") and _lib.load("bf16") is not _lib.load("f16")
    with pytest.raises(_lib.RcfHipError, match="half_storage"):
        ops._dt(torch.zeros(1, dtype=torch.float16))
