"""ctypes bindings to the in-tree native libraries.

libdpamd_hip.so   hand-written gfx950 kernels (C ABI; pointers + hipStream_t)
libdpamd_host.so  C++ host runtime (sampler, IDX reader, bucket planner, TCP store)

torch must be imported before the HIP library is loaded: the library links `libamdhip64.so.7` without an
rpath, so the dynamic loader binds it to the HIP runtime torch has already mapped and both share one
runtime (streams, events and device pointers are interchangeable).  On a machine with a GPU a missing
library is a hard error - there is no silent eager fallback.
"""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL of the HIP library)

# DPA_LIB_DIR: an A/B build of the native libraries (python -m distributedpytorch_amd.csrc.build --libdir DIR -D...)
_LIBDIR = os.environ.get("DPA_LIB_DIR") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib")
_hip = None
_host = None

VP = ctypes.c_void_p
I32 = ctypes.c_int
I64 = ctypes.c_longlong
F32 = ctypes.c_float


class FastDiv(ctypes.Structure):
    _fields_ = [("mul", ctypes.c_uint32), ("shift", ctypes.c_uint32), ("d", ctypes.c_uint32)]


class ConvParams(ctypes.Structure):
    _fields_ = [
        ("src", VP), ("wgt", VP), ("dst", VP), ("bias", VP), ("stats", VP), ("addend", VP), ("stats_shift", VP), ("wgt3", VP), ("wgt3_plane", I64),
        ("bnr_x", VP), ("bnr_y", VP), ("bnr_mean", VP), ("bnr_invstd", VP), ("bnr_mscale", VP), ("bnr_mshift", VP), ("bnr_slab", VP), ("bnr_bits", VP),
        ("bnr2_x", VP), ("bnr2_mean", VP), ("bnr2_invstd", VP), ("bnr2_slab", VP), ("bnr2_cs", I32), ("bnr2_pad", I32),
        ("bnr_cs", I32), ("bnr_y_cs", I32), ("bnr_row0", I32), ("bnr_pad", I32), ("oscale", VP),
        ("addend2", VP), ("a2_H", I32), ("a2_W", I32), ("a2_cs", I32), ("a2_pad", I32),
        ("sN", I32), ("sH", I32), ("sW", I32), ("sC", I32), ("s_cs", I32), ("s_co", I32),
        ("dH", I32), ("dW", I32), ("dC", I32), ("d_cs", I32), ("d_co", I32),
        ("R", I32), ("S", I32), ("sh", I32), ("sw", I32), ("ph", I32), ("pw", I32),
        ("M", I32), ("Kg", I32), ("relu", I32), ("no_uni", I32), ("m_off", I32), ("stats_row0", I32), ("splitK", I32), ("kt_per_split", I32), ("ws", VP), ("tilesM", I32), ("tilesN", I32),
        ("cls_h0", I32), ("cls_w0", I32), ("cls_nh", I32), ("cls_nw", I32), ("base_y", I32), ("base_x", I32),
        ("tap_r0", I32), ("tap_s0", I32), ("nR", I32), ("nS", I32), ("stats_rows_used", I32),
        ("fd_nW", FastDiv), ("fd_nH", FastDiv), ("fd_sC", FastDiv), ("fd_nS", FastDiv),
    ]


class WgradParams(ctypes.Structure):
    _fields_ = [
        ("dy", VP), ("x", VP), ("out", VP),
        ("N", I32), ("H", I32), ("W", I32), ("C", I32), ("c_cs", I32), ("c_co", I32),
        ("Ho", I32), ("Wo", I32), ("K", I32), ("k_cs", I32), ("k_co", I32),
        ("R", I32), ("S", I32), ("sh", I32), ("sw", I32), ("ph", I32), ("pw", I32),
        ("P", I32), ("Ng", I32), ("tilesM", I32), ("tilesN", I32), ("splitK", I32), ("pixPerSplit", I32),
        ("accumulate", I32), ("xcd_order", I32),
        ("fd_Wo", FastDiv), ("fd_Ho", FastDiv), ("fd_C", FastDiv), ("fd_S", FastDiv),
    ]


def lib_path(name: str) -> str:
    return os.path.join(_LIBDIR, name)


# entry points whose int return value is NOT a launch status (sizes, counts, modes) or is checked by its own caller
_NOT_LAUNCH = {"dpa_conv_params_size", "dpa_wgrad_params_size", "dpa_pool_params_size", "dpa_aug_params_size",
               "dpa_bn_slab_rows", "dpa_conv_stats_rows", "dpa_conv_wgrad_workspace", "dpa_get_fp32_mode",
               "dpa_set_fp32_mode", "dpa_x6p_min_tiles", "dpa_x6p_knob", "dpa_wgrad_knob", "dpa_x6s_knob", "dpa_maxpool_bwd_bn", "dpa_x6p_plausible", "dpa_weight_prep_seg_size", "dpa_conv_igemm", "dpa_conv_wgrad", "dpa_init_workspaces"}


class KernelLaunchError(RuntimeError):
    pass


class _CheckedLib:
    """The kernel library with every launching entry point's status checked: each DPA_API launcher returns
    hipGetLastError() (csrc/hip/common.h: dpa_launch_rc), and a rejected launch - too much dynamic LDS, an oversize
    grid - raises here instead of leaving a torch.empty output uninitialised."""

    def __init__(self, lib):
        self._lib = lib
        self._cache = {}

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            raw = getattr(self._lib, name)
            if name in _NOT_LAUNCH:
                fn = raw
            else:
                def fn(*args, _raw=raw, _name=name):
                    rc = _raw(*args)
                    if rc:
                        raise KernelLaunchError(f"{_name}: kernel launch failed with hipError_t {rc}")
                    return 0
            self._cache[name] = fn
        return fn


def hip_lib():
    """The HIP kernel library; raises if it has not been built (python -m distributedpytorch_amd.csrc.build)."""
    global _hip
    if _hip is None:
        path = lib_path("libdpamd_hip.so")
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build the native kernels first with "
                "`python -m distributedpytorch_amd.csrc.build` (the HIP path has no eager fallback)")
        lib = ctypes.CDLL(path)
        for fn in ("dpa_conv_params_size", "dpa_wgrad_params_size"):
            getattr(lib, fn).restype = I32
        if lib.dpa_conv_params_size() != ctypes.sizeof(ConvParams):
            raise RuntimeError("ConvParams ABI mismatch between ops/native.py and conv_common.h")
        if lib.dpa_wgrad_params_size() != ctypes.sizeof(WgradParams):
            raise RuntimeError("WgradParams ABI mismatch between ops/native.py and conv_wgrad.hip")
        lib.dpa_conv_wgrad_workspace.restype = I64
        lib.dpa_get_fp32_mode.restype = I32
        if torch.cuda.is_available():
            # the per-device scratch of the split-K and multi-block-loss paths: allocated here, never inside a launch
            # path (a hipGraph capture must not see a hipMalloc)
            rc = lib.dpa_init_workspaces()
            if rc:
                raise RuntimeError(f"dpa_init_workspaces failed (rc={rc}): cannot allocate the kernel scratch buffers")
        _hip = _CheckedLib(lib)
    return _hip


FP32_MODES = ("exact", "x6")


def set_fp32_mode(mode: str) -> str:
    """Arithmetic of the fp32 convolution kernels: 'exact' = v_mfma_f32_16x16x4_f32, 'x6' (default) = every fp32
    product as six bf16 x bf16 products of exact 3-way bf16 splits on the bf16 matrix pipe, fp32 accumulation
    (csrc/hip/conv_common.h).  Returns the previous mode.  Takes effect for launches issued afterwards."""
    if mode not in FP32_MODES:
        raise ValueError(f"fp32 mode must be one of {FP32_MODES}, got {mode!r}")
    prev = get_fp32_mode()
    hip_lib().dpa_set_fp32_mode(I32(FP32_MODES.index(mode)))
    return prev


def get_fp32_mode() -> str:
    return FP32_MODES[int(hip_lib().dpa_get_fp32_mode())]


def host_lib():
    global _host
    if _host is None:
        path = lib_path("libdpamd_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -m distributedpytorch_amd.csrc.build`")
        _host = ctypes.CDLL(path)
    return _host


def have_hip_lib() -> bool:
    return os.path.exists(lib_path("libdpamd_hip.so"))


def have_host_lib() -> bool:
    return os.path.exists(lib_path("libdpamd_host.so"))


def stream_ptr() -> VP:
    return VP(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> VP:
    return VP(t.data_ptr()) if t is not None else VP(0)
