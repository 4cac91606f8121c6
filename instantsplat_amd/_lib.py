"""ctypes binding of libmi355gs.so (C ABI: include/mi355gs.h).

There is NO fallback: if the hipcc-built library is missing this module raises, and every operator
refuses CPU tensors.  (The CPU test-suite may inject the SIMT-emulated build of the same kernel
sources through `_use_library_for_testing`; nothing in the product path calls that.)
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmi355gs.so")
_LIB = None
_TEST_MODE = False

_P = c_void_p
_SIGNATURES = {
    "mi355gs_abi_version": (c_int, []),
    "mi355gs_error_string": (ctypes.c_char_p, [c_int]),
    "mi355gs_raster_geom_bytes": (c_size_t, [c_int]),
    "mi355gs_raster_tiles_bytes": (c_size_t, [c_int, c_int]),
    "mi355gs_raster_binning_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "mi355gs_raster_grad_scratch_bytes": (c_size_t, [c_int]),
    "mi355gs_raster_grad_gate_offset": (c_size_t, [c_int]),
    "mi355gs_raster_forward_preprocess": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_float, _P, _P,
                                                  _P, _P, _P, c_float, c_float, c_int, _P, _P, _P, _P, _P, _P, c_int]),
    "mi355gs_raster_forward_render": (c_int, [_P, c_int, c_int, c_int, c_int64, _P, _P, _P, _P, _P, c_int]),
    "mi355gs_raster_binning_bytes_render_only": (c_size_t, [c_int64, c_int, c_int]),
    "mi355gs_raster_forward_render_only": (c_int, [_P, c_int, c_int, c_int, c_int64, _P, _P, _P, _P, _P, c_int]),
    "mi355gs_raster_backward": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P,
                                        _P, c_float, c_float, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                        c_int, c_int]),
    "mi355gs_raster_mark_visible": (c_int, [_P, c_int, _P, _P, _P, _P]),
    "mi355gs_raster_frame_stats": (c_int, [_P, c_int, c_int, _P, _P]),
    "mi355gs_tune_min_units": (c_int, [c_int]),
    "mi355gs_tune_scale_grad": (c_int, [c_int]),
    "mi355gs_tune_deterministic": (c_int, [c_int]),
    "mi355gs_profile_begin": (c_int, []),
    "mi355gs_profile_set_period": (c_int, [c_int]),
    "mi355gs_profile_work_counters": (c_int, [_P]),
    "mi355gs_profile_read": (c_int, [c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int)]),
    "mi355gs_profile_end": (c_int, []),
    "mi355gs_profile_ranges": (c_int, [c_int]),
    "mi355gs_ssim_scratch_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mi355gs_ssim_forward": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int]),
    "mi355gs_ssim_backward": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int]),
    "mi355gs_l1_ssim_loss_fused": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, c_float, _P, _P, _P, _P]),
    "mi355gs_l1_ssim_pair_forward": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "mi355gs_l1_ssim_pair_backward": (c_int, [_P, c_int64, _P, _P, _P, _P, c_float, _P, c_float, _P]),
    "mi355gs_loss_program_eval": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_float]),
    "mi355gs_loss_program_eval_grad": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_float, _P, _P, _P, c_float,
                                               c_float, _P]),
    "mi355gs_knn_scratch_bytes": (c_size_t, [c_int]),
    "mi355gs_knn_dist2": (c_int, [_P, c_int, _P, _P, _P]),
    "mi355gs_adam_step": (c_int, [_P, c_int64, c_int, _P, _P, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_int]),
    "mi355gs_adam_multi_step": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, c_float, _P, _P, _P, _P, _P,
                                        ctypes.c_uint32]),
    "mi355gs_pose_forward": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mi355gs_pose_backward": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mi355gs_posed_forward_preprocess": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P,
                                                 c_float, c_float, _P, _P, _P, _P, _P, _P, c_int]),
    "mi355gs_posed_backward": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, c_float,
                                       c_float, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int]),
    "mi355gs_l1_scratch_bytes": (c_size_t, [c_int64]),
    "mi355gs_l1_loss_forward": (c_int, [_P, c_int64, _P, _P, _P, _P]),
    "mi355gs_l1_loss_backward": (c_int, [_P, c_int64, _P, _P, _P, _P]),
    "mi355gs_trainer_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int64]),
    "mi355gs_trainer_create": (c_void_p, [c_int, c_int, c_int, c_int, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mi355gs_trainer_step": (c_int, [_P, _P, c_int, c_int, _P, _P, c_float, c_float, _P, _P, _P, c_float, c_float, c_float, c_float,
                                     c_int, _P, _P]),
    "mi355gs_trainer_optimizer_step": (c_int, [_P, _P, _P, _P, c_float, c_float, c_float, c_int]),
    "mi355gs_trainer_destroy": (None, [_P]),
    "mi355gs_trainer_grad": (_P, [_P, c_int]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


ABI_VERSION = 10   # include/mi355gs.h MI355GS_ABI_VERSION the signatures above were written for


def _bind(path: str):
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = the .so does not match include/mi355gs.h
        fn.restype, fn.argtypes = res, args
    got = lib.mi355gs_abi_version()
    if got != ABI_VERSION:   # a stale build that happens to export every name would be called with the wrong argument lists
        raise RuntimeError(f"{path} implements ABI v{got}, this package binds v{ABI_VERSION}: rebuild it with "
                           "`python -c 'import __graft_entry__ as g; g.build()'`")
    return lib


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). instantsplat_amd has no CPU or PyTorch fallback.")
        _LIB = _bind(LIB_PATH)
        if os.environ.get("MI355GS_DETERMINISTIC", "0") == "1":   # the deterministic backward (include/mi355gs.h), process-wide
            _LIB.mi355gs_tune_deterministic(1)
    return _LIB


def _use_library_for_testing(path: str | None):
    """tests/ only: run the emulated (g++-compiled) build of the kernel sources on CPU tensors."""
    global _LIB, _TEST_MODE, _EXT_BOUND_TO
    if path is None:
        _LIB, _TEST_MODE = None, False
    else:
        _LIB, _TEST_MODE = _bind(path), True
    _EXT_BOUND_TO = None   # the compiled binding is re-bound to the new library's entry points on its next use


# ---- compiled PyTorch binding (csrc_torch/binding.cpp): the same C-ABI calls made from C++ autograd nodes.
# BINDING = "compiled": use it wherever it covers the call (render_posed, fused_l1_ssim_loss, PerPointAdam.step);
# "ctypes": the Python autograd.Function binding everywhere (kept for A/B and as the reference for the compiled one's tests).
BINDING = os.environ.get("MI355GS_BINDING", "compiled")
EXT_PATH = os.path.join(_HERE, "lib", "_mi355gs_torch.so")
_EXT = None
_EXT_BOUND_TO = None
_EXT_SYMBOLS = ("mi355gs_raster_geom_bytes", "mi355gs_raster_tiles_bytes", "mi355gs_raster_binning_bytes",
                "mi355gs_raster_grad_scratch_bytes", "mi355gs_raster_grad_gate_offset", "mi355gs_posed_forward_preprocess", "mi355gs_raster_forward_preprocess", "mi355gs_raster_backward",
                "mi355gs_raster_forward_render", "mi355gs_raster_binning_bytes_render_only", "mi355gs_raster_forward_render_only", "mi355gs_posed_backward", "mi355gs_ssim_scratch_bytes", "mi355gs_l1_ssim_loss_fused", "mi355gs_ssim_forward", "mi355gs_ssim_backward",
                "mi355gs_adam_multi_step", "mi355gs_error_string", "mi355gs_l1_scratch_bytes", "mi355gs_l1_loss_forward", "mi355gs_l1_loss_backward",
                "mi355gs_l1_ssim_pair_forward", "mi355gs_l1_ssim_pair_backward", "mi355gs_loss_program_eval",
                "mi355gs_loss_program_eval_grad")


def compiled():
    """The compiled binding bound to the library `lib()` currently returns, or None when BINDING == "ctypes".
    A missing build is an error, not a silent fallback: both bindings run the same kernels, but the drop-in loop is
    ~30 % slower through ctypes and nothing else would say so."""
    global _EXT, _EXT_BOUND_TO
    if BINDING != "compiled":
        return None
    L = lib()
    if _EXT_BOUND_TO is L:
        return _EXT
    if _EXT is None:
        if not os.path.exists(EXT_PATH):
            raise RuntimeError(f"{EXT_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(or set MI355GS_BINDING=ctypes to use the Python binding)")
        import importlib.util
        spec = importlib.util.spec_from_file_location("_mi355gs_torch", EXT_PATH)
        _EXT = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_EXT)
    _EXT.forget_gates()
    _EXT.bind({name: ctypes.cast(getattr(L, name), ctypes.c_void_p).value for name in _EXT_SYMBOLS}, _TEST_MODE)
    _EXT_BOUND_TO = L
    return _EXT


def check(code: int, what: str):
    if code != 0:
        raise RuntimeError(f"mi355gs: {what} failed: {lib().mi355gs_error_string(code).decode()} ({code})")


def ptr(t: torch.Tensor | None):
    """A tensor's address as ctypes takes it for a void* argument (a plain int; None = NULL)."""
    return None if t is None else t.data_ptr()


_F32 = torch.float32


def f32c_on_one_device(*tensors):
    """f32c + require_device in one pass over the arguments: -> ([contiguous float32 tensors], their common device)."""
    out, dev = [], None
    for t in tensors:
        if t.dtype is not _F32:
            raise RuntimeError(f"expected float32, got {t.dtype}")
        if not t.is_contiguous():
            t = t.contiguous()
        d = t.device
        if dev is None:
            if d.type != "cuda" and not _TEST_MODE:
                raise RuntimeError("instantsplat_amd operators run on the GPU only (got a CPU tensor; there is no CPU fallback)")
            dev = d
        elif d != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {d}")
        out.append(t)
    return out, dev


def require_device(*tensors: torch.Tensor | None):
    """All given tensors must be fp32/int, contiguous, on one HIP device. Returns that device."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda and not _TEST_MODE:
            raise RuntimeError("instantsplat_amd operators run on the GPU only (got a CPU tensor; there is no CPU fallback)")
        if not t.is_contiguous():
            raise RuntimeError("instantsplat_amd operators need contiguous tensors")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NO_GUARD = _NoGuard()


def on_device(device):
    """Context for a library call: the HIP launches inside libmi355gs.so go to the process's CURRENT device, so when the
    tensors live on another one (cuda:1 while cuda:0 is current) the call is wrapped in torch.cuda.device(device).  The
    common case — tensors on the current device — costs one comparison."""
    if device is not None and device.type == "cuda" and device.index is not None and device.index != torch.cuda.current_device():
        return torch.cuda.device(device)
    return _NO_GUARD


def stream_ptr(device):
    if device is not None and device.type == "cuda":
        return torch.cuda.current_stream(device).cuda_stream
    return None


def f32c(t: torch.Tensor | None):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected float32, got {t.dtype}")
    return t.contiguous()
