"""ctypes binding of the C ABI (include/gs2d_rasterizer.h).  Fails loudly when the HIP library is missing:
there is no CPU fallback anywhere in this package."""
import ctypes as C
import os

from . import build as _build

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)

EXPORTS = ["gs2d_forward", "gs2d_backward", "gs2d_forward_posed", "gs2d_backward_posed", "gs2d_mark_visible", "sknn_dist2", "gs2d_geometry_bytes",
           "gs2d_image_bytes", "gs2d_binning_bytes", "gs2d_geometry_layout", "gs2d_binning_layout",
           "gs2d_image_layout", "gs2d_last_error", "gs2d_build_info", "gs2d_stage_timing_enable",
           "gs2d_stage_timing_read", "gs2d_slam_loss", "gs2d_adam_step", "gs2d_pose_quat", "gs2d_backward_staged", "gs2d_set_deterministic",
           "gs2d_get_deterministic", "gs2d_set_reference_binning", "gs2d_get_reference_binning", "gs2d_set_launch_ahead", "gs2d_get_launch_ahead", "gs2d_forward_batch",
           "gs2d_backward_batch", "gs2d_stage_timing_read_abs"]
MAX_FRAMES = 8  # GS2D_MAX_FRAMES


class FrameIO(C.Structure):
    """gs2d_frame_io (include/gs2d_rasterizer.h)."""
    _fields_ = [("geometry_alloc", ALLOC_FN), ("geometry_user", C.c_void_p), ("binning_alloc", ALLOC_FN),
                ("binning_user", C.c_void_p), ("image_alloc", ALLOC_FN), ("image_user", C.c_void_p),
                ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("cam_pos", C.c_void_p),
                ("out_color", C.c_void_p), ("out_others", C.c_void_p), ("radii", C.c_void_p)]


class FrameGrad(C.Structure):
    """gs2d_frame_grad (include/gs2d_rasterizer.h)."""
    _fields_ = [("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p), ("tan_fovx", C.c_float),
                ("tan_fovy", C.c_float), ("radii", C.c_void_p), ("geom_buffer", C.c_void_p), ("binning_buffer", C.c_void_p),
                ("img_buffer", C.c_void_p), ("num_rendered", C.c_int), ("dL_dpix", C.c_void_p), ("dL_depths", C.c_void_p),
                ("dL_dmean2D", C.c_void_p), ("dL_dnormal", C.c_void_p), ("dL_dopacity", C.c_void_p), ("dL_dcolor", C.c_void_p),
                ("dL_dmean3D", C.c_void_p), ("dL_dtransMat", C.c_void_p), ("dL_dsh", C.c_void_p), ("dL_dscale", C.c_void_p),
                ("dL_drot", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            f"gaus_slam_amd: HIP library {path} is missing. Build it with `python -m gaus_slam_amd.build` "
            "(needs hipcc); this package has no CPU fallback.")
    # torch first: its wheel bundles its own HIP runtime, and a process must end up with ONE libamdhip64.  Loading this
    # library before torch pulls in /opt/rocm's copy as a second runtime, which then reports "no ROCm-capable device".
    import torch  # noqa: F401
    L = C.CDLL(path)
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    L.gs2d_forward.restype = i
    L.gs2d_forward.argtypes = [ALLOC_FN, vp, ALLOC_FN, vp, ALLOC_FN, vp, i, i, i, vp, i, i, vp, vp, vp, vp, vp, f, vp, vp,
                               vp, vp, vp, f, f, i, vp, vp, vp, i, i, vp]
    L.gs2d_backward.restype = i
    L.gs2d_backward.argtypes = [i, i, i, i, vp, i, i, vp, vp, vp, vp, f, vp, vp, vp, vp, vp, f, f, vp, vp, vp, vp, vp, vp,
                                vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, vp]
    L.gs2d_forward_posed.restype = i
    L.gs2d_forward_posed.argtypes = L.gs2d_forward.argtypes[:-1] + [vp, vp, vp]
    L.gs2d_backward_posed.restype = i
    L.gs2d_backward_posed.argtypes = L.gs2d_backward.argtypes[:-1] + [vp, vp, vp, vp]
    L.gs2d_backward_staged.restype = i
    L.gs2d_backward_staged.argtypes = [i, i, i] + L.gs2d_backward_posed.argtypes
    L.gs2d_forward_batch.restype = i
    L.gs2d_forward_batch.argtypes = [i, C.POINTER(FrameIO), i, i, i, vp, i, i, vp, vp, vp, vp, vp, f, vp, vp, i, i,
                                     C.POINTER(C.c_int), vp]
    L.gs2d_backward_batch.restype = i
    L.gs2d_backward_batch.argtypes = [i, C.POINTER(FrameGrad), i, i, i, i, vp, i, i, vp, vp, vp, vp, f, vp, vp, i, i, vp]
    L.gs2d_pose_quat.restype = i
    L.gs2d_pose_quat.argtypes = [vp, vp, vp]
    L.gs2d_mark_visible.restype = i
    L.gs2d_mark_visible.argtypes = [i, vp, vp, vp, vp, vp]
    L.sknn_dist2.restype = i
    L.sknn_dist2.argtypes = [i, vp, vp, ALLOC_FN, vp, vp]
    for n in ("gs2d_geometry_bytes", "gs2d_binning_bytes"):
        getattr(L, n).restype = sz
        getattr(L, n).argtypes = [i]
    L.gs2d_image_bytes.restype = sz
    L.gs2d_image_bytes.argtypes = [i, i]
    L.gs2d_geometry_layout.argtypes = [i, C.POINTER(sz)]
    L.gs2d_binning_layout.argtypes = [i, C.POINTER(sz)]
    L.gs2d_image_layout.argtypes = [i, i, C.POINTER(sz)]
    L.gs2d_slam_loss.restype = i
    L.gs2d_slam_loss.argtypes = [i, i, i, vp, vp, vp, vp, f, f, f, f, f, i, i, f, f, f, vp, vp, vp, vp, vp, vp]
    L.gs2d_adam_step.restype = i
    L.gs2d_adam_step.argtypes = [i, vp, vp, f, f, f, i, C.c_ulonglong, vp, vp, vp, vp, vp]
    L.gs2d_set_deterministic.argtypes = [i]
    L.gs2d_get_deterministic.restype = i
    L.gs2d_set_reference_binning.argtypes = [i]
    L.gs2d_set_launch_ahead.argtypes = [i]
    L.gs2d_get_launch_ahead.restype = i
    L.gs2d_get_reference_binning.restype = i
    L.gs2d_stage_timing_enable.argtypes = [i]
    L.gs2d_stage_timing_read.restype = i
    L.gs2d_stage_timing_read.argtypes = [C.POINTER(C.c_float)]
    L.gs2d_stage_timing_read_abs.restype = i
    L.gs2d_stage_timing_read_abs.argtypes = [C.POINTER(C.c_float)]
    L.gs2d_last_error.restype = C.c_char_p
    L.gs2d_build_info.restype = C.c_char_p
    _lib = L
    return L


def last_error():
    return lib().gs2d_last_error().decode()


def build_info():
    """gs2d_build_info(): flags, build date and the hash of the kernel sources the loaded library was compiled from."""
    return lib().gs2d_build_info().decode()


def lib_source_hash():
    """The source hash compiled into the loaded library ("unknown" for a library not built by gaus_slam_amd/build.py)."""
    info = build_info()
    return info.rsplit(" src ", 1)[1] if " src " in info else "unknown"
