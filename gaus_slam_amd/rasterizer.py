"""Host-side mirror of the reference operator surface for the 2D-Gaussian-surfel rasterizer.

Same names, argument order, return order and error behaviour as
RAST/gaus_2dgs_rasterization/__init__.py (GaussianRasterizationSettings :163-176, GaussianRasterizer :178-227,
_RasterizeGaussians :44-161) and the pybind functions of RAST/ext.cpp:15-19 / RAST/rasterize_points.cu:39-260
(rasterize_gaussians, rasterize_gaussians_backward, mark_visible), so render/render_2dgs.py and everything above
it runs unchanged on PyTorch-ROCm.  All compute happens in the HIP library behind the C ABI
(include/gs2d_rasterizer.h); torch only provides device memory and the current stream.
"""
import contextlib
import ctypes as C
import itertools
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib

NUM_CHANNELS = 3


def _ptr(t):
    """Device pointer or NULL for empty tensors (the reference relies on empty tensors having a null data pointer,
    rasterizer_impl.cu:327-328)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def _check_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")  # rasterize_points.cu:27-28


def _f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


class _Chunk:
    """Allocator callback target: the C side asks for N bytes, we hand out a torch uint8 tensor (the resizeFunctional
    lambda of rasterize_points.cu:31-37).  ONE ctypes callback exists per process (creating CFUNCTYPE objects per call
    costs tens of microseconds); the `user` pointer the C side passes back selects the live _Chunk."""

    _live = {}
    _next = itertools.count(1)  # next() on a count is atomic under the GIL: concurrent host threads never share a key

    def __init__(self, device):
        self.device = device
        self.tensor = torch.empty(0, dtype=torch.uint8, device=device)
        self.key = next(_Chunk._next)
        _Chunk._live[self.key] = self
        self.cb = _CHUNK_CB
        self.user = C.c_void_p(self.key)

    def release(self):
        _Chunk._live.pop(self.key, None)


def _chunk_alloc(user, nbytes):
    ch = _Chunk._live[int(user)]
    ch.tensor = torch.empty(int(nbytes), dtype=torch.uint8, device=ch.device)
    return ch.tensor.data_ptr()


_CHUNK_CB = _lib.ALLOC_FN(_chunk_alloc)


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(device):
    """Raw hipStream_t of torch's current stream on `device` (fast path: no Stream object is built)."""
    if _RAW_STREAM is not None:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        return C.c_void_p(_RAW_STREAM(idx))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _on_device:
    """`with torch.cuda.device(dev)` that costs nothing when `dev` already is the current device (the usual case)."""

    def __init__(self, device):
        idx = device.index
        self.ctx = None if idx is None or idx == torch.cuda.current_device() else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, transMat_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        use_sa, prefiltered, debug, pose_Rt=None, pose_quat=None):
    """_C.rasterize_gaussians (rasterize_points.cu:39-138): returns
    (num_rendered, out_color[3,H,W], out_others[7,H,W], radii[P], geomBuffer, binningBuffer, imgBuffer).
    pose_Rt [3,4] / pose_quat [4] (extension, see gs2d_forward_posed): rigid transform fused into the preprocess."""
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    for name, t in (("background", background), ("means3D", means3D), ("colors", colors), ("opacity", opacity),
                    ("scales", scales), ("rotations", rotations), ("transMat_precomp", transMat_precomp),
                    ("viewmatrix", viewmatrix), ("projmatrix", projmatrix), ("sh", sh), ("campos", campos)):
        _check_cuda(t, name)
    L = _lib.lib()
    dev = means3D.device
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    # P == 0 returns zero images (rasterize_points.cu:100-101); otherwise the kernels write every output element,
    # so no fill kernels are spent on them.
    alloc = torch.zeros if P == 0 else torch.empty
    out_color = alloc((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
    out_others = alloc((7, H, W), dtype=torch.float32, device=dev)
    radii = alloc((P,), dtype=torch.int32, device=dev)
    geom, binning, img = _Chunk(dev), _Chunk(dev), _Chunk(dev)
    rendered = 0
    try:
        if P != 0:
            M = sh.size(1) if sh.size(0) != 0 else 0
            keep = [_f32c(t) for t in (background, means3D, sh, colors, opacity, scales, rotations, transMat_precomp,
                                       viewmatrix, projmatrix, campos)]
            bg_, m3_, sh_, col_, op_, sc_, rot_, tm_, vm_, pm_, cp_ = keep
            with _on_device(dev):
                prt_ = _f32c(pose_Rt) if pose_Rt is not None else None
                pq_ = _f32c(pose_quat) if pose_quat is not None else None
                rendered = L.gs2d_forward_posed(
                    geom.cb, geom.user, binning.cb, binning.user, img.cb, img.user, P, int(degree), M, _ptr(bg_), W, H,
                    _ptr(m3_), _ptr(sh_), _ptr(col_), _ptr(op_), _ptr(sc_), float(scale_modifier), _ptr(rot_), _ptr(tm_),
                    _ptr(vm_), _ptr(pm_), _ptr(cp_), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
                    out_color.data_ptr(), out_others.data_ptr(), radii.data_ptr(), int(bool(use_sa)), int(bool(debug)),
                    _ptr(prt_), _ptr(pq_), _stream_ptr(dev))
    finally:  # the callback registry never keeps a chunk of a call that raised
        for ch in (geom, binning, img):
            ch.release()
    if rendered < 0:
        raise RuntimeError(_lib.last_error())
    return rendered, out_color, out_others, radii, geom.tensor, binning.tensor, img.tensor


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                 transMat_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_others, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, use_sa,
                                 debug, pose_Rt=None, pose_quat=None, grad_sink=None, lean=False, pose_only_out=None,
                                 chunk_rows=None, on_chunk=None):
    """_C.rasterize_gaussians_backward (rasterize_points.cu:140-239): returns
    (dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dtransMat[P,9], dL_dsh[P,M,3],
     dL_dscales[P,2], dL_drotations[P,4]).
    grad_sink (optional): dict with any of means3D / opacities / scales / rotations / colors -> contiguous fp32 tensor of
    the gradient's shape; the kernels then write those gradients there (e.g. straight into the all-reduce bucket) instead
    of into fresh tensors.
    lean (used by the autograd node): skip the outputs nobody can observe there -- the internal dL_dnormal and, when no
    cov3D_precomp was given, dL_dtransMat (returned as None): 48 B per Gaussian less to write.
    pose_only_out (tracking): a float32 [4,4] tensor (uninitialised is fine: rows 0-2 receive dL/d[R|t], row 3 zeros -- a
    zero-filled tensor when P == 0); no per-Gaussian gradient
    is computed or allocated and only that tensor is returned.
    chunk_rows / on_chunk (keyframe-sharded BA): run the per-Gaussian stage in chunks of `chunk_rows` Gaussians
    (gs2d_backward_staged) and call on_chunk(g_begin, g_end) after each chunk has been enqueued -- the caller starts that
    chunk's gradient all-reduce there, so it overlaps with the next chunk's kernel."""
    for name, t in (("background", background), ("means3D", means3D), ("radii", radii), ("colors", colors),
                    ("scales", scales), ("rotations", rotations), ("transMat_precomp", transMat_precomp),
                    ("viewmatrix", viewmatrix), ("projmatrix", projmatrix), ("sh", sh), ("campos", campos),
                    ("binningBuffer", binningBuffer), ("imageBuffer", imageBuffer), ("geomBuffer", geomBuffer)):
        _check_cuda(t, name)
    L = _lib.lib()
    dev = means3D.device
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    M = sh.size(1) if sh.size(0) != 0 else 0
    if pose_only_out is not None:
        if pose_Rt is None or M != 0:
            raise RuntimeError("pose_only_out needs a pose and colors_precomp")
        if P != 0:
            keep = [_f32c(t) for t in (background, means3D, colors, scales, rotations, viewmatrix, projmatrix, campos,
                                       dL_dout_color, dL_dout_others, pose_Rt)]
            bg_, m3_, col_, sc_, rot_, vm_, pm_, cp_, dc_, do_, prt_ = keep
            pq_ = _f32c(pose_quat) if pose_quat is not None else None  # None: the kernels derive q_cam from pose_Rt
            with _on_device(dev):
                # stages 1|2|4 (GS2D_BWD_POSE_4X4): all sixteen floats of pose_only_out are written
                rc = L.gs2d_backward_staged(
                    7, 0, P, P, int(degree), 0, int(R), _ptr(bg_), W, H, _ptr(m3_), None, _ptr(col_), _ptr(sc_), float(scale_modifier),
                    _ptr(rot_), None, _ptr(vm_), _ptr(pm_), _ptr(cp_), float(tan_fovx), float(tan_fovy),
                    radii.contiguous().data_ptr(), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), dc_.data_ptr(),
                    do_.data_ptr(), None, None, None, None, None, None, None, None, None, int(bool(use_sa)), int(bool(debug)),
                    _ptr(prt_), _ptr(pq_), pose_only_out.data_ptr(), _stream_ptr(dev))
            if rc < 0:
                raise RuntimeError(_lib.last_error())
        else:
            pose_only_out.zero_()
        return pose_only_out
    # the backward kernels write every element (zeros for culled Gaussians): no torch.zeros fills needed
    z = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dnormal = z(P, 3), z(P, 3), z(P, NUM_CHANNELS), z(P, 3)
    dL_dopacity, dL_dtransMat, dL_dsh, dL_dscales, dL_drotations = z(P, 1), z(P, 9), z(P, M, 3), z(P, 2), z(P, 4)
    if lean:
        dL_dnormal = None
        if transMat_precomp.numel() == 0:
            dL_dtransMat = None
    if grad_sink:
        def sunk(name, t):
            s = grad_sink.get(name)
            if s is None:
                return t
            if s.shape != t.shape or s.dtype != torch.float32 or s.device != dev or not s.is_contiguous():
                raise RuntimeError(f"grad_sink[{name!r}] must be a contiguous fp32 {tuple(t.shape)} tensor on {dev}")
            return s.detach()  # fresh tensor object on the same memory, so autograd can adopt it as .grad without a copy
        dL_dmeans3D, dL_dcolors, dL_dopacity = sunk("means3D", dL_dmeans3D), sunk("colors", dL_dcolors), sunk("opacities", dL_dopacity)
        dL_dscales, dL_drotations = sunk("scales", dL_dscales), sunk("rotations", dL_drotations)
    if P != 0:
        keep = [_f32c(t) for t in (background, means3D, sh, colors, scales, rotations, transMat_precomp, viewmatrix,
                                   projmatrix, campos, dL_dout_color, dL_dout_others)]
        bg_, m3_, sh_, col_, sc_, rot_, tm_, vm_, pm_, cp_, dc_, do_ = keep
        radii_ = radii.contiguous()
        with _on_device(dev):
            prt_ = _f32c(pose_Rt) if pose_Rt is not None else None
            pq_ = _f32c(pose_quat) if pose_quat is not None else None
            dL_dpose = torch.empty((3, 4), dtype=torch.float32, device=dev) if pose_Rt is not None else None
            tail = (P, int(degree), M, int(R), _ptr(bg_), W, H, _ptr(m3_), _ptr(sh_), _ptr(col_), _ptr(sc_),
                    float(scale_modifier), _ptr(rot_), _ptr(tm_), _ptr(vm_), _ptr(pm_), _ptr(cp_), float(tan_fovx),
                    float(tan_fovy), radii_.data_ptr(), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer),
                    dc_.data_ptr(), do_.data_ptr(), dL_dmeans2D.data_ptr(), _ptr(dL_dnormal), dL_dopacity.data_ptr(),
                    dL_dcolors.data_ptr(), dL_dmeans3D.data_ptr(), _ptr(dL_dtransMat), _ptr(dL_dsh),
                    dL_dscales.data_ptr(), dL_drotations.data_ptr(), int(bool(use_sa)), int(bool(debug)), _ptr(prt_),
                    _ptr(pq_), _ptr(dL_dpose), _stream_ptr(dev))
            if on_chunk is None:
                rc = L.gs2d_backward_posed(*tail)
            else:
                rows = max(1, int(chunk_rows or P))
                rc = L.gs2d_backward_staged(1, 0, 0, *tail)  # GS2D_BWD_BLEND
                g0 = 0
                while rc >= 0 and g0 < P:
                    g1 = min(P, g0 + rows)
                    rc = L.gs2d_backward_staged(2, g0, g1, *tail)  # GS2D_BWD_PREPROCESS on [g0, g1)
                    if rc >= 0:
                        on_chunk(g0, g1)
                    g0 = g1
        if rc < 0:
            raise RuntimeError(_lib.last_error())
        if pose_Rt is not None:
            return (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dtransMat, dL_dsh, dL_dscales, dL_drotations,
                    dL_dpose)
    elif pose_Rt is not None:
        return (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dtransMat, dL_dsh, dL_dscales, dL_drotations,
                torch.zeros((3, 4), dtype=torch.float32, device=dev))
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dtransMat, dL_dsh, dL_dscales, dL_drotations


def rasterize_gaussians_batch(background, means3D, colors, opacity, scales, rotations, scale_modifier, transMat_precomp,
                              viewmatrices, projmatrices, image_height, image_width, sh, degree, camposs, use_sa, debug):
    """K frames of the same size over the same Gaussians in one call (gs2d_forward_batch): per frame exactly what
    rasterize_gaussians returns, the blend pass as ONE grid over the tiles of all frames.  viewmatrices / projmatrices:
    [K,4,4] (or [K,16]), camposs: [K,3].  Returns (num_rendered list[K], out_color [K,3,H,W], out_others [K,7,H,W],
    radii [K,P], geomBuffers, binningBuffers, imgBuffers: lists of K uint8 tensors)."""
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    for name, t in (("background", background), ("means3D", means3D), ("colors", colors), ("opacity", opacity),
                    ("scales", scales), ("rotations", rotations), ("transMat_precomp", transMat_precomp),
                    ("viewmatrices", viewmatrices), ("projmatrices", projmatrices), ("sh", sh), ("camposs", camposs)):
        _check_cuda(t, name)
    L = _lib.lib()
    dev = means3D.device
    K = viewmatrices.size(0)
    if not 1 <= K <= _lib.MAX_FRAMES or projmatrices.size(0) != K or camposs.size(0) != K:
        raise RuntimeError(f"need 1..{_lib.MAX_FRAMES} frames with one view matrix, projection matrix and camera position each")
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    alloc = torch.zeros if P == 0 else torch.empty
    out_color = alloc((K, NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
    out_others = alloc((K, 7, H, W), dtype=torch.float32, device=dev)
    radii = alloc((K, P), dtype=torch.int32, device=dev)
    chunks = [(_Chunk(dev), _Chunk(dev), _Chunk(dev)) for _ in range(K)]
    counts = (C.c_int * K)()
    rc = 0
    try:
        if P != 0:
            M = sh.size(1) if sh.size(0) != 0 else 0
            keep = [_f32c(t) for t in (background, means3D, sh, colors, opacity, scales, rotations, transMat_precomp)]
            bg_, m3_, sh_, col_, op_, sc_, rot_, tm_ = keep
            vm_ = _f32c(viewmatrices).reshape(K, 16)
            pm_ = _f32c(projmatrices).reshape(K, 16)
            cp_ = _f32c(camposs).reshape(K, 3)
            io = (_lib.FrameIO * K)()
            for k, (g, b, im) in enumerate(chunks):
                io[k].geometry_alloc, io[k].geometry_user = g.cb, g.user
                io[k].binning_alloc, io[k].binning_user = b.cb, b.user
                io[k].image_alloc, io[k].image_user = im.cb, im.user
                io[k].viewmatrix, io[k].projmatrix, io[k].cam_pos = vm_[k].data_ptr(), pm_[k].data_ptr(), cp_[k].data_ptr()
                io[k].out_color, io[k].out_others, io[k].radii = out_color[k].data_ptr(), out_others[k].data_ptr(), radii[k].data_ptr()
            with _on_device(dev):
                rc = L.gs2d_forward_batch(K, io, P, int(degree), M, _ptr(bg_), W, H, _ptr(m3_), _ptr(sh_), _ptr(col_), _ptr(op_),
                                          _ptr(sc_), float(scale_modifier), _ptr(rot_), _ptr(tm_), int(bool(use_sa)),
                                          int(bool(debug)), counts, _stream_ptr(dev))
    finally:
        for trio in chunks:
            for ch in trio:
                ch.release()
    if rc < 0:
        raise RuntimeError(_lib.last_error())
    return ([int(counts[k]) for k in range(K)], out_color, out_others, radii, [c[0].tensor for c in chunks],
            [c[1].tensor for c in chunks], [c[2].tensor for c in chunks])


def rasterize_gaussians_backward_batch(background, means3D, radii, colors, scales, rotations, scale_modifier, transMat_precomp,
                                       viewmatrices, projmatrices, tan_fovxs, tan_fovys, dL_dout_color, dL_dout_others, sh,
                                       degree, camposs, geomBuffers, Rs, binningBuffers, imageBuffers, use_sa, debug,
                                       grad_sink=None, lean=False, accumulate=False):
    """gs2d_backward_batch: the backward of rasterize_gaussians_batch.  Returns a list of K tuples, frame k's being exactly
    what rasterize_gaussians_backward returns for that frame (per-frame gradients).
    accumulate: frame 0's tensors additionally receive the SUM over all frames (added in frame order, in one kernel).
    grad_sink: as in rasterize_gaussians_backward, for frame 0's parameter gradients."""
    L = _lib.lib()
    dev = means3D.device
    P = means3D.size(0)
    # dL_dout_color / dL_dout_others: [K,3,H,W] / [K,7,H,W] tensors or sequences of K per-frame tensors
    K, H, W = len(dL_dout_color), dL_dout_color[0].size(1), dL_dout_color[0].size(2)
    M = sh.size(1) if sh.size(0) != 0 else 0
    z = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    outs = []
    for k in range(K):
        o = {"means3D": z(P, 3), "means2D": z(P, 3), "colors": z(P, NUM_CHANNELS), "normal": None if lean else z(P, 3),
             "opacities": z(P, 1), "transMat": None if (lean and transMat_precomp.numel() == 0) else z(P, 9), "sh": z(P, M, 3),
             "scales": z(P, 2), "rotations": z(P, 4)}
        if k == 0 and grad_sink:
            for name in ("means3D", "colors", "opacities", "scales", "rotations"):
                sk = grad_sink.get(name)
                if sk is None:
                    continue
                if sk.shape != o[name].shape or sk.dtype != torch.float32 or sk.device != dev or not sk.is_contiguous():
                    raise RuntimeError(f"grad_sink[{name!r}] must be a contiguous fp32 {tuple(o[name].shape)} tensor on {dev}")
                o[name] = sk.detach()
        outs.append(o)
    rc = 0
    if P != 0:
        keep = [_f32c(t) for t in (background, means3D, sh, colors, scales, rotations, transMat_precomp)]
        bg_, m3_, sh_, col_, sc_, rot_, tm_ = keep
        dc_ = [_f32c(dL_dout_color[k]) for k in range(K)]
        do_ = [_f32c(dL_dout_others[k]) for k in range(K)]
        vm_ = _f32c(viewmatrices).reshape(K, 16)
        pm_ = _f32c(projmatrices).reshape(K, 16)
        cp_ = _f32c(camposs).reshape(K, 3)
        radii_ = radii.contiguous()
        fr = (_lib.FrameGrad * K)()
        for k in range(K):
            f, o = fr[k], outs[k]
            f.viewmatrix, f.projmatrix, f.campos = vm_[k].data_ptr(), pm_[k].data_ptr(), cp_[k].data_ptr()
            f.tan_fovx, f.tan_fovy = float(tan_fovxs[k]), float(tan_fovys[k])
            f.radii = radii_[k].data_ptr()
            f.geom_buffer, f.binning_buffer, f.img_buffer = _ptr(geomBuffers[k]), _ptr(binningBuffers[k]), _ptr(imageBuffers[k])
            f.num_rendered = int(Rs[k])
            f.dL_dpix, f.dL_depths = dc_[k].data_ptr(), do_[k].data_ptr()
            f.dL_dmean2D, f.dL_dnormal, f.dL_dopacity = o["means2D"].data_ptr(), _ptr(o["normal"]), o["opacities"].data_ptr()
            f.dL_dcolor, f.dL_dmean3D, f.dL_dtransMat = o["colors"].data_ptr(), o["means3D"].data_ptr(), _ptr(o["transMat"])
            f.dL_dsh, f.dL_dscale, f.dL_drot = _ptr(o["sh"]), o["scales"].data_ptr(), o["rotations"].data_ptr()
        with _on_device(dev):
            rc = L.gs2d_backward_batch(K, fr, int(bool(accumulate)), P, int(degree), M, _ptr(bg_), W, H, _ptr(m3_), _ptr(sh_), _ptr(col_), _ptr(sc_),
                                       float(scale_modifier), _ptr(rot_), _ptr(tm_), int(bool(use_sa)), int(bool(debug)),
                                       _stream_ptr(dev))
    if rc < 0:
        raise RuntimeError(_lib.last_error())
    return [(o["means2D"], o["colors"], o["opacities"], o["means3D"], o["transMat"], o["sh"], o["scales"], o["rotations"])
            for o in outs]


def set_deterministic(on=True):
    """Opt-in deterministic backward (gs2d_set_deterministic): no float atomics, gradients bit-identical from run to run.
    Process-wide; keep it unchanged between a forward and its backward."""
    _lib.lib().gs2d_set_deterministic(int(bool(on)))


def is_deterministic():
    return bool(_lib.lib().gs2d_get_deterministic())


def set_reference_binning(on=True):
    """gs2d_set_reference_binning: one instance for every tile of the reference's 3-sigma square (num_rendered and the
    sorted lists bit-identical to the reference's) instead of only the tiles inside the splat's footprint bound (default;
    same outputs, about a fifth fewer instances).  Process-wide; applies to the next forward."""
    _lib.lib().gs2d_set_reference_binning(int(bool(on)))


def is_reference_binning():
    return bool(_lib.lib().gs2d_get_reference_binning())


def set_launch_ahead(on=True):
    """gs2d_set_launch_ahead (default on): the forward enqueues all its kernels before the host reads num_rendered (the
    stages behind the duplication read the count on the device); off = duplicate, host wait, the rest.  Process-wide."""
    _lib.lib().gs2d_set_launch_ahead(int(bool(on)))


def is_launch_ahead():
    return bool(_lib.lib().gs2d_get_launch_ahead())


def mark_visible(means3D, viewmatrix, projmatrix):
    """_C.mark_visible (rasterize_points.cu:241-260)."""
    L = _lib.lib()
    P = means3D.size(0)
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P != 0:
        m3_, vm_, pm_ = _f32c(means3D), _f32c(viewmatrix), _f32c(projmatrix)
        with _on_device(means3D.device):
            rc = L.gs2d_mark_visible(P, m3_.data_ptr(), vm_.data_ptr(), pm_.data_ptr(), present.data_ptr(),
                                     _stream_ptr(means3D.device))
        if rc < 0:
            raise RuntimeError(_lib.last_error())
    return present


class _Sink:  # process-wide on purpose: autograd runs CUDA backward nodes on its own device thread, not the caller's
    views = None
    chunk_rows = None
    on_chunk = None
    armed = False     # a grad_sink context is active
    consumed = False  # ... and one operator backward has already taken it


_SINK = _Sink()


@contextlib.contextmanager
def grad_sink(views, chunk_rows=None, on_chunk=None):
    """While active, the ONE operator backward that runs in this process writes its parameter gradients into `views`
    (see rasterize_gaussians_backward).  Used by ba_shard to produce gradients directly in the all-reduce bucket.
    A second operator backward inside the same context raises: its gradients would silently land in fresh tensors while
    the caller believes they are in `views` (e.g. a loss function that renders twice)."""
    if _SINK.armed:
        raise RuntimeError("grad_sink contexts cannot be nested")
    _SINK.views, _SINK.armed, _SINK.consumed = views, True, False
    _SINK.chunk_rows, _SINK.on_chunk = chunk_rows, on_chunk
    try:
        yield
    finally:
        _SINK.views, _SINK.armed, _SINK.consumed = None, False, False
        _SINK.chunk_rows, _SINK.on_chunk = None, None


def _take_sink():
    if not _SINK.armed:
        return None, None, None
    if _SINK.consumed:
        raise RuntimeError("a second rasterizer backward ran inside one grad_sink context: only one operator call per "
                           "direct-gradient backward is supported (render once per keyframe, or use direct_grads=False)")
    _SINK.consumed = True
    return _SINK.views, _SINK.chunk_rows, _SINK.on_chunk


def _cpu_copy(args):
    """Host copies of the tensor arguments (the reference's cpu_deep_copy_tuple, RAST/gaus_2dgs_rasterization/__init__.py:17-19)."""
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


class _RasterizeGaussians(torch.autograd.Function):
    """RAST/gaus_2dgs_rasterization/__init__.py:44-161."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        rs = raster_settings
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree,
                rs.campos, rs.use_sa, rs.prefiltered, rs.debug)
        if rs.debug:  # RAST/gaus_2dgs_rasterization/__init__.py:84-91: keep a host copy of the arguments, dump it if the call fails
            cpu_args = _cpu_copy(args)
            try:
                num_rendered, color, depth, radii, geomBuffer, binningBuffer, imgBuffer = rasterize_gaussians(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            num_rendered, color, depth, radii, geomBuffer, binningBuffer, imgBuffer = rasterize_gaussians(*args)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        # autograd would otherwise hand the backward a freshly zero-filled [P] int tensor as the "gradient" of radii on every
        # call (a fill kernel on the stream between the two blend kernels); gradients that really are absent become zeros below
        ctx.set_materialize_grads(False)
        ctx.image_shape = (color.shape, depth.shape)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
         imgBuffer) = ctx.saved_tensors
        if grad_out_color is None:  # only the other image fed the loss
            grad_out_color = torch.zeros(ctx.image_shape[0], dtype=torch.float32, device=means3D.device)
        if grad_depth is None:
            grad_depth = torch.zeros(ctx.image_shape[1], dtype=torch.float32, device=means3D.device)
        sink, chunk_rows, on_chunk = _take_sink()  # one backward per sink; a second one raises
        args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, grad_depth, sh, rs.sh_degree, rs.campos, geomBuffer,
                ctx.num_rendered, binningBuffer, imgBuffer, rs.use_sa, rs.debug)
        kw = dict(grad_sink=sink, lean=True, chunk_rows=chunk_rows, on_chunk=on_chunk)
        if rs.debug:  # RAST/gaus_2dgs_rasterization/__init__.py:135-142
            cpu_args = _cpu_copy(args)
            try:
                res = rasterize_gaussians_backward(*args, **kw)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            res = rasterize_gaussians_backward(*args, **kw)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = res
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, None)


_CAMERA_STACKS = []  # [(settings tuple, vms, pms, cps)]: the stacked per-frame matrices of recently used frame sets


def _stacked_cameras(settings_list):
    """[K,16] view / projection matrices and [K,3] camera positions of a frame set, validated and stacked ONCE per set of
    settings objects (identity-keyed, the last 8 sets are kept): a BA loop renders the same keyframes step after step, and
    neither the three stack kernels nor a device-synchronising background comparison belong on its critical path."""
    key = tuple(settings_list)
    # ... and by the VERSION of their camera tensors: a pose refined in place (rs.viewmatrix.copy_(...) during BA; the
    # NamedTuple's fields cannot be rebound but their tensors can be written) bumps torch's version counter and misses
    ver = tuple((rs.viewmatrix._version, rs.projmatrix._version, rs.campos._version, rs.viewmatrix.data_ptr(),
                 rs.projmatrix.data_ptr(), rs.campos.data_ptr()) for rs in key)
    for ent in _CAMERA_STACKS:
        if len(ent[0]) == len(key) and all(a is b for a, b in zip(ent[0], key)) and ent[4] == ver:
            return ent[1], ent[2], ent[3]
    _CAMERA_STACKS[:] = [ent for ent in _CAMERA_STACKS
                         if not (len(ent[0]) == len(key) and all(a is b for a, b in zip(ent[0], key)))]  # stale entry of this set
    rs0 = key[0]
    for rs in key[1:]:
        same = (rs.image_height == rs0.image_height and rs.image_width == rs0.image_width and rs.use_sa == rs0.use_sa
                and rs.sh_degree == rs0.sh_degree and rs.scale_modifier == rs0.scale_modifier and rs.debug == rs0.debug
                and (rs.bg is rs0.bg or torch.equal(rs.bg, rs0.bg)))
        if not same:
            raise RuntimeError("batched frames must share image size, background, sh_degree, scale_modifier, use_sa and debug")
    vms = torch.stack([rs.viewmatrix.reshape(16) for rs in key]).float().contiguous()
    pms = torch.stack([rs.projmatrix.reshape(16) for rs in key]).float().contiguous()
    cps = torch.stack([rs.campos.reshape(3) for rs in key]).float().contiguous()
    _CAMERA_STACKS.append((key, vms, pms, cps, ver))
    if len(_CAMERA_STACKS) > 8:
        _CAMERA_STACKS.pop(0)
    return vms, pms, cps


class _RasterizeGaussiansBatch(torch.autograd.Function):
    """_RasterizeGaussians over K cameras at once (same Gaussians, same image size): returns (radii [K,P], K colour images,
    K allmaps); the gradient of every input is the sum over the frames (added in frame order, as K separate calls accumulate)."""

    @staticmethod
    def forward(ctx, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings_list, *means2D):
        # means2D: ONE gradient carrier shared by all frames (its gradient is then the sum over the frames -- autograd's
        # semantics for a leaf used K times) or K carriers, one per frame (each receives ITS view's screen-space gradient:
        # what the reference's per-view densification statistics need, scene/Gaussians.py:58-62)
        if len(means2D) not in (1, len(settings_list)):
            raise RuntimeError("means2D: one tensor, or one per frame")
        ctx.n_m2 = len(means2D)
        rs0 = settings_list[0]
        vms, pms, cps = _stacked_cameras(settings_list)
        Rs, color, depth, radii, geoms, bins, imgs = rasterize_gaussians_batch(
            rs0.bg, means3D, colors_precomp, opacities, scales, rotations, rs0.scale_modifier, cov3Ds_precomp, vms, pms,
            rs0.image_height, rs0.image_width, sh, rs0.sh_degree, cps, rs0.use_sa, rs0.debug)
        ctx.settings_list = settings_list
        ctx.Rs = Rs
        K = ctx.K = len(settings_list)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, vms, pms, cps,
                              *geoms, *bins, *imgs)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        ctx.image_shape = (color.shape[1:], depth.shape[1:])
        # one output per frame (slices of the stacked buffers made here, so autograd sees K independent tensors and hands the
        # backward K independent gradients -- a stacked output indexed by the caller would route every frame's gradient through
        # a zero-filled [K, ...] tensor and an add)
        return (radii,) + tuple(color[k] for k in range(K)) + tuple(depth[k] for k in range(K))

    @staticmethod
    def backward(ctx, grad_radii, *grads):
        K = ctx.K
        rs0 = ctx.settings_list[0]
        saved = ctx.saved_tensors
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, vms, pms, cps = saved[:10]
        geoms, bins, imgs = saved[10:10 + K], saved[10 + K:10 + 2 * K], saved[10 + 2 * K:10 + 3 * K]
        zeros = lambda shape: torch.zeros(shape, dtype=torch.float32, device=means3D.device)
        grad_out_color = [g if g is not None else zeros(ctx.image_shape[0]) for g in grads[:K]]
        grad_depth = [g if g is not None else zeros(ctx.image_shape[1]) for g in grads[K:]]
        sink, chunk_rows, on_chunk = _take_sink()
        if on_chunk is not None:
            raise RuntimeError("chunked (overlapped) reduction is a one-keyframe-per-rank feature; the batched backward has none")
        per = rasterize_gaussians_backward_batch(
            rs0.bg, means3D, radii, colors_precomp, scales, rotations, rs0.scale_modifier, cov3Ds_precomp, vms, pms,
            [rs.tanfovx for rs in ctx.settings_list], [rs.tanfovy for rs in ctx.settings_list], grad_out_color, grad_depth, sh,
            rs0.sh_degree, cps, geoms, ctx.Rs, bins, imgs, rs0.use_sa, rs0.debug, grad_sink=sink, lean=True, accumulate=True)
        total = per[0]  # frame order: the same sums K separate backwards accumulate (one kernel, gs2d_backward_batch(accumulate))
        (_g_m2, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = total
        # the screen-space gradients stay per frame in the library (never summed by gs2d_backward_batch)
        if ctx.n_m2 == K:
            g_m2 = tuple(per[k][0] for k in range(K))
        else:
            acc = per[0][0].clone()
            for k in range(1, K):
                acc += per[k][0]
            g_m2 = (acc,)
        return (grad_means3D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, None) + g_m2


def rasterize_gaussians_apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                              raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    use_sa: bool
    prefiltered: bool
    debug: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = lambda: torch.empty(0, dtype=torch.float32, device=means3D.device)
        if shs is None:
            shs = empty()
        if colors_precomp is None:
            colors_precomp = empty()
        if scales is None:
            scales = empty()
        if rotations is None:
            rotations = empty()
        if cov3D_precomp is None:
            cov3D_precomp = empty()
        return rasterize_gaussians_apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                         cov3D_precomp, rs)


class GaussianRasterizerBatch(nn.Module):
    """GaussianRasterizer for K cameras of the same image size in one call (no counterpart in the reference; for BA ranks
    that hold several keyframes, gaus_slam_amd/ba_shard.py).  forward(...) takes the arguments of
    GaussianRasterizer.forward and returns (colors: K tensors [3,H,W], radii [K,P], allmaps: K tensors [7,H,W]).
    means2D: one gradient carrier (receives the SUM of the K views' screen-space gradients) or a list of K carriers (each
    receives its own view's: what add_densification_stats needs per view, scene/Gaussians.py:58-62)."""

    def __init__(self, settings_list):
        super().__init__()
        self.settings_list = list(settings_list)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = lambda: torch.empty(0, dtype=torch.float32, device=means3D.device)
        shs = empty() if shs is None else shs
        colors_precomp = empty() if colors_precomp is None else colors_precomp
        scales = empty() if scales is None else scales
        rotations = empty() if rotations is None else rotations
        cov3D_precomp = empty() if cov3D_precomp is None else cov3D_precomp
        K = len(self.settings_list)
        m2 = tuple(means2D) if isinstance(means2D, (list, tuple)) else (means2D,)
        out = _RasterizeGaussiansBatch.apply(means3D, shs, colors_precomp, opacities, scales, rotations,
                                             cov3D_precomp, self.settings_list, *m2)
        return out[1:1 + K], out[0], out[1 + K:]
