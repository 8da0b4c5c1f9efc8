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
                                       dL_dout_color, dL_dout_others, pose_Rt, pose_quat)]
            bg_, m3_, col_, sc_, rot_, vm_, pm_, cp_, dc_, do_, prt_, pq_ = keep
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


class _RasterizeGaussians(torch.autograd.Function):
    """RAST/gaus_2dgs_rasterization/__init__.py:44-161."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        rs = raster_settings
        num_rendered, color, depth, radii, geomBuffer, binningBuffer, imgBuffer = rasterize_gaussians(
            rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree,
            rs.campos, rs.use_sa, rs.prefiltered, rs.debug)
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
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = rasterize_gaussians_backward(
            rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
            rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, grad_depth, sh, rs.sh_degree, rs.campos, geomBuffer,
            ctx.num_rendered, binningBuffer, imgBuffer, rs.use_sa, rs.debug, grad_sink=sink, lean=True,
            chunk_rows=chunk_rows, on_chunk=on_chunk)
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, None)


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
