"""Fused tracking-regime render.

The reference's tracking renderer (render/__init__.py:17-50) transforms the Gaussians into the camera frame in
PyTorch -- means3D_cam = R x + t with autograd on w2c, rotations = quaternion_multiply(matrix_to_quaternion(R), q)
detached -- and renders with an identity view; the pose gradient is autograd of that transform fed by the
rasterizer's dL/dmeans3D.  Here the transform runs inside the preprocess kernel and the backward reduces
dL/dR = sum_i g_i (x) x_i, dL/dt = sum_i g_i on the GPU (gs2d_forward_posed / gs2d_backward_posed), which removes the
P-sized PyTorch passes from the 40-150-iteration tracking loop.  Gradient semantics follow the reference: the pose
receives the position term only (rotations are detached, render/__init__.py:36).
"""
import torch

from . import rasterizer as _r


def matrix_to_quaternion(R: torch.Tensor) -> torch.Tensor:
    """Rotation matrix [3,3] -> unit quaternion (w,x,y,z) with w >= 0.  Restates the published algorithm of
    pytorch3d.transforms.matrix_to_quaternion (the reference installs pytorch3d@stable, README.md:75): four
    candidate quaternions from the diagonal, pick the best-conditioned one."""
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = R.reshape(9).unbind(0)
    q_abs = torch.sqrt(torch.clamp(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22,
                                                1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22]), min=0.0))
    cand = torch.stack([
        torch.stack([q_abs[0] ** 2, m21 - m12, m02 - m20, m10 - m01]),
        torch.stack([m21 - m12, q_abs[1] ** 2, m10 + m01, m02 + m20]),
        torch.stack([m02 - m20, m10 + m01, q_abs[2] ** 2, m12 + m21]),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[3] ** 2]),
    ])
    cand = cand / (2.0 * q_abs[:, None].clamp(min=0.1))
    q = cand[torch.argmax(q_abs)]  # tensor index: no host synchronisation
    return torch.where(q[0:1] < 0, -q, q)


def pose_quaternion(pose_Rt: torch.Tensor) -> torch.Tensor:
    """matrix_to_quaternion of the rotation block of a contiguous float32 [3,4] device tensor, computed by one tiny kernel
    (gs2d_pose_quat): no host sync and a single launch instead of ~20 PyTorch ops per tracking iteration."""
    from . import _lib
    q = torch.empty(4, dtype=torch.float32, device=pose_Rt.device)
    with _r._on_device(pose_Rt.device):
        rc = _lib.lib().gs2d_pose_quat(pose_Rt.data_ptr(), q.data_ptr(),
                                       _r._stream_ptr(pose_Rt.device))
    if rc < 0:
        raise RuntimeError(_lib.last_error())
    return q


class _RasterizeTracking(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w2c, means3D, colors_precomp, opacities, scales, rotations, raster_settings):
        rs = raster_settings
        pose_Rt = w2c[:3, :4].detach().float().contiguous()
        # q_cam is derived from pose_Rt inside the preprocess kernels (pose_quat=None): the code of gs2d_pose_quat /
        # pose_quaternion(), without its launch
        pose_q = None
        e = torch.empty(0, dtype=torch.float32, device=means3D.device)
        num_rendered, color, allmap, radii, geom, binning, img = _r.rasterize_gaussians(
            rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, e, rs.viewmatrix,
            rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, e, rs.sh_degree, rs.campos,
            rs.use_sa, rs.prefiltered, rs.debug, pose_Rt=pose_Rt, pose_quat=pose_q)
        ctx.rs = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, radii, geom, binning, img, pose_Rt)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)  # no zero-filled [P] "gradient" of radii per call (rasterizer._RasterizeGaussians)
        ctx.image_shape = (color.shape, allmap.shape)
        return color, radii, allmap

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_allmap):
        rs = ctx.rs
        colors_precomp, means3D, scales, rotations, radii, geom, binning, img, pose_Rt = ctx.saved_tensors
        pose_q = None
        e = torch.empty(0, dtype=torch.float32, device=means3D.device)
        if grad_color is None:
            grad_color = torch.zeros(ctx.image_shape[0], dtype=torch.float32, device=means3D.device)
        if grad_allmap is None:
            grad_allmap = torch.zeros(ctx.image_shape[1], dtype=torch.float32, device=means3D.device)
        if not any(ctx.needs_input_grad[1:6]):
            # the reference's tracking renderer detaches every Gaussian parameter (render/__init__.py:31-36): only the
            # pose gradient is wanted; it is accumulated straight into the first three rows of the [4,4] result, whose
            # fourth row the same call clears (GS2D_BWD_POSE_4X4): no fill kernel
            g_w2c = torch.empty((4, 4), dtype=torch.float32, device=means3D.device)
            _r.rasterize_gaussians_backward(
                rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, e, rs.viewmatrix, rs.projmatrix,
                rs.tanfovx, rs.tanfovy, grad_color, grad_allmap, e, rs.sh_degree, rs.campos, geom, ctx.num_rendered, binning,
                img, rs.use_sa, rs.debug, pose_Rt=pose_Rt, pose_quat=pose_q, pose_only_out=g_w2c)
            return g_w2c, None, None, None, None, None, None
        (g_means2D, g_colors, g_opac, g_means3D, g_T, g_sh, g_scales, g_rot, g_pose) = _r.rasterize_gaussians_backward(
            rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, e, rs.viewmatrix, rs.projmatrix,
            rs.tanfovx, rs.tanfovy, grad_color, grad_allmap, e, rs.sh_degree, rs.campos, geom, ctx.num_rendered, binning,
            img, rs.use_sa, rs.debug, pose_Rt=pose_Rt, pose_quat=pose_q, lean=True)
        g_w2c = torch.zeros((4, 4), dtype=torch.float32, device=means3D.device)
        g_w2c[:3, :4] = g_pose
        # rotations are detached in the reference's tracking / BA renderers (render/__init__.py:36,98)
        return g_w2c, g_means3D, g_colors, g_opac, g_scales, None, None


def render_tracking(raster_settings, w2c, means3D, opacities, colors_precomp, scales, rotations):
    """Drop-in for the transform + render part of Renderer_tracking / Renderer_BA (render/__init__.py:31-40,96-102):
    `raster_settings` is the identity-view camera (setup_camera(w, h, K, eye(4))).  Returns the same render_pkg keys
    as render/render_2dgs.py:56-65 (means2D is omitted: the fused path does not need the grad-sink tensor)."""
    color, radii, allmap = _RasterizeTracking.apply(w2c, means3D, colors_precomp, opacities, scales, rotations,
                                                    raster_settings)
    return {"render_color": color, "radius": radii, "allmap": allmap, "render_depth": allmap[0:1],
            "render_alpha": allmap[1:2], "render_normal": allmap[2:5], "render_middepth": allmap[5:6],
            "render_dist": allmap[6:7]}
