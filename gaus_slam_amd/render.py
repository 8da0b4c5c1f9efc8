"""Thin render adapter equal to the reference's render/render_2dgs.py:33-65 (slices the 7-channel allmap) plus the
weight-norm / outlier step of render/__init__.py:46-49.  Provided so benches and tests can drive the operator the
way the reference callers do, without importing the reference."""
import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, GaussianRasterizerBatch


def settings_from_camera(cam, device, bg=None, use_sa=True, sh_degree=0, debug=False):
    """cam: gaus_slam_amd.scene_synth.Camera (same quantities as render/render_2dgs.py:6-31)."""
    bg = torch.zeros(3, dtype=torch.float32, device=device) if bg is None else bg.to(device).float()
    return GaussianRasterizationSettings(
        image_height=cam.H, image_width=cam.W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg, scale_modifier=1.0,
        viewmatrix=cam.viewmatrix.to(device).unsqueeze(0), projmatrix=cam.projmatrix.to(device).unsqueeze(0),
        sh_degree=sh_degree, campos=cam.campos.to(device), use_sa=use_sa, prefiltered=False, debug=debug)


def render(settings, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
           cov3D_precomp=None, use_weight_norm=False, eps=1e-6, depth_near=1e-2, depth_far=1e2):
    color_map, radius, allmap = GaussianRasterizer(settings)(
        means3D, means2D, opacities=opacities, shs=shs, colors_precomp=colors_precomp, scales=scales,
        rotations=rotations, cov3D_precomp=cov3D_precomp)
    pkg = {
        "render_color": color_map, "radius": radius, "means2D": means2D, "allmap": allmap,
        "render_depth": allmap[0:1], "render_alpha": allmap[1:2], "render_normal": allmap[2:5],
        "render_middepth": allmap[5:6], "render_dist": allmap[6:7],
    }
    if use_weight_norm:  # render/__init__.py:46-49
        d = pkg["render_depth"] / (pkg["render_alpha"] + eps)
        outlier = torch.logical_or(d > depth_far, d < depth_near)
        pkg["render_depth"] = torch.where(outlier, torch.zeros_like(d), d)
    return pkg


def render_batch(settings_list, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                 cov3D_precomp=None):
    """render() for K cameras of the same image size in one operator call (GaussianRasterizerBatch): a list of K packages
    with the keys of render().
    means2D: a LIST of K gradient carriers (one per view; package k holds means2D[k], whose .grad is that view's screen-space
    gradient -- feed it to the reference's add_densification_stats(render_pkg) exactly like the package of a single render(),
    slam/Backend.py:117-118, scene/Gaussians.py:58-62) or ONE tensor shared by all views: its .grad is then the SUM of the K
    views' gradients (autograd's semantics for a leaf used K times) and must NOT be fed to per-view densification statistics."""
    color, radius, allmap = GaussianRasterizerBatch(settings_list)(
        means3D, means2D, opacities=opacities, shs=shs, colors_precomp=colors_precomp, scales=scales, rotations=rotations,
        cov3D_precomp=cov3D_precomp)
    per_view = isinstance(means2D, (list, tuple))
    return [{"render_color": color[k], "radius": radius[k], "means2D": means2D[k] if per_view else means2D, "allmap": allmap[k],
             "render_depth": allmap[k][0:1], "render_alpha": allmap[k][1:2], "render_normal": allmap[k][2:5],
             "render_middepth": allmap[k][5:6], "render_dist": allmap[k][6:7]} for k in range(len(settings_list))]
