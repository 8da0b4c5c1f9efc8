"""ORACLE (test infrastructure): plain-PyTorch restatement of the reference's default post-op + loss,
render/__init__.py:46-49 followed by slam/Loss.py:22-58 (use_normal_loss = ignore_outliners = enable_exposure = False).
Autograd of this gives the expected gradients w.r.t. the rasterizer outputs."""
import torch


def post_and_loss(color, allmap, gt_color, gt_depth, mode, w_color, w_depth, w_dist=0.0, silmask_th=0.9, edge_thres=0.4,
                  use_edge_growth=False, use_weight_norm=True, eps=1e-6, depth_near=1e-2, depth_far=1e2):
    """color [3,H,W], allmap [7,H,W], gt_color [H,W,3], gt_depth [H,W,1] -> scalar loss."""
    render_depth, render_alpha, render_dist = allmap[0:1], allmap[1:2], allmap[6:7]
    if use_weight_norm:  # render/__init__.py:46-49
        render_depth = render_depth / (render_alpha + eps)
        outlier = torch.logical_or(render_depth > depth_far, render_depth < depth_near)
        render_depth = torch.where(outlier, torch.zeros_like(render_depth), render_depth)
    a = torch.nan_to_num(render_alpha, 0, 0).permute(1, 2, 0)
    d = torch.nan_to_num(render_depth, 0, 0).permute(1, 2, 0)
    c = torch.nan_to_num(color, 0, 0).permute(1, 2, 0)
    dist = torch.nan_to_num(render_dist, 0, 0).permute(1, 2, 0)
    depth_mask = (gt_depth > 1e-5).view(-1) & (d > 1e-5).view(-1)
    if mode == 0:  # Loss.py:35-49
        m = depth_mask & (a > silmask_th).view(-1)
        lc = (c - gt_color).abs().view(-1, 3)[m].sum()
        ld = (d - gt_depth).abs().view(-1, 1)[m].sum()
        return w_color * lc + w_depth * ld
    cm = (a > edge_thres).reshape(-1) if use_edge_growth else depth_mask  # Loss.py:51-58
    lc = (c - gt_color).abs().view(-1, 3)[cm].mean()
    ld = (d - gt_depth).abs().view(-1, 1)[depth_mask].mean()
    ldist = dist.view(-1, 1)[cm].mean()
    return w_color * lc + w_depth * ld + w_dist * ldist
