"""Fused post-op + loss for the SLAM iterations.

Equivalent to the reference's default-configuration pipeline between the rasterizer and `loss.backward()`:
render/__init__.py:46-49 (weight-normalised depth, near/far outliers zeroed) followed by slam/Loss.py:22-58
(nan_to_num, depth / silhouette masks, masked L1 sums for tracking, masked means + 0.1*dist-style term for mapping).
One autograd node, two HIP kernels (csrc/gs2d_loss.hip).  Settings outside the default configuration
(use_normal_loss, ignore_outliners, enable_exposure) are not covered -- use the PyTorch formulation for those."""
import torch

from . import _lib
from .rasterizer import _on_device, _stream_ptr

_WS_DOUBLES = 2560  # GS2D_LOSS_WS_DOUBLES (include/gs2d_rasterizer.h)


def _call(cfg, W, H, color_, allmap_, gtc, gtd, ws, out, g_color, g_allmap, upstream, dev):
    p = lambda t: None if t is None else t.data_ptr()
    with _on_device(dev):
        rc = _lib.lib().gs2d_slam_loss(
            int(cfg["mode"]), W, H, color_.data_ptr(), allmap_.data_ptr(), gtc.data_ptr(), gtd.data_ptr(),
            float(cfg["w_color"]), float(cfg["w_depth"]), float(cfg.get("w_dist", 0.0)), float(cfg.get("silmask_th", 0.9)),
            float(cfg.get("edge_thres", 0.4)), int(bool(cfg.get("use_edge_growth", False))),
            int(bool(cfg.get("use_weight_norm", True))), float(cfg.get("eps", 1e-6)), float(cfg.get("depth_near", 1e-2)),
            float(cfg.get("depth_far", 1e2)), ws.data_ptr(), p(out), p(g_color), p(g_allmap), p(upstream),
            _stream_ptr(dev))
    if rc < 0:
        raise RuntimeError("gs2d_slam_loss failed")


class _SlamLoss(torch.autograd.Function):
    """forward = the reduction pass only (loss value); backward = the gradient pass, scaled in-kernel by dL/dloss.  The
    partial sums of the forward stay in `ws` for the backward."""

    @staticmethod
    def forward(ctx, color, allmap, gt_color, gt_depth, cfg):
        if not color.is_cuda:
            raise RuntimeError("color must be a CUDA tensor")
        dev = color.device
        H, W = color.shape[1], color.shape[2]
        color_, allmap_ = color.detach().float().contiguous(), allmap.detach().float().contiguous()
        gtc = gt_color.detach().float().contiguous().reshape(H, W, 3)
        gtd = gt_depth.detach().float().contiguous().reshape(H, W)
        ws = torch.empty(_WS_DOUBLES, dtype=torch.float64, device=dev)
        out = torch.empty(8, dtype=torch.float32, device=dev)
        _call(cfg, W, H, color_, allmap_, gtc, gtd, ws, out, None, None, None, dev)
        ctx.save_for_backward(color_, allmap_, gtc, gtd, ws)
        ctx.cfg = cfg
        ctx.terms = out
        return out[0]

    @staticmethod
    def backward(ctx, grad_loss):
        color_, allmap_, gtc, gtd, ws = ctx.saved_tensors
        dev = color_.device
        H, W = color_.shape[1], color_.shape[2]
        g_color, g_allmap = torch.empty_like(color_), torch.empty_like(allmap_)
        up = grad_loss.detach().to(dtype=torch.float32).contiguous()
        _call(ctx.cfg, W, H, color_, allmap_, gtc, gtd, ws, None, g_color, g_allmap, up, dev)
        return g_color, g_allmap, None, None, None


def _loss_and_grads(color, allmap, gt_color, gt_depth, cfg):
    if not color.is_cuda:
        raise RuntimeError("color must be a CUDA tensor")
    dev = color.device
    H, W = color.shape[1], color.shape[2]
    color_, allmap_ = color.detach().float().contiguous(), allmap.detach().float().contiguous()
    gtc = gt_color.detach().float().contiguous().reshape(H, W, 3)
    gtd = gt_depth.detach().float().contiguous().reshape(H, W)
    ws = torch.empty(_WS_DOUBLES, dtype=torch.float64, device=dev)
    out = torch.empty(8, dtype=torch.float32, device=dev)
    g_color, g_allmap = torch.empty_like(color_), torch.empty_like(allmap_)
    _call(cfg, W, H, color_, allmap_, gtc, gtd, ws, out, g_color, g_allmap, None, dev)
    return out[0], g_color, g_allmap


def tracking_loss_and_grads(color, allmap, gt_color, gt_depth, w_color, w_depth, silmask_th=0.9, use_weight_norm=True, eps=1e-6,
                            depth_near=1e-2, depth_far=1e2):
    """tracking_loss and its gradients w.r.t. the rasterizer outputs in ONE call -- two kernels (reduce; gradients + loss)
    instead of the three of the autograd node (reduce, loss, and the gradient pass again in backward), and no second autograd
    node.  For iteration loops that seed the rasterizer's backward themselves:
        loss, g_color, g_allmap = tracking_loss_and_grads(pkg["render_color"], pkg["allmap"], ...)
        torch.autograd.backward([pkg["render_color"], pkg["allmap"]], [g_color, g_allmap])
    Returns (loss, dL_dcolor [3, H, W], dL_dallmap [7, H, W]); the values equal tracking_loss(...) and its .backward()."""
    return _loss_and_grads(color, allmap, gt_color, gt_depth, dict(
        mode=0, w_color=w_color, w_depth=w_depth, silmask_th=silmask_th, use_weight_norm=use_weight_norm, eps=eps,
        depth_near=depth_near, depth_far=depth_far))


def mapping_loss_and_grads(color, allmap, gt_color, gt_depth, w_color, w_depth, w_dist, use_edge_growth=False, edge_thres=0.4,
                           use_weight_norm=True, eps=1e-6, depth_near=1e-2, depth_far=1e2):
    """mapping_loss and its gradients in one call (see tracking_loss_and_grads)."""
    return _loss_and_grads(color, allmap, gt_color, gt_depth, dict(
        mode=1, w_color=w_color, w_depth=w_depth, w_dist=w_dist, use_edge_growth=use_edge_growth, edge_thres=edge_thres,
        use_weight_norm=use_weight_norm, eps=eps, depth_near=depth_near, depth_far=depth_far))


def tracking_loss(color, allmap, gt_color, gt_depth, w_color, w_depth, silmask_th=0.9, use_weight_norm=True, eps=1e-6,
                  depth_near=1e-2, depth_far=1e2):
    """slam/Loss.py:35-49 on top of render/__init__.py:46-49: masked (depth-valid & alpha > silmask_th) L1 SUMS."""
    return _SlamLoss.apply(color, allmap, gt_color, gt_depth, dict(
        mode=0, w_color=w_color, w_depth=w_depth, silmask_th=silmask_th, use_weight_norm=use_weight_norm, eps=eps,
        depth_near=depth_near, depth_far=depth_far))


def mapping_loss(color, allmap, gt_color, gt_depth, w_color, w_depth, w_dist, use_edge_growth=False, edge_thres=0.4,
                 use_weight_norm=True, eps=1e-6, depth_near=1e-2, depth_far=1e2):
    """slam/Loss.py:51-58: masked L1 MEANS for colour / depth plus the mean of render_dist over the colour mask."""
    return _SlamLoss.apply(color, allmap, gt_color, gt_depth, dict(
        mode=1, w_color=w_color, w_depth=w_depth, w_dist=w_dist, use_edge_growth=use_edge_growth, edge_thres=edge_thres,
        use_weight_norm=use_weight_norm, eps=eps, depth_near=depth_near, depth_far=depth_far))
