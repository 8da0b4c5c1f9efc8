"""Fused post-op + loss (gs2d_slam_loss, gaus_slam_amd/loss.py) against the plain-PyTorch restatement
of the reference formulas (oracle/loss_ref.py): loss value and gradients w.r.t. the rasterizer outputs."""
import numpy as np
import pytest
import torch

from tests import util


def _inputs(W, H, seed):
    g = torch.Generator().manual_seed(seed)
    color = torch.rand(3, H, W, generator=g)
    allmap = torch.zeros(7, H, W)
    alpha = torch.rand(H, W, generator=g)
    alpha[torch.rand(H, W, generator=g) < 0.2] = 0.0
    depth = (0.5 + 5 * torch.rand(H, W, generator=g)) * alpha
    allmap[0], allmap[1], allmap[6] = depth, alpha, 0.01 * torch.rand(H, W, generator=g)
    allmap[0, 0, :5] = float("nan"); allmap[0, 1, :5] = float("inf"); allmap[1, 2, :5] = float("nan")
    allmap[0, 3, :5] = 500.0  # beyond depth_far after normalisation
    color[0, 4, :5] = float("nan"); allmap[6, 5, :5] = float("inf")
    gt_color = torch.rand(H, W, 3, generator=g)
    gt_depth = 0.5 + 5 * torch.rand(H, W, 1, generator=g)
    gt_depth[torch.rand(H, W, 1, generator=g) < 0.1] = 0.0
    return color, allmap, gt_color, gt_depth


def test_loss_oracle_matches_hand_computation():
    from oracle import loss_ref
    color = torch.full((3, 2, 2), 0.5)
    allmap = torch.zeros(7, 2, 2)
    allmap[0] = torch.tensor([[1.9, 0.0], [0.95, 3.0]])
    allmap[1] = torch.tensor([[0.95, 0.0], [0.5, 1.0]])
    gt_color = torch.zeros(2, 2, 3)
    gt_depth = torch.tensor([[2.5, 1.0], [2.0, 0.0]]).reshape(2, 2, 1)
    # tracking: only pixel (0,0) passes (depth valid both ways, alpha > 0.9): |0.5-0|*3 + |1.9/0.950001 - 2.5|
    loss = loss_ref.post_and_loss(color, allmap, gt_color, gt_depth, 0, 0.5, 1.0)
    assert float(loss) == pytest.approx(0.5 * 1.5 + abs(1.9 / (0.95 + 1e-6) - 2.5), rel=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,edge", [(0, False), (1, False), (1, True)])
@pytest.mark.parametrize("weight_norm", [True, False])
def test_fused_loss_matches_reference_formulation(mode, edge, weight_norm):
    from gaus_slam_amd import loss as gl
    from oracle import loss_ref
    W, H = 200, 136
    color, allmap, gt_color, gt_depth = _inputs(W, H, seed=mode * 2 + edge)
    c = color.double().clone().requires_grad_(True)
    a = allmap.double().clone().requires_grad_(True)
    kw = dict(w_color=0.5, w_depth=1.0, use_weight_norm=weight_norm)
    ref = loss_ref.post_and_loss(c, a, gt_color.double(), gt_depth.double(), mode, w_dist=0.1, use_edge_growth=edge, **kw)
    ref.backward()
    dev = torch.device("cuda")
    cg = color.to(dev).requires_grad_(True)
    ag = allmap.to(dev).requires_grad_(True)
    if mode == 0:
        out = gl.tracking_loss(cg, ag, gt_color.to(dev), gt_depth.to(dev), **kw)
    else:
        out = gl.mapping_loss(cg, ag, gt_color.to(dev), gt_depth.to(dev), w_dist=0.1, use_edge_growth=edge, **kw)
    (2.0 * out).backward()
    assert float(out.detach()) == pytest.approx(float(ref.detach()), rel=2e-6)
    # torch autograd turns 0 * inf / 0 * nan into NaN gradients at the poisoned pixels; the fused kernel writes 0 there
    for mine, ref_g in ((cg.grad.cpu(), 2.0 * c.grad), (ag.grad.cpu(), 2.0 * a.grad)):
        ok = torch.isfinite(ref_g)
        assert ok.float().mean() > 0.99
        assert util.grad_err(mine[ok].numpy(), ref_g[ok].numpy()) < 1e-5
    assert torch.isfinite(cg.grad).all() and torch.isfinite(ag.grad).all()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("size", [(200, 136), (640, 480)])
def test_one_call_loss_and_grads_equal_the_autograd_node(mode, size):
    """tracking_/mapping_loss_and_grads (value + gradients in one library call: two kernels) against the autograd node (three):
    same value, same gradients, bit for bit -- both run the same two passes over the same partial sums.  640x480 has more
    pixels than reduce workgroups x 256, so the strided pixel loop and the 512-partial fold are exercised."""
    from gaus_slam_amd import loss as gl
    W, H = size
    color, allmap, gt_color, gt_depth = _inputs(W, H, seed=7 + mode)
    dev = torch.device("cuda")
    cg = color.to(dev).requires_grad_(True)
    ag = allmap.to(dev).requires_grad_(True)
    gc, gd = gt_color.to(dev), gt_depth.to(dev)
    if mode == 0:
        out = gl.tracking_loss(cg, ag, gc, gd, 0.5, 1.0)
        loss, g_c, g_a = gl.tracking_loss_and_grads(cg, ag, gc, gd, 0.5, 1.0)
    else:
        out = gl.mapping_loss(cg, ag, gc, gd, 0.5, 1.0, 0.1)
        loss, g_c, g_a = gl.mapping_loss_and_grads(cg, ag, gc, gd, 0.5, 1.0, 0.1)
    out.backward()
    assert torch.equal(loss, out.detach())
    assert torch.equal(g_c, cg.grad) and torch.equal(g_a, ag.grad)
    assert not g_c.requires_grad and not g_a.requires_grad
