"""The two SLAM inner loops the operator exists for, run end to end on synthetic data: they have to CONVERGE.

Tracking (slam/Frontend.py / Backend.py `tracking`: 40-150 iterations per frame): the map is fixed, the camera pose is
optimised against the observed colour + depth with the masked L1 sums of slam/Loss.py:35-49.  Mapping (`mapping`): the pose
is fixed, Gaussian parameters are optimised against the observation.  Neither loop compares numbers with the oracle (the
gradient parity tests do); what is checked is that the gradients, fused losses and fused optimiser steps work TOGETHER the
way the SLAM system uses them: the pose error / the loss must drop by a large factor."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _pose_error(a, b):
    d = a.double() @ torch.inverse(b.double())
    ang = torch.rad2deg(torch.arccos(torch.clamp((torch.trace(d[:3, :3]) - 1) / 2, -1, 1)))
    return float(ang), float(d[:3, 3].norm())


def test_tracking_loop_recovers_a_perturbed_pose():
    from gaus_slam_amd import loss as gl, render as gs_render, tracking
    from gaus_slam_amd.scene_synth import random_w2c
    P, W, H = 60000, 320, 240
    dev = torch.device("cuda")
    sc = util.make_scene(P, W, H, seed=2, regime="tracking")  # camera-space scene: the true w2c is the identity
    settings = gs_render.settings_from_camera(sc["cam"], dev, use_sa=True)
    p = {k: sc[k].to(dev) for k in ("means3D", "opacities", "scales", "rotations", "colors")}
    true_w2c = torch.eye(4, device=dev)
    with torch.no_grad():
        obs = tracking.render_tracking(settings, true_w2c, p["means3D"], p["opacities"], p["colors"], p["scales"], p["rotations"])
        gt_color = obs["render_color"].permute(1, 2, 0).contiguous()
        gt_depth = (obs["allmap"][0] / (obs["allmap"][1] + 1e-6)).unsqueeze(-1).contiguous()  # render/__init__.py:46
    # pose parameters as the reference keeps them (scene/Frame.py:84-92): unit quaternion (w, x, y, z) + translation
    start = random_w2c(np.random.default_rng(5), max_rot_deg=1.5, max_trans=0.03)
    cam_rot = tracking.matrix_to_quaternion(start[:3, :3]).to(dev).requires_grad_(True)
    cam_tran = start[:3, 3].clone().to(dev).requires_grad_(True)

    def build_w2c():
        w, x, y, z = torch.nn.functional.normalize(cam_rot, dim=0)
        R = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)]),
                         torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)]),
                         torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)])])
        top = torch.cat([R, cam_tran[:, None]], 1)
        return torch.cat([top, torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=dev)], 0)
    ang0, tr0 = _pose_error(build_w2c().detach().cpu(), true_w2c.cpu())
    opt = torch.optim.Adam([{"params": [cam_rot], "lr": 4e-4}, {"params": [cam_tran], "lr": 2e-3}])  # configs: cam_rot / cam_trans lrs
    losses = []
    for _ in range(150):
        opt.zero_grad(set_to_none=True)
        pkg = tracking.render_tracking(settings, build_w2c(), p["means3D"], p["opacities"], p["colors"], p["scales"], p["rotations"])
        loss = gl.tracking_loss(pkg["render_color"], pkg["allmap"], gt_color, gt_depth, 0.5, 1.0)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    ang1, tr1 = _pose_error(build_w2c().detach().cpu(), true_w2c.cpu())
    print(f"tracking: rotation error {ang0:.3f} -> {ang1:.3f} deg, translation error {tr0:.4f} -> {tr1:.4f} m, "
          f"loss {losses[0]:.1f} -> {losses[-1]:.1f}")
    assert losses[-1] < 0.25 * losses[0]
    assert tr1 < 0.4 * tr0 and ang1 < 0.5 * ang0


def test_mapping_loop_fits_the_observation():
    from gaus_slam_amd import ba_shard, loss as gl, optim as gs_optim, render as gs_render
    P, W, H = 60000, 320, 240
    dev = torch.device("cuda")
    sc = util.make_scene(P, W, H, seed=3, regime="mapping")
    settings = gs_render.settings_from_camera(sc["cam"], dev, use_sa=True)
    names = ("means3D", "opacities", "scales", "rotations", "colors")
    truth = {k: sc[k].to(dev) for k in names}

    def rasterize(q):
        m2 = torch.zeros_like(q["means3D"], requires_grad=True)
        return gs_render.render(settings, q["means3D"], m2, q["opacities"], colors_precomp=q["colors"], scales=q["scales"],
                                rotations=q["rotations"])
    with torch.no_grad():
        obs = rasterize(truth)
        gt_color = obs["render_color"].permute(1, 2, 0).contiguous()
        gt_depth = (obs["allmap"][0] / (obs["allmap"][1] + 1e-6)).unsqueeze(-1).contiguous()
    g = torch.Generator().manual_seed(0)
    start = dict(truth)
    start["colors"] = (truth["colors"] + 0.25 * torch.randn(P, 3, generator=g).to(dev)).clamp(0, 1)
    start["means3D"] = truth["means3D"] + 0.01 * torch.randn(P, 3, generator=g).to(dev)
    soa = gs_optim.GaussianSoA({k: v.clone() for k, v in start.items()})
    leaves = dict(soa.leaves())
    fopt = gs_optim.FusedGaussianAdam(soa, {"means3D": 1e-4, "colors": 2.5e-3, "opacities": 0.0, "scales": 0.0, "rotations": 0.0})
    ba = ba_shard.KeyframeShardedBA(leaves, lambda q, _kf: gl.mapping_loss(*(lambda pk: (pk["render_color"], pk["allmap"]))(rasterize(q)),
                                                                         gt_color, gt_depth, 0.5, 1.0, 0.0), direct_grads=True)
    losses = []
    for _ in range(120):
        ba.step([0])
        fopt.step(ba.bucket.flat, leaves)
        with torch.no_grad():
            pk = rasterize(leaves)
            losses.append(float(gl.mapping_loss(pk["render_color"], pk["allmap"], gt_color, gt_depth, 0.5, 1.0, 0.0)))
    print(f"mapping: loss {losses[0]:.4f} -> {losses[-1]:.4f}")
    assert losses[-1] < 0.4 * losses[0]
