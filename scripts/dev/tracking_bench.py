"""Tracking-iteration micro-bench: reference formulation (PyTorch transform + autograd, render/__init__.py:31-40)
vs the fused posed operator.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gaus_slam_amd import render as gsr, tracking
from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads, random_w2c
import numpy as np
P, W, H = 500000, 640, 480
dev = torch.device("cuda")
sc = make_scene(P, W, H, seed=0, regime="tracking")
w2c0 = random_w2c(np.random.default_rng(1), 2.0, 0.05).to(dev)
settings = gsr.settings_from_camera(sc["cam"], dev, use_sa=True)
p = {k: sc[k].to(dev) for k in ("means3D", "scales", "rotations", "opacities", "colors")}
dc, da = make_upstream_grads(W, H, channels=(0, 1)); dc, da = dc.to(dev), da.to(dev)
def fused():
    w = w2c0.clone().requires_grad_(True)
    pkg = tracking.render_tracking(settings, w, p["means3D"], p["opacities"], p["colors"], p["scales"], p["rotations"])
    torch.autograd.backward([pkg["render_color"], pkg["allmap"]], [dc, da]); return w.grad
def unfused():
    w = w2c0.clone().requires_grad_(True)
    means_cam = (w[:3, :3] @ p["means3D"].T + w[:3, 3:]).T
    qc = tracking.matrix_to_quaternion(w[:3, :3].detach())
    aw, ax, ay, az = qc; bw, bx, by, bz = p["rotations"].unbind(1)
    rot = torch.stack([aw*bw-ax*bx-ay*by-az*bz, aw*bx+ax*bw+ay*bz-az*by, aw*by-ax*bz+ay*bw+az*bx, aw*bz+ax*by-ay*bx+az*bw], 1)
    m2 = torch.zeros_like(means_cam, requires_grad=True)
    pkg = gsr.render(settings, means_cam, m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"], rotations=rot)
    torch.autograd.backward([pkg["render_color"], pkg["allmap"]], [dc, da]); return w.grad
for name, fn in (("unfused (reference formulation)", unfused), ("fused posed op", fused)):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter()-t0)/50*1e3:.3f} ms / tracking iteration")
