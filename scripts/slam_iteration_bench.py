"""SLAM-iteration micro-bench on top of the rasterizer (run on the GPU box): one TRACKING iteration and one MAPPING
iteration, each in the reference's formulation (PyTorch transform / post-op / loss / torch.optim.Adam around the
operator: render/__init__.py:17-50, slam/Loss.py:22-58, scene/Gaussians.py:121-137) and in the fused formulation this
package offers (render_tracking + tracking_loss; mapping_loss + gradients written into the bucket + FusedGaussianAdam).
Prints ms per iteration.  The operator itself is identical in both columns."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gaus_slam_amd import ba_shard, loss as gl, optim, render as gsr, tracking
from gaus_slam_amd.scene_synth import make_scene, random_w2c

P, W, H = 500000, 640, 480
dev = torch.device("cuda")
LRS = dict(xyz=1e-4, opacity=0.05, scaling=1e-3, rotation=1e-3, rgb=2.5e-3)  # configs/replica/config_fast.py:115-122
g = torch.Generator().manual_seed(0)
gt_color = torch.rand(H, W, 3, generator=g).to(dev)
gt_depth = (0.5 + 5 * torch.rand(H, W, 1, generator=g)).to(dev)
names = ("means3D", "opacities", "scales", "rotations", "colors")


def post_and_loss_ref(pkg, mode):
    """render/__init__.py:46-49 + slam/Loss.py:22-58 in plain PyTorch (default configuration)."""
    d = pkg["allmap"][0:1] / (pkg["allmap"][1:2] + 1e-6)
    d = torch.where((d > 1e2) | (d < 1e-2), torch.zeros_like(d), d)
    a = torch.nan_to_num(pkg["allmap"][1:2], 0, 0).permute(1, 2, 0)
    d = torch.nan_to_num(d, 0, 0).permute(1, 2, 0)
    c = torch.nan_to_num(pkg["render_color"], 0, 0).permute(1, 2, 0)
    dist = torch.nan_to_num(pkg["allmap"][6:7], 0, 0).permute(1, 2, 0)
    dm = (gt_depth > 1e-5).view(-1) & (d > 1e-5).view(-1)
    if mode == 0:
        m = dm & (a > 0.9).view(-1)
        return 0.5 * (c - gt_color).abs().view(-1, 3)[m].sum() + (d - gt_depth).abs().view(-1, 1)[m].sum()
    return (0.5 * (c - gt_color).abs().view(-1, 3)[dm].mean() + (d - gt_depth).abs().view(-1, 1)[dm].mean()
            + 0.1 * dist.view(-1, 1)[dm].mean())


def timeit(fn, n=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


# ------------------------------------------------------------------ tracking
sc = make_scene(P, W, H, seed=0, regime="tracking")
settings = gsr.settings_from_camera(sc["cam"], dev, use_sa=True)
p = {k: sc[k].to(dev) for k in names}
w_ref = random_w2c(np.random.default_rng(1), 2.0, 0.05).to(dev).requires_grad_(True)
w_fus = w_ref.detach().clone().requires_grad_(True)
opt_ref, opt_fus = torch.optim.Adam([w_ref], lr=0.0), torch.optim.Adam([w_fus], lr=0.0)


def track_ref():
    opt_ref.zero_grad(set_to_none=True)
    means_cam = (w_ref[:3, :3] @ p["means3D"].T + w_ref[:3, 3:]).T
    aw, ax, ay, az = tracking.matrix_to_quaternion(w_ref[:3, :3].detach())
    bw, bx, by, bz = p["rotations"].unbind(1)
    rot = torch.stack([aw*bw-ax*bx-ay*by-az*bz, aw*bx+ax*bw+ay*bz-az*by, aw*by-ax*bz+ay*bw+az*bx, aw*bz+ax*by-ay*bx+az*bw], 1)
    m2 = torch.zeros_like(means_cam, requires_grad=True)
    pkg = gsr.render(settings, means_cam, m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"], rotations=rot)
    post_and_loss_ref(pkg, 0).backward()
    opt_ref.step()


def track_fused():
    opt_fus.zero_grad(set_to_none=True)
    pkg = tracking.render_tracking(settings, w_fus, p["means3D"], p["opacities"], p["colors"], p["scales"], p["rotations"])
    gl.tracking_loss(pkg["render_color"], pkg["allmap"], gt_color, gt_depth, 0.5, 1.0).backward()
    opt_fus.step()


FUSED_ONLY = "--fused-only" in sys.argv  # for profiling the fused path alone
t_ref = float("nan") if FUSED_ONLY else timeit(track_ref)
print(f"tracking iteration @ {W}x{H}, {P} Gaussians:  reference formulation {t_ref:.3f} ms   fused {timeit(track_fused):.3f} ms")

# ------------------------------------------------------------------ mapping
sc = make_scene(P, W, H, seed=0, regime="mapping")
settings = gsr.settings_from_camera(sc["cam"], dev, use_sa=True)
ref = {k: sc[k].to(dev).clone().requires_grad_(True) for k in names}
ropt = torch.optim.Adam([{"params": [ref[k]], "lr": 0.0} for k in names], lr=0.0, eps=1e-15)


def rasterize(q):
    m2 = torch.zeros_like(q["means3D"], requires_grad=True)
    return gsr.render(settings, q["means3D"], m2, q["opacities"], colors_precomp=q["colors"], scales=q["scales"],
                      rotations=q["rotations"])


def map_ref():
    ropt.zero_grad(set_to_none=True)
    post_and_loss_ref(rasterize(ref), 1).backward()
    ropt.step()


soa = optim.GaussianSoA({k: sc[k].to(dev) for k in names})
leaves = dict(soa.leaves())
fopt = optim.FusedGaussianAdam(soa, {})  # lr = 0: the scene stays fixed, the full moment update runs
ba = ba_shard.KeyframeShardedBA(leaves, lambda q, _kf: gl.mapping_loss(*(lambda pk: (pk["render_color"], pk["allmap"]))(rasterize(q)),
                                                                     gt_color, gt_depth, 0.5, 1.0, 0.1), direct_grads=True)


def map_fused():
    ba.step([0])
    fopt.step(ba.bucket.flat)


m_ref = float("nan") if FUSED_ONLY else timeit(map_ref)
print(f"mapping  iteration @ {W}x{H}, {P} Gaussians:  reference formulation {m_ref:.3f} ms   fused {timeit(map_fused):.3f} ms")
