"""Dev tool behind DESIGN.md section 6 (K-keyframe optimisation policy): mapping loops on a synthetic scene with M keyframes,
K keyframes per optimizer step (gradients summed or averaged, learning rates scaled or not), mean loss over all keyframes.
usage: kpolicy_probe.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaus_slam_amd import ba_shard, loss as gl, optim as gs_optim, render as gs_render
from gaus_slam_amd.scene_synth import make_scene, random_w2c, setup_camera
dev = torch.device("cuda", 0)
P, W, H, M = 60000, 320, 240, 8
sc = make_scene(P, W, H, seed=3, regime="mapping")
names = ("means3D", "opacities", "scales", "rotations", "colors")
truth = {k: sc[k].to(dev) for k in names}
rng = np.random.default_rng(7)
cams = [sc["cam"]] + [setup_camera(W, H, sc["cam"].K, random_w2c(rng, 4.0, 0.15) @ sc["cam"].w2c) for _ in range(M - 1)]
sts = [gs_render.settings_from_camera(c, dev, use_sa=True) for c in cams]
def rasterize(q, kf):
    m2 = torch.zeros_like(q["means3D"], requires_grad=True)
    return gs_render.render(sts[kf], q["means3D"], m2, q["opacities"], colors_precomp=q["colors"], scales=q["scales"], rotations=q["rotations"])
gts = []
with torch.no_grad():
    for kf in range(M):
        obs = rasterize(truth, kf)
        gts.append((obs["render_color"].permute(1, 2, 0).contiguous(), (obs["allmap"][0] / (obs["allmap"][1] + 1e-6)).unsqueeze(-1).contiguous()))
g = torch.Generator().manual_seed(0)
start = dict(truth)
start["colors"] = (truth["colors"] + 0.25 * torch.randn(P, 3, generator=g).to(dev)).clamp(0, 1)
start["means3D"] = truth["means3D"] + 0.01 * torch.randn(P, 3, generator=g).to(dev)
LR = {"means3D": 1e-4, "colors": 2.5e-3, "opacities": 0.0, "scales": 0.0, "rotations": 0.0}
def loss_of(q, kf):
    pk = rasterize(q, kf)
    return gl.mapping_loss(pk["render_color"], pk["allmap"], gts[kf][0], gts[kf][1], 0.5, 1.0, 0.0)
def run(K, steps, average=False, lr_scale=1.0, seed=1):
    soa = gs_optim.GaussianSoA({k: v.clone() for k, v in start.items()})
    leaves = dict(soa.leaves())
    fopt = gs_optim.FusedGaussianAdam(soa, {k: v * lr_scale for k, v in LR.items()})
    ba = ba_shard.KeyframeShardedBA(leaves, loss_of, direct_grads=True, average=average)
    order = np.random.default_rng(seed)
    for s in range(steps):
        kfs = [int(k) for k in order.choice(M, size=K, replace=False)]  # the reference draws a random keyframe per step (Backend.py:103)
        ba.step(kfs)
        if average and K > 1: ba.bucket.flat.div_(K)   # world size 1: emulate the mean
        fopt.step(ba.bucket.flat, leaves)
    with torch.no_grad():
        return float(np.mean([float(loss_of(leaves, kf)) for kf in range(M)]))
with torch.no_grad():
    l0 = float(np.mean([float(loss_of({k: v for k, v in start.items()}, kf)) for kf in range(M)]))
print(f"start loss {l0:.5f}")
S = 240
for name, kw in [("K=1, S steps", dict(K=1, steps=S)), ("K=1, S/2 steps", dict(K=1, steps=S // 2)), ("K=2 sum, S steps", dict(K=2, steps=S)),
                 ("K=2 mean, S steps", dict(K=2, steps=S, average=True)), ("K=2 sum, S/2 steps", dict(K=2, steps=S // 2)),
                 ("K=2 sum, S/2 steps, lr x sqrt2", dict(K=2, steps=S // 2, lr_scale=2 ** 0.5)), ("K=2 sum, S/2 steps, lr x 2", dict(K=2, steps=S // 2, lr_scale=2.0)),
                 ("K=4 sum, S steps", dict(K=4, steps=S)), ("K=4 sum, S/4 steps", dict(K=4, steps=S // 4)), ("K=4 sum, S/4 steps, lr x 2", dict(K=4, steps=S // 4, lr_scale=2.0)),
                 ("K=4 sum, S/4 steps, lr x 4", dict(K=4, steps=S // 4, lr_scale=4.0))]:
    vals = [run(seed=sd, **kw) for sd in (1, 2)]
    print(f"{name:34s} loss {np.mean(vals):.5f}  ({vals[0]:.5f}, {vals[1]:.5f})  = {np.mean(vals) / l0:.3f} of start")
