"""Dev tool: default backward (LDS plain-store / atomic accumulate, global float atomics) against the deterministic one on random scenes."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from gaus_slam_amd import render as gs_render, rasterizer
from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads
dev = torch.device('cuda', 0)
rng = np.random.default_rng(0)
names = ("means3D", "opacities", "scales", "rotations", "colors")
worst = 0.0
for it in range(40):
    W = int(rng.integers(3, 60)) * 16 - int(rng.integers(0, 16)); H = int(rng.integers(3, 45)) * 16 - int(rng.integers(0, 16))
    P = int(rng.choice([50, 700, 5000, 40000, 150000]))
    regime = ["mapping", "tracking"][it % 2]
    use_sa = bool(it % 3)
    sc = make_scene(P, W, H, seed=100 + it, regime=regime, scale_lo=0.3, scale_hi=float(rng.choice([4.0, 12.0, 40.0])))
    chans = (0, 1, 5, 6) if it % 4 else (0, 1, 2, 3, 4, 5, 6)
    dc, da = make_upstream_grads(W, H, seed=it, channels=chans); dc, da = (dc * W * H).to(dev), (da * W * H).to(dev)
    st = gs_render.settings_from_camera(sc['cam'], dev, use_sa=use_sa)
    res = []
    for det in (False, True):
        rasterizer.set_deterministic(det)
        p = {k: sc[k].to(dev).requires_grad_(True) for k in names}
        m2 = torch.zeros_like(p['means3D'], requires_grad=True)
        pkg = gs_render.render(st, p['means3D'], m2, p['opacities'], colors_precomp=p['colors'], scales=p['scales'], rotations=p['rotations'])
        torch.autograd.backward([pkg['render_color'], pkg['allmap']], [dc, da])
        res.append([p[k].grad.double().cpu() for k in names] + [m2.grad.double().cpu()])
    rasterizer.set_deterministic(False)
    err = 0.0
    for a, b in zip(*res):
        assert torch.isfinite(a).all() and torch.isfinite(b).all()
        scale = float(b.abs().max())
        if scale > 0:
            err = max(err, float((a - b).abs().max()) / scale)
    worst = max(worst, err)
    print(f"{it:2d} P={P:6d} {W}x{H} {regime:8s} sa={int(use_sa)} dn={int(len(chans) == 7)}  max rel err {err:.2e}", flush=True)
print("worst", worst)
assert worst < 1e-4
