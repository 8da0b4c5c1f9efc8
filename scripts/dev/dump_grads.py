"""Dev tool: deterministic backward on the headline scene, all gradient tensors to an .npz (to compare two library builds bit for bit)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from gaus_slam_amd import render as gs_render, rasterizer
from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads
dev = torch.device('cuda', 0)
P, W, H = 200000, 640, 480
out = {}
for regime in ("mapping", "tracking"):
    sc = make_scene(P, W, H, seed=0, regime=regime)
    names = ("means3D", "opacities", "scales", "rotations", "colors")
    p = {k: sc[k].to(dev).requires_grad_(True) for k in names}
    dc, da = make_upstream_grads(W, H, seed=1); dc, da = dc.to(dev), da.to(dev)
    st = gs_render.settings_from_camera(sc['cam'], dev, use_sa=True)
    rasterizer.set_deterministic(True)
    m2 = torch.zeros_like(p['means3D'], requires_grad=True)
    pkg = gs_render.render(st, p['means3D'], m2, p['opacities'], colors_precomp=p['colors'], scales=p['scales'], rotations=p['rotations'])
    torch.autograd.backward([pkg['render_color'], pkg['allmap']], [dc, da])
    for k in names:
        out[f"{regime}_{k}"] = p[k].grad.cpu().numpy()
    out[f"{regime}_means2D"] = m2.grad.cpu().numpy()
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1], {k: float(np.abs(v).max()) for k, v in out.items() if k.endswith("means2D")})
