"""Dev: cumulative host profile of one fwd+bwd step at a tiny problem size (GPU time negligible)."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, ".")
import torch
from gaus_slam_amd import ba_shard, render as gs_render
from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads
P, W, H = 2000, 64, 48
dev = torch.device("cuda")
sc = make_scene(P, W, H, seed=0, regime="mapping")
params = {k: sc[k].to(dev).requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations", "colors")}
dc, da = make_upstream_grads(W, H); dc, da = dc.to(dev), da.to(dev)
settings = gs_render.settings_from_camera(sc["cam"], dev)
def fwd():
    m2 = torch.zeros_like(params["means3D"], requires_grad=True)
    return gs_render.render(settings, params["means3D"], m2, params["opacities"], colors_precomp=params["colors"], scales=params["scales"], rotations=params["rotations"])
def step():
    for p in params.values(): p.grad = None
    pkg = fwd()
    torch.autograd.backward([pkg["render_color"], pkg["allmap"]], [dc, da])
for _ in range(50): step()
torch.cuda.synchronize()
for name, fn in (("forward only", lambda: fwd()), ("forward+backward", step)):
    t0 = time.perf_counter()
    for _ in range(300): fn()
    torch.cuda.synchronize()
    print(name, "%.1f us" % ((time.perf_counter() - t0) / 300 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:4500])
