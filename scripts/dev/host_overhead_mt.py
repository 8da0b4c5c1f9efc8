"""Dev tool: host cost of one fwd+bwd step (tiny problem) with the autograd engine's device thread on and off
(torch.autograd.set_multithreading_enabled)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gaus_slam_amd import ba_shard, render as gs_render
from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads
P, W, H = 2000, 64, 48
dev = torch.device("cuda")
sc = make_scene(P, W, H, seed=0, regime="mapping")
params = {k: sc[k].to(dev).requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations", "colors")}
dc, da = make_upstream_grads(W, H); dc, da = dc.to(dev), da.to(dev)
settings = gs_render.settings_from_camera(sc["cam"], dev)
def render_fn(p, _):
    m2 = torch.empty_like(p["means3D"]).requires_grad_(True)
    pkg = gs_render.render(settings, p["means3D"], m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"], rotations=p["rotations"])
    return (pkg["render_color"], pkg["allmap"]), (dc, da)
ba = ba_shard.KeyframeShardedBA(params, render_fn)
for mt in (True, False, True, False):
    torch.autograd.set_multithreading_enabled(mt)
    for _ in range(50): ba.step([0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(400): ba.step([0])
    torch.cuda.synchronize()
    print(f"autograd multithreading {mt}: host-bound step {(time.perf_counter() - t0) / 400 * 1e6:.1f} us", flush=True)
