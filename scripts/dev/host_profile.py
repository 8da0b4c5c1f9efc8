"""Host-overhead breakdown of one bench step (run on the GPU box)."""
import cProfile, pstats, sys, os, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gaus_slam_amd import ba_shard, render as gs_render
from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads
P, W, H = 500000, 640, 480
dev = torch.device("cuda")
sc = make_scene(P, W, H, seed=0, regime="mapping")
params = {k: sc[k].to(dev).requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations", "colors")}
dc, da = make_upstream_grads(W, H); dc, da = dc.to(dev), da.to(dev)
settings = gs_render.settings_from_camera(sc["cam"], dev)
def render_fn(p, _):
    m2 = torch.zeros_like(p["means3D"], requires_grad=True)
    pkg = gs_render.render(settings, p["means3D"], m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"], rotations=p["rotations"])
    return (pkg["render_color"], pkg["allmap"]), (dc, da)
ba = ba_shard.KeyframeShardedBA(params, render_fn)
for _ in range(10): ba.step([0])
torch.cuda.synchronize()
for mode in ("async", "sync-each"):
    t0 = time.perf_counter()
    for _ in range(50):
        ba.step([0])
        if mode == "sync-each": torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(mode, "ms/step", (time.perf_counter() - t0) / 50 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(50): ba.step([0])
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
print("nproc", os.cpu_count(), "load", os.getloadavg())
