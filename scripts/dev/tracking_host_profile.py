"""Dev tool: host cost of one tracking iteration (bench.py --workload tracking's loop) on a problem so small that the GPU
time is negligible, and its cProfile."""
import cProfile, io, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaus_slam_amd import loss as gl, render as gs_render, tracking
from gaus_slam_amd.scene_synth import make_scene, random_w2c
P, W, H = 2000, 64, 48
dev = torch.device("cuda")
sc = make_scene(P, W, H, seed=0, regime="tracking")
settings = gs_render.settings_from_camera(sc["cam"], dev, use_sa=True)
p = {k: sc[k].to(dev) for k in ("means3D", "opacities", "scales", "rotations", "colors")}
g = torch.Generator().manual_seed(0)
gt_color = torch.rand(H, W, 3, generator=g).to(dev); gt_depth = (0.5 + 5 * torch.rand(H, W, 1, generator=g)).to(dev)
w2c = random_w2c(np.random.default_rng(1), 2.0, 0.05).to(dev).requires_grad_(True)
opt = torch.optim.Adam([w2c], lr=0.0, fused=True)
def one_step():
    opt.zero_grad(set_to_none=True)
    pkg = tracking.render_tracking(settings, w2c, p["means3D"], p["opacities"], p["colors"], p["scales"], p["rotations"])
    _l, g_c, g_a = gl.tracking_loss_and_grads(pkg["render_color"], pkg["allmap"], gt_color, gt_depth, 0.5, 1.0)
    torch.autograd.backward([pkg["render_color"], pkg["allmap"]], [g_c, g_a])
    opt.step()
for _ in range(50): one_step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): one_step()
torch.cuda.synchronize(); print("host-bound tracking iteration: %.1f us" % ((time.perf_counter() - t0) / 300 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): one_step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(22); print(s.getvalue()[:5000])
