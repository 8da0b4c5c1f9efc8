"""Dev tool: ms per step of the bench's op loop as a function of the number of timed steps K (sync on both sides)."""
import sys, time, torch
sys.path.insert(0, '.')
from gaus_slam_amd import render as gs_render
from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads
dev = torch.device('cuda', 0)
P, W, H = 500000, 640, 480
sc = make_scene(P, W, H, seed=0, regime='mapping')
names = ("means3D", "opacities", "scales", "rotations", "colors")
p = {k: sc[k].to(dev).requires_grad_(True) for k in names}
dc, da = make_upstream_grads(W, H, seed=1); dc, da = dc.to(dev), da.to(dev)
st = gs_render.settings_from_camera(sc['cam'], dev, use_sa=True)
def step():
    m2 = torch.zeros_like(p['means3D'], requires_grad=True)
    pkg = gs_render.render(st, p['means3D'], m2, p['opacities'], colors_precomp=p['colors'], scales=p['scales'], rotations=p['rotations'])
    torch.autograd.backward([pkg['render_color'], pkg['allmap']], [dc, da])
    for v in p.values(): v.grad = None
for _ in range(20): step()
for K in (1, 2, 5, 10, 20, 50, 200, 1000):
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K): step()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"K={K:5d}  total {best*1e3:9.3f} ms   per step {best/K*1e3:.4f} ms", flush=True)
