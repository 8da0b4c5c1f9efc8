"""Dev tool: ms per step of the op loop in consecutive windows, from the first GPU work of the process on (clock ramp of a cold card)."""
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
torch.cuda.synchronize()
T0 = time.perf_counter()
for w in range(30):
    t0 = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"t={t1 - T0:6.2f} s  window {w:2d}: {(t1 - t0) / 200 * 1e3:.4f} ms/step", flush=True)
