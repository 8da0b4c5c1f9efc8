import sys, time, torch
sys.path.insert(0, '.')
from gaus_slam_amd import render as gs_render
from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads
dev = torch.device('cuda', 0)
P, W, H = 500000, 640, 480
sc = make_scene(P, W, H, seed=0, regime='mapping')
names = ("means3D", "opacities", "scales", "rotations", "colors")
p = {k: sc[k].to(dev).requires_grad_(True) for k in names}
dc, da = make_upstream_grads(W, H, seed=1, channels=(0, 1, 5, 6)); dc, da = dc.to(dev), da.to(dev)
st = gs_render.settings_from_camera(sc['cam'], dev, use_sa=True)
def step():
    m2 = torch.zeros_like(p['means3D'], requires_grad=True)
    pkg = gs_render.render(st, p['means3D'], m2, p['opacities'], colors_precomp=p['colors'], scales=p['scales'], rotations=p['rotations'])
    torch.autograd.backward([pkg['render_color'], pkg['allmap']], [dc, da])
    g = p['means3D'].grad
    for v in p.values(): v.grad = None
    return g
for _ in range(20): step()
torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated(); r0 = torch.cuda.memory_reserved()
g0 = step().clone(); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3000): g = step()
torch.cuda.synchronize(); t = time.perf_counter() - t0
print('3000 steps', t / 3000 * 1e3, 'ms/step; allocated delta', torch.cuda.memory_allocated() - m0, 'reserved delta', torch.cuda.memory_reserved() - r0)
print('max |g - g0| / max|g0| =', float((g - g0).abs().max() / g0.abs().max()), 'finite', bool(torch.isfinite(g).all()))
