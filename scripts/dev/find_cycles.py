"""Dev tool: which objects of one op step only the cyclic garbage collector frees."""
import gc, sys, collections, torch
sys.path.insert(0, '.')
from gaus_slam_amd import render as gs_render
from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads
dev = torch.device('cuda', 0)
P, W, H = 20000, 320, 240
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
for _ in range(3): step()
gc.collect(); gc.disable()
m0 = torch.cuda.memory_allocated()
for _ in range(5): step()
torch.cuda.synchronize()
print("allocated after 5 steps without the collector: +%.1f MB" % ((torch.cuda.memory_allocated() - m0) / 1e6))
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
cnt = collections.Counter(type(o).__name__ for o in gc.garbage)
print("collected", n, cnt.most_common(12))
for o in gc.garbage:
    if type(o).__name__ in ("dict", "function", "cell", "tuple", "list") :
        continue
    print(type(o), str(o)[:120])
    break
for o in gc.garbage[:40]:
    if isinstance(o, dict):
        print("dict keys:", list(o.keys())[:12])
