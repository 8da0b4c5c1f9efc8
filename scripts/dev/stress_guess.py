"""Dev tool: duplicate-before-num_rendered under a count that jumps around.  Same problem shape, scene scale drawn at random per
iteration (num_rendered varies by more than 5x), single-frame and batched calls interleaved: every call is repeated at once
(the repeat's guess fits by construction) and must give identical lists and images."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from gaus_slam_amd import render as gs_render
from gaus_slam_amd.scene_synth import make_scene, random_w2c, setup_camera
dev = torch.device('cuda', 0)
rng = np.random.default_rng(0)
W, H, P = 320, 240, 8000
names = ("means3D", "opacities", "scales", "rotations", "colors")
redo = 0
last_R = None
for it in range(120):
    hi = float(np.exp(rng.uniform(np.log(0.8), np.log(30.0))))
    sc = make_scene(P, W, H, seed=1000 + it, regime="mapping", scale_lo=0.3, scale_hi=hi)
    p = {k: sc[k].to(dev) for k in names}
    m2 = torch.zeros_like(p["means3D"])
    cams = [sc["cam"]] + [setup_camera(W, H, sc["cam"].K, random_w2c(np.random.default_rng(it * 7 + i), 3.0, 0.1) @ sc["cam"].w2c) for i in range(2)]
    sts = [gs_render.settings_from_camera(c, dev, use_sa=bool(it % 2)) for c in cams]
    def call():
        if it % 3 == 2:
            pk = gs_render.render_batch(sts, p["means3D"], m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"], rotations=p["rotations"])
            return [t for q in pk for t in (q["render_color"], q["allmap"], q["radius"])]
        q = gs_render.render(sts[0], p["means3D"], m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"], rotations=p["rotations"])
        return [q["render_color"], q["allmap"], q["radius"]]
    a = call(); b = call()
    for x, y in zip(a, b):
        assert torch.equal(x, y), it
    assert all(torch.isfinite(x.float()).all() for x in a)
torch.cuda.synchronize()
print("120 iterations, single and batched calls interleaved: repeated calls identical, all outputs finite")
