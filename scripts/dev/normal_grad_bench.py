"""Dev tool: backward stage time with and without upstream gradients on the normal channels (any_dn path)."""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from gaus_slam_amd import _lib, render as gs_render
from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads
P, W, H = 500000, 640, 480
dev = torch.device("cuda")
sc = make_scene(P, W, H, seed=0, regime="mapping")
params = {k: sc[k].to(dev).requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations", "colors")}
settings = gs_render.settings_from_camera(sc["cam"], dev, use_sa=True)
L = _lib.lib()
for chans in ((0, 1, 5, 6), (0, 1, 2, 3, 4, 5, 6)):
    dcolor, dallmap = [t.to(dev) for t in make_upstream_grads(W, H, seed=1, channels=chans)]
    L.gs2d_stage_timing_enable(1)
    acc = [0.0] * 8
    buf = (C.c_float * 9)()
    for it in range(13):
        for p in params.values(): p.grad = None
        m2 = torch.zeros_like(params["means3D"], requires_grad=True)
        pkg = gs_render.render(settings, params["means3D"], m2, params["opacities"], colors_precomp=params["colors"],
                               scales=params["scales"], rotations=params["rotations"])
        torch.autograd.backward([pkg["render_color"], pkg["allmap"]], [dcolor, dallmap])
        L.gs2d_stage_timing_read(buf)
        if it >= 3:
            for i in range(8): acc[i] += max(buf[i], 0.0)
    L.gs2d_stage_timing_enable(0)
    print(chans, "blend_fwd %.1f us, blend_bwd %.1f us, preprocess_bwd %.1f us" % (acc[5] * 100, acc[6] * 100, acc[7] * 100))
