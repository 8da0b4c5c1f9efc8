"""Dev: cost of the opt-in deterministic backward at the bench size (stage timings, default vs deterministic)."""
import ctypes as C, json, sys
sys.path.insert(0, ".")
import torch
from gaus_slam_amd import _lib, rasterizer, render as gs_render
from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads
P, W, H = 500000, 640, 480
dev = torch.device("cuda")
sc = make_scene(P, W, H, seed=0, regime="mapping")
p = {k: sc[k].to(dev).requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations", "colors")}
dc, da = [t.to(dev) for t in make_upstream_grads(W, H, seed=1)]
settings = gs_render.settings_from_camera(sc["cam"], dev, use_sa=True)
L = _lib.lib()
res = {}
for det in (False, True):
    rasterizer.set_deterministic(det)
    L.gs2d_stage_timing_enable(1)
    buf = (C.c_float * 9)(); acc = [0.0] * 9
    for it in range(13):
        m2 = torch.zeros_like(p["means3D"], requires_grad=True)
        pkg = gs_render.render(settings, p["means3D"], m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"], rotations=p["rotations"])
        torch.autograd.backward([pkg["render_color"], pkg["allmap"]], [dc, da])
        L.gs2d_stage_timing_read(buf)
        if it >= 3:
            for i in range(9): acc[i] += max(buf[i], 0.0) / 10
    L.gs2d_stage_timing_enable(0)
    res["deterministic" if det else "default"] = {"blend_bwd_ms": round(acc[6], 4), "blend_fwd_ms": round(acc[5], 4), "preprocess_bwd_ms": round(acc[7], 4)}
rasterizer.set_deterministic(False)
print(json.dumps({"bench": "deterministic backward, 640x480 / 500k (blend_bwd_ms includes slot memset + inverse map + ordered reduce)", "results": res}))
