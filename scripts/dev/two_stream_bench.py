"""Dev experiment: two keyframes per step rendered fwd+bwd (a) back to back on one stream, (b) on two HIP streams.
Kernels of the two streams can fill each other's tails (4800 waves on 1024 SIMDs leave ~18 % of a blend kernel idle)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from gaus_slam_amd import render as gs_render
from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads, random_w2c, setup_camera

P, W, H = 500000, 640, 480
dev = torch.device("cuda")
sc = make_scene(P, W, H, seed=0, regime="mapping")
names = ("means3D", "opacities", "scales", "rotations", "colors")
params = {k: sc[k].to(dev).requires_grad_(True) for k in names}
dcolor, dallmap = [t.to(dev) for t in make_upstream_grads(W, H, seed=1)]
cams = [sc["cam"], setup_camera(W, H, sc["cam"].K, random_w2c(np.random.default_rng(5), 3.0, 0.1) @ sc["cam"].w2c)]
settings = [gs_render.settings_from_camera(c, dev, use_sa=True) for c in cams]


def one(k):
    m2 = torch.zeros_like(params["means3D"], requires_grad=True)
    pkg = gs_render.render(settings[k], params["means3D"], m2, params["opacities"], colors_precomp=params["colors"],
                           scales=params["scales"], rotations=params["rotations"])
    return torch.autograd.grad([pkg["render_color"], pkg["allmap"]], [params[n] for n in names], [dcolor, dallmap])


def sequential():
    return one(0), one(1)


streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def concurrent():
    cur = torch.cuda.current_stream()
    out = []
    for k in (0, 1):
        streams[k].wait_stream(cur)
        with torch.cuda.stream(streams[k]):
            out.append(one(k))
    for s in streams:
        cur.wait_stream(s)
    return out


def interleaved():
    """forward A, forward B, backward A, backward B on two streams."""
    cur = torch.cuda.current_stream()
    pk = []
    for k in (0, 1):
        streams[k].wait_stream(cur)
        with torch.cuda.stream(streams[k]):
            m2 = torch.zeros_like(params["means3D"], requires_grad=True)
            pk.append(gs_render.render(settings[k], params["means3D"], m2, params["opacities"], colors_precomp=params["colors"],
                                       scales=params["scales"], rotations=params["rotations"]))
    out = []
    for k in (0, 1):
        with torch.cuda.stream(streams[k]):
            out.append(torch.autograd.grad([pk[k]["render_color"], pk[k]["allmap"]], [params[n] for n in names], [dcolor, dallmap]))
    for s in streams:
        cur.wait_stream(s)
    return out


for name, fn in (("sequential, one stream", sequential), ("two streams, fwd+bwd each", concurrent), ("two streams, interleaved", interleaved)):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms per 2 keyframes")
g1 = sequential(); g2 = interleaved(); torch.cuda.synchronize()
print("max grad diff:", max(float((a - b).abs().max()) for x, y in zip(g1, g2) for a, b in zip(x, y)))
