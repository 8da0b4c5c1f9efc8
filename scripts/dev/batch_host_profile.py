"""Dev tool: is a K-keyframe step host-bound?  Times one BA step three ways on the bench scene: wall time with a device sync per
step, host enqueue time (no sync), and cProfile of the host side.  usage: batch_host_profile.py K [--no-batch]"""
import cProfile, pstats, sys, time, io, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaus_slam_amd import ba_shard, render as gs_render
from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads, random_w2c, setup_camera
K = int(sys.argv[1]); nobatch = "--no-batch" in sys.argv
dev = torch.device("cuda", 0)
P, W, H = 500000, 640, 480
sc = make_scene(P, W, H, seed=0, regime="mapping")
params = {k: sc[k].to(dev).requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations", "colors")}
dc, da = make_upstream_grads(W, H, seed=1); dc, da = dc.to(dev), da.to(dev)
sts = [gs_render.settings_from_camera(sc["cam"] if i == 0 else setup_camera(W, H, sc["cam"].K, random_w2c(np.random.default_rng(2000 + i), 3.0, 0.1) @ sc["cam"].w2c), dev) for i in range(K)]
def one(p, kf):
    m2 = torch.empty_like(p["means3D"]).requires_grad_(True)
    pk = gs_render.render(sts[kf], p["means3D"], m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"], rotations=p["rotations"])
    return (pk["render_color"], pk["allmap"]), (dc, da)
def batch(p, kfs):
    m2 = torch.empty_like(p["means3D"]).requires_grad_(True)
    pks = gs_render.render_batch([sts[k] for k in kfs], p["means3D"], m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"], rotations=p["rotations"])
    outs, ups = [], []
    for pk in pks:
        outs += [pk["render_color"], pk["allmap"]]; ups += [dc, da]
    return outs, ups
ba = ba_shard.KeyframeShardedBA(params, one, batch_fn=None if nobatch else batch)
kfs = list(range(K))
for _ in range(300): ba.step(kfs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): ba.step(kfs)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 100
host = []
for _ in range(50):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ba.step(kfs); host.append(time.perf_counter() - t0)
print(f"K={K} {'sequential' if nobatch else 'batch'}: wall {wall*1e3:.3f} ms/step ({K/wall:.0f} frames/s), host call on an idle GPU {np.median(host)*1e3:.3f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(50): ba.step(kfs)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print("\n".join(s.getvalue().splitlines()[:40]))
# the bench's protocol: sync, 5 warm-up steps, sync, 20 timed steps -- and per-step host timestamps of such a window
for rep in range(3):
    torch.cuda.synchronize()
    for _ in range(5): ba.step(kfs)
    torch.cuda.synchronize()
    ts = [time.perf_counter()]
    for _ in range(20):
        ba.step(kfs); ts.append(time.perf_counter())
    torch.cuda.synchronize(); tend = time.perf_counter()
    d = np.diff(ts) * 1e3
    print(f"20-step window: {(tend - ts[0]) / 20 * 1e3:.3f} ms/step; host return times per step (ms): " + " ".join(f"{x:.2f}" for x in d) + f"; drain {1e3 * (tend - ts[-1]):.2f}")

# stream timeline of one step from the library's stage events (begin/end relative to the step's first kernel)
import ctypes as C
from gaus_slam_amd import _lib
L = _lib.lib()
L.gs2d_stage_timing_enable(1)
names = ["preprocess", "scan", "duplicate", "sort", "ranges", "blend_fwd", "blend_bwd", "preprocess_bwd", "cull"]
for rep in range(3):
    for _ in range(6): ba.step(kfs)
    buf = (C.c_float * 18)()
    L.gs2d_stage_timing_read_abs(buf)
    ev = sorted((buf[2 * i], buf[2 * i + 1], n) for i, n in enumerate(names) if buf[2 * i + 1] >= 0 or buf[2 * i] != -1.0)
    print("timeline (us): " + "  ".join(f"{n} {b * 1e3:.0f}..{e * 1e3:.0f}" for b, e, n in ev if n not in ("ranges", "cull")))
L.gs2d_stage_timing_enable(0)
