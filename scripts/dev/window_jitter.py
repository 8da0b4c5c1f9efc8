"""Dev tool: distribution of the bench protocol's 20-step windows (sync, 5 warm-up steps, sync, 20 timed steps) over many
repetitions, for the library selected by GS2D_LIB_PATH: how often does a window come out slow, and how slow?
usage: window_jitter.py [windows]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaus_slam_amd import ba_shard, render as gs_render
from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
P, W, H = 500000, 640, 480
sc = make_scene(P, W, H, seed=0, regime="mapping")
params = {k: sc[k].to(dev).requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations", "colors")}
dc, da = make_upstream_grads(W, H, seed=1); dc, da = dc.to(dev), da.to(dev)
st = gs_render.settings_from_camera(sc["cam"], dev)
def one(p, kf):
    m2 = torch.empty_like(p["means3D"]).requires_grad_(True)
    pk = gs_render.render(st, p["means3D"], m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"], rotations=p["rotations"])
    return (pk["render_color"], pk["allmap"]), (dc, da)
ba = ba_shard.KeyframeShardedBA(params, one)
for _ in range(3000): ba.step([0])
torch.cuda.synchronize()
win, worst = [], []
for rep in range(n):
    torch.cuda.synchronize()
    for _ in range(5): ba.step([0])
    torch.cuda.synchronize()
    ts = [time.perf_counter()]
    for _ in range(20):
        ba.step([0]); ts.append(time.perf_counter())
    torch.cuda.synchronize(); tend = time.perf_counter()
    win.append((tend - ts[0]) / 20 * 1e3); worst.append(np.diff(ts).max() * 1e3)
win, worst = np.array(win), np.array(worst)
print(f"{os.path.basename(os.environ.get('GS2D_LIB_PATH', 'product'))}: {n} windows of 20 steps: ms/step median {np.median(win):.4f} mean {win.mean():.4f} "
      f"p90 {np.percentile(win, 90):.4f} max {win.max():.4f}; windows > 1.05 x median: {(win > 1.05 * np.median(win)).sum()}; "
      f"slowest host step per window (ms): median {np.median(worst):.2f} p90 {np.percentile(worst, 90):.2f} max {worst.max():.2f}", flush=True)
