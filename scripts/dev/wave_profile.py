"""Dev tool: per-wave start/end/trips of the two blend kernels (variant library built with -DGS2D_PROFILE_WAVES).
Run on the GPU box:  GS2D_LIB_PATH=scripts/dev/variants/libprof.so python scripts/dev/wave_profile.py"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from gaus_slam_amd import _lib, render as gs_render
from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads

P = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
W, H = 640, 480
dev = torch.device("cuda")
sc = make_scene(P, W, H, seed=0, regime="mapping")
params = {k: sc[k].to(dev).requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations", "colors")}
dcolor, dallmap = [t.to(dev) for t in make_upstream_grads(W, H, seed=1)]
settings = gs_render.settings_from_camera(sc["cam"], dev, use_sa=True)
for _ in range(5):
    m2 = torch.zeros_like(params["means3D"], requires_grad=True)
    pkg = gs_render.render(settings, params["means3D"], m2, params["opacities"], colors_precomp=params["colors"],
                           scales=params["scales"], rotations=params["rotations"])
    torch.autograd.backward([pkg["render_color"], pkg["allmap"]], [dcolor, dallmap])
torch.cuda.synchronize()
L = _lib.lib()
ntiles = ((W + 15) // 16) * ((H + 15) // 16)
out = {}
for k, name in ((0, "fwd"), (1, "bwd")):
    buf = np.zeros(ntiles * 4 * 4, np.uint64)
    assert L.gs2d_debug_read_wave_profile(k, buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size)) == 0
    a = buf.reshape(ntiles * 4, 4)
    t0, t1, hw = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 3]
    trips = (a[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
    stage_cyc = (a[:, 2] >> np.uint64(32)).astype(np.int64)  # shader cycles spent staging batches (s_memtime)
    out[name] = a
    start = t0.min()
    span = (t1.max() - start) / 100.0  # wall_clock64 ticks at 100 MHz -> us
    dur = (t1 - t0) / 100.0
    end = (t1 - start) / 100.0
    hwid = (hw & 0xffffffff).astype(np.int64); xcc = (hw >> 32).astype(np.int64) & 0xf
    simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
    simd_key = ((xcc * 8 + se) * 2 + sh) * 64 + cu * 4 + simd
    uniq, inv = np.unique(simd_key, return_inverse=True)
    per_simd_trips = np.bincount(inv, weights=trips)
    per_simd_end = np.zeros(len(uniq)); np.maximum.at(per_simd_end, inv, end)
    per_simd_n = np.bincount(inv)
    print(f"== {name}: span {span:.1f} us, waves {len(a)}, total trips {trips.sum()}, trips/wave mean {trips.mean():.1f} "
          f"min {trips.min()} max {trips.max()} cv {trips.std() / trips.mean():.3f}")
    print(f"   wave duration us: mean {dur.mean():.1f} p50 {np.median(dur):.1f} p95 {np.percentile(dur, 95):.1f} max {dur.max():.1f}; "
          f"start spread {(t0.max() - start) / 100.0:.1f} us")
    print(f"   SIMDs used {len(uniq)}, waves/SIMD min {per_simd_n.min()} max {per_simd_n.max()}; trips/SIMD mean {per_simd_trips.mean():.0f} "
          f"max {per_simd_trips.max():.0f} (max/mean {per_simd_trips.max() / per_simd_trips.mean():.3f})")
    print(f"   SIMD end time us: mean {per_simd_end.mean():.1f} p5 {np.percentile(per_simd_end, 5):.1f} p50 {np.median(per_simd_end):.1f} "
          f"p95 {np.percentile(per_simd_end, 95):.1f} max {per_simd_end.max():.1f}")
    # busy fraction over time: number of waves alive in 10 slices
    edges = np.linspace(0, span, 11)
    alive = [(((t0 - start) / 100.0 < e1) & (end > e0)).sum() for e0, e1 in zip(edges[:-1], edges[1:])]
    print("   waves alive per decile:", alive)
    print(f"   staging: mean {stage_cyc.mean():.0f} shader cycles per wave = {stage_cyc.mean() / 2.1e3:.1f} us at 2.1 GHz = {stage_cyc.sum() / 2.1e3 / dur.sum() * 100:.1f} % of wave time")
    print(f"   corr(trips, duration) {np.corrcoef(trips, dur)[0, 1]:.3f}; ns per trip (sum dur / sum trips) {dur.sum() / trips.sum() * 1e3:.1f}")
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/wave_profile.npz", **out)
