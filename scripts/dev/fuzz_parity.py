"""Dev: randomised end-to-end parity run -- the HIP path through the C ABI against the CPU oracle on random image sizes (ragged
borders, tile counts on both sides of the binning policy's thresholds), Gaussian counts, splat sizes, regimes, use_sa, backgrounds,
SH degrees / coefficient counts, scale modifiers, the deterministic backward and upstream-gradient channel sets.  Per case: every check of tests/test_gpu_parity.py (bit-exact radii / offsets / keys / lists /
ranges / contributor counts on stable pixels, images within 1e-4, knife-edge pixels matched to one variant) in reference-binning
mode, the default footprint mode against the oracle's blend on ITS lists (images within 1e-4, lists an ordered subsequence), and
the gradients (1e-4 of each tensor's magnitude).  usage: fuzz_parity.py [cases=40] [seconds=420] [seed=2024] [only=<case>]; prints one line per
case and a summary; stops at the time limit."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gs2d_oracle as orc  # noqa: E402
from gaus_slam_amd import rasterizer  # noqa: E402
from tests import util  # noqa: E402
from tests.test_gpu_footprint import _assert_subsequence  # noqa: E402
from tests.test_gpu_parity import GRAD_TOL, IMG_TOL, KNIFE, _compare_forward  # noqa: E402

orc.set_threads(os.cpu_count() or 1)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
limit = float(sys.argv[2]) if len(sys.argv) > 2 else 420.0
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 2024)
only = int(sys.argv[4]) if len(sys.argv) > 4 else -1  # run this case alone (same parameters as in the full run)
GRADS = ["dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dtransMat", "dL_dscales", "dL_drotations", "dL_dmeans2D"]
t0 = time.time()
done = 0
worst_img, worst_grad = 0.0, 0.0
for it in range(n_cases):
    if time.time() - t0 > limit:
        break
    # image: a fifth of the cases large enough for > 1536 / > 2560 tiles (8192-pair binning workgroups need >= 1M pairs too)
    if it % 5 == 4:
        W, H = int(rng.integers(900, 1300)), int(rng.integers(600, 900))
        P = int(rng.choice([150000, 400000, 900000]))
    else:
        W, H = int(rng.integers(17, 700)), int(rng.integers(17, 500))
        P = int(rng.choice([1, 37, 600, 5000, 40000, 120000]))
    regime = ["tracking", "mapping"][int(rng.integers(2))]
    use_sa = bool(rng.integers(2))
    scale_hi = float(rng.choice([1.5, 4.0, 12.0, 30.0]))
    bg = tuple(float(x) for x in rng.uniform(0, 1, 3)) if rng.integers(2) else (0.0, 0.0, 0.0)
    cull_frac = float(rng.choice([0.0, 0.03, 0.3]))
    chans = [(0, 1, 2, 3, 4, 5, 6), (0, 1, 5, 6), (0, 1)][int(rng.integers(3))]
    sh_degree = int(rng.integers(0, 4)) if rng.integers(4) == 0 else -1  # a quarter of the cases: colours from SH
    sh_full = bool(rng.integers(2))
    scale_modifier = float(rng.choice([1.0, 1.0, 1.0, 0.6, 1.7]))
    det = rng.integers(5) == 0  # a fifth: the deterministic backward
    if only >= 0 and it != only:
        continue  # (every random draw of the case is above: the sequence stays the same)
    sc = util.make_scene(P, W, H, seed=1000 + it, regime=regime, scale_lo=0.3, scale_hi=scale_hi, cull_frac=cull_frac)
    kw = dict(use_sa=use_sa, bg=bg, scale_modifier=scale_modifier)
    if sh_degree >= 0:
        M = 16 if sh_full else (sh_degree + 1) ** 2
        shs = np.random.default_rng(5000 + it).normal(0, 0.35, (P, M, 3)).astype(np.float32)
        shs[:, 0] += 0.8
        shs[::7, 0] -= 2.5  # some colours clamp at 0
        kw.update(shs=shs, sh_degree=sh_degree)
    o = util.oracle_forward(orc, sc, **kw)
    rasterizer.set_deterministic(bool(det))  # (the forward sizes its binning chunk for the mode: set before it)
    try:
        h = util.hip_forward(sc, **kw)
    except Exception:
        rasterizer.set_deterministic(False)
        raise
    try:
        stable = _compare_forward(o, h, W, H, orc, knife_frac=0.05)  # (crowded random scenes: thousands of splats per pixel)
    except AssertionError:
        print(f"FAILED case {it}: {W}x{H} P={P} {regime} sa={int(use_sa)} scale_hi={scale_hi} bg={bg} cull_frac={cull_frac}")
        st = (o["stability"] > KNIFE).reshape(H, W)
        for name, a, b in (("color", h["color"], o["color"]), ("allmap", h["allmap"], o["allmap"])):
            d = np.abs(a - b) * st[None]
            c, y, x = np.unravel_index(int(np.argmax(d)), d.shape)
            print(f"  {name}: worst channel {c} at pixel ({x}, {y}): hip {a[c, y, x]!r} oracle {b[c, y, x]!r} diff {d[c, y, x]:.3e}; "
                  f"sa_amp {o['sa_amp'].reshape(H, W)[y, x]:.3e}; "
                  f"alpha there {o['allmap'][1, y, x]:.6f}, stability {o['stability'].reshape(H, W)[y, x]:.3e}, "
                  f"contributors {o['n_contrib'][:H * W].reshape(H, W)[y, x]}")
        raise
    img = max(float(np.abs(h["color"] - o["color"])[:, stable].max(initial=0.0)),
              float(util.allmap_dev(h, o, stable).max()))
    dc, da = util.make_upstream_grads(W, H, seed=it, channels=chans)
    dc = (dc * W * H).numpy(); da = (da * W * H).numpy()
    dc[:, ~stable] = 0; da[:, ~stable] = 0
    go = orc.backward(o, dc, da)
    try:
        gh = util.hip_backward(h, dc, da)
    finally:
        rasterizer.set_deterministic(False)
    gerr = 0.0
    for k in (GRADS if sh_degree < 0 else [g for g in GRADS if g != "dL_dcolors"] + ["dL_dsh"]):
        e = util.grad_err(gh[k], go[k].reshape(gh[k].shape))
        assert e <= GRAD_TOL, (it, k, e)
        gerr = max(gerr, e)
    # the library's default lists (footprint rectangles): ordered subsequences, same images
    hf = util.hip_forward(sc, binning="footprint", **kw)
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    _assert_subsequence(h, hf, tiles)
    dimg = max(float(np.abs(hf["color"] - o["color"])[:, stable].max(initial=0.0)),
               float(util.allmap_dev(hf, o, stable).max()))
    assert dimg <= IMG_TOL, (it, dimg)
    worst_img, worst_grad = max(worst_img, img, dimg), max(worst_grad, gerr)
    done += 1
    print(f"case {it:3d}: {W}x{H} ({tiles} tiles) P={P} {regime} sa={int(use_sa)} scale_hi={scale_hi} R={o['num_rendered']} "
          f"(footprint {hf['num_rendered']}) knife={int((~stable).sum())} img {max(img, dimg):.2e} grad {gerr:.2e} "
          f"channels={len(chans)} sh={sh_degree} smod={scale_modifier} det={int(det)}", flush=True)
print(f"{done} cases in {time.time() - t0:.0f} s: all checks passed; worst image deviation {worst_img:.2e} (limit {IMG_TOL}), "
      f"worst gradient deviation {worst_grad:.2e} of the tensor's magnitude (limit {GRAD_TOL}); knife-edge margin {KNIFE}")
