"""Dev: randomised robustness run of the footprint binning and the cull bits -- many seeds, image sizes, regimes and
distorted parameter distributions: forward outputs of the two binning modes must be bit-identical, the lists a subsequence,
and no cull bit may be missing against the brute-force per-pixel test (oracle/cull_exact.c)."""
import ctypes as C, os, subprocess, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import util
from tests.test_gpu_footprint import _assert_subsequence
from tests.test_gpu_cull import _bits
from oracle import gs2d_oracle as orc
orc.set_threads(os.cpu_count() or 1)
td = tempfile.mkdtemp(); so = os.path.join(td, "cull_exact.so")
subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", os.path.join(ROOT, "oracle", "cull_exact.c"), "-o", so, "-lm"])
exact = C.CDLL(so)
rng = np.random.default_rng(123)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
worst = 0.0
for it in range(n):
    W = int(rng.choice([96, 160, 333, 640, 1000])); H = int(rng.choice([64, 120, 250, 480, 700]))
    P = int(rng.integers(500, 40000)); regime = str(rng.choice(["mapping", "tracking"]))
    sc = util.make_scene(P, W, H, seed=1000 + it, regime=regime)
    g = torch.Generator().manual_seed(it)
    if it % 3 == 1:
        sc["scales"] = sc["scales"] * torch.exp(2.0 * torch.randn(P, 2, generator=g))     # anisotropic, tiny ... huge
    if it % 3 == 2:
        sc["opacities"] = torch.clamp(torch.rand(P, 1, generator=g) ** 5, 5e-4, 0.9999)
    o = util.oracle_forward(orc, sc)
    hr = util.hip_forward(sc, binning="reference"); ht = util.hip_forward(sc, binning="footprint")
    assert np.array_equal(hr["point_list"], o["point_list"]) and np.array_equal(hr["ranges"], o["ranges"])
    for k in ("color", "allmap", "final_T"):
        assert np.array_equal(ht[k].view(np.uint32), hr[k].view(np.uint32)), (it, k)
    _assert_subsequence(o, ht, o["ranges"].shape[0])
    groups, rows, eg, er = _bits(exact, o, hr, W, H)
    assert np.count_nonzero(eg & ~groups) == 0 and np.count_nonzero(er & ~rows) == 0, it
    err = float(np.abs(hr["color"] - o["color"]).max()); worst = max(worst, err)
    print(f"{it:2d} {W}x{H} P={P} {regime:8s} instances {o['num_rendered']} -> {ht['num_rendered']}  max |colour - oracle| {err:.2e}", flush=True)
print("all good; worst colour error vs oracle", worst)
