"""Round-2 GPU parity additions (through the C ABI): every SH degree with both coefficient layouts, the remaining
BASELINE.md shapes, the 3072-element step of the per-tile LDS sort, knife-edge pixels resolved against the oracle's
decision variants at the full bench size, and an operator-level check that culling granularity never changes a result."""
import os

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
IMG_TOL, GRAD_TOL, KNIFE = 1e-4, 1e-4, 2e-5


def _fwd_bwd_vs_oracle(oracle, sc, W, H, use_sa=True, channels=(0, 1, 2, 3, 4, 5, 6), grad_keys=None, knife_check=True, **kw):
    o = util.oracle_forward(oracle, sc, use_sa=use_sa, **kw)
    h = util.hip_forward(sc, use_sa=use_sa, **kw)
    assert h["num_rendered"] == o["num_rendered"]
    np.testing.assert_array_equal(h["radii"], o["radii"])
    np.testing.assert_array_equal(h["point_list"], o["point_list"])
    np.testing.assert_array_equal(h["ranges"], o["ranges"])
    stable = (o["stability"] > KNIFE).reshape(H, W)
    assert (~stable).mean() < 5e-3
    HW = H * W
    np.testing.assert_array_equal(h["last_contributor"][stable], o["n_contrib"][:HW].reshape(H, W)[stable])
    np.testing.assert_array_equal(h["median_contributor"][stable], o["n_contrib"][HW:].reshape(H, W)[stable])
    assert np.abs(h["color"] - o["color"])[:, stable].max() <= IMG_TOL
    assert util.allmap_dev(h, o, stable).max() <= IMG_TOL
    if knife_check:
        util.check_knife_pixels(oracle, o, h, stable, IMG_TOL, KNIFE)
    dc, da = util.make_upstream_grads(W, H, channels=channels)
    dc, da = (dc * W * H).numpy(), (da * W * H).numpy()
    dc[:, ~stable] = 0; da[:, ~stable] = 0
    go = oracle.backward(o, dc, da)
    gh = util.hip_backward(h, dc, da)
    for k in grad_keys or ["dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dtransMat", "dL_dmeans2D"]:
        assert util.grad_err(gh[k], go[k].reshape(gh[k].shape)) <= GRAD_TOL, k
    return o, h, go, gh


@pytest.mark.parametrize("degree", [0, 1, 2, 3])
@pytest.mark.parametrize("full_m", [True, False])
def test_sh_degrees_and_coefficient_counts(oracle, degree, full_m):
    """computeColorFromSH forward/backward (forward.cu:20-71, backward.cu:20-139) for every degree, with the coefficient
    array either the full 16 rows (M = 16: rows beyond (D+1)^2 are ignored and their gradient must be exactly zero) or
    exactly (D+1)^2 rows (M = 1 for degree 0)."""
    P, W, H = 1500, 128, 96
    sc = util.make_scene(P, W, H, seed=30 + degree, regime="mapping")
    M = 16 if full_m else (degree + 1) ** 2
    rng = np.random.default_rng(100 + degree)
    shs = rng.normal(0, 0.35, (P, M, 3)).astype(np.float32)
    shs[:, 0] += 0.8
    shs[::7, 0] -= 2.5  # some colours clamp at 0 (the `clamped` mask gates the backward)
    o, h, go, gh = _fwd_bwd_vs_oracle(oracle, sc, W, H, use_sa=True, shs=shs, sh_degree=degree,
                                      grad_keys=["dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh"])
    vis = o["radii"] > 0  # the clamp mask is only written for Gaussians that pass the culls (forward.cu:183-184, 231)
    np.testing.assert_array_equal(h["clamped"][vis], o["clamped"][vis])
    assert o["clamped"][vis].any()
    used = (degree + 1) ** 2
    assert gh["dL_dsh"].shape == (P, M, 3)
    if M > used:
        assert np.abs(gh["dL_dsh"][:, used:]).max() == 0.0  # backward.cu:20-139 never touches the unused rows
    assert np.abs(gh["dL_dsh"][:, :used]).max() > 0


def test_baseline_config1_640x480_200k(oracle):
    """BASELINE.json configs[1]: single frame fwd+bwd at 640x480 with ~200k surfels (the bench itself runs 500k)."""
    oracle.set_threads(os.cpu_count() or 1)
    sc = util.make_scene(200000, 640, 480, seed=8, regime="mapping")
    _fwd_bwd_vs_oracle(oracle, sc, 640, 480, use_sa=True, channels=(0, 1, 5, 6),
                       grad_keys=["dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations"])
    oracle.set_threads(1)


def test_scannetpp_reference_config_876x584_2M(oracle):
    """The shape the reference's ScanNet++ config renders (configs/scannetpp/config.py:15-16: 876x584 -> 55x37 = 2035
    tiles, 11 tile-id bits) with 2M surfels; BASELINE.json quotes 1168x876 for the same scene (tests/test_gpu_more.py)."""
    P, W, H = 2000000, 876, 584
    oracle.set_threads(os.cpu_count() or 1)
    sc = util.make_scene(P, W, H, seed=9, regime="mapping")
    o, h, _, _ = _fwd_bwd_vs_oracle(oracle, sc, W, H, use_sa=True, channels=(0, 1, 6),
                                    grad_keys=["dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations"])
    assert o["nbits"] == 32 + 11
    oracle.set_threads(1)


def test_lds_sort_capacity_step_3072(oracle):
    """The per-tile depth sort sizes its LDS segment as the smallest of 1536 / 2048 / 3072 / 4096 that is >= 1.3x the mean
    list length (gs2d_binning.hip launch_tile_depth_sort); earlier tests hit 1536, 2048 and the > 4096 global path.  This
    scene's mean list length selects 3072 -- since round 4 sorted inside the forward blend kernel by the INDEX sort (8 bytes of LDS
    per element, gs2d_tile_sort.h), its longest lists through global scratch."""
    W, H = 160, 128
    P = 66000
    sc = util.make_scene(P, W, H, seed=12, regime="mapping")
    sc["means3D"][::4, 2] = sc["means3D"][::4, 2].round(decimals=1)  # many exact depth ties: order must stay stable
    o = util.oracle_forward(oracle, sc, use_sa=True)
    tiles = o["ranges"].shape[0]
    want = o["num_rendered"] * 13 // (tiles * 10)
    assert 2048 < want <= 3072, want
    lens = o["ranges"][:, 1].astype(np.int64) - o["ranges"][:, 0]
    h = util.hip_forward(sc, use_sa=True)
    np.testing.assert_array_equal(h["keys"], o["keys"])
    np.testing.assert_array_equal(h["point_list"], o["point_list"])
    np.testing.assert_array_equal(h["ranges"], o["ranges"])
    stable = (o["stability"] > KNIFE).reshape(H, W)
    np.testing.assert_array_equal(h["last_contributor"][stable], o["n_contrib"][:H * W].reshape(H, W)[stable])
    assert np.abs(h["color"] - o["color"])[:, stable].max() <= IMG_TOL
    print(f"mean list {lens.mean():.0f}, max {lens.max()}, capacity step 3072 (want {want})")


def test_knife_edge_pixels_at_full_bench_size(oracle):
    """640x480 / 500k (the bench workload): every pixel excluded from the L-inf comparison as knife-edge equals the oracle
    under one outcome of its near-threshold decisions -- count and max error are printed (pytest -s)."""
    P, W, H = 500000, 640, 480
    oracle.set_threads(os.cpu_count() or 1)
    sc = util.make_scene(P, W, H, seed=0, regime="mapping")
    o = util.oracle_forward(oracle, sc, use_sa=True)
    h = util.hip_forward(sc, use_sa=True)
    stable = (o["stability"] > KNIFE).reshape(H, W)
    n, worst = util.check_knife_pixels(oracle, o, h, stable, IMG_TOL, KNIFE)
    assert n < 5e-3 * H * W and worst <= IMG_TOL
    oracle.set_threads(1)


def test_backward_runs_without_a_forward_on_the_same_stream_state(oracle):
    """The backward takes its sub-block queues from the cull pass (not from the forward blend): two backwards from one
    forward give identical queues, and a backward after an unrelated forward of another scene still matches the oracle."""
    W, H = 256, 192
    sc = util.make_scene(8000, W, H, seed=41, regime="mapping")
    o = util.oracle_forward(oracle, sc, use_sa=True)
    h = util.hip_forward(sc, use_sa=True)
    other = util.make_scene(3000, 128, 96, seed=42, regime="tracking")
    util.hip_forward(other, use_sa=False)  # unrelated work in between
    stable = (o["stability"] > KNIFE).reshape(H, W)
    dc, da = util.make_upstream_grads(W, H, channels=(0, 1, 5, 6))
    dc, da = (dc * W * H).numpy(), (da * W * H).numpy()
    dc[:, ~stable] = 0; da[:, ~stable] = 0
    go = oracle.backward(o, dc, da)
    g1 = util.hip_backward(h, dc, da)
    g2 = util.hip_backward(h, dc, da)
    for k in ["dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations"]:
        assert util.grad_err(g1[k], go[k].reshape(g1[k].shape)) <= GRAD_TOL, k
        assert util.grad_err(g2[k], g1[k]) <= 1e-5, k  # float atomics: equal up to summation order


@pytest.mark.parametrize("P,W,H,channels", [(8000, 256, 192, (0, 1, 2, 3, 4, 5, 6)), (150000, 640, 480, (0, 1, 5, 6))])
def test_deterministic_backward_is_bit_identical(oracle, P, W, H, channels):
    """Opt-in deterministic backward (gs2d_set_deterministic; SURVEY.md section 5: the reference's float atomics,
    backward.cu:441-460, make its gradients run-to-run non-deterministic): repeated backwards -- also from a second,
    independent forward -- give bit-identical gradients, which still match the oracle; the default (atomic) mode agrees
    with them to summation-order accuracy."""
    from gaus_slam_amd import rasterizer
    keys = ["dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dtransMat", "dL_dmeans2D"]
    sc = util.make_scene(P, W, H, seed=51, regime="mapping")
    sc["scales"][: P // 50] *= 0.02  # sub-pixel splats: the low-pass (mean2D) branch of the backward
    oracle.set_threads(os.cpu_count() or 1)
    o = util.oracle_forward(oracle, sc, use_sa=True)
    stable = (o["stability"] > KNIFE).reshape(H, W)
    dc, da = util.make_upstream_grads(W, H, channels=channels)
    dc, da = (dc * W * H).numpy(), (da * W * H).numpy()
    dc[:, ~stable] = 0; da[:, ~stable] = 0
    go = oracle.backward(o, dc, da)
    oracle.set_threads(1)
    assert not rasterizer.is_deterministic()
    h0 = util.hip_forward(sc, use_sa=True)
    g_atomic = util.hip_backward(h0, dc, da)
    rasterizer.set_deterministic(True)
    try:
        h1 = util.hip_forward(sc, use_sa=True)
        g1 = util.hip_backward(h1, dc, da)
        g2 = util.hip_backward(h1, dc, da)
        h2 = util.hip_forward(sc, use_sa=True)
        g3 = util.hip_backward(h2, dc, da)
    finally:
        rasterizer.set_deterministic(False)
    for k in keys:
        assert np.array_equal(g1[k].view(np.uint32), g2[k].view(np.uint32)), k
        assert np.array_equal(g1[k].view(np.uint32), g3[k].view(np.uint32)), k
        assert util.grad_err(g1[k], go[k].reshape(g1[k].shape)) <= GRAD_TOL, k
        assert util.grad_err(g_atomic[k], g1[k]) <= 1e-5, k
    assert np.abs(g1["dL_dmeans2D"]).max() > 0


def test_second_backward_on_the_same_forward_is_clean():
    """The first backward relies on the forward's cull kernel having cleared the gradient accumulator; a second backward on
    the same forward state (retain_graph) must clear it itself: both give the same gradients."""
    P, W, H = 20000, 320, 240
    sc = util.make_scene(P, W, H, seed=4, regime="mapping")
    dc, da = util.make_upstream_grads(W, H, channels=(0, 1, 2, 3, 4, 5, 6))
    dc = (dc * W * H).numpy(); da = (da * W * H).numpy()
    h = util.hip_forward(sc, binning="footprint")
    g1 = util.hip_backward(h, dc, da)
    g2 = util.hip_backward(h, dc, da)
    g3 = util.hip_backward(h, 2 * dc, 2 * da)
    for k in ["dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dmeans2D"]:
        assert np.abs(g1[k]).max() > 0
        assert util.grad_err(g2[k], g1[k]) <= 1e-5, k
        assert util.grad_err(g3[k], 2 * g1[k]) <= 1e-5, k


@pytest.mark.parametrize("case", range(10))
def test_default_and_deterministic_backward_agree_on_random_scenes(case):
    """The default backward (plain-store / LDS-atomic accumulate per trip, global float atomics per batch) against the
    atomic-free deterministic one on random image sizes (ragged edge tiles), scene sizes, splat scales, regimes, both
    distortion modes, with and without gradients on the normal channels: equal up to summation order."""
    from gaus_slam_amd import rasterizer, render as gs_render
    from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(1000 + case)
    W = int(rng.integers(3, 40)) * 16 - int(rng.integers(0, 16))
    H = int(rng.integers(3, 30)) * 16 - int(rng.integers(0, 16))
    P = int(rng.choice([50, 700, 5000, 40000]))
    sc = make_scene(P, W, H, seed=200 + case, regime=["mapping", "tracking"][case % 2], scale_lo=0.3,
                    scale_hi=float(rng.choice([4.0, 12.0, 40.0])))
    chans = (0, 1, 5, 6) if case % 3 else (0, 1, 2, 3, 4, 5, 6)
    dc, da = make_upstream_grads(W, H, seed=case, channels=chans)
    dc, da = (dc * W * H).to(dev), (da * W * H).to(dev)
    st = gs_render.settings_from_camera(sc["cam"], dev, use_sa=bool(case % 4))
    names = ("means3D", "opacities", "scales", "rotations", "colors")
    res = []
    try:
        for det in (False, True):
            rasterizer.set_deterministic(det)
            p = {k: sc[k].to(dev).requires_grad_(True) for k in names}
            m2 = torch.zeros_like(p["means3D"], requires_grad=True)
            pkg = gs_render.render(st, p["means3D"], m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"],
                                   rotations=p["rotations"])
            torch.autograd.backward([pkg["render_color"], pkg["allmap"]], [dc, da])
            res.append([p[k].grad.double().cpu() for k in names] + [m2.grad.double().cpu()])
    finally:
        rasterizer.set_deterministic(False)
    for name, a, b in zip(names + ("means2D",), *res):
        assert torch.isfinite(a).all() and torch.isfinite(b).all(), name
        scale = float(b.abs().max())
        if scale > 0:
            assert float((a - b).abs().max()) <= 1e-5 * scale, (name, W, H, P)
