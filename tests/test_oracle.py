"""CPU tests that pin the oracle itself (no GPU): closed-form micro-cases, structural invariants, the independent
pure-PyTorch restatement, autograd of that restatement for the hand-written backward, and the committed golden
vectors.  The reference has no tests or golden vectors of its own, so these are the pins the
header of oracle/gs2d_oracle.c refers to."""
import glob
import math
import os

import numpy as np
import pytest
import torch

from gaus_slam_amd.scene_synth import intrinsics_for, setup_camera
from tests import util

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def _single_surfel_scene(W, H, z, opacities, scale_px=6.0, colors=None):
    """Fronto-parallel surfels stacked on the optical axis through the centre pixel."""
    K = intrinsics_for(W, H)
    # the reference's ndc->pixel map (auxiliary.h:61-64) puts pixel i's centre at i + 0.5 in K's coordinates
    K[0, 2] = (W - 1) / 2.0 + 0.5
    K[1, 2] = (H - 1) / 2.0 + 0.5
    cam = setup_camera(W, H, K, torch.eye(4))
    f = float(K[0, 0])
    n = len(z)
    z = np.asarray(z, np.float32)
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    means = np.stack([np.zeros(n), np.zeros(n), z], 1).astype(np.float32)
    scales = np.stack([z / f * scale_px, z / f * scale_px], 1).astype(np.float32)
    rots = np.tile(np.array([[1.0, 0, 0, 0]], np.float32), (n, 1))  # normal = +z, flipped towards the camera
    colors = np.asarray(colors if colors is not None else np.tile([[1.0, 0.5, 0.25]], (n, 1)), np.float32)
    sc = dict(means3D=torch.from_numpy(means), scales=torch.from_numpy(scales), rotations=torch.from_numpy(rots),
              opacities=torch.tensor(opacities, dtype=torch.float32)[:, None], colors=torch.from_numpy(colors), cam=cam)
    return sc


def test_single_surfel_closed_form(oracle):
    W, H = 33, 33  # centre pixel (16,16) exactly on the optical axis
    sc = _single_surfel_scene(W, H, [2.0], [0.6])
    st = util.oracle_forward(oracle, sc, use_sa=False)
    c = (16, 16)
    assert st["allmap"][1][c] == pytest.approx(0.6, abs=1e-6)          # alpha = opacity at the centre
    assert st["allmap"][0][c] == pytest.approx(0.6 * 2.0, abs=1e-5)    # depth accumulates alpha*z
    assert st["allmap"][5][c] == pytest.approx(2.0, abs=1e-5)          # median depth
    np.testing.assert_allclose(st["allmap"][2:5, 16, 16], [0, 0, -0.6], atol=1e-6)  # normal faces the camera
    np.testing.assert_allclose(st["color"][:, 16, 16], 0.6 * np.array([1.0, 0.5, 0.25]), atol=1e-6)
    # gaussian falloff along a row: alpha = o * exp(-0.5 * (dx/scale_px)^2) for a fronto-parallel surfel
    dx = 3
    assert st["allmap"][1][16, 16 + dx] == pytest.approx(0.6 * math.exp(-0.5 * (dx / 6.0) ** 2), rel=1e-4)


def test_two_stacked_surfels_closed_form(oracle):
    W, H = 33, 33
    o1, o2, z1, z2 = 0.5, 0.8, 1.5, 3.0
    sc = _single_surfel_scene(W, H, [z2, z1], [o2, o1], colors=[[0, 1, 0], [1, 0, 0]])  # given back-to-front
    st = util.oracle_forward(oracle, sc, use_sa=False, bg=(0.0, 0.0, 1.0))
    T = (1 - o1) * (1 - o2)
    np.testing.assert_allclose(st["color"][:, 16, 16], [o1, (1 - o1) * o2, T], atol=1e-6)
    assert st["allmap"][1][16, 16] == pytest.approx(1 - T, abs=1e-6)
    assert st["allmap"][0][16, 16] == pytest.approx(o1 * z1 + (1 - o1) * o2 * z2, abs=1e-5)
    assert st["n_contrib"][16 * W + 16] == 2
    # depth order, not submission order
    tile0 = st["point_list"][st["ranges"][1 * 3 + 1][0]:st["ranges"][1 * 3 + 1][1]]
    assert list(tile0) == [1, 0]


@pytest.mark.parametrize("regime", ["tracking", "mapping"])
def test_structural_invariants(oracle, regime):
    W, H, P = 200, 136, 3000
    sc = util.make_scene(P, W, H, seed=11, regime=regime)
    st = util.oracle_forward(oracle, sc, use_sa=True)
    R = st["num_rendered"]
    assert st["tiles_touched"].sum() == R == st["point_offsets"][-1]
    assert (st["tiles_touched"][st["radii"] == 0] == 0).all()
    mask = (1 << st["nbits"]) - 1
    k = st["keys"] & np.uint64(mask)
    assert (k[1:] >= k[:-1]).all()
    gx, gy = (W + 15) // 16, (H + 15) // 16
    lens = st["ranges"][:, 1].astype(np.int64) - st["ranges"][:, 0]
    assert lens.sum() == R and (lens >= 0).all()
    tiles = (st["keys"] >> np.uint64(32)).astype(np.int64)
    for t in np.unique(tiles):
        a, b = st["ranges"][t]
        assert (tiles[a:b] == t).all()
    alpha = st["allmap"][1]
    assert (alpha >= 0).all() and (alpha < 1).all()
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    tile_of_pix = (ys // 16) * gx + xs // 16
    assert (st["n_contrib"][:H * W].reshape(H, W) <= lens[tile_of_pix]).all()
    # culled Gaussians: 3% were placed behind the near plane / far off-screen
    assert (st["radii"] == 0).sum() >= int(0.02 * P)


@pytest.mark.parametrize("regime", ["tracking", "mapping"])
@pytest.mark.parametrize("use_sa", [True, False])
def test_c_oracle_matches_pure_pytorch_forward(oracle, regime, use_sa):
    """BASELINE.json configs[0]: 160x120 / 256 Gaussians via the pure-PyTorch CPU path."""
    from oracle import torch_ref
    W, H, P = 160, 120, 256
    sc = util.make_scene(P, W, H, seed=0, regime=regime)
    cam = sc["cam"]
    st = util.oracle_forward(oracle, sc, use_sa=use_sa)
    dt = torch.float64
    with torch.no_grad():
        r = torch_ref.render(sc["means3D"].to(dt), sc["scales"].to(dt), sc["rotations"].to(dt), sc["opacities"].to(dt),
                             sc["colors"].to(dt), cam.viewmatrix.to(dt), cam.projmatrix.to(dt), W, H, use_sa=use_sa)
    np.testing.assert_array_equal(r["radii"].numpy(), st["radii"])
    np.testing.assert_array_equal(r["point_list"].numpy(), st["point_list"])
    np.testing.assert_array_equal(r["ranges"].numpy(), st["ranges"])
    np.testing.assert_array_equal(r["n_contrib"].numpy().reshape(-1), st["n_contrib"])
    assert np.abs(r["color"].numpy() - st["color"]).max() < 1e-4
    assert np.abs(r["allmap"].numpy() - st["allmap"]).max() < 3e-4  # float32 oracle vs float64 restatement


@pytest.mark.parametrize("regime", ["tracking", "mapping"])
def test_backward_matches_autograd_where_exact(oracle, regime):
    """use_sa=False, unit quaternions, ray-splat branch: the reference backward is the exact gradient there
    (SURVEY.md 'Hard parts'), so autograd of the independent PyTorch forward must reproduce the oracle's
    hand-written backward."""
    from oracle import torch_ref
    W, H, P = 96, 64, 120
    sc = util.make_scene(P, W, H, seed=3, regime=regime)
    cam = sc["cam"]
    bg = np.array([0.3, 0.1, 0.7], np.float32)
    st = util.oracle_forward(oracle, sc, use_sa=False, bg=bg)
    dc, da = util.make_upstream_grads(W, H, channels=(0, 1, 2, 3, 4, 5, 6))
    dc, da = dc * W * H, da * W * H
    g = oracle.backward(st, dc.numpy(), da.numpy())
    assert np.abs(g["dL_dmeans2D_blend"]).max() == 0.0  # no low-pass contribution in this scene
    dt = torch.float64
    leaves = {k: sc[k].to(dt).clone().requires_grad_(True) for k in ["means3D", "scales", "rotations", "opacities", "colors"]}
    r = torch_ref.render(leaves["means3D"], leaves["scales"], leaves["rotations"], leaves["opacities"], leaves["colors"],
                         cam.viewmatrix.to(dt), cam.projmatrix.to(dt), W, H, use_sa=False, bg=torch.from_numpy(bg))
    ((r["color"] * dc.to(dt)).sum() + (r["allmap"] * da.to(dt)).sum()).backward()
    for k, gk in [("means3D", "dL_dmeans3D"), ("scales", "dL_dscales"), ("rotations", "dL_drotations"),
                  ("opacities", "dL_dopacity"), ("colors", "dL_dcolors")]:
        a = leaves[k].grad.numpy()
        assert util.grad_err(g[gk].reshape(a.shape), a) < 2e-5, k


def test_mean2d_densification_hack(oracle):
    """backward.cu:660-663: dL_dmeans2D.xy is overwritten with dL_dT[0].z*depth*W/2, dL_dT[1].z*depth*H/2."""
    W, H, P = 96, 64, 120
    sc = util.make_scene(P, W, H, seed=3, regime="tracking")
    st = util.oracle_forward(oracle, sc, use_sa=True)
    dc, da = util.make_upstream_grads(W, H)
    g = oracle.backward(st, dc.numpy(), da.numpy())
    vis = st["radii"] > 0
    depth = st["transMats"][:, 8]
    np.testing.assert_allclose(g["dL_dmeans2D"][vis, 0], (g["dL_dtransMat_blend"][vis, 2] * depth[vis]) * 0.5 * W, rtol=1e-6)
    np.testing.assert_allclose(g["dL_dmeans2D"][vis, 1], (g["dL_dtransMat_blend"][vis, 5] * depth[vis]) * 0.5 * H, rtol=1e-6)
    assert (g["dL_dmeans2D"][:, 2] == 0).all()
    for k in ("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dopacity", "dL_dcolors"):
        assert (g[k][~vis] == 0).all()  # dense zero grads for invisible Gaussians (rasterize_points.cu:192-200)


def _run_golden(oracle, d):
    sc_kw = {}
    kind = str(d["kind"])
    if kind == "sh3":
        sc_kw.update(shs=d["shs"], sh_degree=3)
    else:
        sc_kw.update(colors_precomp=d["colors"])
    if kind == "precomp":
        sc_kw.update(transMat_precomp=d["transMat_precomp"])
    else:
        sc_kw.update(scales=d["scales"], rotations=d["rotations"])
    st = oracle.forward(d["means3D"], d["opacities"], d["viewmatrix"], d["projmatrix"], d["campos"], int(d["W"]),
                        int(d["H"]), float(d["tanfovx"]), float(d["tanfovy"]), bg=d["bg"], use_sa=bool(d["use_sa"]),
                        **sc_kw)
    g = oracle.backward(st, d["dL_dcolor"], d["dL_dallmap"])
    return st, g


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(oracle, path):
    d = np.load(path)
    oracle.set_threads(1)
    st, g = _run_golden(oracle, d)
    for k in ("radii", "point_list", "ranges", "n_contrib"):
        np.testing.assert_array_equal(st[k], d[k])
    assert st["num_rendered"] == int(d["num_rendered"])
    # the fixture was produced by this code with this compiler: identical up to libm/FMA codegen differences
    np.testing.assert_allclose(st["color"], d["color"], atol=2e-6)
    np.testing.assert_allclose(st["allmap"], d["allmap"], atol=2e-5)
    for k in ("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dopacity", "dL_dcolors", "dL_dtransMat", "dL_dsh"):
        assert util.grad_err(g[k], d[k]) < 1e-5, k


def test_sh_color_and_clamp(oracle):
    """forward.cu:20-71: degree-0 SH gives SH_C0*c + 0.5, negative results clamp to 0 and kill the gradient."""
    W, H = 33, 33
    sc = _single_surfel_scene(W, H, [2.0], [0.9])
    shs = np.zeros((1, 1, 3), np.float32)
    shs[0, 0] = [1.0, -3.0, 0.0]
    st = util.oracle_forward(oracle, sc, use_sa=False, shs=shs, sh_degree=0)
    exp = np.maximum(0.28209479177387814 * shs[0, 0] + 0.5, 0)
    np.testing.assert_allclose(st["rgb"][0], exp, atol=1e-6)
    assert list(st["clamped"][0]) == [0, 1, 0]
    g = oracle.backward(st, np.ones((3, H, W), np.float32), np.zeros((7, H, W), np.float32))
    assert g["dL_dsh"][0, 0, 1] == 0 and g["dL_dsh"][0, 0, 0] > 0


def test_knn_oracle_matches_kdtree(oracle):
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(2000, 3)).astype(np.float32)
    d = oracle.dist2_knn3(pts)
    dd, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    np.testing.assert_allclose(d, (dd[:, 1:] ** 2).mean(1), rtol=1e-4)


def test_higher_msb(oracle):
    # rasterizer_impl.cu:35-50: bits needed for the tile id (7/11/12/12/11 for configs A/B/R/S)
    assert [oracle.higher_msb(n) for n in (80, 1200, 3225, 4015, 2035)] == [7, 11, 12, 12, 11]


def test_batched_pure_pytorch_render_matches_c_oracle(oracle):
    """oracle/torch_batched.py (the pure-PyTorch CPU baseline of bench.py, all tiles in lock-step) restates the same
    forward as the C oracle: indices bit-exact, images to float32 conditioning on stable pixels, and it is differentiable."""
    import torch
    from oracle import torch_batched
    from tests import util
    for P, W, H, use_sa in ((256, 160, 120, True), (1500, 150, 100, False)):
        sc = util.make_scene(P, W, H, seed=3, regime="mapping")
        cam = sc["cam"]
        bg = (0.2, 0.5, 0.1)
        o = util.oracle_forward(oracle, sc, use_sa=use_sa, bg=bg)
        leaves = {k: sc[k].clone().requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "colors")}
        r = torch_batched.render(leaves["means3D"], leaves["scales"], leaves["rotations"], leaves["opacities"], leaves["colors"],
                                 cam.viewmatrix, cam.projmatrix, W, H, bg=torch.tensor(bg), use_sa=use_sa)
        assert np.array_equal(r["point_list"].numpy(), o["point_list"]) and np.array_equal(r["ranges"].numpy(), o["ranges"])
        np.testing.assert_array_equal(r["radii"].numpy(), o["radii"])
        stable = (o["stability"] > 2e-5).reshape(H, W)
        np.testing.assert_array_equal(r["n_contrib"][0].numpy()[stable], o["n_contrib"][:H * W].reshape(H, W)[stable])
        assert np.abs(r["color"].detach().numpy() - o["color"])[:, stable].max() <= 1e-4
        assert np.abs(r["allmap"].detach().numpy() - o["allmap"])[:, stable].max() <= 1e-3
        (r["color"].sum() + r["allmap"][:2].sum()).backward()
        assert all(t.grad is not None and torch.isfinite(t.grad).all() for t in leaves.values())


def test_pixel_override_backward_reproduces_the_plain_backward():
    """oracle.backward(pixel_overrides=...) (orc_blend_bwd_pixel: the backward of one pixel on RECORDED forward outcomes)
    with no decision flipped must give what the plain backward gives -- for every pixel of a small image, both distortion
    modes, all ten upstream channels."""
    from oracle import gs2d_oracle as orc
    from tests import util
    W, H, P = 64, 48, 600
    sc = util.make_scene(P, W, H, seed=9, regime="mapping")
    for use_sa in (True, False):
        o = util.oracle_forward(orc, sc, use_sa=use_sa, bg=(0.3, 0.1, 0.6))
        dc, da = util.make_upstream_grads(W, H, channels=(0, 1, 2, 3, 4, 5, 6))
        dc, da = (dc * W * H).numpy(), (da * W * H).numpy()
        g0 = orc.backward(o, dc, da)
        every = [(x, y, 0) for y in range(H) for x in range(W)]
        g1 = orc.backward(o, dc, da, pixel_overrides=every, knife=0.0)
        for k in ("dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dtransMat", "dL_dscales", "dL_drotations", "dL_dmeans2D", "dL_dnormal"):
            assert np.abs(g0[k]).max() > 0, k
            assert util.grad_err(g1[k], g0[k]) <= 1e-6, (use_sa, k)
        # a flipped decision changes the result (the override is not a no-op): pick a pixel with a near-threshold decision
        ys, xs = np.nonzero((o["stability"] < 5e-2).reshape(H, W))
        if len(ys):
            x, y = int(xs[0]), int(ys[0])
            nk, variants = orc.pixel_variants(o, x, y, 5e-2)
            assert nk >= 1 and len(variants) >= 2 and variants[1]["mask"] == 1
            g2 = orc.backward(o, dc, da, pixel_overrides=[(x, y, 1)], knife=5e-2)
            assert any(util.grad_err(g2[k], g0[k]) > 1e-7 for k in ("dL_dcolors", "dL_dopacity", "dL_dtransMat"))


@pytest.mark.parametrize("use_sa", [True, False])
def test_float64_backward_is_the_float32_oracle_up_to_rounding(oracle, use_sa):
    """oracle.backward_f64 (gs2d_oracle_f64.c: the backward in double on the float32 paths' inputs and decisions) is the
    yardstick of the GPU rounding-error tests, so it is pinned here twice: (1) against the float32 oracle -- separately written
    source; a formula slip in either would show as an O(1) difference, rounding shows as ~1e-7 of a tensor's maximum;
    (2) where the reference backward is the exact gradient (use_sa=False, ray-splat branch) against float64 autograd of the
    independent PyTorch forward, much tighter than the float32 oracle can be held to."""
    W, H, P = 96, 64, 150
    sc = util.make_scene(P, W, H, seed=4, regime="mapping")
    sc["scales"] = sc["scales"].clone()
    sc["scales"][::7] *= 0.02  # some splats live on the low-pass branch (backward.cu:450-457, 538-563)
    bg = np.array([0.3, 0.1, 0.7], np.float32)
    st = util.oracle_forward(oracle, sc, use_sa=use_sa, bg=bg)
    dc, da = util.make_upstream_grads(W, H, channels=(0, 1, 2, 3, 4, 5, 6))
    dc, da = (dc * W * H).numpy(), (da * W * H).numpy()
    stable = (st["stability"] > 2e-5).reshape(H, W)
    dc[:, ~stable] = 0; da[:, ~stable] = 0
    g32, g64 = oracle.backward(st, dc, da), oracle.backward_f64(st, dc, da)
    assert np.abs(g64["dL_dmeans2D_blend"]).max() > 0
    for k in ("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dopacity", "dL_dcolors", "dL_dnormal", "dL_dtransMat_blend",
              "dL_dmeans2D_blend"):
        assert g64[k].dtype == np.float64 and np.abs(g64[k]).max() > 0, k
        assert util.grad_err(g32[k], g64[k].reshape(g32[k].shape)) <= 2e-6, (k, util.grad_err(g32[k], g64[k].reshape(g32[k].shape)))


def test_float64_backward_matches_float64_autograd_where_exact(oracle):
    from oracle import torch_ref
    W, H, P = 96, 64, 120
    sc = util.make_scene(P, W, H, seed=3, regime="mapping")
    cam = sc["cam"]
    bg = np.array([0.3, 0.1, 0.7], np.float32)
    st = util.oracle_forward(oracle, sc, use_sa=False, bg=bg)
    dc, da = util.make_upstream_grads(W, H, channels=(0, 1, 2, 3, 4, 5, 6))
    dc, da = dc * W * H, da * W * H
    g = oracle.backward_f64(st, dc.numpy(), da.numpy())
    dt = torch.float64
    leaves = {k: sc[k].to(dt).clone().requires_grad_(True) for k in ["means3D", "scales", "rotations", "opacities", "colors"]}
    r = torch_ref.render(leaves["means3D"], leaves["scales"], leaves["rotations"], leaves["opacities"], leaves["colors"],
                         cam.viewmatrix.to(dt), cam.projmatrix.to(dt), W, H, use_sa=False, bg=torch.from_numpy(bg))
    ((r["color"] * dc.to(dt)).sum() + (r["allmap"] * da.to(dt)).sum()).backward()
    for k, gk in [("means3D", "dL_dmeans3D"), ("scales", "dL_dscales"), ("rotations", "dL_drotations"),
                  ("opacities", "dL_dopacity"), ("colors", "dL_dcolors")]:
        a = leaves[k].grad.numpy()
        # what is left between the two float64 evaluations (2-6e-6, the same as float32 oracle vs autograd): the forward state
        # (T_final, M1, M2 ...) reaches the backward rounded to float32 and the splat records are float32 -- inputs of the
        # function, not rounding inside it (float32 oracle vs backward_f64 on the SAME inputs: 1.5-2.7e-7)
        assert util.grad_err(g[gk].reshape(a.shape), a) < 1e-5, (k, util.grad_err(g[gk].reshape(a.shape), a))
