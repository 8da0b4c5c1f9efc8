"""More GPU parity tests (through the C ABI / operator mirror): committed golden vectors incl. the SH and
precomputed-transMat paths, edge cases, BASELINE.json's full bench size, a culling stress scene, simple_knn,
mark_visible and the autograd operator surface."""
import glob
import os

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
IMG_TOL, GRAD_TOL, KNIFE = 1e-4, 1e-4, 2e-5


def _scene_from_golden(d):
    from gaus_slam_amd.scene_synth import Camera
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    cam = Camera(W=int(d["W"]), H=int(d["H"]), tanfovx=float(d["tanfovx"]), tanfovy=float(d["tanfovy"]),
                 viewmatrix=t(d["viewmatrix"]), projmatrix=t(d["projmatrix"]), campos=t(d["campos"]), K=None, w2c=None)
    return dict(means3D=t(d["means3D"]), scales=t(d["scales"]), rotations=t(d["rotations"]), opacities=t(d["opacities"]),
                colors=t(d["colors"]), cam=cam)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_vectors(path):
    d = np.load(path)
    sc = _scene_from_golden(d)
    kind = str(d["kind"])
    kw = {}
    if kind == "sh3":
        kw.update(shs=d["shs"], sh_degree=3)
    if kind == "precomp":
        kw.update(transMat_precomp=d["transMat_precomp"])
    h = util.hip_forward(sc, use_sa=bool(d["use_sa"]), bg=d["bg"], **kw)
    assert h["num_rendered"] == int(d["num_rendered"])
    np.testing.assert_array_equal(h["radii"], d["radii"])
    np.testing.assert_array_equal(h["point_list"], d["point_list"])
    np.testing.assert_array_equal(h["ranges"], d["ranges"])
    H, W = int(d["H"]), int(d["W"])
    assert d["stability"].min() > KNIFE  # the fixtures contain no knife-edge pixel: compare everything
    np.testing.assert_array_equal(h["last_contributor"].reshape(-1), d["n_contrib"][:H * W])
    np.testing.assert_array_equal(h["median_contributor"].reshape(-1), d["n_contrib"][H * W:])
    assert np.abs(h["color"] - d["color"]).max() <= IMG_TOL
    assert np.abs(h["allmap"] - d["allmap"]).max() <= IMG_TOL
    g = util.hip_backward(h, d["dL_dcolor"], d["dL_dallmap"])
    keys = ["dL_dcolors", "dL_dopacity", "dL_dtransMat", "dL_dmeans2D"]
    keys += ["dL_dmeans3D"] if kind != "precomp" or True else []
    if kind != "precomp":
        keys += ["dL_dscales", "dL_drotations"]
    if kind == "sh3":
        keys += ["dL_dsh"]
    for k in keys:
        assert util.grad_err(g[k], d[k].reshape(g[k].shape)) <= GRAD_TOL, k


def test_empty_and_fully_culled(oracle):
    from gaus_slam_amd import rasterizer
    dev = torch.device("cuda")
    e = torch.empty(0, device=dev)
    sc = util.make_scene(64, 64, 48, seed=1, regime="tracking")
    cam = sc["cam"]
    z3, z1 = torch.zeros(0, 3, device=dev), torch.zeros(0, 1, device=dev)
    R, color, allmap, radii, *_ = rasterizer.rasterize_gaussians(
        torch.tensor([0.1, 0.2, 0.3], device=dev), z3, z3, z1, torch.zeros(0, 2, device=dev), torch.zeros(0, 4, device=dev),
        1.0, e, cam.viewmatrix.to(dev), cam.projmatrix.to(dev), cam.tanfovx, cam.tanfovy, 48, 64, e, 0,
        cam.campos.to(dev), True, False, False)
    assert R == 0 and radii.numel() == 0 and float(color.abs().max()) == 0 and float(allmap.abs().max()) == 0  # rasterize_points.cu:100
    # every Gaussian behind the camera: R = 0, image = background
    sc["means3D"][:, 2] = -sc["means3D"][:, 2].abs() - 1.0
    bg = (0.1, 0.2, 0.3)
    h = util.hip_forward(sc, bg=bg)
    o = util.oracle_forward(oracle, sc, bg=bg)
    assert h["num_rendered"] == 0 == o["num_rendered"] and (h["radii"] == 0).all()
    np.testing.assert_allclose(h["color"], o["color"])
    np.testing.assert_allclose(h["color"][:, 5, 7], bg)
    g = util.hip_backward(h, np.ones((3, 48, 64), np.float32), np.ones((7, 48, 64), np.float32))
    assert all(np.abs(v).max() == 0 for v in g.values() if v.size)


def test_debug_flag_and_scale_modifier(oracle):
    sc = util.make_scene(500, 96, 80, seed=5, regime="mapping")
    o = util.oracle_forward(oracle, sc, scale_modifier=1.7)
    h = util.hip_forward(sc, scale_modifier=1.7, debug=True)
    np.testing.assert_array_equal(h["point_list"], o["point_list"])
    stable = (o["stability"] > KNIFE).reshape(80, 96)
    assert util.allmap_dev(h, o, stable).max() <= IMG_TOL


def test_backward_with_a_scale_modifier(oracle):
    """scale_modifier != 1: the backward ignores the modifier when it rebuilds the transform (backward.cu:504) but its
    means2D output uses the FORWARD's Tw.z (backward.cu:660-663) -- the one case in which the library's preprocess backward
    still reads the geometry records instead of recomputing that word."""
    W, H = 128, 96
    sc = util.make_scene(800, W, H, seed=7, regime="mapping")
    o = util.oracle_forward(oracle, sc, scale_modifier=1.7)
    h = util.hip_forward(sc, scale_modifier=1.7)
    np.testing.assert_array_equal(h["point_list"], o["point_list"])
    stable = (o["stability"] > KNIFE).reshape(H, W)
    dc, da = util.make_upstream_grads(W, H, channels=(0, 1, 5, 6))
    dc = (dc * W * H).numpy(); da = (da * W * H).numpy()
    dc[:, ~stable] = 0; da[:, ~stable] = 0
    go = oracle.backward(o, dc, da)
    gh = util.hip_backward(h, dc, da)
    for k in ["dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dmeans2D"]:
        assert util.grad_err(gh[k], go[k].reshape(gh[k].shape)) <= 1e-4, k
    assert np.abs(gh["dL_dmeans2D"]).max() > 0


def test_culling_stress_extreme_splats(oracle):
    """Grazing, huge, tiny, near-plane and behind-the-eye-crossing surfels: the quadrant cull must never drop a
    contributing pair (bit-exact n_contrib / images vs the oracle, which has no culling)."""
    P, W, H = 3000, 208, 160
    sc = util.make_scene(P, W, H, seed=21, regime="mapping", max_tilt_deg=89.9, scale_lo=0.05, scale_hi=60.0)
    rng = np.random.default_rng(3)
    sc["means3D"][: P // 10, 2] = torch.from_numpy(rng.uniform(0.2, 0.35, P // 10).astype(np.float32))  # hugging the near plane
    sc["opacities"][P // 10: P // 5] = 1.0  # opacity above the 0.99 alpha clamp
    sc["opacities"][P // 5: P // 4] = 0.004  # around the 1/255 visibility threshold
    sc["scales"][P // 4: P // 3] *= 30.0  # discs that cross the eye plane
    o = util.oracle_forward(oracle, sc, use_sa=True)
    h = util.hip_forward(sc, use_sa=True)
    np.testing.assert_array_equal(h["point_list"], o["point_list"])
    stable = (o["stability"] > KNIFE).reshape(H, W)
    assert (~stable).mean() < 0.01
    np.testing.assert_array_equal(h["last_contributor"][stable], o["n_contrib"][:H * W].reshape(H, W)[stable])
    assert np.abs(h["color"] - o["color"])[:, stable].max() <= IMG_TOL
    # depth-like channels scale with the (unbounded) depths of this scene: compare relative to magnitude
    for c in range(7):
        scale = max(1.0, float(np.abs(o["allmap"][c]).max()))
        assert np.abs(h["allmap"][c] - o["allmap"][c])[stable].max() <= IMG_TOL * scale, c
    dc, da = util.make_upstream_grads(W, H, channels=(0, 1, 2, 3, 4, 5, 6))
    dc, da = (dc * W * H).numpy(), (da * W * H).numpy()
    dc[:, ~stable] = 0; da[:, ~stable] = 0
    go = oracle.backward(o, dc, da)
    gh = util.hip_backward(h, dc, da)
    for k in ["dL_dcolors", "dL_dopacity", "dL_dtransMat", "dL_dmeans3D", "dL_dscales", "dL_drotations"]:
        assert util.grad_err(gh[k], go[k].reshape(gh[k].shape)) <= 5e-4, k


@pytest.mark.parametrize("regime", ["mapping"])
def test_full_bench_size_500k(oracle, regime):
    """BASELINE.json metric size: 640x480, 500k Gaussians, checked against the oracle (multi-threaded) plus
    size-independent properties."""
    P, W, H = 500000, 640, 480
    sc = util.make_scene(P, W, H, seed=0, regime=regime)
    oracle.set_threads(os.cpu_count() or 1)
    o = util.oracle_forward(oracle, sc, use_sa=True)
    h = util.hip_forward(sc, use_sa=True)
    assert h["num_rendered"] == o["num_rendered"]
    np.testing.assert_array_equal(h["radii"], o["radii"])
    np.testing.assert_array_equal(h["point_list"], o["point_list"])
    np.testing.assert_array_equal(h["ranges"], o["ranges"])
    # properties: keys sorted, ranges partition [0,R), alpha in [0,1)
    mask = np.uint64((1 << o["nbits"]) - 1)
    k = h["keys"] & mask
    assert (k[1:] >= k[:-1]).all()
    lens = h["ranges"][:, 1].astype(np.int64) - h["ranges"][:, 0]
    assert lens.sum() == h["num_rendered"]
    assert (h["allmap"][1] >= 0).all() and (h["allmap"][1] < 1).all()
    stable = (o["stability"] > KNIFE).reshape(H, W)
    assert (~stable).mean() < 5e-3
    HW = H * W
    np.testing.assert_array_equal(h["last_contributor"][stable], o["n_contrib"][:HW].reshape(H, W)[stable])
    assert np.abs(h["color"] - o["color"])[:, stable].max() <= IMG_TOL
    assert util.allmap_dev(h, o, stable).max() <= IMG_TOL
    dc, da = util.make_upstream_grads(W, H)
    dc, da = (dc * W * H).numpy(), (da * W * H).numpy()
    dc[:, ~stable] = 0; da[:, ~stable] = 0
    go = oracle.backward(o, dc, da)
    gh = util.hip_backward(h, dc, da)
    for k in ["dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dtransMat"]:
        assert util.grad_err(gh[k], go[k].reshape(gh[k].shape)) <= GRAD_TOL, k
    # size-independent property: the backward is LINEAR in the upstream gradients (every term of backward.cu:143-664
    # is), including on the knife-edge pixels and all ten channels -- checked at the full size without the oracle
    u1 = util.make_upstream_grads(W, H, seed=11, channels=(0, 1, 2, 3, 4, 5, 6))
    u2 = util.make_upstream_grads(W, H, seed=12, channels=(0, 1, 2, 3, 4, 5, 6))
    c1, a1 = (u1[0] * W * H).numpy(), (u1[1] * W * H).numpy()
    c2, a2 = (u2[0] * W * H).numpy(), (u2[1] * W * H).numpy()
    g1, g2 = util.hip_backward(h, c1, a1), util.hip_backward(h, c2, a2)
    g12 = util.hip_backward(h, 0.75 * c1 - 1.5 * c2, 0.75 * a1 - 1.5 * a2)
    for k in ["dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dmeans2D"]:
        assert util.grad_err(g12[k], 0.75 * g1[k] - 1.5 * g2[k]) <= GRAD_TOL, k
    oracle.set_threads(1)


def test_mark_visible(oracle):
    from gaus_slam_amd import rasterizer
    sc = util.make_scene(5000, 320, 240, seed=2, regime="mapping")
    cam = sc["cam"]
    dev = torch.device("cuda")
    got = rasterizer.mark_visible(sc["means3D"].to(dev), cam.viewmatrix.to(dev), cam.projmatrix.to(dev)).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.mark_visible(sc["means3D"].numpy(), cam.viewmatrix.numpy()))
    assert 0 < got.sum() < 5000


@pytest.mark.parametrize("N", [1, 3, 5, 1000, 40000])
def test_simple_knn_dist2(oracle, N):
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(N)
    pts = rng.normal(size=(N, 3)).astype(np.float32)
    if N >= 1000:
        pts[: N // 10] = pts[N // 10: 2 * (N // 10)]  # exact duplicates
    got = distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.dist2_knn3(pts))  # exact k-NN set, same expression order: bit-exact


def test_autograd_operator_surface(oracle):
    """Drive the op the way render/render_2dgs.py + slam/Loss.py do: settings NamedTuple, GaussianRasterizer module,
    means2D grad sink, weight-normalised depth, masked L1 loss; gradients vs the oracle fed with the same upstream
    gradients."""
    from gaus_slam_amd import render as gsr
    W, H, P = 160, 120, 2000
    sc = util.make_scene(P, W, H, seed=9, regime="tracking")
    dev = torch.device("cuda")
    params = {k: sc[k].to(dev).requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "colors")}
    means2D = torch.zeros_like(params["means3D"], requires_grad=True)
    settings = gsr.settings_from_camera(sc["cam"], dev, use_sa=True)
    pkg = gsr.render(settings, params["means3D"], means2D, params["opacities"], colors_precomp=params["colors"],
                     scales=params["scales"], rotations=params["rotations"], use_weight_norm=True)
    for t in (pkg["render_color"], pkg["allmap"]):
        t.retain_grad()
    target_c = torch.full((3, H, W), 0.4, device=dev)
    target_d = torch.full((1, H, W), 2.5, device=dev)
    mask = (pkg["render_alpha"] > 0.5).float()
    loss = ((pkg["render_color"] - target_c).abs() * mask).sum() + ((pkg["render_depth"] - target_d).abs() * mask).sum() \
        + 0.1 * pkg["render_dist"].mean()
    loss.backward()
    assert pkg["radius"].dtype == torch.int32 and not pkg["radius"].requires_grad
    o = util.oracle_forward(oracle, sc, use_sa=True)
    go = oracle.backward(o, pkg["render_color"].grad.cpu().numpy(), pkg["allmap"].grad.cpu().numpy())
    for name, k in (("means3D", "dL_dmeans3D"), ("scales", "dL_dscales"), ("rotations", "dL_drotations"),
                    ("opacities", "dL_dopacity"), ("colors", "dL_dcolors")):
        assert util.grad_err(params[name].grad.cpu().numpy(), go[k].reshape(params[name].shape)) <= 2e-4, name
    assert util.grad_err(means2D.grad.cpu().numpy(), go["dL_dmeans2D"]) <= 2e-4
    assert float(means2D.grad[:, 2].abs().max()) == 0


def test_long_tile_lists_take_the_global_sort_path(oracle):
    """Tile lists longer than the LDS capacity of the per-tile depth sort (4096) use the global-memory variant;
    the order must still be bit-exact.  Many equal depths stress stability (ties keep Gaussian order)."""
    P, W, H = 40000, 48, 32
    sc = util.make_scene(P, W, H, seed=4, regime="tracking", scale_lo=2.0, scale_hi=12.0)
    sc["means3D"][::3, 2] = 2.0  # exact depth ties
    o = util.oracle_forward(oracle, sc, use_sa=True)
    lens = o["ranges"][:, 1].astype(np.int64) - o["ranges"][:, 0]
    assert lens.max() > 4096
    h = util.hip_forward(sc, use_sa=True)
    np.testing.assert_array_equal(h["keys"], o["keys"])
    np.testing.assert_array_equal(h["point_list"], o["point_list"])
    np.testing.assert_array_equal(h["ranges"], o["ranges"])
    stable = (o["stability"] > KNIFE).reshape(H, W)
    np.testing.assert_array_equal(h["last_contributor"][stable], o["n_contrib"][:H * W].reshape(H, W)[stable])
    assert np.abs(h["color"] - o["color"])[:, stable].max() <= IMG_TOL


def test_replica_shaped_1200x680(oracle):
    """Replica native 1200x680 (75x43 = 3225 tiles -> 12 tile-id bits), 600k surfels,
    tracking regime (identity view)."""
    P, W, H = 600000, 1200, 680
    sc = util.make_scene(P, W, H, seed=2, regime="tracking")
    oracle.set_threads(os.cpu_count() or 1)
    o = util.oracle_forward(oracle, sc, use_sa=True)
    assert o["nbits"] == 32 + 12
    h = util.hip_forward(sc, use_sa=True)
    assert h["num_rendered"] == o["num_rendered"]
    np.testing.assert_array_equal(h["point_list"], o["point_list"])
    np.testing.assert_array_equal(h["ranges"], o["ranges"])
    stable = (o["stability"] > KNIFE).reshape(H, W)
    assert (~stable).mean() < 5e-3
    np.testing.assert_array_equal(h["last_contributor"][stable], o["n_contrib"][:H * W].reshape(H, W)[stable])
    assert np.abs(h["color"] - o["color"])[:, stable].max() <= IMG_TOL
    assert util.allmap_dev(h, o, stable).max() <= IMG_TOL
    dc, da = util.make_upstream_grads(W, H, channels=(0, 1))  # tracking loss touches colour, depth, alpha
    dc, da = (dc * W * H).numpy(), (da * W * H).numpy()
    dc[:, ~stable] = 0; da[:, ~stable] = 0
    go = oracle.backward(o, dc, da)
    gh = util.hip_backward(h, dc, da)
    for k in ["dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations"]:
        assert util.grad_err(gh[k], go[k].reshape(gh[k].shape)) <= GRAD_TOL, k
    oracle.set_threads(1)


def test_scannetpp_shaped_1168x876_2M(oracle):
    """BASELINE.json configs[4] shape: 1168x876 (73x55 = 4015 tiles, 12 tile-id bits), 2M surfels.  Forward indices
    bit-exact, images within tolerance on stable pixels; gradients on the parameters the mapping loss touches."""
    P, W, H = 2000000, 1168, 876
    sc = util.make_scene(P, W, H, seed=6, regime="mapping")
    oracle.set_threads(os.cpu_count() or 1)
    o = util.oracle_forward(oracle, sc, use_sa=True)
    h = util.hip_forward(sc, use_sa=True)
    assert h["num_rendered"] == o["num_rendered"] and o["nbits"] == 32 + 12
    np.testing.assert_array_equal(h["point_list"], o["point_list"])
    np.testing.assert_array_equal(h["ranges"], o["ranges"])
    stable = (o["stability"] > KNIFE).reshape(H, W)
    assert (~stable).mean() < 5e-3
    assert np.abs(h["color"] - o["color"])[:, stable].max() <= IMG_TOL
    assert util.allmap_dev(h, o, stable).max() <= IMG_TOL
    dc, da = util.make_upstream_grads(W, H, channels=(0, 1, 6))
    dc, da = (dc * W * H).numpy(), (da * W * H).numpy()
    dc[:, ~stable] = 0; da[:, ~stable] = 0
    go = oracle.backward(o, dc, da)
    gh = util.hip_backward(h, dc, da)
    for k in ["dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations"]:
        assert util.grad_err(gh[k], go[k].reshape(gh[k].shape)) <= GRAD_TOL, k
    oracle.set_threads(1)


def test_masked_upstream_gradients(oracle):
    """Upstream gradients that are exactly zero on most of the image (SLAM's masked losses): the backward's
    zero-quadrant early exit must not change any gradient."""
    W, H, P = 320, 240, 6000
    sc = util.make_scene(P, W, H, seed=17, regime="mapping")
    o = util.oracle_forward(oracle, sc, use_sa=True)
    h = util.hip_forward(sc, use_sa=True)
    stable = (o["stability"] > KNIFE).reshape(H, W)
    dc, da = util.make_upstream_grads(W, H, channels=(0, 1, 6))
    dc, da = (dc * W * H).numpy(), (da * W * H).numpy()
    mask = np.zeros((H, W), bool)
    mask[60:150, 100:260] = True      # only a window carries gradient; cuts through tiles and quadrants
    mask[::7, ::5] |= True            # plus isolated pixels
    mask &= stable
    dc[:, ~mask] = 0; da[:, ~mask] = 0
    go = oracle.backward(o, dc, da)
    gh = util.hip_backward(h, dc, da)
    for k in ["dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dtransMat", "dL_dmeans2D"]:
        assert util.grad_err(gh[k], go[k].reshape(gh[k].shape)) <= GRAD_TOL, k


def test_slam_iteration_fused_loss_adam_and_direct_bucket_grads(oracle):
    """One mapping iteration the fused way -- rasterizer -> gs2d_slam_loss -> gradients written straight into the all-reduce
    bucket -> gs2d_adam_step over the flat SoA -- against the same iteration in the reference's formulation (operator +
    PyTorch post-op/loss from oracle/loss_ref.py + torch.optim.Adam over five tensors)."""
    from gaus_slam_amd import ba_shard, loss as gl, optim, render as gsr
    from oracle import loss_ref
    W, H, P = 160, 120, 3000
    sc = util.make_scene(P, W, H, seed=21, regime="mapping")
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    gt_color, gt_depth = torch.rand(H, W, 3, generator=g).to(dev), (0.5 + 5 * torch.rand(H, W, 1, generator=g)).to(dev)
    settings = gsr.settings_from_camera(sc["cam"], dev, use_sa=True)
    names = ("means3D", "opacities", "scales", "rotations", "colors")
    lrs = dict(xyz=1e-4, opacity=0.05, scaling=1e-3, rotation=1e-3, rgb=2.5e-3)

    def rasterize(p):
        m2 = torch.zeros_like(p["means3D"], requires_grad=True)
        return gsr.render(settings, p["means3D"], m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"],
                          rotations=p["rotations"])

    # reference formulation
    ref = {k: sc[k].to(dev).clone().requires_grad_(True) for k in names}
    ropt = torch.optim.Adam([{"params": [ref[k]], "lr": lrs[optim.GROUP_NAMES[k]]} for k in names], lr=0.0, eps=1e-15)
    pkg = rasterize(ref)
    loss_ref.post_and_loss(pkg["render_color"], pkg["allmap"], gt_color, gt_depth, 1, 0.5, 1.0, 0.1).backward()
    ref_grads = {k: ref[k].grad.clone() for k in names}
    ropt.step()

    # fused formulation
    soa = optim.GaussianSoA({k: sc[k].to(dev) for k in names})
    params = dict(soa.leaves())
    fopt = optim.FusedGaussianAdam(soa, lrs)

    def fn(p, _kf):
        pk = rasterize(p)
        return gl.mapping_loss(pk["render_color"], pk["allmap"], gt_color, gt_depth, 0.5, 1.0, 0.1)

    ba = ba_shard.KeyframeShardedBA(params, fn, direct_grads=True)
    grads = ba.step([0])
    for k in names:
        assert grads[k].data_ptr() == ba.bucket.views[k].data_ptr(), k  # written in place, no pack copy
        assert util.grad_err(grads[k].cpu().numpy(), ref_grads[k].cpu().numpy()) < 2e-4, k  # float atomics reorder
    fopt.step(ba.bucket.flat)
    for k in names:
        assert (soa.views[k] - ref[k].detach()).abs().max().item() < 2e-5, k


def test_more_than_4096_tiles_takes_the_radix_pass_path(oracle):
    """Images with more than GS2D_BIN_MAX_TILES (4096) tiles bin by tile id with the generic 8-bit radix passes (64-bit
    keys + separate ids, tile_ranges kernel) instead of the single counting-sort pass; order and images must not change."""
    P, W, H = 30000, 1600, 1104  # 100 x 69 = 6900 tiles
    sc = util.make_scene(P, W, H, seed=12, regime="mapping", scale_lo=1.0, scale_hi=6.0)
    o = util.oracle_forward(oracle, sc, use_sa=True)
    h = util.hip_forward(sc, use_sa=True)
    assert h["num_rendered"] == o["num_rendered"] and h["ranges"].shape[0] == 6900
    np.testing.assert_array_equal(h["keys"], o["keys"])
    np.testing.assert_array_equal(h["point_list"], o["point_list"])
    np.testing.assert_array_equal(h["ranges"], o["ranges"])
    stable = (o["stability"] > KNIFE).reshape(H, W)
    assert np.abs(h["color"] - o["color"])[:, stable].max() <= IMG_TOL
    dc, da = util.make_upstream_grads(W, H, channels=(0, 1, 6))
    dc, da = (dc * W * H).numpy(), (da * W * H).numpy()
    dc[:, ~stable] = 0; da[:, ~stable] = 0
    gh = util.hip_backward(h, dc, da)
    go = oracle.backward(o, dc, da)
    for k in ("dL_dmeans3D", "dL_dopacity", "dL_dcolors", "dL_dscales", "dL_drotations"):
        assert util.grad_err(gh[k], go[k].reshape(gh[k].shape)) <= GRAD_TOL, k


@pytest.mark.filterwarnings("ignore:The AccumulateGrad node's stream")
def test_two_keyframes_per_rank_on_two_streams_match_sequential(oracle):
    """A rank holding two keyframes renders them on two HIP streams (ba_shard.KeyframeShardedBA, streams=2); the summed
    gradients must equal the sequential single-stream result (float atomics reorder only)."""
    from gaus_slam_amd import ba_shard, render as gsr
    from gaus_slam_amd.scene_synth import random_w2c, setup_camera
    W, H, P = 160, 120, 4000
    sc = util.make_scene(P, W, H, seed=31, regime="mapping")
    dev = torch.device("cuda")
    cams = [sc["cam"], setup_camera(W, H, sc["cam"].K, random_w2c(np.random.default_rng(3), 3.0, 0.1) @ sc["cam"].w2c)]
    settings = [gsr.settings_from_camera(c, dev, use_sa=True) for c in cams]
    dc, da = [t.to(dev) * W * H for t in util.make_upstream_grads(W, H, channels=(0, 1, 6))]
    names = ("means3D", "opacities", "scales", "rotations", "colors")

    def fn(p, kf):
        m2 = torch.zeros_like(p["means3D"], requires_grad=True)
        pkg = gsr.render(settings[kf], p["means3D"], m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"],
                         rotations=p["rotations"])
        return (pkg["render_color"], pkg["allmap"]), (dc, da)

    res = []
    for streams in (1, 2):
        params = {k: sc[k].to(dev).clone().requires_grad_(True) for k in names}
        ba = ba_shard.KeyframeShardedBA(params, fn, streams=streams)
        g = ba.step([0, 1])
        torch.cuda.synchronize()
        res.append({k: v.clone() for k, v in g.items()})
    for k in names:
        assert util.grad_err(res[1][k].cpu().numpy(), res[0][k].cpu().numpy()) < 1e-5, k
        assert float(res[0][k].abs().max()) > 0


def test_backward_through_one_output_only():
    """A loss that uses only one of the two images: autograd hands the operator's backward no gradient for the other one
    (the function does not let it materialise zeros, which would also cost a [P] fill for `radii` on every call)."""
    from gaus_slam_amd import render as gs_render
    dev = torch.device("cuda", 0)
    W, H = 112, 80
    sc = util.make_scene(900, W, H, seed=9, regime="mapping")
    st = gs_render.settings_from_camera(sc["cam"], dev, use_sa=True)
    names = ("means3D", "opacities", "scales", "rotations", "colors")

    def grads(which):
        p = {k: sc[k].to(dev).requires_grad_(True) for k in names}
        m2 = torch.empty_like(p["means3D"]).requires_grad_(True)
        pkg = gs_render.render(st, p["means3D"], m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"],
                               rotations=p["rotations"])
        wc = torch.linspace(0.5, 1.5, 3 * H * W, device=dev).reshape(3, H, W)
        wa = torch.linspace(-1.0, 1.0, 7 * H * W, device=dev).reshape(7, H, W)
        if which == "color":
            (pkg["render_color"] * wc).sum().backward()
        elif which == "allmap":
            (pkg["allmap"] * wa).sum().backward()
        else:
            torch.autograd.backward([pkg["render_color"], pkg["allmap"]],
                                    [wc if which == "color+0" else torch.zeros_like(wc),
                                     torch.zeros_like(wa) if which == "color+0" else wa])
        return [p[k].grad.double().cpu() for k in names] + [m2.grad.double().cpu()]

    for one, both in (("color", "color+0"), ("allmap", "0+allmap")):
        for a, b in zip(grads(one), grads(both)):
            scale = float(b.abs().max())  # 0 for the colours under the allmap-only loss
            assert float((a - b).abs().max()) <= 1e-5 * scale
