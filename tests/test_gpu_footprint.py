"""Footprint binning (the library default) against the reference's 3-sigma rectangles.

By default the library makes a (tile, Gaussian) instance only for the tiles inside the splat's footprint bound
(include/gs2d_rasterizer.h, gs2d_set_reference_binning).  The chain of evidence that this changes no result:
  1. reference mode reproduces the oracle's lists bit for bit (tests/test_gpu_parity.py and friends);
  2. the footprint lists are ordered subsequences of the reference lists, tile by tile (radii untouched);
  3. the ORACLE's blend run on the footprint lists gives the oracle's image and gradients bit for bit -- i.e. under the
     reference's own arithmetic every dropped instance contributes nothing to any pixel;
  4. the HIP forward outputs are bit-identical between the two modes, and so are the deterministic-mode gradients;
  5. the HIP path in footprint mode matches the oracle (on the footprint lists) within the usual tolerances, knife-edge
     pixels resolved against the oracle's decision variants."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

IMG_TOL = 1e-4
GRAD_TOL = 1e-4
KNIFE = 2e-5
GRADS = ["dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dtransMat", "dL_dscales", "dL_drotations", "dL_dmeans2D"]


def _assert_subsequence(ref, sub, ntiles):
    """every tile's list in `sub` is the tile's list in `ref` with elements removed (order kept)."""
    def keyed(h):
        lens = h["ranges"][:, 1].astype(np.int64) - h["ranges"][:, 0]
        assert lens.sum() == h["num_rendered"] and (lens >= 0).all()
        tile = np.repeat(np.arange(ntiles, dtype=np.int64), lens)
        return (tile << 32) | h["point_list"].astype(np.int64)
    kr, ks = keyed(ref), keyed(sub)
    assert len(np.unique(kr)) == len(kr)  # a Gaussian appears once per tile
    order = np.argsort(kr, kind="stable")
    where = np.searchsorted(kr[order], ks)
    assert (where < len(kr)).all() and (kr[order][np.minimum(where, len(kr) - 1)] == ks).all(), "instance not in the reference list"
    pos = order[where]  # position of every kept instance in the reference's (tile-major) list
    assert (np.diff(pos) > 0).all(), "order inside a tile changed"
    return pos


@pytest.mark.parametrize("regime", ["tracking", "mapping"])
@pytest.mark.parametrize("use_sa", [True, False])
@pytest.mark.parametrize("P,W,H", [(256, 160, 120), (4000, 320, 240), (3000, 150, 100)])
def test_footprint_lists_change_nothing(oracle, regime, use_sa, P, W, H):
    sc = util.make_scene(P, W, H, seed=0, regime=regime)
    bg = (0.2, 0.5, 0.1)
    o = util.oracle_forward(oracle, sc, use_sa=use_sa, bg=bg)
    hr = util.hip_forward(sc, use_sa=use_sa, bg=bg, binning="reference")
    ht = util.hip_forward(sc, use_sa=use_sa, bg=bg, binning="footprint")
    ntiles = o["ranges"].shape[0]
    # 1 + 2
    np.testing.assert_array_equal(hr["point_list"], o["point_list"])
    np.testing.assert_array_equal(hr["ranges"], o["ranges"])
    np.testing.assert_array_equal(ht["radii"], o["radii"])
    assert (ht["tiles_touched"] <= o["tiles_touched"]).all()
    assert ht["num_rendered"] == int(ht["tiles_touched"].sum()) <= o["num_rendered"]
    _assert_subsequence(o, ht, ntiles)
    print(f"instances: reference {o['num_rendered']}, footprint {ht['num_rendered']} "
          f"({1 - ht['num_rendered'] / max(o['num_rendered'], 1):.1%} dropped)")
    # 3: the oracle's arithmetic on the shorter lists
    ot = oracle.reblend(o, ht["ranges"], ht["point_list"])
    for k in ("color", "allmap", "final_T", "median_depth", "depth_std"):
        np.testing.assert_array_equal(ot[k].view(np.uint32), o[k].view(np.uint32), err_msg=k)
    dc, da = util.make_upstream_grads(W, H, channels=(0, 1, 2, 3, 4, 5, 6))
    dc = (dc * W * H).numpy(); da = (da * W * H).numpy()
    stable = (ot["stability"] > KNIFE).reshape(H, W)
    dc[:, ~stable] = 0; da[:, ~stable] = 0
    go, gt = oracle.backward(o, dc, da), oracle.backward(ot, dc, da)
    for k in GRADS:
        np.testing.assert_array_equal(gt[k].view(np.uint32), go[k].view(np.uint32), err_msg=k)
    # 4: HIP forward bit-identical between the modes
    np.testing.assert_array_equal(ht["color"].view(np.uint32), hr["color"].view(np.uint32))
    np.testing.assert_array_equal(ht["allmap"].view(np.uint32), hr["allmap"].view(np.uint32))
    np.testing.assert_array_equal(ht["final_T"].view(np.uint32), hr["final_T"].view(np.uint32))
    # 5: HIP (footprint lists) against the oracle on the same lists
    HW = H * W
    np.testing.assert_array_equal(ht["last_contributor"][stable], ot["n_contrib"][:HW].reshape(H, W)[stable])
    np.testing.assert_array_equal(ht["median_contributor"][stable], ot["n_contrib"][HW:].reshape(H, W)[stable])
    assert np.abs(ht["color"] - ot["color"])[:, stable].max() <= IMG_TOL
    assert (util.allmap_dev(ht, ot, stable) <= IMG_TOL).all()
    util.check_knife_pixels(oracle, ot, ht, stable, IMG_TOL, KNIFE)
    gh = util.hip_backward(ht, dc, da)
    for k in GRADS:
        assert util.grad_err(gh[k], go[k].reshape(gh[k].shape)) <= GRAD_TOL, k


def test_footprint_deterministic_gradients_bit_identical_between_modes():
    from gaus_slam_amd import rasterizer
    P, W, H = 20000, 320, 240
    sc = util.make_scene(P, W, H, seed=3, regime="mapping")
    dc, da = util.make_upstream_grads(W, H, channels=(0, 1, 2, 3, 4, 5, 6))
    dc = (dc * W * H).numpy(); da = (da * W * H).numpy()
    rasterizer.set_deterministic(True)
    try:
        g = {}
        for mode in ("reference", "footprint"):
            h = util.hip_forward(sc, binning=mode)
            g[mode] = util.hip_backward(h, dc, da)
    finally:
        rasterizer.set_deterministic(False)
    for k in GRADS:
        np.testing.assert_array_equal(g["footprint"][k].view(np.uint32), g["reference"][k].view(np.uint32), err_msg=k)


def test_footprint_full_size_bit_identical_and_smaller():
    """BASELINE shape (640x480 / 500k): outputs bit-identical between the modes, lists a subsequence, fewer instances."""
    P, W, H = 500000, 640, 480
    sc = util.make_scene(P, W, H, seed=0, regime="mapping")
    hr = util.hip_forward(sc, binning="reference")
    ht = util.hip_forward(sc, binning="footprint")
    np.testing.assert_array_equal(ht["radii"], hr["radii"])
    np.testing.assert_array_equal(ht["color"].view(np.uint32), hr["color"].view(np.uint32))
    np.testing.assert_array_equal(ht["allmap"].view(np.uint32), hr["allmap"].view(np.uint32))
    np.testing.assert_array_equal(ht["final_T"].view(np.uint32), hr["final_T"].view(np.uint32))
    _assert_subsequence(hr, ht, hr["ranges"].shape[0])
    assert ht["num_rendered"] < 0.9 * hr["num_rendered"]
    print(f"instances: reference {hr['num_rendered']}, footprint {ht['num_rendered']}")
    # the gradients agree up to the order of the float atomics (all ten upstream channels live)
    dc, da = util.make_upstream_grads(W, H, seed=21, channels=(0, 1, 2, 3, 4, 5, 6))
    dc = (dc * W * H).numpy(); da = (da * W * H).numpy()
    gr, gt = util.hip_backward(hr, dc, da), util.hip_backward(ht, dc, da)
    for k in GRADS:
        assert util.grad_err(gt[k], gr[k]) <= 2e-5, k


def test_footprint_edge_cases(oracle):
    """Opacities at and below the 1/255 threshold (footprint empty: no instance at all, radii still the reference's),
    NaN opacity, splats behind / across the eye plane and huge splats: same image bit for bit in both modes."""
    P, W, H = 3000, 160, 120
    sc = util.make_scene(P, W, H, seed=5, regime="mapping")
    op = sc["opacities"].clone()
    op[0:200] = 1.0 / 255.0
    op[200:400] = 0.5 / 255.0
    op[400:500] = 0.0
    op[500:520] = float("nan")
    op[520:600] = 1.0
    sc["opacities"] = op
    scl = sc["scales"].clone(); scl[600:700] *= 40.0; scl[700:800] *= 1e-3
    sc["scales"] = scl
    o = util.oracle_forward(oracle, sc)
    hr = util.hip_forward(sc, binning="reference")
    ht = util.hip_forward(sc, binning="footprint")
    np.testing.assert_array_equal(hr["point_list"], o["point_list"])
    np.testing.assert_array_equal(ht["radii"], o["radii"])
    _assert_subsequence(o, ht, o["ranges"].shape[0])
    vis = o["radii"] > 0
    never = vis & (np.nan_to_num(op.numpy().reshape(-1), nan=1.0) < 1.0 / 255.0)
    assert never.any() and (ht["tiles_touched"][never] == 0).all()  # can never reach alpha 1/255: no instance at all
    ot = oracle.reblend(o, ht["ranges"], ht["point_list"])
    for k in ("color", "allmap"):
        np.testing.assert_array_equal(ot[k].view(np.uint32), o[k].view(np.uint32), err_msg=k)
        np.testing.assert_array_equal(ht[k].view(np.uint32), hr[k].view(np.uint32), err_msg=k)


@pytest.mark.parametrize("P,W,H,regime", [(600000, 1200, 680, "tracking"), (2000000, 1168, 876, "mapping"),
                                          (300000, 3840, 2160, "mapping")])
def test_footprint_large_images_bit_identical(P, W, H, regime):
    """Centres thousands of pixels from the origin (where a cancellation-prone bound would be least accurate) and tile
    counts on both sides of the single-pass binning limit: forward outputs bit-identical between the modes, lists a
    subsequence."""
    sc = util.make_scene(P, W, H, seed=1, regime=regime)
    hr = util.hip_forward(sc, binning="reference")
    ht = util.hip_forward(sc, binning="footprint")
    np.testing.assert_array_equal(ht["radii"], hr["radii"])
    for k in ("color", "allmap", "final_T"):
        np.testing.assert_array_equal(ht[k].view(np.uint32), hr[k].view(np.uint32), err_msg=k)
    _assert_subsequence(hr, ht, hr["ranges"].shape[0])
    assert ht["num_rendered"] < hr["num_rendered"]
    print(f"{W}x{H}: reference {hr['num_rendered']}, footprint {ht['num_rendered']}")
