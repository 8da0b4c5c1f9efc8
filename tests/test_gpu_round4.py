"""Round-4 GPU tests: invariants of the backward's queues that round 3 only met by accident."""
import os

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("use_sa", [True, False])
@pytest.mark.parametrize("W,H", [(64, 48), (37, 21)])
def test_backward_with_exhausted_rows_beside_live_ones(oracle, use_sa, W, H):
    """The single-frame backward keeps one queue per 4x4 sub-block (DPP row) of a quadrant; a row whose queue is exhausted keeps
    evaluating the batch's deepest staged record with all its lanes inactive while the other rows work on (gs2d_blend.hip,
    BwdBatchT).  Round 3's abort (an exhausted row read past the staged records and corrupted a Gaussian id) was only caught by
    a batch test; this scene forces the situation in the plain single-frame path: every splat is small and sits in the TOP-LEFT
    4x4 sub-block of its 8x8 quadrant, so rows 1-3 of nearly every quadrant are exhausted from the first trip on while row 0
    has many -- plus a few large splats so that some batches do fill all rows, and a ragged image whose border quadrants have
    rows and lanes outside the picture.  Gradients against the oracle; a fault or a NaN fails it."""
    rng = np.random.default_rng(7)
    P = 3000
    sc = util.make_scene(P, W, H, seed=51, regime="tracking", cull_frac=0.0)
    cam = sc["cam"]
    f, cx, cy = float(cam.K[0, 0]), float(cam.K[0, 2]), float(cam.K[1, 2])
    z = rng.uniform(1.0, 4.0, P)
    # pixel centres inside the top-left sub-block of a random quadrant
    u = rng.integers(0, (W + 7) // 8, P) * 8 + rng.uniform(0.8, 2.2, P)
    v = rng.integers(0, (H + 7) // 8, P) * 8 + rng.uniform(0.8, 2.2, P)
    means = np.stack([(u - cx) / f * z, (v - cy) / f * z, z], 1)
    sigma_px = np.where(rng.uniform(size=P) < 0.004, rng.uniform(3.0, 6.0, P), rng.uniform(0.3, 0.6, P))  # a dozen large splats
    sc["means3D"] = torch.from_numpy(means).float()
    sc["scales"] = torch.from_numpy(np.stack([z / f * sigma_px, z / f * sigma_px], 1)).float()
    sc["rotations"] = torch.tensor([[0.0, 1.0, 0.0, 0.0]]).repeat(P, 1)  # normals facing the camera
    sc["opacities"] = torch.from_numpy(rng.uniform(0.05, 0.6, (P, 1))).float()
    oracle.set_threads(os.cpu_count() or 1)
    o = util.oracle_forward(oracle, sc, use_sa=use_sa)
    h = util.hip_forward(sc, use_sa=use_sa)
    np.testing.assert_array_equal(h["point_list"], o["point_list"])
    # the situation is really there: most pixels outside the top-left sub-blocks have no contributor, those inside have many
    last = o["n_contrib"][:H * W].reshape(H, W)
    ys, xs = np.mgrid[0:H, 0:W]
    inside = ((xs % 8) < 4) & ((ys % 8) < 4)
    assert np.median(last[inside]) >= 8
    if W == 64:  # (the small ragged image is covered by its few large splats; it is there for the lanes outside the picture)
        assert (last[~inside] == 0).mean() > 0.25
    stable = (o["stability"] > 2e-5).reshape(H, W)
    dc, da = util.make_upstream_grads(W, H, channels=(0, 1, 2, 3, 4, 5, 6))
    dc, da = (dc * W * H).numpy(), (da * W * H).numpy()
    dc[:, ~stable] = 0; da[:, ~stable] = 0
    go = oracle.backward(o, dc, da)
    gh = util.hip_backward(h, dc, da)
    oracle.set_threads(1)
    for k in ("dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dtransMat", "dL_dscales", "dL_drotations", "dL_dmeans2D"):
        assert np.isfinite(gh[k]).all(), k
        assert util.grad_err(gh[k], go[k].reshape(gh[k].shape)) <= 1e-4, (k, util.grad_err(gh[k], go[k].reshape(gh[k].shape)))


def test_launch_ahead_forward_zero_instances_and_jumps(oracle):
    """The launch-ahead forward (gs2d_set_launch_ahead, default) sizes its grids and its binning chunk from the previous call's
    count for the same problem shape and reads the count on the device.  Same shape, three very different counts in a row:
    a regular scene, the same Gaussians all moved behind the camera (zero instances: every kernel behind the duplication runs
    with R = 0 in a chunk sized for many), then huge splats (a count far beyond the capacity: those kernels do nothing, the host
    runs the stages again).  Every call must give the oracle's lists and image."""
    from gaus_slam_amd import rasterizer
    assert rasterizer.is_launch_ahead()
    W, H, P = 256, 192, 5000
    base = util.make_scene(P, W, H, seed=61, regime="tracking")
    gone = dict(base); gone["means3D"] = base["means3D"].clone(); gone["means3D"][:, 2] = -1.0   # behind the near plane
    huge = dict(base); huge["scales"] = base["scales"] * 30.0
    for name, sc in (("regular", base), ("nothing visible", gone), ("regular again", base), ("huge splats", huge), ("regular once more", base)):
        o = util.oracle_forward(oracle, sc, use_sa=True)
        h = util.hip_forward(sc, use_sa=True)
        assert h["num_rendered"] == o["num_rendered"], name
        if name == "nothing visible":
            assert h["num_rendered"] == 0 and float(np.abs(h["color"]).max()) == 0.0 and float(np.abs(h["allmap"]).max()) == 0.0
            continue
        np.testing.assert_array_equal(h["point_list"], o["point_list"], err_msg=name)
        np.testing.assert_array_equal(h["ranges"], o["ranges"], err_msg=name)
        stable = (o["stability"] > 2e-5).reshape(H, W)
        assert np.abs(h["color"] - o["color"])[:, stable].max() <= 1e-4, name
        assert util.allmap_dev(h, o, stable).max() <= 1e-4, name
    assert util.oracle_forward(oracle, huge, use_sa=True)["num_rendered"] > 8 * util.oracle_forward(oracle, base, use_sa=True)["num_rendered"]


@pytest.mark.parametrize("items", [8192, 16384])
def test_binning_with_large_workgroups_on_crowded_tiles(items):
    """bin_scatter ranks a pair among its wave's pairs of the same tile with the LDS atomic's return value and re-ranks the lanes
    of one instruction that share a tile in lane order (gs2d_binning.hip).  Images of more than 1536 tiles take 8192 / 16384
    pairs per workgroup (several rounds per wave, the pairs read twice); the scenes of that size in this suite have few pairs
    per tile, so the re-ranking hardly runs there.  GS2D_BIN_ITEMS_FORCE (read once per process, hence the child process) sends
    the crowded scenes -- 40 000 splats on six tiles with exact depth ties, where every instruction holds dozens of pairs of
    one tile -- and the golden vectors through those instantiations; lists, keys and ranges must stay bit-exact."""
    import subprocess
    import sys
    env = dict(os.environ, GS2D_BIN_ITEMS_FORCE=str(items))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_more.py", "-x", "-q", "-m", "gpu", "-k",
                        "long_tile_lists or golden_vectors or culling_stress"], env=env, capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.parametrize("huge", [True, False])
def test_non_finite_and_degenerate_inputs_do_not_disturb_the_rest(oracle, huge):
    """NaN, +-Inf, 0, negative and 1e+-30 sprinkled over every input tensor (plus all-zero quaternions and scales): the reference
    does not trap them ("NaNs are not trapped"; slam/Loss.py:22-25 cleans up afterwards).  Every index
    the kernels form comes from clamped, saturating conversions (getRect, f2i_sat: CUDA's cvt semantics, NaN -> 0), so nothing may
    fault or hang, the structural outputs must equal the oracle's bit for bit (num_rendered, radii, tiles_touched, offsets, tile
    ranges, every tile's SET of Gaussians -- the order among splats whose depth key is a NaN follows the NaN's sign / payload bits,
    which differ between processors), and the tiles whose lists hold no poisoned Gaussian must come out as if the poison were
    not there: image within 1e-4 of the oracle's, gradients of the Gaussians that only touch such tiles within 1e-4.
    huge: with +-Inf / +-1e30 among the poison some splats cover every tile (no clean tile is left: structural checks and the
    backward's survival only); without them most tiles stay clean."""
    P, W, H = 12000, 480, 352
    sc = util.make_scene(P, W, H, seed=9, regime="mapping")
    rng = np.random.default_rng(1)
    bad = [float("nan"), float("inf"), -float("inf"), 0.0, -1.0, 1e30, -1e30, 1e-30] if huge else [float("nan"), 0.0, -1.0, 1e-30]
    poisoned = np.zeros(P, bool)
    for name, per in (("means3D", 3), ("scales", 2), ("rotations", 4), ("opacities", 1), ("colors", 3)):
        t = sc[name].clone()
        idx = rng.choice(t.numel(), 80 if huge else 16, replace=False)
        for j, i in enumerate(idx):
            v = bad[j % len(bad)]
            # (a scale of -1 is a splat a metre wide: it covers every tile; the mild set uses a negative scale of ordinary size)
            t.view(-1)[int(i)] = -0.004 if (not huge and name == "scales" and v == -1.0) else v
        poisoned[idx // per] = True
        sc[name] = t
    sc["rotations"][10:14] = 0; sc["scales"][20:24] = 0
    poisoned[10:14] = True; poisoned[20:24] = True
    oracle.set_threads(os.cpu_count() or 1)
    o = util.oracle_forward(oracle, sc, use_sa=True)
    h = util.hip_forward(sc, use_sa=True)  # (reference-binning mode)
    assert h["num_rendered"] == o["num_rendered"] > 0
    np.testing.assert_array_equal(h["radii"], o["radii"])
    np.testing.assert_array_equal(h["tiles_touched"], o["tiles_touched"])
    np.testing.assert_array_equal(h["point_offsets"], o["point_offsets"])
    np.testing.assert_array_equal(h["ranges"], o["ranges"])
    gx = (W + 15) // 16
    clean_tile = np.ones(o["ranges"].shape[0], bool)
    for t, (a, b) in enumerate(o["ranges"]):
        lo, lh = o["point_list"][a:b], h["point_list"][a:b]
        np.testing.assert_array_equal(np.sort(lo), np.sort(lh))
        clean_tile[t] = not poisoned[lo].any()
        if clean_tile[t]:
            np.testing.assert_array_equal(lo, lh)  # (finite keys: the reference's order)
    if huge:
        assert clean_tile.mean() < 0.2
        dc, da = util.make_upstream_grads(W, H, channels=(0, 1, 2, 3, 4, 5, 6))
        gh = util.hip_backward(h, (dc * W * H).numpy(), (da * W * H).numpy())  # (a fault or a hang fails the test here)
        assert gh["dL_dmeans3D"].shape == (P, 3)
        return
    assert 0.2 < clean_tile.mean() < 1.0
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    clean_px = clean_tile[(ys // 16) * gx + xs // 16] & (o["stability"] > 2e-5).reshape(H, W)
    assert np.isfinite(o["color"][:, clean_px]).all() and np.isfinite(o["allmap"][:, clean_px]).all()
    assert np.abs(h["color"] - o["color"])[:, clean_px].max() <= 1e-4
    assert util.allmap_dev(h, o, clean_px).max() <= 1e-4
    # backward: upstream gradients on the clean pixels only (a NaN times a zero gradient is still a NaN: the poisoned tiles
    # would otherwise flood the atomics of every Gaussian they share with a clean tile)
    dc, da = util.make_upstream_grads(W, H, channels=(0, 1, 2, 3, 4, 5, 6))
    dc, da = (dc * W * H).numpy(), (da * W * H).numpy()
    dc[:, ~clean_px] = 0; da[:, ~clean_px] = 0
    go = oracle.backward(o, dc, da)
    gh = util.hip_backward(h, dc, da)  # (a fault or a hang fails the test here)
    # Gaussians all of whose instances lie in clean tiles
    only_clean = np.ones(P, bool)
    for t, (a, b) in enumerate(o["ranges"]):
        if not clean_tile[t]:
            only_clean[o["point_list"][a:b]] = False
    only_clean &= o["radii"] > 0
    assert only_clean.sum() > 100
    for k in ["dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dtransMat"]:
        a, b = gh[k].reshape(P, -1)[only_clean], go[k].reshape(P, -1)[only_clean]
        assert np.isfinite(b).all() and np.isfinite(a).all(), k
        assert util.grad_err(a, b) <= 1e-4, k
