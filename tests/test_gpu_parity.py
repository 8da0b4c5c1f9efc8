"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Tolerances (BASELINE.json north_star): rendered colour/depth within 1e-4 L-inf; radii, sorted point_list and
tile ranges bit-exact.  Gradients are float atomics sums in the reference (run-to-run non-deterministic), so
they are compared relative to each tensor's magnitude (1e-4).
Pixels whose discrete decisions sit within 2e-5 (relative) of a threshold in the oracle ("knife-edge",
oracle.stability) may legitimately flip under 1-ulp differences of expf; they are excluded from the plain L-inf
check, their count is bounded, and each of them must equal the oracle's pixel under ONE outcome of those
near-threshold decisions (util.check_knife_pixels / oracle.pixel_variants).  In the backward comparison they
receive zero upstream gradient."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

IMG_TOL = 1e-4
GRAD_TOL = 1e-4
KNIFE = 2e-5


def _compare_forward(o, h, W, H, oracle=None, knife_frac=2e-3):
    assert h["num_rendered"] == o["num_rendered"]
    np.testing.assert_array_equal(h["radii"], o["radii"])
    vis = o["radii"] > 0
    np.testing.assert_array_equal(h["tiles_touched"], o["tiles_touched"])
    np.testing.assert_array_equal(h["point_offsets"], o["point_offsets"])
    np.testing.assert_array_equal(h["depths"][vis].view(np.uint32), o["depths"][vis].view(np.uint32))
    np.testing.assert_array_equal(h["rec"][vis][:, [0, 1, 2, 4, 5, 6, 8, 9, 10]].view(np.uint32),
                                  o["transMats"][vis].view(np.uint32))
    np.testing.assert_array_equal(h["rec"][vis][:, [3, 7]].view(np.uint32), o["means2D"][vis].view(np.uint32))
    np.testing.assert_array_equal(h["keys"], o["keys"])
    np.testing.assert_array_equal(h["point_list"], o["point_list"])
    np.testing.assert_array_equal(h["ranges"], o["ranges"])
    stable = (o["stability"] > KNIFE).reshape(H, W)
    assert (~stable).mean() < knife_frac  # (a sanity bound on the scene, not on the kernels)
    HW = H * W
    np.testing.assert_array_equal(h["last_contributor"][stable], o["n_contrib"][:HW].reshape(H, W)[stable])
    np.testing.assert_array_equal(h["median_contributor"][stable], o["n_contrib"][HW:].reshape(H, W)[stable])
    dc = np.abs(h["color"] - o["color"])[:, stable].max()
    da = util.allmap_dev(h, o, stable)  # (incl. the conditioning allowance of use_sa's depth channels)
    assert dc <= IMG_TOL, dc
    assert (da <= IMG_TOL).all(), da
    if oracle is not None:  # the excluded pixels must equal the oracle under one outcome of their near-threshold decisions
        util.check_knife_pixels(oracle, o, h, stable, IMG_TOL, KNIFE)
    return stable


@pytest.mark.parametrize("regime", ["tracking", "mapping"])
@pytest.mark.parametrize("use_sa", [True, False])
@pytest.mark.parametrize("P,W,H", [(256, 160, 120), (4000, 320, 240), (3000, 150, 100)])
def test_forward_backward_parity(oracle, regime, use_sa, P, W, H):
    sc = util.make_scene(P, W, H, seed=0, regime=regime)
    bg = (0.2, 0.5, 0.1)
    o = util.oracle_forward(oracle, sc, use_sa=use_sa, bg=bg)
    h = util.hip_forward(sc, use_sa=use_sa, bg=bg)
    stable = _compare_forward(o, h, W, H, oracle)
    dc, da = util.make_upstream_grads(W, H, channels=(0, 1, 2, 3, 4, 5, 6))
    dc = (dc * W * H).numpy(); da = (da * W * H).numpy()
    # knife-edge pixels get no upstream gradient so a flipped decision cannot leak into the comparison
    dc[:, ~stable] = 0; da[:, ~stable] = 0
    go = oracle.backward(o, dc, da)
    gh = util.hip_backward(h, dc, da)
    for k in ["dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dtransMat", "dL_dscales", "dL_drotations", "dL_dmeans2D"]:
        assert util.grad_err(gh[k], go[k].reshape(gh[k].shape)) <= GRAD_TOL, k
