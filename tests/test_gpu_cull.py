"""The cull bits of the forward (64 per instance: the 2x2 pixel groups of the tile a splat can reach) and of the backward
(their ORs over the 4x4 sub-blocks) against a brute-force per-pixel evaluation of the reference's alpha test
(oracle/cull_exact.c).  A missed bit would silently drop a contribution; the cull may only ever be conservative."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exact():
    td = tempfile.mkdtemp()
    so = os.path.join(td, "cull_exact.so")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", os.path.join(ROOT, "oracle", "cull_exact.c"), "-o", so, "-lm"])
    return C.CDLL(so)


def _bits(exact, o, h, W, H):
    R = h["num_rendered"]
    al = lambda n: (n + 255) // 256 * 256
    b = h["buffers"][1].cpu().numpy().tobytes()
    groups = np.frombuffer(b[al(4 * R):al(4 * R) + 8 * R], dtype=np.uint64)      # private layout: point_list, group bits, row bits
    rows = np.frombuffer(b[al(4 * R) + al(8 * R):al(4 * R) + al(8 * R) + 4 * R], dtype=np.uint32)
    p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    keep = [np.ascontiguousarray(o[k]) for k in ("ranges", "point_list", "means2D", "transMats", "normal_opacity")]
    eg, er = np.zeros(R, np.uint64), np.zeros(R, np.uint32)
    exact.exact_group_bits(W, H, *[p(a) for a in keep], p(eg))
    exact.exact_bits(W, H, *[p(a) for a in keep], p(er))
    return groups, rows, eg, er


def _check(exact, oracle, sc, W, H, max_looseness):
    o = util.oracle_forward(oracle, sc)
    h = util.hip_forward(sc, binning="reference")
    np.testing.assert_array_equal(h["point_list"], o["point_list"])
    groups, rows, eg, er = _bits(exact, o, h, W, H)
    pc = lambda a: int(np.unpackbits(a.view(np.uint8)).sum())
    assert np.count_nonzero(eg & ~groups) == 0, "a 2x2 group the splat reaches is missing from the forward's cull bits"
    assert np.count_nonzero(er & ~rows) == 0, "a 4x4 sub-block the splat reaches is missing from the backward's cull bits"
    print(f"group bits: exact {pc(eg)}, cull {pc(groups)} ({pc(groups) / max(pc(eg), 1):.4f}x); row bits: exact {pc(er)}, cull {pc(rows)}")
    assert pc(groups) <= max_looseness * pc(eg) + 64


def test_cull_bits_are_conservative_and_tight(exact, oracle):
    W, H, P = 640, 480, 60000
    _check(exact, oracle, util.make_scene(P, W, H, seed=6, regime="mapping"), W, H, 1.02)


def test_cull_bits_on_extreme_splats(exact, oracle):
    """grazing / huge / tiny / near-plane / barely visible splats: never a missed bit (tightness is not asserted here)."""
    W, H, P = 320, 240, 6000
    sc = util.make_scene(P, W, H, seed=7, regime="mapping")
    g = torch.Generator().manual_seed(1)
    sc["scales"] = sc["scales"] * torch.exp(3.0 * torch.randn(P, 1, generator=g))      # tiny ... huge
    sc["opacities"] = torch.clamp(torch.rand(P, 1, generator=g) ** 4, 1e-3, 0.999)          # many barely visible
    _check(exact, oracle, sc, W, H, 1e9)
