#!/usr/bin/env python3
"""Generates the committed golden fixtures (tests/golden/*.npz) from the CPU oracle.

The reference cannot be executed in this environment (CUDA-only, no tests or golden vectors of its own --
SURVEY.md section 4 / section 8(c)), so these vectors are produced by oracle/gs2d_oracle.c, itself pinned by the pure-PyTorch
restatement and autograd (tests/test_oracle.py).  Each file holds the inputs and every output/intermediate
the parity tests compare.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import gs2d_oracle as orc  # noqa: E402
from tests import util  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [
    # name, P, W, H, regime, use_sa, kind
    ("a_tracking_sa", 200, 96, 64, "tracking", True, "rgb"),
    ("a_mapping_sa", 200, 96, 64, "mapping", True, "rgb"),
    ("a_mapping_nosa", 200, 96, 64, "mapping", False, "rgb"),
    ("ragged_tracking_nosa", 150, 70, 50, "tracking", False, "rgb"),
    ("sh3_mapping_sa", 120, 64, 48, "mapping", True, "sh3"),
    ("precomp_tracking_sa", 120, 64, 48, "tracking", True, "precomp"),
]


def make_case(name, P, W, H, regime, use_sa, kind, seed=7):
    sc = util.make_scene(P, W, H, seed=seed, regime=regime)
    rng = np.random.default_rng(seed + 1)
    bg = np.array([0.1, 0.3, 0.6], np.float32)
    extra = {}
    kw = {}
    if kind == "sh3":
        shs = (rng.normal(0, 0.4, (P, 16, 3))).astype(np.float32)
        shs[:, 0] += 1.0
        kw.update(shs=shs, sh_degree=3)
        extra["shs"] = shs
    if kind == "precomp":
        st0 = util.oracle_forward(orc, sc, use_sa=use_sa, bg=bg)
        tm = st0["transMats"].copy()
        tm[st0["radii"] == 0] = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1], np.float32)
        kw.update(transMat_precomp=tm)
        extra["transMat_precomp"] = tm
    st = util.oracle_forward(orc, sc, use_sa=use_sa, bg=bg, **kw)
    dc, da = util.make_upstream_grads(W, H, seed=seed + 2, channels=(0, 1, 2, 3, 4, 5, 6))
    dc = (dc * W * H).numpy()
    da = (da * W * H).numpy()
    g = orc.backward(st, dc, da)
    cam = sc["cam"]
    out = dict(
        P=P, W=W, H=H, use_sa=use_sa, kind=kind, bg=bg, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
        means3D=sc["means3D"].numpy(), scales=sc["scales"].numpy(), rotations=sc["rotations"].numpy(),
        opacities=sc["opacities"].numpy(), colors=sc["colors"].numpy(), viewmatrix=cam.viewmatrix.numpy(),
        projmatrix=cam.projmatrix.numpy(), campos=cam.campos.numpy(), dL_dcolor=dc, dL_dallmap=da,
        color=st["color"], allmap=st["allmap"], radii=st["radii"], num_rendered=st["num_rendered"],
        point_list=st["point_list"], ranges=st["ranges"], n_contrib=st["n_contrib"], stability=st["stability"],
        **extra, **{k: v for k, v in g.items() if not k.endswith("_blend")})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    return out


if __name__ == "__main__":
    orc.set_threads(1)
    for c in CASES:
        o = make_case(*c)
        print(c[0], "R =", o["num_rendered"], "min stability", float(o["stability"].min()))
