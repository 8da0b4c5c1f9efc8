"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/gs2d_rasterizer.h declares
(no compute calls without a GPU), the operator mirror validates arguments like the reference, and the product
never falls back to the CPU."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hiplib():
    from gaus_slam_amd import build, _lib
    build.build()
    return _lib.lib()


def test_library_exports_every_declared_symbol(hiplib):
    hdr = open(os.path.join(ROOT, "include", "gs2d_rasterizer.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b((?:gs2d|sknn)_[a-z0-9_]+)\s*\(", hdr)) - {"gs2d_alloc_fn"}
    assert {"gs2d_forward", "gs2d_backward", "gs2d_mark_visible", "sknn_dist2"} <= names
    for n in sorted(names):
        assert hasattr(hiplib, n), n
    from gaus_slam_amd import _lib
    assert set(_lib.EXPORTS) == names


def test_layout_queries_are_consistent(hiplib):
    go = (C.c_size_t * 5)()
    hiplib.gs2d_geometry_layout(1000, go)
    assert list(go) == sorted(go) and go[3] - go[2] >= 4000 and hiplib.gs2d_geometry_bytes(1000) > go[4]
    assert hiplib.gs2d_geometry_bytes(500000) >= 500000 * (4 + 4 + 4 + 80 + 3 + 80)
    assert hiplib.gs2d_image_bytes(640, 480) >= 1200 * (8 + 7 * 256 * 4)
    assert hiplib.gs2d_binning_bytes(1000000) >= 1000000 * 24
    assert b"gfx950" in hiplib.gs2d_build_info()


def test_argument_validation_matches_reference():
    from gaus_slam_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    rs = GaussianRasterizationSettings(image_height=8, image_width=8, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3),
                                       scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0,
                                       campos=torch.zeros(3), use_sa=True, prefiltered=False, debug=False)
    r = GaussianRasterizer(rs)
    m, o = torch.zeros(4, 3), torch.zeros(4, 1)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(m, m, o, scales=torch.ones(4, 2), rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(m, m, o, shs=torch.zeros(4, 1, 3), colors_precomp=m, scales=torch.ones(4, 2), rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed"):
        r(m, m, o, colors_precomp=m, scales=torch.ones(4, 2))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed"):
        r(m, m, o, colors_precomp=m, scales=torch.ones(4, 2), rotations=torch.ones(4, 4), cov3D_precomp=torch.ones(4, 9))


def test_no_cpu_fallback():
    """CPU tensors are rejected exactly like the reference's CHECK_INPUT (rasterize_points.cu:27-28)."""
    from gaus_slam_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    rs = GaussianRasterizationSettings(image_height=8, image_width=8, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3),
                                       scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0,
                                       campos=torch.zeros(3), use_sa=True, prefiltered=False, debug=False)
    m, o = torch.zeros(4, 3), torch.zeros(4, 1)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        GaussianRasterizer(rs)(m, m, o, colors_precomp=m, scales=torch.ones(4, 2), rotations=torch.ones(4, 4))
    from gaus_slam_amd.knn import distCUDA2
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        distCUDA2(torch.zeros(10, 3))


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "gaus_slam_amd")
    offenders = []
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle|#include\s+\"[^\"]*oracle", txt, flags=re.M):
                    offenders.append(f)
    for top in ("gaus_2dgs_rasterization", "simple_knn"):
        for f in os.listdir(os.path.join(ROOT, top)):
            if f.endswith(".py") and re.search(r"^\s*(from|import)\s+oracle", open(os.path.join(ROOT, top, f)).read(), flags=re.M):
                offenders.append(f)
    assert not offenders, offenders


def test_dropin_import_surface():
    import gaus_2dgs_rasterization as g
    assert {"image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
            "sh_degree", "campos", "use_sa", "prefiltered", "debug"} == set(g.GaussianRasterizationSettings._fields)
    assert callable(g._C.rasterize_gaussians) and callable(g._C.rasterize_gaussians_backward) and callable(g._C.mark_visible)
    from simple_knn._C import distCUDA2  # noqa: F401


def test_camera_matches_reference_formulas():
    from gaus_slam_amd.scene_synth import intrinsics_for, setup_camera
    W, H = 640, 480
    K = intrinsics_for(W, H)
    cam = setup_camera(W, H, K, torch.eye(4))
    assert cam.tanfovx == pytest.approx(W / (2 * 525.0)) and cam.tanfovy == pytest.approx(H / (2 * 525.0))
    P = cam.projmatrix.t()  # true projection: row 3 = (0,0,1,0) so clip w = view z
    assert torch.allclose(P[3], torch.tensor([0.0, 0.0, 1.0, 0.0]))
    assert P[0, 0] == pytest.approx(2 * 525.0 / W) and P[0, 2] == pytest.approx(-(W - 2 * 319.5) / W)


def test_pixel_state_index_map_is_a_bijection():
    """tests/util.pix_index_map mirrors the forward's lane order (2x2 pixel groups): every pixel of a tile-aligned image
    gets its own slot and whole 64-slot runs belong to one 8x8 quadrant."""
    import numpy as np
    from tests import util
    W, H = 48, 32
    idx = util.pix_index_map(W, H)
    assert sorted(idx.ravel().tolist()) == list(range(W * H))
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    quad_of_pixel = (ys // 8) * (W // 8) + xs // 8
    for run in range(W * H // 64):
        assert len(set(quad_of_pixel[(idx // 64) == run].tolist())) == 1
    # the four lanes of a group are a 2x2 pixel block
    g0 = np.argwhere((idx // 4) == 5)
    assert g0[:, 0].max() - g0[:, 0].min() == 1 and g0[:, 1].max() - g0[:, 1].min() == 1


def test_staged_backward_rejects_bad_stage_masks(hiplib):
    """gs2d_backward_staged validates `stages` before it touches any buffer or the GPU: unknown bits, no stage at all, and
    GS2D_BWD_POSE_4X4 (4) without both stages are errors with a message."""
    nargs = len(hiplib.gs2d_backward_staged.argtypes)

    def call(stages):
        args = [stages, 0, 0] + [None if t in (C.c_void_p, C.c_char_p) else 0 for t in hiplib.gs2d_backward_staged.argtypes[3:]]
        assert len(args) == nargs
        return hiplib.gs2d_backward_staged(*args)

    for bad in (0, 4, 8, 5, 6, 16 | 3):
        assert call(bad) < 0, bad
        assert b"stages" in hiplib.gs2d_last_error() or b"GS2D_BWD_POSE_4X4" in hiplib.gs2d_last_error()


def test_deterministic_backward_refuses_chunks_without_a_forward_record(hiplib):
    """The deterministic backward appends 320 B x R of partial records to the binning chunk, so it only ever runs on a chunk
    whose forward this library recorded (mode, R, size).  With the flag on, a backward on unknown chunks must fail with a
    message BEFORE anything is launched or written -- checkable without a GPU because the pointers are never touched."""
    argt = hiplib.gs2d_backward_staged.argtypes
    names_at = {3: 1000, 6: 5000, 8: 64, 9: 48}  # P, R, width, height
    args = [1, 0, 0] + [(None if t in (C.c_void_p, C.c_char_p) else 0) for t in argt[3:]]
    for k, v in names_at.items():
        args[k] = v
    fake = (C.c_char * 64)()
    addr = C.addressof(fake)
    # geom_buffer, binning_buffer, img_buffer: positions of the three char* arguments
    char_ps = [i for i, t in enumerate(argt) if t is C.c_char_p]
    ptr_args = list(args)
    if char_ps:
        for i in char_ps:
            ptr_args[i] = C.cast(addr, C.c_char_p)
    else:  # all pointers are void*: geom/binning/img follow `radii`
        base = 3 + 20
        for i in range(base, base + 3):
            ptr_args[i] = addr
    hiplib.gs2d_set_deterministic(1)
    try:
        assert hiplib.gs2d_backward_staged(*ptr_args) < 0
        assert b"without a forward record" in hiplib.gs2d_last_error()
    finally:
        hiplib.gs2d_set_deterministic(0)


def test_debug_mode_dumps_the_arguments_of_a_failing_call(tmp_path, monkeypatch):
    """debug=True mirrors RAST/gaus_2dgs_rasterization/__init__.py:84-91: a host copy of the arguments is taken before the
    call and written to snapshot_fw.dump when the call raises (here: CPU tensors, rejected like the reference's CHECK_INPUT)."""
    from gaus_slam_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    monkeypatch.chdir(tmp_path)
    P = 5
    rs = GaussianRasterizationSettings(image_height=16, image_width=16, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3), scale_modifier=1.0,
                                       viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0, campos=torch.zeros(3),
                                       use_sa=True, prefiltered=False, debug=True)
    means = torch.rand(P, 3)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        GaussianRasterizer(rs)(means, torch.zeros(P, 3), torch.rand(P, 1), colors_precomp=torch.rand(P, 3), scales=torch.rand(P, 2),
                               rotations=torch.rand(P, 4))
    dump = torch.load(tmp_path / "snapshot_fw.dump", weights_only=True)
    assert torch.equal(dump[1], means) and dump[-1] is True


def test_build_info_carries_the_source_hash(hiplib):
    """gs2d_build_info() names the hash of the kernel sources the library was compiled from (gaus_slam_amd/build.py:
    csrc/* + the C-ABI header); for the in-tree library that is the hash of the tree."""
    from gaus_slam_amd import build, _lib
    if os.environ.get("GS2D_LIB_PATH"):
        pytest.skip("an experiment library is loaded")
    h = build.source_hash()
    assert re.fullmatch(r"[0-9a-f]{16}", h)
    assert _lib.lib_source_hash() == h, (_lib.build_info(), h)


def test_kept_artifacts_of_this_round_come_from_the_trees_kernels():
    """VERDICT r3, hygiene: one build, one set of artifacts.  Every JSON kept under profiles/ for the current round (name
    contains _r04) says which kernel sources produced it (bench.py's `build.source_hash`, the PMC summaries' `source_hash`),
    and that hash is the hash of the sources in this tree -- a kernel change after the artifacts were taken fails here until
    scripts/gpu_round4_artifacts.sh has run again."""
    import glob
    import json
    from gaus_slam_amd import build
    h = build.source_hash()
    stale = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_r04*.json"))):
        with open(path) as fh:
            d = json.loads(fh.read().strip().splitlines()[-1])
        got = (d.get("build") or {}).get("source_hash") or d.get("source_hash")
        if got != h:
            stale.append((os.path.basename(path), got))
    assert not stale, f"artifacts not from the tree's kernels ({h}): {stale}"
