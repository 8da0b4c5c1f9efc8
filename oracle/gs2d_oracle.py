"""ctypes/numpy front-end of the CPU ORACLE (oracle/gs2d_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The shipped package (gaus_slam_amd/) never
imports this module.  Parity status: "parity unpinned" against the reference
binary (see the header of gs2d_oracle.c and DESIGN.md).

The call structure mirrors the reference stage driver
RAST/cuda_rasterizer/rasterizer_impl.cu:201-350 (forward) and :354-460
(backward), and the tensor conventions of RAST/rasterize_points.cu:39-239.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgs2d_oracle.so")
_lib = None

TILE = 16


def build(force=False):
    """Compile the C oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("gs2d_oracle.c", "gs2d_oracle_f64.c", "gs2d_oracle_forms.c", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libgs2d_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_preprocess.restype = C.c_int64
        _lib.orc_higher_msb.restype = C.c_uint32
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be contiguous"
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def higher_msb(n):
    return int(lib().orc_higher_msb(C.c_uint32(n)))


def set_float_accumulation(on):
    """orc_set_float_accumulation: per-Gaussian gradient sums in float32 (the reference's float atomics) instead of double."""
    lib().orc_set_float_accumulation(C.c_int(int(bool(on))))


def set_threads(n):
    lib().orc_set_threads(C.c_int(int(n)))


def forward(means3D, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy,
            scales=None, rotations=None, colors_precomp=None, shs=None, sh_degree=0,
            transMat_precomp=None, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, use_sa=True,
            want_stability=True):
    """Full forward.  Returns a dict with outputs and every intermediate.

    viewmatrix / projmatrix: flat 16 floats, column-major (the transposed
    matrices render/render_2dgs.py:10-24 builds)."""
    L = lib()
    means3D = _f32(means3D).reshape(-1, 3)
    P = means3D.shape[0]
    opacities = _f32(opacities).reshape(-1)
    scales = _f32(scales)
    rotations = _f32(rotations)
    colors_precomp = _f32(colors_precomp)
    transMat_precomp = _f32(transMat_precomp)
    shs = _f32(shs)
    M = 0 if shs is None else shs.shape[1]
    vm = _f32(viewmatrix).reshape(16)
    pm = _f32(projmatrix).reshape(16)
    campos = _f32(campos).reshape(3)
    bg = _f32(np.asarray(bg)).reshape(3)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    HW = H * W
    st = dict(P=P, W=W, H=H, M=M, D=sh_degree, use_sa=bool(use_sa), tanfovx=float(tanfovx),
              tanfovy=float(tanfovy), scale_modifier=float(scale_modifier),
              means3D=means3D, opacities=opacities, scales=scales, rotations=rotations,
              colors_precomp=colors_precomp, transMat_precomp=transMat_precomp, shs=shs,
              viewmatrix=vm, projmatrix=pm, campos=campos, bg=bg)
    radii = np.zeros(P, np.int32)
    means2D = np.zeros((P, 2), np.float32)
    depths = np.zeros(P, np.float32)
    transMats = np.zeros((P, 9), np.float32)
    rgb = np.zeros((P, 3), np.float32)
    normal_opacity = np.zeros((P, 4), np.float32)
    tiles_touched = np.zeros(P, np.uint32)
    clamped = np.zeros((P, 3), np.uint8)
    point_offsets = np.zeros(P, np.uint32)
    R = 0
    if P > 0:
        R = int(L.orc_preprocess(
            C.c_int(P), C.c_int(sh_degree), C.c_int(M), _p(means3D), _p(scales),
            C.c_float(scale_modifier), _p(rotations), _p(opacities), _p(shs), _p(transMat_precomp),
            _p(colors_precomp), _p(vm), _p(pm), _p(campos), C.c_int(W), C.c_int(H),
            _p(radii), _p(means2D), _p(depths), _p(transMats), _p(rgb), _p(normal_opacity),
            _p(tiles_touched), _p(clamped), _p(point_offsets)))
    keys_unsorted = np.zeros(R, np.uint64)
    vals_unsorted = np.zeros(R, np.uint32)
    keys = np.zeros(R, np.uint64)
    point_list = np.zeros(R, np.uint32)
    ranges = np.zeros((gx * gy, 2), np.uint32)
    nbits = 32 + higher_msb(gx * gy)
    if P > 0:
        L.orc_duplicate(C.c_int(P), _p(means2D), _p(depths), _p(point_offsets), _p(radii),
                        C.c_int(W), C.c_int(H), _p(keys_unsorted), _p(vals_unsorted))
        L.orc_sort_pairs(C.c_int64(R), _p(keys_unsorted), _p(vals_unsorted), _p(keys), _p(point_list),
                         C.c_int(nbits))
        L.orc_tile_ranges(C.c_int64(R), _p(keys), C.c_int(gx * gy), _p(ranges))
    out_color = np.zeros((3, H, W), np.float32)
    out_others = np.zeros((7, H, W), np.float32)
    final_T = np.zeros(3 * HW, np.float32)
    n_contrib = np.zeros(2 * HW, np.uint32)
    median_depth = np.zeros(HW, np.float32)
    depth_std = np.zeros(HW, np.float32)
    stab = np.concatenate([np.full(HW, 1e30, np.float32), np.zeros(HW, np.float32)]) if want_stability else None
    features = colors_precomp if colors_precomp is not None else rgb
    tm = transMat_precomp if transMat_precomp is not None else transMats
    if P > 0:
        L.orc_blend_fwd(C.c_int(W), C.c_int(H), _p(ranges), _p(point_list), _p(means2D), _p(features),
                        _p(tm), _p(normal_opacity), _p(bg), C.c_int(int(use_sa)),
                        _p(out_color), _p(out_others), _p(final_T), _p(n_contrib),
                        _p(median_depth), _p(depth_std), _p(stab))
    st.update(num_rendered=R, radii=radii, means2D=means2D, depths=depths, transMats=transMats, rgb=rgb,
              normal_opacity=normal_opacity, tiles_touched=tiles_touched, clamped=clamped,
              point_offsets=point_offsets, keys_unsorted=keys_unsorted, vals_unsorted=vals_unsorted,
              keys=keys, point_list=point_list, ranges=ranges, nbits=nbits,
              color=out_color, allmap=out_others, final_T=final_T, n_contrib=n_contrib,
              median_depth=median_depth, depth_std=depth_std, stability=None if stab is None else stab[:HW],
              sa_amp=None if stab is None else stab[HW:])
    return st


def reblend(st, ranges, point_list, want_stability=True):
    """The blend of forward() run again on OTHER per-tile lists (same preprocessed splats): returns a copy of the state
    with ranges / point_list replaced and every blend output recomputed.  Used to show that lists with non-contributing
    instances removed give the same image bit for bit (tests/test_gpu_footprint.py)."""
    L = lib()
    W, H = st["W"], st["H"]
    HW = H * W
    ranges = np.ascontiguousarray(ranges, dtype=np.uint32).reshape(-1, 2)
    point_list = np.ascontiguousarray(point_list, dtype=np.uint32)
    assert ranges.shape == st["ranges"].shape
    out_color = np.zeros((3, H, W), np.float32)
    out_others = np.zeros((7, H, W), np.float32)
    final_T = np.zeros(3 * HW, np.float32)
    n_contrib = np.zeros(2 * HW, np.uint32)
    median_depth = np.zeros(HW, np.float32)
    depth_std = np.zeros(HW, np.float32)
    stab = np.concatenate([np.full(HW, 1e30, np.float32), np.zeros(HW, np.float32)]) if want_stability else None
    features = st["colors_precomp"] if st["colors_precomp"] is not None else st["rgb"]
    tm = st["transMat_precomp"] if st["transMat_precomp"] is not None else st["transMats"]
    if st["P"] > 0:
        L.orc_blend_fwd(C.c_int(W), C.c_int(H), _p(ranges), _p(point_list), _p(st["means2D"]), _p(features),
                        _p(tm), _p(st["normal_opacity"]), _p(st["bg"]), C.c_int(int(st["use_sa"])),
                        _p(out_color), _p(out_others), _p(final_T), _p(n_contrib),
                        _p(median_depth), _p(depth_std), _p(stab))
    new = dict(st)
    new.update(num_rendered=int(len(point_list)), ranges=ranges, point_list=point_list, color=out_color, allmap=out_others,
               final_T=final_T, n_contrib=n_contrib, median_depth=median_depth, depth_std=depth_std,
               stability=None if stab is None else stab[:HW], sa_amp=None if stab is None else stab[HW:])
    return new


def pixel_variants(st, px, py, knife, max_decisions=6):
    """All outcomes of the forward blend at pixel (px, py) when the decisions within `knife` (relative) of their threshold
    are flipped in every combination (orc_blend_fwd_pixel).  Returns (n_decisions, list of dicts color[3], others[7],
    last_contributor, median_contributor, final_T); n_decisions > max_decisions returns only the unflipped variant."""
    L = lib()
    L.orc_blend_fwd_pixel.restype = C.c_int
    features = st["colors_precomp"] if st["colors_precomp"] is not None else st["rgb"]
    tm = st["transMat_precomp"] if st["transMat_precomp"] is not None else st["transMats"]

    def one(mask):
        out = np.zeros(13, np.float32)
        nk = L.orc_blend_fwd_pixel(C.c_int(st["W"]), C.c_int(st["H"]), C.c_int(int(px)), C.c_int(int(py)), _p(st["ranges"]),
                                   _p(st["point_list"]), _p(st["means2D"]), _p(features), _p(tm), _p(st["normal_opacity"]),
                                   _p(st["bg"]), C.c_int(int(st["use_sa"])), C.c_float(knife), C.c_uint32(mask), _p(out))
        return nk, dict(color=out[0:3].copy(), others=out[3:10].copy(), last_contributor=int(out[10]),
                        median_contributor=int(out[11]), final_T=float(out[12]))

    nk, base = one(0)
    base["mask"] = 0
    if nk == 0 or nk > max_decisions:
        return nk, [base]
    out = [base]
    for m in range(1, 1 << nk):
        v = one(m)[1]
        v["mask"] = m
        out.append(v)
    return nk, out


FORM_EXP2, FORM_RCP, FORM_MERGED, FORM_CLOSED, FORM_CONF, FORM_EXPAND = 1, 2, 4, 8, 16, 32  # gs2d_oracle_forms.c


def backward(st, dL_dcolor, dL_dallmap, pixel_overrides=None, knife=0.0, forms=None):
    """Backward for a forward() state.  Returns the 8 gradients of
    RAST/rasterize_points.cu:238 (+ the internal dL_dnormal).
    pixel_overrides (optional): list of (px, py, flipmask) -- those pixels take part under the given outcome of their
    near-threshold decisions (the numbering of pixel_variants, same `knife`) instead of the oracle's own outcome: their
    upstream gradient is removed from the plain pass and their contribution comes from orc_blend_bwd_pixel.
    forms (optional, bit set of FORM_*): the blend stage in float32 with the HIP kernel's algebraic rewrites switched on one
    by one (gs2d_oracle_forms.c; 0 = this oracle's own operation order) -- for pricing each rewrite against backward_f64."""
    L = lib()
    P, W, H, M = st["P"], st["W"], st["H"], st["M"]
    dL_dcolor = _f32(dL_dcolor).reshape(3, H, W)
    dL_dallmap = _f32(dL_dallmap).reshape(7, H, W)
    extra = None
    if pixel_overrides and P > 0:
        features0 = st["colors_precomp"] if st["colors_precomp"] is not None else st["rgb"]
        tm0 = st["transMat_precomp"] if st["transMat_precomp"] is not None else st["transMats"]
        extra = np.zeros((P, 20), np.float64)
        dL_dcolor, dL_dallmap = dL_dcolor.copy(), dL_dallmap.copy()
        L.orc_blend_bwd_pixel.restype = C.c_int
        for (x, y, mask) in pixel_overrides:
            gp = np.ascontiguousarray(dL_dcolor[:, y, x]); go = np.ascontiguousarray(dL_dallmap[:, y, x])
            L.orc_blend_bwd_pixel(C.c_int(W), C.c_int(H), C.c_int(int(x)), C.c_int(int(y)), _p(st["ranges"]), _p(st["point_list"]),
                                  _p(st["means2D"]), _p(features0), _p(tm0), _p(st["normal_opacity"]), _p(st["bg"]),
                                  C.c_int(int(st["use_sa"])), C.c_float(knife), C.c_uint32(int(mask)), _p(gp), _p(go), _p(extra))
            dL_dcolor[:, y, x] = 0
            dL_dallmap[:, y, x] = 0
    g = dict(
        dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dmeans2D=np.zeros((P, 3), np.float32),
        dL_dcolors=np.zeros((P, 3), np.float32), dL_dnormal=np.zeros((P, 3), np.float32),
        dL_dopacity=np.zeros((P, 1), np.float32), dL_dtransMat=np.zeros((P, 9), np.float32),
        dL_dsh=np.zeros((P, M, 3), np.float32), dL_dscales=np.zeros((P, 2), np.float32),
        dL_drotations=np.zeros((P, 4), np.float32))
    if P == 0:
        return g
    features = st["colors_precomp"] if st["colors_precomp"] is not None else st["rgb"]
    tm = st["transMat_precomp"] if st["transMat_precomp"] is not None else st["transMats"]
    if forms is not None:
        assert extra is None
        L.orc_blend_bwd_forms(C.c_int(int(forms)), C.c_int(P), C.c_int(W), C.c_int(H), _p(st["ranges"]), _p(st["point_list"]),
                              _p(st["bg"]), _p(st["means2D"]), _p(st["normal_opacity"]), _p(tm), _p(features),
                              _p(st["final_T"]), _p(st["n_contrib"]), _p(dL_dcolor), _p(dL_dallmap),
                              _p(st["median_depth"]), _p(st["depth_std"]), C.c_int(int(st["use_sa"])),
                              _p(g["dL_dtransMat"]), _p(g["dL_dmeans2D"]), _p(g["dL_dnormal"]),
                              _p(g["dL_dopacity"]), _p(g["dL_dcolors"]))
    else:
        L.orc_blend_bwd(C.c_int(P), C.c_int(W), C.c_int(H), _p(st["ranges"]), _p(st["point_list"]),
                        _p(st["bg"]), _p(st["means2D"]), _p(st["normal_opacity"]), _p(tm), _p(features),
                        _p(st["final_T"]), _p(st["n_contrib"]), _p(dL_dcolor), _p(dL_dallmap),
                        _p(st["median_depth"]), _p(st["depth_std"]), C.c_int(int(st["use_sa"])),
                        _p(g["dL_dtransMat"]), _p(g["dL_dmeans2D"]), _p(g["dL_dnormal"]),
                        _p(g["dL_dopacity"]), _p(g["dL_dcolors"]), _p(extra))
    g["dL_dtransMat_blend"] = g["dL_dtransMat"].copy()
    g["dL_dmeans2D_blend"] = g["dL_dmeans2D"].copy()
    L.orc_preprocess_bwd(C.c_int(P), C.c_int(st["D"]), C.c_int(M), _p(st["means3D"]), _p(tm),
                         _p(st["radii"]), _p(st["shs"]), _p(st["clamped"]), _p(st["scales"]),
                         _p(st["rotations"]), C.c_float(st["scale_modifier"]), _p(st["viewmatrix"]),
                         _p(st["projmatrix"]), C.c_int(W), C.c_int(H), C.c_float(st["tanfovx"]),
                         C.c_float(st["tanfovy"]), _p(st["campos"]),
                         _p(g["dL_dtransMat"]), _p(g["dL_dnormal"]), _p(g["dL_dcolors"]), _p(g["dL_dsh"]),
                         _p(g["dL_dmeans2D"]), _p(g["dL_dmeans3D"]), _p(g["dL_dscales"]),
                         _p(g["dL_drotations"]))
    return g


def backward_f64(st, dL_dcolor, dL_dallmap):
    """The backward of a forward() state evaluated in FLOAT64 on the float32 paths' inputs and discrete decisions
    (oracle/gs2d_oracle_f64.c): the yardstick that tells a float32 path's rounding error from a formula difference.
    colors_precomp path only.  Returns float64 arrays: dL_dmeans3D, dL_dscales, dL_drotations, dL_dopacity, dL_dcolors,
    dL_dnormal and the blend-stage dL_dtransMat_blend / dL_dmeans2D_blend."""
    L = lib()
    P, W, H = st["P"], st["W"], st["H"]
    assert st["shs"] is None, "backward_f64 covers the colors_precomp path"
    dL_dcolor = _f32(dL_dcolor).reshape(3, H, W)
    dL_dallmap = _f32(dL_dallmap).reshape(7, H, W)
    z = lambda *s: np.zeros(s, np.float64)
    g = dict(dL_dmeans3D=z(P, 3), dL_dmeans2D_blend=z(P, 3), dL_dcolors=z(P, 3), dL_dnormal=z(P, 3), dL_dopacity=z(P, 1),
             dL_dtransMat=z(P, 9), dL_dscales=z(P, 2), dL_drotations=z(P, 4))
    if P == 0:
        return g
    tm = st["transMat_precomp"] if st["transMat_precomp"] is not None else st["transMats"]
    L.orc_blend_bwd_f64(C.c_int(P), C.c_int(W), C.c_int(H), _p(st["ranges"]), _p(st["point_list"]), _p(st["bg"]),
                        _p(st["means2D"]), _p(st["normal_opacity"]), _p(tm), _p(st["colors_precomp"]), _p(st["final_T"]),
                        _p(st["n_contrib"]), _p(dL_dcolor), _p(dL_dallmap), _p(st["median_depth"]), _p(st["depth_std"]),
                        C.c_int(int(st["use_sa"])), _p(g["dL_dtransMat"]), _p(g["dL_dmeans2D_blend"]), _p(g["dL_dnormal"]),
                        _p(g["dL_dopacity"]), _p(g["dL_dcolors"]))
    g["dL_dtransMat_blend"] = g["dL_dtransMat"].copy()
    L.orc_preprocess_bwd_f64(C.c_int(P), _p(st["means3D"]), _p(tm), _p(st["radii"]), _p(st["scales"]), _p(st["rotations"]),
                             _p(st["viewmatrix"]), _p(st["projmatrix"]), C.c_int(W), C.c_int(H), C.c_float(st["tanfovx"]),
                             C.c_float(st["tanfovy"]), _p(g["dL_dtransMat"]), _p(g["dL_dnormal"]), _p(g["dL_dmeans2D_blend"]),
                             _p(g["dL_dmeans3D"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g


def mark_visible(means3D, viewmatrix):
    means3D = _f32(means3D).reshape(-1, 3)
    out = np.zeros(means3D.shape[0], np.uint8)
    if means3D.shape[0]:
        lib().orc_mark_visible(C.c_int(means3D.shape[0]), _p(means3D), _p(_f32(viewmatrix).reshape(16)), _p(out))
    return out.astype(bool)


def dist2_knn3(points):
    """simple_knn.distCUDA2 semantics (mean squared distance to the 3 nearest other points)."""
    pts = _f32(points).reshape(-1, 3)
    out = np.zeros(pts.shape[0], np.float32)
    if pts.shape[0]:
        lib().orc_dist2_knn3(C.c_int(pts.shape[0]), _p(pts), _p(out))
    return out


# ---------------------------------------------------------------------------------------------------------------
# Tracking-regime composition (gs2d_forward_posed / gs2d_backward_posed): the reference does this part in PyTorch
# (render/__init__.py:31-40).  Expression order below is the bit-exact contract with the HIP preprocess kernels.
def compose_pose(means3D, rotations, pose_Rt, pose_q):
    """means_cam = R x + t ; rot = standardize(q_cam (x) q)   (float32, fixed association order)."""
    f = np.float32
    x, y, z = (means3D[:, i].astype(f) for i in range(3))
    Rt = np.asarray(pose_Rt, f).reshape(3, 4)
    cam = np.stack([((Rt[r, 0] * x + Rt[r, 1] * y) + Rt[r, 2] * z) + Rt[r, 3] for r in range(3)], 1).astype(f)
    aw, ax, ay, az = (f(v) for v in np.asarray(pose_q, f))
    bw, bx, by, bz = (rotations[:, i].astype(f) for i in range(4))
    ow = ((aw * bw - ax * bx) - ay * by) - az * bz
    ox = ((aw * bx + ax * bw) + ay * bz) - az * by
    oy = ((aw * by - ax * bz) + ay * bw) + az * bx
    oz = ((aw * bz + ax * by) - ay * bx) + az * bw
    q = np.stack([ow, ox, oy, oz], 1).astype(f)
    sign = np.where(q[:, 0] < 0, f(-1), f(1)).astype(f)
    return cam, (q * sign[:, None]).astype(f), sign


def forward_posed(means3D, rotations, pose_Rt, pose_q, *args, **kw):
    cam, q, sign = compose_pose(_f32(means3D).reshape(-1, 3), _f32(rotations).reshape(-1, 4), pose_Rt, pose_q)
    st = forward(cam, *args, rotations=q, **kw)
    st["pose_world_means"] = _f32(means3D).reshape(-1, 3)
    st["pose_Rt"] = np.asarray(pose_Rt, np.float32).reshape(3, 4)
    st["pose_q"] = np.asarray(pose_q, np.float32).reshape(4)
    st["pose_sign"] = sign
    return st


def backward_posed(st, dL_dcolor, dL_dallmap):
    """Backward of forward_posed: the plain backward in the camera frame, then
    dL/dR = sum g (x) x, dL/dt = sum g, dL/dx = R^T g, dL/dq = sign * L(q_cam)^T dL/dq'  (double accumulation)."""
    g = backward(st, dL_dcolor, dL_dallmap)
    gc = g["dL_dmeans3D"].astype(np.float64)
    x = st["pose_world_means"].astype(np.float64)
    Rt = st["pose_Rt"].astype(np.float64)
    dpose = np.zeros((3, 4))
    dpose[:, :3] = gc.T @ x
    dpose[:, 3] = gc.sum(0)
    g["dL_dpose"] = dpose.astype(np.float32)
    g["dL_dmeans3D_cam"] = g["dL_dmeans3D"]
    g["dL_dmeans3D"] = (gc @ Rt[:, :3]).astype(np.float32)
    aw, ax, ay, az = st["pose_q"].astype(np.float64)
    Lm = np.array([[aw, -ax, -ay, -az], [ax, aw, -az, ay], [ay, az, aw, -ax], [az, -ay, ax, aw]])
    g["dL_drotations"] = ((g["dL_drotations"].astype(np.float64) @ Lm) * st["pose_sign"][:, None]).astype(np.float32)
    return g
