"""Shared helpers for the parity tests: run the same seeded scene through the CPU oracle and through the HIP
library (via the C ABI wrappers in gaus_slam_amd.rasterizer) and expose comparable views of both."""
import ctypes as C

import numpy as np
import torch

from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads  # noqa: F401

TILE = 16


def oracle_forward(orc, sc, use_sa=True, bg=(0.0, 0.0, 0.0), shs=None, sh_degree=0, transMat_precomp=None,
                   scale_modifier=1.0):
    cam = sc["cam"]
    kw = {}
    if transMat_precomp is None:
        kw.update(scales=sc["scales"].numpy(), rotations=sc["rotations"].numpy())
    else:
        kw.update(transMat_precomp=transMat_precomp)
    if shs is None:
        kw.update(colors_precomp=sc["colors"].numpy())
    else:
        kw.update(shs=shs, sh_degree=sh_degree)
    return orc.forward(sc["means3D"].numpy(), sc["opacities"].numpy(), cam.viewmatrix.numpy(), cam.projmatrix.numpy(),
                       cam.campos.numpy(), cam.W, cam.H, cam.tanfovx, cam.tanfovy, bg=bg, use_sa=use_sa,
                       scale_modifier=scale_modifier, **kw)


def pix_index_map(W, H):
    """state index (tile*256 + quadrant*64 + group*4 + (y%2)*2 + x%2, group = the 2x2 pixel group of the 8x8 quadrant,
    row-major 4x4) for every pixel, as [H,W] int64 (see include/gs2d_rasterizer.h)."""
    gx = (W + TILE - 1) // TILE
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    tile = (ys // TILE) * gx + (xs // TILE)
    ly, lx = ys % TILE, xs % TILE
    quad = (ly // 8) * 2 + (lx // 8)
    group = ((ly % 8) // 2) * 4 + (lx % 8) // 2
    return tile * 256 + quad * 64 + group * 4 + (ly % 2) * 2 + (lx % 2)


def hip_forward(sc, use_sa=True, bg=(0.0, 0.0, 0.0), shs=None, sh_degree=0, transMat_precomp=None, scale_modifier=1.0,
                debug=False, device="cuda", binning="reference"):
    """Calls the C-ABI forward through gaus_slam_amd.rasterizer.rasterize_gaussians and unpacks the private
    scratch layout for comparison.  binning: "reference" = the reference's 3-sigma tile rectangles (gs2d_set_reference_binning,
    what the bit-exact list comparisons need), "footprint" = the library's default (tests/test_gpu_footprint.py)."""
    from gaus_slam_amd import rasterizer
    assert binning in ("reference", "footprint")
    rasterizer.set_reference_binning(binning == "reference")
    try:
        return _hip_forward(sc, use_sa, bg, shs, sh_degree, transMat_precomp, scale_modifier, debug, device)
    finally:
        rasterizer.set_reference_binning(False)  # the library default


def _hip_forward(sc, use_sa, bg, shs, sh_degree, transMat_precomp, scale_modifier, debug, device):
    from gaus_slam_amd import _lib, rasterizer
    cam = sc["cam"]
    dev = torch.device(device)
    e = torch.empty(0, dtype=torch.float32, device=dev)
    t = lambda a: (torch.as_tensor(a).float().to(dev).contiguous() if a is not None else e)
    means3D, opac = t(sc["means3D"]), t(sc["opacities"])
    colors = t(sc["colors"]) if shs is None else e
    sh = t(shs) if shs is not None else e
    scales = t(sc["scales"]) if transMat_precomp is None else e
    rots = t(sc["rotations"]) if transMat_precomp is None else e
    tm = t(transMat_precomp) if transMat_precomp is not None else e
    args = (t(np.asarray(bg, np.float32)), means3D, colors, opac, scales, rots, scale_modifier, tm,
            t(cam.viewmatrix), t(cam.projmatrix), cam.tanfovx, cam.tanfovy, cam.H, cam.W, sh, sh_degree,
            t(cam.campos), use_sa, False, debug)
    R, color, allmap, radii, geom, binning, img = rasterizer.rasterize_gaussians(*args)
    torch.cuda.synchronize()
    L = _lib.lib()
    P, W, H = means3D.shape[0], cam.W, cam.H
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    go = (C.c_size_t * 5)(); bo = (C.c_size_t * 2)(); io = (C.c_size_t * 2)()
    L.gs2d_geometry_layout(P, go); L.gs2d_binning_layout(R, bo); L.gs2d_image_layout(W, H, io)
    g = geom.cpu().numpy(); b = binning.cpu().numpy(); im = img.cpu().numpy()
    view = lambda buf, off, dt, n: np.frombuffer(buf.tobytes()[off:off + n * np.dtype(dt).itemsize], dtype=dt)
    out = dict(num_rendered=R, color=color.cpu().numpy(), allmap=allmap.cpu().numpy(), radii=radii.cpu().numpy(),
               args=args, buffers=(geom, binning, img))
    if P > 0:
        out["depths"] = view(g, go[0], np.float32, P)
        out["tiles_touched"] = view(g, go[1], np.uint32, P)
        out["point_offsets"] = view(g, go[2], np.uint32, P)
        out["rec"] = view(g, go[3], np.float32, P * 20).reshape(P, 20)
        out["clamped"] = view(g, go[4], np.uint8, P * 3).reshape(P, 3)
        out["point_list"] = view(b, bo[0], np.uint32, R) if R > 0 else np.zeros(0, np.uint32)
        # the full 64-bit sorted keys are materialised only in debug mode (the product path keeps packed (depth, id)
        # pairs and writes just the point list): fetch them from a second, debug-mode forward of the same inputs
        if R > 0 and not debug:
            dbg_args = args[:-1] + (True,)
            R2, _, _, _, _, binning2, _ = rasterizer.rasterize_gaussians(*dbg_args)
            torch.cuda.synchronize()
            assert R2 == R
            out["keys"] = view(binning2.cpu().numpy(), bo[1], np.uint64, R)
        else:
            out["keys"] = view(b, bo[1], np.uint64, R) if R > 0 else np.zeros(0, np.uint64)
        out["ranges"] = view(im, io[0], np.uint32, gx * gy * 2).reshape(gx * gy, 2)
        plane = gx * gy * 256
        ps = view(im, io[1], np.float32, 7 * plane).reshape(7, plane)
        idx = pix_index_map(W, H)
        out["final_T"] = ps[0][idx]; out["M1"] = ps[1][idx]; out["M2"] = ps[2][idx]
        out["median_depth"] = ps[3][idx]; out["depth_std"] = ps[4][idx]
        psu = ps.view(np.uint32)
        out["last_contributor"] = psu[5][idx]; out["median_contributor"] = psu[6][idx]
    return out


def hip_backward(fw, dL_dcolor, dL_dallmap, device="cuda"):
    from gaus_slam_amd import rasterizer
    a = fw["args"]
    geom, binning, img = fw["buffers"]
    dev = torch.device(device)
    dc = torch.as_tensor(dL_dcolor).float().to(dev).contiguous()
    da = torch.as_tensor(dL_dallmap).float().to(dev).contiguous()
    radii = torch.as_tensor(fw["radii"]).to(dev)
    res = rasterizer.rasterize_gaussians_backward(
        a[0], a[1], radii, a[2], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], dc, da, a[14], a[15], a[16], geom,
        fw["num_rendered"], binning, img, a[17], a[19])
    torch.cuda.synchronize()
    names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dtransMat", "dL_dsh", "dL_dscales",
             "dL_drotations"]
    return {n: r.cpu().numpy() for n, r in zip(names, res)}


def grad_err(a, b):
    """max |a-b| normalised by max |b| (per tensor)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if b.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def grad_err_mid(a, b, floor=1e-3):
    """max RELATIVE error over the entries with |b| >= floor * max|b|: constrains the mid-magnitude entries that the
    per-tensor max-norm of grad_err leaves free (an entry of 1e-3 max may be 10 % off under grad_err <= 1e-4)."""
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    if b.size == 0 or np.abs(b).max() == 0:
        return 0.0
    sel = np.abs(b) >= floor * np.abs(b).max()
    return float((np.abs(a[sel] - b[sel]) / np.abs(b[sel])).max())


def rounding_report(gh, go, g64, keys, floor=1e-3):
    """Per tensor: the mid-magnitude relative error (grad_err_mid) and the max-norm error (grad_err) of the HIP gradients and
    of the float32 oracle's, both measured against the float64 evaluation of the same backward on the same inputs and
    decisions (oracle.backward_f64).  Returns {key: (hip_mid, orc_mid, hip_max, orc_max, hip_rms, orc_rms)}; rms = root mean
    square of the relative error over the mid-magnitude entries (the max is an extreme-value statistic of ~1e6 entries)."""
    out = {}
    for k in keys:
        ref = np.asarray(g64[k], np.float64).ravel()
        a = np.asarray(gh[k], np.float64).ravel()
        b = np.asarray(go[k], np.float64).ravel()
        sel = np.abs(ref) >= floor * np.abs(ref).max()
        ra, rb = np.abs(a[sel] - ref[sel]) / np.abs(ref[sel]), np.abs(b[sel] - ref[sel]) / np.abs(ref[sel])
        out[k] = (float(ra.max()), float(rb.max()), grad_err(a, ref), grad_err(b, ref), float(np.sqrt((ra ** 2).mean())),
                  float(np.sqrt((rb ** 2).mean())))
        print(f"{k}: vs float64 -- mid-magnitude relative error HIP {out[k][0]:.2e} / oracle-f32 {out[k][1]:.2e}; rms of it "
              f"{out[k][4]:.2e} / {out[k][5]:.2e}; max-norm {out[k][2]:.2e} / {out[k][3]:.2e}")
    return out


SA_EPS = 2.0 ** -22  # four float32 ulps: the relative rounding allowed on the cancelling terms of use_sa's depth variance


def allmap_dev(h, o, stable, scale=None):
    """Per channel: the largest deviation of the HIP allmap from the oracle's on the `stable` pixels, after the allowance the
    CONDITIONING of the reference's own formula grants on the depth channels of use_sa (oracle `sa_amp`, gs2d_oracle.c
    orc_blend_fwd: forward.cu:405-416 divides by a variance formed by cancellation; where the splats in front of a pixel lie at
    nearly one depth two correct float32 evaluations differ by sa_amp x their relative rounding).  Allowance: SA_EPS x sa_amp on
    channel 0 (depth), 8 x that on channel 6 (its error is 2 (d - m) times the depth's); 0 elsewhere and without use_sa.  Found by
    scripts/dev/fuzz_parity.py (1 pixel of 1088x606 at 2.4e-4, sa_amp 1.0e4; the 99.9th percentile of that image is 22).
    `h`, `o`: dicts with "allmap" [7,H,W]; o may lack "sa_amp" (fixtures): no allowance then."""
    d = np.abs(h["allmap"] - o["allmap"]).astype(np.float64)
    amp = o.get("sa_amp")
    if amp is not None:
        a = np.asarray(amp, np.float64).reshape(d.shape[1:]) * SA_EPS
        d[0] = np.maximum(d[0] - a, 0.0)
        d[6] = np.maximum(d[6] - 8.0 * a, 0.0)
    if scale is not None:
        d = d / np.asarray(scale, np.float64)[:, None, None]
    return d[:, stable].max(axis=1, initial=0.0)


def _sa_allow(o, x, y):
    """allmap_dev's allowance for one pixel: [7] (non-zero on channels 0 and 6 of an ill-conditioned use_sa pixel)"""
    out = np.zeros(7)
    amp = o.get("sa_amp")
    if amp is not None:
        a = float(np.asarray(amp).reshape(o["H"], o["W"])[y, x]) * SA_EPS
        out[0], out[6] = a, 8.0 * a
    return out


def match_knife_variants(orc, o, h, stable, tol, knife, scale=None):
    """For every knife-edge pixel: the outcome of its near-threshold decisions (flip mask of oracle.pixel_variants) under which
    the oracle's pixel equals the HIP pixel (same contributors, smallest error <= tol).  Returns [(x, y, mask)] -- the input of
    oracle.backward(pixel_overrides=...), which lets those pixels take part in a gradient comparison."""
    ys, xs = np.nonzero(~stable)
    sc = np.ones(7) if scale is None else np.asarray(scale, np.float64)
    out = []
    for y, x in zip(ys.tolist(), xs.tolist()):
        nk, variants = orc.pixel_variants(o, x, y, knife)
        best = None
        for v in variants:
            if v["last_contributor"] != int(h["last_contributor"][y, x]) or v["median_contributor"] != int(h["median_contributor"][y, x]):
                continue
            e = max(float(np.abs(h["color"][:, y, x] - v["color"]).max()),
                    float((np.maximum(np.abs(h["allmap"][:, y, x] - v["others"]) - _sa_allow(o, x, y), 0.0) / sc).max()))
            if best is None or e < best[0]:
                best = (e, v["mask"])
        assert best is not None and best[0] <= tol, f"knife-edge pixel ({x},{y}) matches no oracle outcome"
        out.append((x, y, best[1]))
    return out


def check_knife_pixels(orc, o, h, stable, tol, knife, scale=None):
    """Knife-edge pixels (a discrete decision of the blend within `knife` of its threshold in the oracle) are excluded from
    the plain L-inf comparison because a 1-ulp difference of v_exp_f32 / v_rcp_f32 may flip that decision.  They are not
    unchecked: the HIP pixel must equal the oracle's pixel under ONE of the possible outcomes of those decisions
    (oracle.pixel_variants enumerates them).  Returns (n_pixels, max_err over the matched variants); raises on a pixel
    that matches no variant.  scale: optional per-channel magnitudes for the 7 allmap channels (stress scenes)."""
    H, W = stable.shape
    HW = H * W
    ys, xs = np.nonzero(~stable)
    worst = 0.0
    sc = np.ones(7) if scale is None else np.asarray(scale, np.float64)
    for y, x in zip(ys.tolist(), xs.tolist()):
        nk, variants = orc.pixel_variants(o, x, y, knife)
        best = None
        for v in variants:
            if v["last_contributor"] != int(h["last_contributor"][y, x]) or v["median_contributor"] != int(h["median_contributor"][y, x]):
                continue
            e = max(float(np.abs(h["color"][:, y, x] - v["color"]).max()),
                    float((np.maximum(np.abs(h["allmap"][:, y, x] - v["others"]) - _sa_allow(o, x, y), 0.0) / sc).max()))
            best = e if best is None else min(best, e)
        assert best is not None and best <= tol, (
            f"knife-edge pixel ({x},{y}) with {nk} near-threshold decisions matches none of the {len(variants)} oracle outcomes "
            f"(best err {best})")
        worst = max(worst, best)
    print(f"knife-edge pixels: {len(ys)} of {HW} ({len(ys) / HW:.2e}), all matched an oracle outcome, max err {worst:.2e}")
    return len(ys), worst


def free_port():
    """A TCP port that is free right now on 127.0.0.1 (rendezvous of the multi-process tests: a fixed port fails when two runs of
    the suite overlap, or when the previous run's socket is still in TIME_WAIT)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
