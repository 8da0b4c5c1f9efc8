"""gs2d_forward_batch / gs2d_backward_batch: K keyframes over the same Gaussians in one call, one blend grid over the tiles of
all frames.  Per frame the results must be those of K separate calls: forward outputs and the private lists bit for bit,
gradients up to the summation order of the float atomics (the separate calls are not bit-reproducible run to run either)."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

GRAD_KEYS = ["means2D", "colors", "opacities", "means3D", "transMat", "sh", "scales", "rotations"]


def _cameras(sc, K, seed):
    from gaus_slam_amd.scene_synth import random_w2c, setup_camera
    cam = sc["cam"]
    rng = np.random.default_rng(seed)
    cams = [cam]
    for _ in range(K - 1):
        cams.append(setup_camera(cam.W, cam.H, cam.K, random_w2c(rng, max_rot_deg=4.0, max_trans=0.15) @ cam.w2c))
    return cams


def _inputs(sc, dev):
    e = torch.empty(0, dtype=torch.float32, device=dev)
    t = lambda a: torch.as_tensor(a).float().to(dev).contiguous()
    return dict(bg=torch.zeros(3, device=dev), means3D=t(sc["means3D"]), colors=t(sc["colors"]), opac=t(sc["opacities"]),
                scales=t(sc["scales"]), rots=t(sc["rotations"]), e=e)


@pytest.mark.parametrize("K,P,W,H,use_sa", [(2, 6000, 256, 192, True), (3, 20000, 333, 250, False), (8, 1500, 128, 96, True),
                                            (4, 200000, 640, 480, True)])
def test_batch_equals_separate_calls(K, P, W, H, use_sa):
    from gaus_slam_amd import rasterizer, _lib
    import ctypes as C
    dev = torch.device("cuda", 0)
    sc = util.make_scene(P, W, H, seed=90 + K, regime="mapping")
    cams = _cameras(sc, K, 7)
    x = _inputs(sc, dev)
    t = lambda a: torch.as_tensor(a).float().to(dev).contiguous()
    vms = torch.stack([t(c.viewmatrix) for c in cams]); pms = torch.stack([t(c.projmatrix) for c in cams])
    cps = torch.stack([t(c.campos) for c in cams])
    # separate calls
    sep = [rasterizer.rasterize_gaussians(x["bg"], x["means3D"], x["colors"], x["opac"], x["scales"], x["rots"], 1.0, x["e"],
                                          vms[k], pms[k], cams[k].tanfovx, cams[k].tanfovy, H, W, x["e"], 0, cps[k], use_sa, False,
                                          False) for k in range(K)]
    Rs, color, others, radii, geoms, bins, imgs = rasterizer.rasterize_gaussians_batch(
        x["bg"], x["means3D"], x["colors"], x["opac"], x["scales"], x["rots"], 1.0, x["e"], vms, pms, H, W, x["e"], 0, cps, use_sa,
        False)
    torch.cuda.synchronize()
    L = _lib.lib()
    for k in range(K):
        R, c1, o1, r1, g1, b1, i1 = sep[k]
        assert Rs[k] == R and R > 0
        assert torch.equal(color[k].view(torch.int32), c1.view(torch.int32)), k
        assert torch.equal(others[k].view(torch.int32), o1.view(torch.int32)), k
        assert torch.equal(radii[k], r1), k
        bo = (C.c_size_t * 2)(); L.gs2d_binning_layout(R, bo)
        assert torch.equal(bins[k][bo[0]:bo[0] + 4 * R], b1[bo[0]:bo[0] + 4 * R]), k   # sorted point lists
        io = (C.c_size_t * 2)(); L.gs2d_image_layout(W, H, io)
        ntiles = ((W + 15) // 16) * ((H + 15) // 16)
        assert torch.equal(imgs[k][io[0]:io[0] + 8 * ntiles], i1[io[0]:io[0] + 8 * ntiles]), k       # tile ranges
        # the 7 pixel-state planes, at the slots of real pixels (lanes outside a ragged image are never written)
        idx = torch.as_tensor(util.pix_index_map(W, H).ravel(), device=dev)
        pa = imgs[k][io[1]:io[1] + 4 * 7 * ntiles * 256].view(torch.int32).view(7, ntiles * 256)
        pb = i1[io[1]:io[1] + 4 * 7 * ntiles * 256].view(torch.int32).view(7, ntiles * 256)
        assert torch.equal(pa[:, idx], pb[:, idx]), k
    # backward: same upstream gradients per frame
    dcs, das = [], []
    for k in range(K):
        dc, da = util.make_upstream_grads(W, H, seed=k, channels=(0, 1, 2, 3, 4, 5, 6) if k % 2 else (0, 1, 5, 6))
        dcs.append((dc * W * H).to(dev)); das.append((da * W * H).to(dev))
    g_sep = [rasterizer.rasterize_gaussians_backward(
        x["bg"], x["means3D"], sep[k][3], x["colors"], x["scales"], x["rots"], 1.0, x["e"], vms[k], pms[k], cams[k].tanfovx,
        cams[k].tanfovy, dcs[k], das[k], x["e"], 0, cps[k], sep[k][4], sep[k][0], sep[k][5], sep[k][6], use_sa, False)
        for k in range(K)]
    g_bat = rasterizer.rasterize_gaussians_backward_batch(
        x["bg"], x["means3D"], radii, x["colors"], x["scales"], x["rots"], 1.0, x["e"], vms, pms, [c.tanfovx for c in cams],
        [c.tanfovy for c in cams], torch.stack(dcs), torch.stack(das), x["e"], 0, cps, geoms, Rs, bins, imgs, use_sa, False)
    torch.cuda.synchronize()
    for k in range(K):
        for name, a, b in zip(GRAD_KEYS, g_bat[k], g_sep[k]):
            if name == "sh":
                continue
            a, b = a.double().cpu().numpy(), b.double().cpu().numpy()
            assert np.isfinite(a).all(), (k, name)
            if name not in ("transMat",):
                assert np.abs(b).max() > 0, (k, name)
            assert util.grad_err(a, b) <= 1e-5, (k, name)
    # accumulate: frame 0's tensors hold the sum over the frames, added in frame order (== the tensor sums of the per-frame
    # gradients of the same call, bit for bit: same operands, same order)
    g_acc = rasterizer.rasterize_gaussians_backward_batch(
        x["bg"], x["means3D"], radii, x["colors"], x["scales"], x["rots"], 1.0, x["e"], vms, pms, [c.tanfovx for c in cams],
        [c.tanfovy for c in cams], torch.stack(dcs), torch.stack(das), x["e"], 0, cps, geoms, Rs, bins, imgs, use_sa, False,
        accumulate=True)
    torch.cuda.synchronize()
    for i, name in enumerate(GRAD_KEYS):
        if name == "sh":
            continue
        tot = g_acc[1][i].clone()          # frames 1..K-1 keep their own gradients: rebuild the sum from this same call
        for k in range(2, K):
            tot = tot + g_acc[k][i]
        if name == "means2D":  # the screen-space gradient is per view: never summed, frame 0 keeps its own
            ref = g_bat[0][i].double()
        else:
            ref = g_bat[0][i].double() + sum(g_bat[k][i].double() for k in range(1, K))
        assert util.grad_err(g_acc[0][i].double().cpu().numpy(), ref.cpu().numpy()) <= 1e-5, name
    # a second batched backward on the same forward state (accumulators no longer known clean) gives the same result
    g_bat2 = rasterizer.rasterize_gaussians_backward_batch(
        x["bg"], x["means3D"], radii, x["colors"], x["scales"], x["rots"], 1.0, x["e"], vms, pms, [c.tanfovx for c in cams],
        [c.tanfovy for c in cams], torch.stack(dcs), torch.stack(das), x["e"], 0, cps, geoms, Rs, bins, imgs, use_sa, False)
    for k in range(K):
        for name, a, b in zip(GRAD_KEYS, g_bat2[k], g_bat[k]):
            if name != "sh":
                assert util.grad_err(a.double().cpu().numpy(), b.double().cpu().numpy()) <= 1e-5, (k, name)


def test_batch_of_one_and_argument_errors():
    from gaus_slam_amd import rasterizer
    dev = torch.device("cuda", 0)
    W, H, P = 160, 120, 900
    sc = util.make_scene(P, W, H, seed=3, regime="mapping")
    x = _inputs(sc, dev)
    t = lambda a: torch.as_tensor(a).float().to(dev).contiguous()
    cam = sc["cam"]
    vm, pm, cp = t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos)
    R, c1, o1, r1, *_ = rasterizer.rasterize_gaussians(x["bg"], x["means3D"], x["colors"], x["opac"], x["scales"], x["rots"], 1.0,
                                                       x["e"], vm, pm, cam.tanfovx, cam.tanfovy, H, W, x["e"], 0, cp, True, False, False)
    Rs, c, o, r, *_ = rasterizer.rasterize_gaussians_batch(x["bg"], x["means3D"], x["colors"], x["opac"], x["scales"], x["rots"], 1.0,
                                                           x["e"], vm[None], pm[None], H, W, x["e"], 0, cp[None], True, False)
    assert Rs == [R] and torch.equal(c[0], c1) and torch.equal(o[0], o1) and torch.equal(r[0], r1)
    with pytest.raises(RuntimeError, match="frames"):
        rasterizer.rasterize_gaussians_batch(x["bg"], x["means3D"], x["colors"], x["opac"], x["scales"], x["rots"], 1.0, x["e"],
                                             vm[None].repeat(9, 1, 1), pm[None].repeat(9, 1, 1), H, W, x["e"], 0,
                                             cp[None].repeat(9, 1), True, False)
    # P == 0: zero images, no instances
    z = torch.zeros(0, 3, device=dev)
    Rs0, c0, o0, r0, *_ = rasterizer.rasterize_gaussians_batch(x["bg"], z, z, torch.zeros(0, 1, device=dev), torch.zeros(0, 2, device=dev),
                                                              torch.zeros(0, 4, device=dev), 1.0, x["e"], vm[None].repeat(2, 1, 1),
                                                              pm[None].repeat(2, 1, 1), H, W, x["e"], 0, cp[None].repeat(2, 1), True, False)
    assert Rs0 == [0, 0] and float(c0.abs().max()) == 0 and float(o0.abs().max()) == 0
    # no deterministic variant: refused with a message
    Rs, c, o, r, geoms, bins, imgs = rasterizer.rasterize_gaussians_batch(
        x["bg"], x["means3D"], x["colors"], x["opac"], x["scales"], x["rots"], 1.0, x["e"], vm[None].repeat(2, 1, 1),
        pm[None].repeat(2, 1, 1), H, W, x["e"], 0, cp[None].repeat(2, 1), True, False)
    dc, da = util.make_upstream_grads(W, H)
    dc, da = dc.to(dev)[None].repeat(2, 1, 1, 1), da.to(dev)[None].repeat(2, 1, 1, 1)
    rasterizer.set_deterministic(True)
    try:
        with pytest.raises(RuntimeError, match="deterministic"):
            rasterizer.rasterize_gaussians_backward_batch(
                x["bg"], x["means3D"], r, x["colors"], x["scales"], x["rots"], 1.0, x["e"], vm[None].repeat(2, 1, 1),
                pm[None].repeat(2, 1, 1), [cam.tanfovx] * 2, [cam.tanfovy] * 2, dc, da, x["e"], 0, cp[None].repeat(2, 1), geoms, Rs,
                bins, imgs, True, False)
    finally:
        rasterizer.set_deterministic(False)


def test_batched_operator_and_ba_step_match_keyframe_by_keyframe():
    """The autograd surface (GaussianRasterizerBatch / render_batch) and KeyframeShardedBA with a batch_fn: summed gradients of a
    rank's keyframes equal those of rendering them one operator call after the other."""
    from gaus_slam_amd import ba_shard, render as gs_render
    dev = torch.device("cuda", 0)
    W, H, P, K = 320, 240, 30000, 3
    sc = util.make_scene(P, W, H, seed=17, regime="mapping")
    cams = _cameras(sc, K, 11)
    sts = [gs_render.settings_from_camera(c, dev, use_sa=True) for c in cams]
    names = ("means3D", "opacities", "scales", "rotations", "colors")
    ups = []
    for k in range(K):
        dc, da = util.make_upstream_grads(W, H, seed=20 + k, channels=(0, 1, 5, 6))
        ups.append(((dc * W * H).to(dev), (da * W * H).to(dev)))

    def one(p, kf):
        m2 = torch.empty_like(p["means3D"]).requires_grad_(True)
        pk = gs_render.render(sts[kf], p["means3D"], m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"],
                              rotations=p["rotations"])
        return (pk["render_color"], pk["allmap"]), ups[kf]

    def batch(p, kfs):
        m2 = torch.empty_like(p["means3D"]).requires_grad_(True)
        pks = gs_render.render_batch([sts[k] for k in kfs], p["means3D"], m2, p["opacities"], colors_precomp=p["colors"],
                                     scales=p["scales"], rotations=p["rotations"])
        outs, gs = [], []
        for k, pk in zip(kfs, pks):
            outs += [pk["render_color"], pk["allmap"]]
            gs += list(ups[k])
        return outs, gs

    res = {}
    for mode, direct in (("seq", False), ("batch", False), ("batch_direct", True)):
        params = {k: sc[k].to(dev).requires_grad_(True) for k in names}
        ba = ba_shard.KeyframeShardedBA(params, one, direct_grads=direct, batch_fn=None if mode == "seq" else batch)
        g = ba.step(list(range(K)))
        torch.cuda.synchronize()
        res[mode] = {n: g[n].double().cpu().numpy().copy() for n in names}
        if direct:  # frame 0's gradients were written straight into the bucket
            assert g["means3D"].data_ptr() == ba.bucket.views["means3D"].data_ptr()
    for n in names:
        assert np.abs(res["seq"][n]).max() > 0
        assert util.grad_err(res["batch"][n], res["seq"][n]) <= 1e-5, n
        assert util.grad_err(res["batch_direct"][n], res["seq"][n]) <= 1e-5, n


def test_batch_screen_space_gradients_stay_per_view():
    """the reference's mapping step feeds every rendered view's means2D.grad to add_densification_stats
    (slam/Backend.py:117-118; scene/Gaussians.py:58-62 accumulates norm(means2D.grad[:, :2]) per view).  render_batch with one
    gradient carrier PER VIEW gives each carrier exactly the gradient a separate render() of that view gives; with one shared
    carrier the gradient is the sum over the views (autograd's semantics), which those statistics must not be fed."""
    from gaus_slam_amd import render as gs_render
    dev = torch.device("cuda", 0)
    W, H, P, K = 320, 240, 20000, 3
    sc = util.make_scene(P, W, H, seed=19, regime="mapping")
    cams = _cameras(sc, K, 13)
    sts = [gs_render.settings_from_camera(c, dev, use_sa=True) for c in cams]
    p = {k: sc[k].to(dev).requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations", "colors")}
    ups = []
    for k in range(K):
        dc, da = util.make_upstream_grads(W, H, seed=30 + k, channels=(0, 1, 5, 6))
        ups.append(((dc * W * H).to(dev), (da * W * H).to(dev)))
    sep = []
    for k in range(K):
        m2 = torch.zeros_like(p["means3D"]).requires_grad_(True)
        pk = gs_render.render(sts[k], p["means3D"], m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"],
                              rotations=p["rotations"])
        torch.autograd.backward([pk["render_color"], pk["allmap"]], list(ups[k]))
        assert pk["means2D"] is m2 and float(m2.grad.abs().max()) > 0
        sep.append(m2.grad.clone())
    g_param_sep = {k: v.grad.clone() for k, v in p.items()}
    for v in p.values():
        v.grad = None
    # one carrier per view
    m2s = [torch.zeros_like(p["means3D"]).requires_grad_(True) for _ in range(K)]
    pks = gs_render.render_batch(sts, p["means3D"], m2s, p["opacities"], colors_precomp=p["colors"], scales=p["scales"],
                                 rotations=p["rotations"])
    outs, gs = [], []
    for k, pk in enumerate(pks):
        assert pk["means2D"] is m2s[k]
        outs += [pk["render_color"], pk["allmap"]]
        gs += list(ups[k])
    torch.autograd.backward(outs, gs)
    for k in range(K):
        assert util.grad_err(m2s[k].grad.cpu().numpy(), sep[k].cpu().numpy()) <= 1e-5, k
        # what the reference's statistic sees for view k
        assert torch.allclose(m2s[k].grad[:, :2].norm(dim=-1), sep[k][:, :2].norm(dim=-1), rtol=1e-4, atol=1e-7 * float(sep[k].abs().max()))
    for k2, v in p.items():  # the parameter gradients are the sums over the views, as K separate calls accumulate them
        assert util.grad_err(v.grad.cpu().numpy(), g_param_sep[k2].cpu().numpy()) <= 1e-5, k2
        v.grad = None
    # one shared carrier: the sum over the views
    m2 = torch.zeros_like(p["means3D"]).requires_grad_(True)
    pks = gs_render.render_batch(sts, p["means3D"], m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"],
                                 rotations=p["rotations"])
    outs = []
    for pk in pks:
        outs += [pk["render_color"], pk["allmap"]]
    torch.autograd.backward(outs, gs)
    assert util.grad_err(m2.grad.cpu().numpy(), sum(sep).cpu().numpy()) <= 1e-5


def test_batch_camera_cache_sees_an_in_place_pose_update():
    """the stacked camera matrices of a frame set are cached by the identity of the settings objects; a pose refined
    IN PLACE (the tensors of a NamedTuple can be written) must not render with the stale stack."""
    from gaus_slam_amd import render as gs_render
    from gaus_slam_amd.scene_synth import random_w2c, setup_camera
    dev = torch.device("cuda", 0)
    W, H, P = 160, 120, 3000
    sc = util.make_scene(P, W, H, seed=23, regime="mapping")
    cams = _cameras(sc, 2, 5)
    sts = [gs_render.settings_from_camera(c, dev, use_sa=True) for c in cams]
    p = {k: sc[k].to(dev) for k in ("means3D", "opacities", "scales", "rotations", "colors")}
    m2 = torch.zeros_like(p["means3D"])
    kw = dict(colors_precomp=p["colors"], scales=p["scales"], rotations=p["rotations"])
    before = gs_render.render_batch(sts, p["means3D"], m2, p["opacities"], **kw)[1]["render_color"].clone()
    new = setup_camera(W, H, sc["cam"].K, random_w2c(np.random.default_rng(77), 6.0, 0.2) @ sc["cam"].w2c)
    sts[1].viewmatrix.copy_(new.viewmatrix.to(dev).unsqueeze(0))
    sts[1].projmatrix.copy_(new.projmatrix.to(dev).unsqueeze(0))
    sts[1].campos.copy_(new.campos.to(dev))
    after = gs_render.render_batch(sts, p["means3D"], m2, p["opacities"], **kw)[1]["render_color"]
    single = gs_render.render(sts[1], p["means3D"], m2, p["opacities"], **kw)["render_color"]
    assert torch.equal(after, single) and not torch.equal(after, before)


def test_ba_step_with_a_batch_fn_in_deterministic_mode():
    """the batched backward has no deterministic variant; with gs2d_set_deterministic(1) a rank that holds several
    keyframes must take them one by one (bit-identical from run to run) instead of failing in the middle of the step."""
    from gaus_slam_amd import ba_shard, rasterizer, render as gs_render
    dev = torch.device("cuda", 0)
    W, H, P, K = 256, 192, 12000, 3
    sc = util.make_scene(P, W, H, seed=29, regime="mapping")
    cams = _cameras(sc, K, 17)
    sts = [gs_render.settings_from_camera(c, dev, use_sa=True) for c in cams]
    names = ("means3D", "opacities", "scales", "rotations", "colors")
    ups = []
    for k in range(K):
        dc, da = util.make_upstream_grads(W, H, seed=40 + k, channels=(0, 1, 5, 6))
        ups.append(((dc * W * H).to(dev), (da * W * H).to(dev)))

    def one(p, kf):
        m2 = torch.empty_like(p["means3D"]).requires_grad_(True)
        pk = gs_render.render(sts[kf], p["means3D"], m2, p["opacities"], colors_precomp=p["colors"], scales=p["scales"],
                              rotations=p["rotations"])
        return (pk["render_color"], pk["allmap"]), ups[kf]

    def batch(p, kfs):
        m2 = torch.empty_like(p["means3D"]).requires_grad_(True)
        pks = gs_render.render_batch([sts[k] for k in kfs], p["means3D"], m2, p["opacities"], colors_precomp=p["colors"],
                                     scales=p["scales"], rotations=p["rotations"])
        outs, gs = [], []
        for k, pk in zip(kfs, pks):
            outs += [pk["render_color"], pk["allmap"]]
            gs += list(ups[k])
        return outs, gs

    def run():
        params = {k: sc[k].to(dev).requires_grad_(True) for k in names}
        ba = ba_shard.KeyframeShardedBA(params, one, direct_grads=True, batch_fn=batch)
        g = ba.step(list(range(K)))
        torch.cuda.synchronize()
        return {n: g[n].cpu().numpy().copy() for n in names}

    ref = run()  # (batched, atomics)
    rasterizer.set_deterministic(True)
    try:
        a, b = run(), run()
    finally:
        rasterizer.set_deterministic(False)
    for n in names:
        assert np.array_equal(a[n].view(np.uint32), b[n].view(np.uint32)), n
        assert util.grad_err(a[n], ref[n]) <= 1e-5, n
