"""Fused dense Adam (gs2d_adam_step / gaus_slam_amd/optim.py) against torch.optim.Adam configured as the reference does
(scene/Gaussians.py:121-137), and the PLY map layout (scene/Gaussians.py:435-500) reader/writer."""
import numpy as np
import pytest
import torch


def test_ply_roundtrip_and_header(tmp_path):
    from gaus_slam_amd import ply
    rng = np.random.default_rng(0)
    P = 37
    f = dict(xyz=rng.normal(size=(P, 3)), opacity=rng.normal(size=(P, 1)), scaling=rng.normal(size=(P, 2)),
             rotation=rng.normal(size=(P, 4)), rgb=rng.random((P, 3)))
    path = str(tmp_path / "map" / "g.ply")
    ply.save_ply(path, **f)
    raw = open(path, "rb").read()
    head = raw[:raw.index(b"end_header\n") + 11].decode()
    # attribute order of construct_list_of_attributes (scene/Gaussians.py:435-450) for the RGB (non-SH) map
    want = ["x", "y", "z", "nx", "ny", "nz", "opacity", "scale_0", "scale_1", "rot_0", "rot_1", "rot_2", "rot_3", "r", "g", "b"]
    assert [l.split()[2] for l in head.splitlines() if l.startswith("property")] == want
    assert "format binary_little_endian 1.0" in head and f"element vertex {P}" in head
    assert len(raw) == len(head) + P * 16 * 4
    back = ply.load_ply(path)
    for k, v in f.items():
        np.testing.assert_array_equal(back[k], v.astype(np.float32))
    names, table = ply.read_vertex_table(path)
    assert np.all(table["nx"] == 0) and np.all(table["nz"] == 0)


def test_ply_sh_layout_follows_loader(tmp_path):
    from gaus_slam_amd import ply
    rng = np.random.default_rng(1)
    P, M = 5, 16
    f_dc, f_rest = rng.normal(size=(P, 1, 3)).astype(np.float32), rng.normal(size=(P, M - 1, 3)).astype(np.float32)
    path = str(tmp_path / "sh.ply")
    ply.save_ply(path, rng.normal(size=(P, 3)), np.zeros((P, 1)), np.zeros((P, 2)), np.ones((P, 4)), f_dc=f_dc, f_rest=f_rest)
    names, table = ply.read_vertex_table(path)
    assert names[-45:] == [f"f_rest_{i}" for i in range(45)] and names[13:16] == ["f_dc_0", "f_dc_1", "f_dc_2"]
    # loader rule: f_rest_i is element i of the [3, M-1] (channel-major) block
    assert table["f_rest_1"][2] == f_rest[2, 1, 0] and table["f_rest_15"][2] == f_rest[2, 0, 1]
    back = ply.load_ply(path)
    np.testing.assert_array_equal(back["f_dc"], f_dc)
    np.testing.assert_array_equal(back["f_rest"], f_rest)


def test_ply_reads_ascii(tmp_path):
    from gaus_slam_amd import ply
    path = tmp_path / "a.ply"
    path.write_text("ply\nformat ascii 1.0\ncomment x\nelement vertex 2\n" + "".join(
        f"property float {n}\n" for n in ply.attribute_names()) + "end_header\n" +
        " ".join(str(i) for i in range(16)) + "\n" + " ".join(str(i + 100) for i in range(16)) + "\n")
    back = ply.load_ply(str(path))
    assert back["xyz"].tolist() == [[0, 1, 2], [100, 101, 102]] and back["rgb"][1].tolist() == [113, 114, 115]


def test_adam_wrapper_refuses_cpu():
    from gaus_slam_amd.optim import FusedGaussianAdam, GaussianSoA
    P = 4
    soa = GaussianSoA(dict(means3D=torch.zeros(P, 3), opacities=torch.zeros(P, 1), scales=torch.zeros(P, 2),
                           rotations=torch.zeros(P, 4), colors=torch.zeros(P, 3)))
    opt = FusedGaussianAdam(soa, dict(xyz=1e-4))
    with pytest.raises(RuntimeError):
        opt.step(torch.zeros(13 * P))


LRS = dict(xyz=0.0001, opacity=0.05, scaling=0.001, rotation=0.001, rgb=0.0025)  # configs/replica/config_fast.py:115-122


def _torch_adam(fields, grads_per_step):
    names = dict(means3D="xyz", opacities="opacity", scales="scaling", rotations="rotation", colors="rgb")
    ps = {n: torch.nn.Parameter(t.clone().double()) for n, t in fields.items()}
    opt = torch.optim.Adam([{"params": [ps[n]], "lr": LRS[names[n]], "name": names[n]} for n in ps], lr=0.0, eps=1e-15)
    for g in grads_per_step:
        for n, p in ps.items():
            p.grad = g[n].clone().double()
        opt.step()
    return ps, opt


@pytest.mark.gpu
@pytest.mark.parametrize("P", [1, 1003, 50000])
def test_fused_adam_matches_torch_adam(P):
    from gaus_slam_amd.ba_shard import GradBucket
    from gaus_slam_amd.optim import FusedGaussianAdam, GaussianSoA
    g = torch.Generator().manual_seed(P)
    ks = dict(means3D=3, opacities=1, scales=2, rotations=4, colors=3)
    fields = {n: torch.randn(P, k, generator=g) for n, k in ks.items()}
    steps = []
    for s in range(6):
        gr = {n: torch.randn(P, k, generator=g) * 10.0 ** float(torch.randint(-6, 1, (1,), generator=g)) for n, k in ks.items()}
        if s == 2:
            gr["opacities"].zero_()  # zero gradients: moments decay, dense Adam still moves the parameter
        steps.append(gr)
    ref, _ = _torch_adam(fields, steps)  # float64 torch.optim.Adam: the formulas the reference runs
    dev = torch.device("cuda")
    soa = GaussianSoA({n: t.to(dev) for n, t in fields.items()})
    opt = FusedGaussianAdam(soa, LRS)
    bucket = GradBucket(P, dev)
    for gr in steps:
        bucket.pack({n: t.to(dev) for n, t in gr.items()})
        opt.step(bucket.flat)
    for n in ks:
        got, want = soa.views[n].cpu().double(), ref[n].detach()
        # float32 arithmetic vs float64: tolerance 2e-6 absolute on O(1) parameters moved by <= 6 * lr
        assert (got - want).abs().max() < 2e-6, n
    # the step really moved things by ~lr per step
    assert (soa.views["opacities"].cpu() - fields["opacities"]).abs().max() > 0.05


@pytest.mark.gpu
def test_fused_adam_prune_and_cat_keep_moments_aligned():
    from gaus_slam_amd.ba_shard import GradBucket
    from gaus_slam_amd.optim import FusedGaussianAdam, GaussianSoA
    dev = torch.device("cuda")
    P = 300
    g = torch.Generator().manual_seed(3)
    ks = dict(means3D=3, opacities=1, scales=2, rotations=4, colors=3)
    fields = {n: torch.randn(P, k, generator=g) for n, k in ks.items()}
    g1 = {n: torch.randn(P, k, generator=g) for n, k in ks.items()}
    keep = torch.rand(P, generator=g) > 0.3
    new = {n: torch.randn(17, k, generator=g) for n, k in ks.items()}
    P2 = int(keep.sum()) + 17
    g2 = {n: torch.randn(P2, k, generator=g) for n, k in ks.items()}
    # expected: torch Adam with state surgery as prune_optimizer / cat_tensors_to_optimizer do it
    ref, ropt = _torch_adam(fields, [g1])
    ps2 = {}
    for grp in ropt.param_groups:
        p = grp["params"][0]
        n = [k for k, v in ref.items() if v is p][0]
        st = ropt.state.pop(p)
        st["exp_avg"] = torch.cat([st["exp_avg"][keep], torch.zeros(17, ks[n], dtype=torch.float64)])
        st["exp_avg_sq"] = torch.cat([st["exp_avg_sq"][keep], torch.zeros(17, ks[n], dtype=torch.float64)])
        q = torch.nn.Parameter(torch.cat([p.detach()[keep], new[n].double()]))
        grp["params"][0] = q
        ropt.state[q] = st
        q.grad = g2[n].double()
        ps2[n] = q
    ropt.step()
    soa = GaussianSoA({n: t.to(dev) for n, t in fields.items()})
    opt = FusedGaussianAdam(soa, LRS)
    b = GradBucket(P, dev); b.pack({n: t.to(dev) for n, t in g1.items()}); opt.step(b.flat)
    opt.prune(keep.to(dev)); opt.cat({n: t.to(dev) for n, t in new.items()})
    assert soa.P == P2
    b = GradBucket(P2, dev); b.pack({n: t.to(dev) for n, t in g2.items()}); opt.step(b.flat)
    for n in ks:
        assert (soa.views[n].cpu().double() - ps2[n].detach()).abs().max() < 2e-6, n


@pytest.mark.gpu
def test_render_through_leaves_after_prune_and_cat():
    """prune() / cat() re-allocate the flat SoA: leaves fetched before are stale (they alias the OLD buffer, so rendering
    from them would silently ignore every later Adam step).  Fresh leaves render the new map; stale ones are rejected."""
    from gaus_slam_amd import optim, render as gs_render
    from gaus_slam_amd.scene_synth import make_scene, make_upstream_grads
    dev = torch.device("cuda")
    P, W, H = 3000, 160, 120
    sc = make_scene(P, W, H, seed=5, regime="mapping")
    names = ("means3D", "opacities", "scales", "rotations", "colors")
    soa = optim.GaussianSoA({k: sc[k].to(dev) for k in names})
    opt = optim.FusedGaussianAdam(soa, dict(xyz=1e-3, opacity=1e-2, scaling=1e-3, rotation=1e-3, rgb=1e-2))
    settings = gs_render.settings_from_camera(sc["cam"], dev, use_sa=True)
    dc, da = [t.to(dev) for t in make_upstream_grads(W, H, seed=2)]

    def render_and_grads(leaves):
        m2 = torch.zeros_like(leaves["means3D"], requires_grad=True)
        pkg = gs_render.render(settings, leaves["means3D"], m2, leaves["opacities"], colors_precomp=leaves["colors"],
                               scales=leaves["scales"], rotations=leaves["rotations"])
        torch.autograd.backward([pkg["render_color"], pkg["allmap"]], [dc, da])
        return pkg["render_color"].detach().clone(), torch.cat([leaves[n].grad.reshape(-1) for n in optim.BUCKET_FIELDS])

    old = soa.leaves()
    img0, g0 = render_and_grads(old)
    opt.step(g0, old)
    keep = torch.ones(P, dtype=torch.bool, device=dev)
    keep[::3] = False
    opt.prune(keep)                                             # P changes ...
    opt.cat({k: sc[k][: int((~keep).sum())].to(dev) for k in names})  # ... and is restored: same P, new buffer
    assert soa.P == P
    with pytest.raises(RuntimeError, match="stale Gaussian leaf"):
        soa.assert_current(old)
    with pytest.raises(RuntimeError, match="stale Gaussian leaf"):
        opt.step(g0, old)
    new = soa.leaves()
    soa.assert_current(new)
    img1, g1 = render_and_grads(new)
    assert float((img1 - img0).abs().max()) > 1e-3  # the map did change
    before = soa.flat.clone()
    opt.step(g1, new)
    assert not torch.equal(before, soa.flat)
    # the fresh leaves alias the buffer Adam just updated: a second render sees the update without re-fetching
    img2, _ = render_and_grads({k: v.detach().requires_grad_(True) for k, v in new.items()})
    assert float((img2 - img1).abs().max()) > 0
