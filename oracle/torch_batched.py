"""Pure-PyTorch CPU render, vectorised over ALL tiles at once (ORACLE / BASELINE side).

TEST INFRASTRUCTURE ONLY (see oracle/gs2d_oracle.c header): never imported by the product.  This is the
"pure-PyTorch CPU render timed on the same box's host cores" that BASELINE.json's north_star and BASELINE.md
section 4 ask for as the CPU baseline of the GPU numbers.  oracle/torch_ref.py (one tile at a time, splats
sequential) states the same semantics but is far too slow beyond a few thousand Gaussians; here every tile walks its
depth-sorted list in lock-step: step j composites the j-th splat of EVERY tile onto that tile's 256 pixels, so one
frame is `max list length` steps of ~60 elementwise ops on [tiles, 256] tensors.  The backward is torch.autograd of
this forward (a CPU timing baseline; the reference's hand-derived backward is restated in oracle/gs2d_oracle.c).

Follows RAST/cuda_rasterizer/forward.cu:75-147,150-253 (preprocess, via oracle/torch_ref.py),
rasterizer_impl.cu:70-138 (duplicate, sort, ranges) and forward.cu:258-467 (blend)."""
import torch

from . import torch_ref

TILE = torch_ref.TILE
NEAR_N, FAR_N, FILTER_INV_SQ = torch_ref.NEAR_N, torch_ref.FAR_N, torch_ref.FILTER_INV_SQ


def bin_and_sort(p_view, center, radius, ok, W, H):
    """rasterizer_impl.cu:70-138 vectorised: -> point_list [R], ranges [tiles, 2] (long)."""
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    r = radius.float()
    cf = center.float()
    minx = ((cf[:, 0] - r) / TILE).trunc().clamp(0, gx).long()
    miny = ((cf[:, 1] - r) / TILE).trunc().clamp(0, gy).long()
    maxx = ((cf[:, 0] + r + (TILE - 1)) / TILE).trunc().clamp(0, gx).long()
    maxy = ((cf[:, 1] + r + (TILE - 1)) / TILE).trunc().clamp(0, gy).long()
    nx, ny = (maxx - minx).clamp(min=0), (maxy - miny).clamp(min=0)
    cnt = torch.where(ok, nx * ny, torch.zeros_like(nx))
    ok = ok & (cnt > 0)
    cnt = torch.where(ok, cnt, torch.zeros_like(cnt))
    R = int(cnt.sum())
    ids = torch.repeat_interleave(torch.arange(cnt.shape[0]), cnt)
    start = torch.cumsum(cnt, 0) - cnt
    local = torch.arange(R) - start[ids]
    ty = miny[ids] + local // nx[ids].clamp(min=1)
    tx = minx[ids] + local % nx[ids].clamp(min=1)
    tiles = ty * gx + tx
    dbits = p_view[:, 2].detach().float().view(torch.int32).long()[ids]
    order = torch.argsort(tiles * (1 << 32) + dbits, stable=True)
    point_list, tiles_sorted = ids[order], tiles[order]
    cntt = torch.bincount(tiles_sorted, minlength=gx * gy)
    ends = torch.cumsum(cntt, 0)
    ranges = torch.stack([ends - cntt, ends], 1)
    return ok, point_list, tiles_sorted, ranges


def render(means3D, scales, rotations, opacities, colors, viewmatrix, projmatrix, W, H, bg=None, use_sa=True,
           scale_modifier=1.0, max_steps=None):
    """Forward render -> dict(color[3,H,W], allmap[7,H,W], radii, point_list, ranges, n_contrib[2,H,W], steps).
    max_steps (bench only): stop after that many list positions and report how many a full frame needs."""
    dt = means3D.dtype
    P = means3D.shape[0]
    bg = torch.zeros(3, dtype=dt) if bg is None else bg.to(dt)
    opacities = opacities.reshape(-1)
    T, normal, p_view = torch_ref.preprocess(means3D, scales, rotations, viewmatrix, projmatrix, W, H, scale_modifier)
    cosv = -(p_view * normal).sum(-1)
    normal = normal * torch.where(cosv > 0, 1.0, -1.0).to(dt)[:, None]
    dist, center, extent = torch_ref.aabb(T)
    radius = torch.ceil(extent.max(-1).values)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    ntiles = gx * gy
    with torch.no_grad():
        ok = (p_view[:, 2] > 0.2) & (cosv != 0) & (dist != 0)
        ok, point_list, tiles_sorted, ranges = bin_and_sort(p_view, center, radius, ok, W, H)
        radii = torch.where(ok, radius, torch.zeros_like(radius)).to(torch.int32)
        lens = ranges[:, 1] - ranges[:, 0]
        L = int(lens.max()) if ntiles else 0
        # padded per-tile lists: slot P = a dummy splat that never contributes
        pad = torch.full((ntiles, max(L, 1)), P, dtype=torch.long)
        pos = torch.arange(point_list.shape[0]) - ranges[tiles_sorted, 0]
        pad[tiles_sorted, pos] = point_list
        ys, xs = torch.meshgrid(torch.arange(TILE), torch.arange(TILE), indexing="ij")
        tix = torch.arange(ntiles)
        px = ((tix % gx) * TILE)[:, None] + xs.flatten()[None]  # [tiles, 256]
        py = ((tix // gx) * TILE)[:, None] + ys.flatten()[None]
        inside = (px < W) & (py < H)
        pxf, pyf = px.to(dt), py.to(dt)
    # per-Gaussian tables with the dummy row appended (opacity 0 -> alpha < 1/255 -> skipped)
    z1 = torch.zeros(1, dtype=dt)
    Tt_ = torch.cat([T.reshape(P, 9), torch.ones(1, 9, dtype=dt)])
    cen_ = torch.cat([center.detach(), torch.zeros(1, 2, dtype=dt)])
    nrm_ = torch.cat([normal, torch.zeros(1, 3, dtype=dt)])
    col_ = torch.cat([colors, torch.zeros(1, 3, dtype=dt)])
    opa_ = torch.cat([opacities, z1])
    n = TILE * TILE
    zero = torch.zeros(ntiles, n, dtype=dt)
    Tr = torch.ones(ntiles, n, dtype=dt)
    C = torch.zeros(ntiles, n, 3, dtype=dt)
    N = torch.zeros(ntiles, n, 3, dtype=dt)
    D, D2, M1, M2, dist_acc, median = zero, zero, zero, zero, zero, zero
    median_c = torch.zeros(ntiles, n, dtype=torch.long)
    last = torch.zeros(ntiles, n, dtype=torch.long)
    done = ~inside
    steps = L if max_steps is None else min(L, int(max_steps))
    for j in range(steps):
        g = pad[:, j]
        if bool(done.all()):
            break
        Tg = Tt_[g]                                        # [tiles, 9]
        Tu, Tv, Tw = Tg[:, None, 0:3], Tg[:, None, 3:6], Tg[:, None, 6:9]
        k = pxf[..., None] * Tw - Tu                       # [tiles, 256, 3]
        l = pyf[..., None] * Tw - Tv
        p = torch.cross(k, l, dim=-1)
        pz_ok = p[..., 2] != 0
        pz = torch.where(pz_ok, p[..., 2], torch.ones_like(p[..., 2]))
        s0, s1 = p[..., 0] / pz, p[..., 1] / pz
        rho3d = s0 * s0 + s1 * s1
        cg = cen_[g]
        d0, d1 = cg[:, None, 0] - pxf, cg[:, None, 1] - pyf
        rho2d = FILTER_INV_SQ * (d0 * d0 + d1 * d1)
        ray = rho3d <= rho2d
        rho = torch.where(ray, rho3d, rho2d)
        depth = torch.where(ray, s0 * Tg[:, None, 6] + s1 * Tg[:, None, 7] + Tg[:, None, 8], Tg[:, None, 8].expand(-1, n))
        alpha = torch.clamp(opa_[g][:, None] * torch.exp(-0.5 * rho), max=0.99)
        valid = (~done) & pz_ok & ~(depth < NEAR_N) & ~(alpha < 1.0 / 255.0)
        test_T = Tr * (1 - alpha)
        stop = valid & (test_T < 0.0001)
        done = done | stop
        valid = valid & ~stop
        w = alpha * Tr
        upd = valid & (Tr > 0.5)
        median = torch.where(upd, depth, median)
        median_c = torch.where(upd, torch.full_like(median_c, j + 1), median_c)
        if use_sa:
            has = D > 0
            one_m_T = torch.where(has, 1 - Tr, torch.ones_like(Tr))
            exp_std = torch.clamp((D2 - 2 * D * median) / one_m_T + median * median, min=1e-7)
            conf = torch.exp(-((median - depth) ** 2) / (4 * exp_std))
            depth_sa = torch.where(has, conf * depth + (1 - conf) * median, depth)
            D = torch.where(valid, D + depth_sa * w, D)
            D2 = torch.where(valid, D2 + depth_sa * depth_sa * w, D2)
        else:
            A = 1 - Tr
            safe_depth = torch.where(valid, depth, torch.ones_like(depth))
            m = FAR_N / (FAR_N - NEAR_N) * (1 - NEAR_N / safe_depth)
            dist_acc = torch.where(valid, dist_acc + (m * m * A + M2 - 2 * m * M1) * w, dist_acc)
            D = torch.where(valid, D + depth * w, D)
            M1 = torch.where(valid, M1 + m * w, M1)
            M2 = torch.where(valid, M2 + m * m * w, M2)
        wv = torch.where(valid, w, torch.zeros_like(w))
        N = N + nrm_[g][:, None, :] * wv[..., None]
        C = C + col_[g][:, None, :] * wv[..., None]
        Tr = torch.where(valid, test_T, Tr)
        last = torch.where(valid, torch.full_like(last, j + 1), last)
    reg = D2 - 2 * median * D + (1 - Tr) * median * median if use_sa else dist_acc
    col = C + Tr[..., None] * bg
    am = torch.stack([D, 1 - Tr, N[..., 0], N[..., 1], N[..., 2], median, reg], 0)  # [7, tiles, 256]

    def to_image(t):  # [..., tiles, 256] -> [..., H, W]
        lead = t.shape[:-2]
        t = t.reshape(*lead, gy, gx, TILE, TILE).transpose(-3, -2).reshape(*lead, gy * TILE, gx * TILE)
        return t[..., :H, :W]

    color = to_image(col.permute(2, 0, 1))
    allmap = to_image(am)
    n_contrib = torch.stack([to_image(last), to_image(median_c)], 0)
    return dict(color=color, allmap=allmap, radii=radii, point_list=point_list, ranges=ranges, n_contrib=n_contrib,
                final_T=to_image(Tr.detach()), steps=L, steps_run=steps)
