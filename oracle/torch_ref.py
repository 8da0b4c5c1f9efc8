"""Pure-PyTorch CPU render of the 2D-Gaussian-surfel rasterizer (ORACLE side).

TEST INFRASTRUCTURE ONLY (see oracle/gs2d_oracle.c header).  This is the
independent second statement of the forward pass: written with vectorised
torch ops (one tile at a time, pixels vectorised, splats sequential) so that
  (1) forward values can be cross-checked against the scalar C oracle, and
  (2) torch.autograd of it pins the hand-written backward formulae of the C
      oracle on the branches where the reference gradient is exact
      (use_sa=False; unit quaternions; ray-splat branch).
It is also BASELINE.json's configs[0]: "160x120 / 256-Gaussian forward render
via a pure-PyTorch CPU path".

Follows RAST/cuda_rasterizer/forward.cu:75-147 (transmat, aabb), :150-253
(preprocess), rasterizer_impl.cu:70-138 (keys, sort, ranges) and
forward.cu:258-467 (blend).  SH colours are not restated here (the C oracle
has them); pass colours explicitly.
"""
import torch

TILE = 16
NEAR_N, FAR_N, FILTER_INV_SQ = 0.2, 100.0, 100.0


def _quat_to_R(q, detach_norm=True):
    s = 1.0 / torch.sqrt((q * q).sum(-1, keepdim=True))
    if detach_norm:
        s = s.detach()  # the reference vjp (auxiliary.h:237-281) ignores the normalisation
    w, x, y, z = (q * s).unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)
    return R.reshape(-1, 3, 3)


def preprocess(means3D, scales, rotations, viewmatrix, projmatrix, W, H, scale_modifier=1.0):
    """-> T[P,3,3] (rows Tu,Tv,Tw), normal[P,3], p_view[P,3].  viewmatrix/projmatrix: [4,4] transposed
    (column-major) matrices as in render/render_2dgs.py:10-24."""
    V = viewmatrix.reshape(4, 4).t()   # true w2c
    Pm = projmatrix.reshape(4, 4).t()  # true full projection
    R = _quat_to_R(rotations)
    L0 = R[:, :, 0] * (scale_modifier * scales[:, 0:1])
    L1 = R[:, :, 1] * (scale_modifier * scales[:, 1:2])
    ones = torch.ones_like(means3D[:, :1])
    zeros = torch.zeros_like(ones)
    h = torch.stack([torch.cat([L0, zeros], 1), torch.cat([L1, zeros], 1), torch.cat([means3D, ones], 1)], 1)  # [P,3,4]
    q = h @ Pm.t()  # [P,3,4] clip coords of each column
    Tu = q[..., 0] * (W / 2.0) + q[..., 3] * ((W - 1) / 2.0)
    Tv = q[..., 1] * (H / 2.0) + q[..., 3] * ((H - 1) / 2.0)
    Tw = q[..., 3]
    T = torch.stack([Tu, Tv, Tw], 1)
    normal = R[:, :, 2] @ V[:3, :3].t()
    p_view = means3D @ V[:3, :3].t() + V[:3, 3]
    return T, normal, p_view


def aabb(T, cutoff=3.0):
    Tu, Tv, Tw = T[:, 0], T[:, 1], T[:, 2]
    temp = torch.tensor([cutoff * cutoff, cutoff * cutoff, -1.0], dtype=T.dtype)
    dist = (Tw * Tw * temp).sum(-1)
    f = temp / dist[:, None]
    center = torch.stack([(f * Tu * Tw).sum(-1), (f * Tv * Tw).sum(-1)], -1)
    tmp = torch.stack([(f * Tu * Tu).sum(-1), (f * Tv * Tv).sum(-1)], -1)
    extent = torch.sqrt(torch.clamp(center * center - tmp, min=1e-4))
    return dist, center, extent


def render(means3D, scales, rotations, opacities, colors, viewmatrix, projmatrix, W, H,
           bg=None, use_sa=True, scale_modifier=1.0, detach_center=True):
    """Forward render.  Returns dict(color[3,H,W], allmap[7,H,W], radii[P], point_list, ranges,
    n_contrib[2,H,W], final_T[H,W])."""
    dt = means3D.dtype
    P = means3D.shape[0]
    bg = torch.zeros(3, dtype=dt) if bg is None else bg.to(dt)
    opacities = opacities.reshape(-1)
    T, normal, p_view = preprocess(means3D, scales, rotations, viewmatrix, projmatrix, W, H, scale_modifier)
    cosv = -(p_view * normal).sum(-1)
    normal = normal * torch.where(cosv > 0, 1.0, -1.0).to(dt)[:, None]
    dist, center, extent = aabb(T)
    radius = torch.ceil(extent.max(-1).values)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    with torch.no_grad():
        ok = (p_view[:, 2] > 0.2) & (cosv != 0) & (dist != 0)
        r = radius.float()
        cf = center.float()
        minx = ((cf[:, 0] - r) / TILE).trunc().clamp(0, gx).long()
        miny = ((cf[:, 1] - r) / TILE).trunc().clamp(0, gy).long()
        maxx = ((cf[:, 0] + r + (TILE - 1)) / TILE).trunc().clamp(0, gx).long()
        maxy = ((cf[:, 1] + r + (TILE - 1)) / TILE).trunc().clamp(0, gy).long()
        ok &= ((maxx - minx) * (maxy - miny)) > 0
        radii = torch.where(ok, radius, torch.zeros_like(radius)).to(torch.int32)
        # duplicate + stable sort by (tile, depth bits, idx)
        ids, tiles = [], []
        for i in torch.nonzero(ok).flatten().tolist():
            ys = torch.arange(int(miny[i]), int(maxy[i]))
            xs = torch.arange(int(minx[i]), int(maxx[i]))
            t = (ys[:, None] * gx + xs[None, :]).flatten()
            tiles.append(t)
            ids.append(torch.full_like(t, i))
        if ids:
            ids = torch.cat(ids)
            tiles = torch.cat(tiles)
            dbits = p_view[:, 2].detach().float().view(torch.int32).long()[ids]
            key = tiles * (1 << 32) + dbits
            order = torch.argsort(key, stable=True)
            point_list = ids[order]
            tiles_sorted = tiles[order]
        else:
            point_list = torch.zeros(0, dtype=torch.long)
            tiles_sorted = torch.zeros(0, dtype=torch.long)
        ranges = torch.zeros(gx * gy, 2, dtype=torch.long)
        if len(point_list):
            cnt = torch.bincount(tiles_sorted, minlength=gx * gy)
            ends = torch.cumsum(cnt, 0)
            nz = cnt > 0
            ranges[nz, 0] = (ends - cnt)[nz]
            ranges[nz, 1] = ends[nz]
    xy_center = center.detach() if detach_center else center
    color = torch.zeros(3, H, W, dtype=dt)
    allmap = torch.zeros(7, H, W, dtype=dt)
    n_contrib = torch.zeros(2, H, W, dtype=torch.long)
    final_T = torch.zeros(H, W, dtype=dt)
    out_c, out_a = [], []
    for tile in range(gx * gy):
        tx, ty = tile % gx, tile // gx
        x0, y0 = tx * TILE, ty * TILE
        x1, y1 = min(x0 + TILE, W), min(y0 + TILE, H)
        ys, xs = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
        px = xs.flatten().to(dt)
        py = ys.flatten().to(dt)
        n = px.shape[0]
        z = torch.zeros(n, dtype=dt)
        Tt = torch.ones(n, dtype=dt)
        C = torch.zeros(n, 3, dtype=dt)
        N = torch.zeros(n, 3, dtype=dt)
        D, D2, M1, M2, dist_acc, median = z, z, z, z, z, z
        median_c = torch.full((n,), -1, dtype=torch.long)
        last = torch.zeros(n, dtype=torch.long)
        done = torch.zeros(n, dtype=torch.bool)
        r0, r1 = int(ranges[tile, 0]), int(ranges[tile, 1])
        for j, g in enumerate(point_list[r0:r1].tolist()):
            contributor = j + 1
            Tu, Tv, Tw = T[g, 0], T[g, 1], T[g, 2]
            k = px[:, None] * Tw[None] - Tu[None]
            l = py[:, None] * Tw[None] - Tv[None]
            p = torch.cross(k, l, dim=-1)
            valid = (~done) & (p[:, 2] != 0)
            pz = torch.where(p[:, 2] != 0, p[:, 2], torch.ones_like(p[:, 2]))
            s0, s1 = p[:, 0] / pz, p[:, 1] / pz
            rho3d = s0 * s0 + s1 * s1
            d0, d1 = xy_center[g, 0] - px, xy_center[g, 1] - py
            rho2d = FILTER_INV_SQ * (d0 * d0 + d1 * d1)
            ray = rho3d <= rho2d
            rho = torch.where(ray, rho3d, rho2d)
            depth = torch.where(ray, s0 * Tw[0] + s1 * Tw[1] + Tw[2], Tw[2].expand(n))
            valid = valid & ~(depth < NEAR_N)
            alpha = torch.clamp(opacities[g] * torch.exp(-0.5 * rho), max=0.99)
            valid = valid & ~(alpha < 1.0 / 255.0)
            test_T = Tt * (1 - alpha)
            stop = valid & (test_T < 0.0001)
            done = done | stop
            valid = valid & ~stop
            if not bool(valid.any()):
                if bool(done.all()):
                    break
                continue
            w = alpha * Tt
            upd_med = valid & (Tt > 0.5)
            median = torch.where(upd_med, depth, median)
            median_c = torch.where(upd_med, torch.full_like(median_c, contributor), median_c)
            if use_sa:
                has = D > 0
                one_m_T = torch.where(has, 1 - Tt, torch.ones_like(Tt))
                exp_std = torch.clamp((D2 - 2 * D * median) / one_m_T + median * median, min=1e-7)
                conf = torch.exp(-((median - depth) ** 2) / (4 * exp_std))
                depth_sa = torch.where(has, conf * depth + (1 - conf) * median, depth)
                D = torch.where(valid, D + depth_sa * w, D)
                D2 = torch.where(valid, D2 + depth_sa * depth_sa * w, D2)
            else:
                A = 1 - Tt
                safe_depth = torch.where(valid, depth, torch.ones_like(depth))
                m = FAR_N / (FAR_N - NEAR_N) * (1 - NEAR_N / safe_depth)
                dist_acc = torch.where(valid, dist_acc + (m * m * A + M2 - 2 * m * M1) * w, dist_acc)
                D = torch.where(valid, D + depth * w, D)
                M1 = torch.where(valid, M1 + m * w, M1)
                M2 = torch.where(valid, M2 + m * m * w, M2)
            wv = torch.where(valid, w, torch.zeros_like(w))
            N = N + normal[g][None] * wv[:, None]
            C = C + colors[g][None] * wv[:, None]
            Tt = torch.where(valid, test_T, Tt)
            last = torch.where(valid, torch.full_like(last, contributor), last)
        reg = D2 - 2 * median * D + (1 - Tt) * median * median if use_sa else dist_acc
        col = C + Tt[:, None] * bg[None]
        am = torch.stack([D, 1 - Tt, N[:, 0], N[:, 1], N[:, 2], median, reg], 0)
        hh, ww = y1 - y0, x1 - x0
        out_c.append((tile, col.t().reshape(3, hh, ww)))
        out_a.append((tile, am.reshape(7, hh, ww)))
        n_contrib[0, y0:y1, x0:x1] = last.reshape(hh, ww)
        n_contrib[1, y0:y1, x0:x1] = median_c.clamp(min=0).reshape(hh, ww)
        final_T[y0:y1, x0:x1] = Tt.detach().reshape(hh, ww)
    # assemble without in-place writes so autograd works
    rows_c, rows_a = [], []
    for ty in range(gy):
        rows_c.append(torch.cat([out_c[ty * gx + tx][1] for tx in range(gx)], 2))
        rows_a.append(torch.cat([out_a[ty * gx + tx][1] for tx in range(gx)], 2))
    color = torch.cat(rows_c, 1)
    allmap = torch.cat(rows_a, 1)
    return dict(color=color, allmap=allmap, radii=radii, point_list=point_list, ranges=ranges,
                n_contrib=n_contrib, final_T=final_T, center=center, T=T)
