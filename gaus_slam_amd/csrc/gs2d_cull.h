// Per-instance sub-block cull: ONE pass over a tile's depth-sorted instance list that decides, for every (Gaussian, tile)
// instance, which of the tile's sixteen 4x4 pixel sub-blocks the splat can reach with alpha >= 1/255.  Both blend
// kernels build their per-sub-block queues from these bits, so neither evaluates a cull test nor loads the record of a
// splat that touches nothing in its quadrant.  (Round 1 ran the test once per quadrant wave, i.e. the conic of every
// instance was rebuilt four times: about a third of blend_fwd's vector instructions.)  The pass runs as phase 0 of
// blend_fwd_kernel (each workgroup culls its own tile's list, then blends it): as a kernel of its own it cost 30 us plus
// a launch boundary, inline its latency stalls are filled by other workgroups' blending (about 20 us).
//
// Culling never changes a result: a culled (sub-block, splat) pair is one whose every pixel the reference would
// `continue` past (alpha < 1/255, forward.cu:385-387 / backward.cu:319-320); tests: bit-exact n_contrib on a stress scene.
#pragma once
#include "gs2d_common.h"

namespace {

__device__ __forceinline__ float cull_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float cull_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// Bit layout of the result (shared with the blend kernels): byte q = quadrant q of the tile (q = 2*(y>=8) + (x>=8), the
// wave that owns it), bit r of that byte's low nibble = 4x4 sub-block r of the quadrant (r = 2*(y%8>=4) + (x%8>=4), the
// wave's DPP row).  With sub-block columns sx and rows sy in 0..3 the bit index is 8*(sy>>1) + 4*(sx>>1) + 2*(sy&1) + (sx&1),
// so "column sx" and "row sy" are the constant masks below and a separable decision (x-interval AND y-interval) is one AND.
// bits4: bit sx set = column sx selected.  The four column patterns are 0x0505 << {0, 1, 4, 5}: disjoint, so the selection
// is one multiplication by the bits moved to those positions.
__device__ __forceinline__ uint32_t col_mask(uint32_t bits4)
{
    return 0x0505u * ((bits4 & 3u) | ((bits4 & 12u) << 2));
}
// row patterns: 0x0033 << {0, 2, 8, 10}
__device__ __forceinline__ uint32_t row_mask(uint32_t bits4)
{
    return 0x0033u * ((bits4 & 1u) | ((bits4 & 2u) << 1) | ((bits4 & 4u) << 6) | ((bits4 & 8u) << 7));
}
// Which of the four 4-pixel-wide sub-block columns (first pixels o, o+4, o+8, o+12; a column spans [x0, x0+3]) does the
// interval [lo, hi] meet?  Column s is met iff x0_s + 3 >= lo and x0_s <= hi, i.e. s in [ceil((lo-o-3)/4), floor((hi-o)/4)]:
// a contiguous run of bits.  NaN bounds select every column (conservative).
__device__ __forceinline__ uint32_t interval_cols(float lo, float hi, float o)
{
    const float fa = ceilf((lo - o - 3.f) * 0.25f), fb = floorf((hi - o) * 0.25f);
    const int a = (int)fminf(fmaxf(fa, 0.f), 4.f);   // NaN -> 0
    const int b = (int)fminf(fmaxf(fb, -1.f), 3.f);  // NaN -> -1 ...
    const uint32_t run = ((2u << b) - 1u) & ~((1u << a) - 1u);  // bits a..b (empty when b < a; b = -1: 2u << -1 is avoided below)
    return (fb == fb && fa == fa) ? (b < 0 ? 0u : run) : 0xFu;
}
__device__ __forceinline__ int sub_bit(int sx, int sy) { return 8 * (sy >> 1) + 4 * (sx >> 1) + 2 * (sy & 1) + (sx & 1); }

// Conservative test "can this splat reach alpha >= 1/255 on any pixel of sub-block (sx, sy)?" for all sixteen sub-blocks
// of the tile whose first pixel is (tx0, ty0).  rho_max = 2 ln(255 opacity) (+margin, from the preprocess kernel):
//   alpha >= 1/255  <=>  min(rho3d, rho2d) <= rho_max.
//  (1), (2) the disc {rho2d <= rho_max} and the AABB of the ellipse {rho3d <= rho_max}: gs2d_footprint (gs2d_common.h).
//  (3) Inside the AABB the ellipse itself is tested: with k = x Tw - Tu, l = y Tw - Tv the kernel's p = k x l is LINEAR in
//      the pixel, p = A dx + B dy + C (A = Tw x l, B = k x Tw, C = k x l at the tile centre), so
//      F = p.x^2 + p.y^2 - rho_max p.z^2 is an exact quadratic whose sign is the sign of rho3d - rho_max.  For a convex F
//      the minimum over a rectangle lies at the ellipse centre (if inside) or on one of the 4 edges, where F is a 1-D
//      parabola; the 8 + 8 edge lines of the 16 sub-blocks are shared, so a sub-block costs four clamped evaluations.
//      A sub-block is dropped only if that minimum exceeds a rounding margin.
// Anything that cannot be bounded safely is kept.
__device__ __forceinline__ uint32_t splat_touch_mask16(const float4 q0, const float4 q1, const float4 q2, float rho_max,
                                                       float tx0, float ty0)
{
    const Gs2dFootprint fp = gs2d_footprint(q0, q1, q2, rho_max);  // disc + ellipse AABB, shared with the preprocess kernel
    if (fp.kind == 0) return 0u;
    const float rl = fp.rl;
    // low-pass disc {rho2d <= rho_max}: bounding box of the disc against the sub-block columns / rows
    const uint32_t lp = col_mask(interval_cols(q0.w - rl, q0.w + rl, tx0)) & row_mask(interval_cols(q1.w - rl, q1.w + rl, ty0));
    if (fp.kind == 2) return 0xFFFFu;
    const float cx = fp.cx, cy = fp.cy, ex = fp.ex, ey = fp.ey, mx = fp.mx, my = fp.my;
    // AABB of the ellipse / its centre against the sub-block columns and rows
    const uint32_t in_aabb = col_mask(interval_cols(cx - ex - mx, cx + ex + mx, tx0)) & row_mask(interval_cols(cy - ey - my, cy + ey + my, ty0));
    if (in_aabb == 0u) return lp;  // the ellipse's AABB misses the tile
    // ellipse centre inside (or within the margin of) the sub-block
    const uint32_t centre_in = col_mask(interval_cols(cx - mx, cx + mx, tx0)) & row_mask(interval_cols(cy - my, cy + my, ty0));
    // exact conic in coordinates local to the tile centre (|dx|, |dy| <= 7.5: well conditioned)
    const float xm = tx0 + 7.5f, ym = ty0 + 7.5f;
    const float k0 = fmaf(xm, q2.x, -q0.x), k1 = fmaf(xm, q2.y, -q0.y), k2 = fmaf(xm, q2.z, -q0.z);
    const float l0 = fmaf(ym, q2.x, -q1.x), l1 = fmaf(ym, q2.y, -q1.y), l2 = fmaf(ym, q2.z, -q1.z);
    const float Cx = k1 * l2 - k2 * l1, Cy = k2 * l0 - k0 * l2, Cz = k0 * l1 - k1 * l0;          // k x l
    const float Ax = q2.y * l2 - q2.z * l1, Ay = q2.z * l0 - q2.x * l2, Az = q2.x * l1 - q2.y * l0;  // Tw x l
    const float Bx = k1 * q2.z - k2 * q2.y, By = k2 * q2.x - k0 * q2.z, Bz = k0 * q2.y - k1 * q2.x;  // k x Tw
    const float c = rho_max;
    const float Fxx = Ax * Ax + Ay * Ay - c * (Az * Az), Fyy = Bx * Bx + By * By - c * (Bz * Bz);
    const float Fxy = Ax * Bx + Ay * By - c * (Az * Bz);
    const float Fx = Ax * Cx + Ay * Cy - c * (Az * Cz), Fy = Bx * Cx + By * Cy - c * (Bz * Cz);
    const float F0 = Cx * Cx + Cy * Cy - c * (Cz * Cz);
    if (!(Fxx > 0.f) || !(Fyy > 0.f)) return 0xFFFFu;  // not the convex (ellipse) case after rounding: keep
    // rounding margin: 1e-4 of the largest magnitude the terms of F can reach on the tile
    const float Px = (fabsf(Ax) + fabsf(Bx)) * 7.5f + fabsf(Cx), Py = (fabsf(Ay) + fabsf(By)) * 7.5f + fabsf(Cy);
    const float Pz = (fabsf(Az) + fabsf(Bz)) * 7.5f + fabsf(Cz);
    const float margin = 1e-4f * (Px * Px + Py * Py + c * (Pz * Pz));
    if (!(margin < 1e30f)) return 0xFFFFu;  // overflow / NaN in the coefficients: keep (below this every term of F is finite)
    const float nhx = -0.5f * cull_rcp(Fxx), nhy = -0.5f * cull_rcp(Fyy);
    // the 8 horizontal and 8 vertical edge lines of the sub-blocks (local coordinate e of line i: -7.5 + 4 (i/2) + 3 (i%2)):
    //   F(t, e) = Fxx t^2 + hb t + hc   (horizontal, y = e)      F(e, t) = Fyy t^2 + vb t + vc   (vertical, x = e)
    float hb[8], hc[8], ht[8], vb[8], vc[8], vt[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float e = -7.5f + 4.f * (i >> 1) + 3.f * (i & 1);
        hb[i] = 2.f * fmaf(Fxy, e, Fx); hc[i] = fmaf(fmaf(Fyy, e, 2.f * Fy), e, F0); ht[i] = hb[i] * nhx;  // unconstrained minimiser
        vb[i] = 2.f * fmaf(Fxy, e, Fy); vc[i] = fmaf(fmaf(Fxx, e, 2.f * Fx), e, F0); vt[i] = vb[i] * nhy;
    }
    // every sub-block is evaluated (straight-line code: a wave's 64 instances never agree on which ones are needed)
    uint32_t edge_hit = 0u;
#pragma unroll
    for (int sy = 0; sy < 4; sy++) {
#pragma unroll
        for (int sx = 0; sx < 4; sx++) {
            const float xa = -7.5f + 4.f * sx, xb = xa + 3.f, ya = -7.5f + 4.f * sy, yb = ya + 3.f;
            const float t0 = __builtin_amdgcn_fmed3f(ht[2 * sy], xa, xb), t1 = __builtin_amdgcn_fmed3f(ht[2 * sy + 1], xa, xb);
            const float u0 = __builtin_amdgcn_fmed3f(vt[2 * sx], ya, yb), u1 = __builtin_amdgcn_fmed3f(vt[2 * sx + 1], ya, yb);
            const float f0_ = fmaf(fmaf(Fxx, t0, hb[2 * sy]), t0, hc[2 * sy]);
            const float f1_ = fmaf(fmaf(Fxx, t1, hb[2 * sy + 1]), t1, hc[2 * sy + 1]);
            const float g0_ = fmaf(fmaf(Fyy, u0, vb[2 * sx]), u0, vc[2 * sx]);
            const float g1_ = fmaf(fmaf(Fyy, u1, vb[2 * sx + 1]), u1, vc[2 * sx + 1]);
            const float fmin_edges = fminf(fminf(f0_, f1_), fminf(g0_, g1_));  // all finite (margin guard above)
            if (!(fmin_edges > margin)) edge_hit |= 1u << sub_bit(sx, sy);
        }
    }
    const uint32_t m = lp | (in_aabb & (centre_in | edge_hit));
    return m;
}


#ifndef GS2D_CULL_T
#define GS2D_CULL_T 256
#endif
// The tile's sorted segment walked by 256 threads, thread t takes instances t, t+256, ...; the two dependent gathers of
// the NEXT instance (id, then 52 bytes of its record) are issued before the ~700 instructions of the current one, so only
// a wave's first gather is exposed.  hits[i] = the 16 sub-block bits of instance i, spread over four bytes (byte q = nibble
// of quadrant q).  (Tried and dropped: 1024-thread workgroups, -25 %; a flat one-instance-per-thread grid fed by
// per-instance tile ids from the depth sort, -30 %: every wave then pays both gather latencies for one instance.)
__device__ __forceinline__ void cull_tile_list(const uint2 range, float tx0, float ty0, const uint32_t* __restrict__ point_list,
                                               const float4* __restrict__ rec, uint32_t* __restrict__ hits)
{
    uint32_t i = range.x + threadIdx.x;
    if (i >= range.y) return;
    const uint32_t last = range.y - 1;
    // software pipeline: ids are fetched two instances ahead, records one ahead.  Every load is unconditional (indices
    // clamped to the segment): a load inside a branch makes the compiler drain ALL pending loads at the join, i.e. wait
    // for the prefetch before the work it is meant to overlap.
    uint32_t id_next = point_list[min(i + GS2D_CULL_T, last)];
    const float4* rp = rec + (size_t)point_list[i] * GS2D_REC_F4;
    float4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
    float rho_max = reinterpret_cast<const float*>(rp)[18];  // q4.z
    for (;;) {
        const uint32_t id_next2 = point_list[min(i + 2 * GS2D_CULL_T, last)];
        const float4* np = rec + (size_t)id_next * GS2D_REC_F4;
        const float4 n0 = np[0], n1 = np[1], n2 = np[2];
        const float nrho = reinterpret_cast<const float*>(np)[18];
        const uint32_t m = splat_touch_mask16(r0, r1, r2, rho_max, tx0, ty0);
        hits[i] = (m & 0xFu) | ((m & 0xF0u) << 4) | ((m & 0xF00u) << 8) | ((m & 0xF000u) << 12);
        i += GS2D_CULL_T;
        if (i >= range.y) break;
        r0 = n0; r1 = n1; r2 = n2; rho_max = nrho; id_next = id_next2;
    }
}

}  // namespace
