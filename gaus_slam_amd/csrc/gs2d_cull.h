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

// Bit layout of the result (shared with the blend kernels).  A tile is 8 x 8 GROUPS of 2 x 2 pixels; group column GX and
// group row GY (0..7) belong to quadrant q = 2 (GY >> 2) + (GX >> 2) (the wave that owns it) and are group
// g = 4 (GY & 3) + (GX & 3) of that quadrant (the wave's lanes 4g .. 4g+3, one DPP quad).  Result bit 16 q + g.
//
// cm = the group columns hit in group row GY (bit GX) -> the two nibbles go to the two quadrants of that row
__device__ __forceinline__ uint64_t place_group_row(uint32_t cm, int GY)
{
    const int sh = ((GY >> 2) * 2) * 16 + (GY & 3) * 4;
    return ((uint64_t)(cm & 0xFu) << sh) | ((uint64_t)((cm >> 4) & 0xFu) << (sh + 16));
}
// Which of the eight 2-pixel-wide group columns (first pixels o, o+2, ..., o+14) does the interval [lo, hi] meet?  Column G
// holds pixels o + 2G and o + 2G + 1: met iff o + 2G + 1 >= lo and o + 2G <= hi, i.e. G in [ceil((lo-o-1)/2), floor((hi-o)/2)]:
// a contiguous run of bits.  NaN bounds select every column (conservative).
__device__ __forceinline__ uint32_t interval_groups(float lo, float hi, float o)
{
    const float fa = ceilf((lo - o - 1.f) * 0.5f), fb = floorf((hi - o) * 0.5f);
    const int a = (int)fminf(fmaxf(fa, 0.f), 8.f);   // NaN -> 0
    const int b = (int)fminf(fmaxf(fb, -1.f), 7.f);  // NaN -> -1 ...
    const uint32_t run = ((2u << b) - 1u) & ~((1u << a) - 1u);  // bits a..b (empty when b < a)
    return (fb == fb && fa == fa) ? (b < 0 ? 0u : run) : 0xFFu;
}

// Conservative test "can this splat reach alpha >= 1/255 on any pixel of group (GX, GY)?" for all 64 groups of the tile
// whose first pixel is (tx0, ty0).  rho_max = 2 ln(255 opacity) (+margin, from the preprocess kernel):
//   alpha >= 1/255  <=>  min(rho3d, rho2d) <= rho_max.
//  (1) {rho2d <= rho_max} is a disc of radius sqrt(rho_max/100) px around the stored centre: its bounding box against the
//      group columns / rows (gs2d_footprint, gs2d_common.h, for the radius).
//  (2) {rho3d <= rho_max} is the image of the surfel's disc u^2+v^2 <= rho_max.  With k = x Tw - Tu, l = y Tw - Tv the
//      kernel's p = k x l is LINEAR in the pixel, p = A dx + B dy + C (A = Tw x l, B = k x Tw, C = k x l at the tile centre),
//      so F = p.x^2 + p.y^2 - rho_max p.z^2 is an exact quadratic whose sign is the sign of rho3d - rho_max.  When the disc
//      lies safely in front of the eye F is convex and {F <= 0} an ellipse, which is RASTERISED: on each of the tile's 16
//      pixel rows F is a parabola in x, its two roots bound the pixels of that row the splat can reach (rounding margin on
//      F, 1e-3 px on the roots); two pixel rows make one group row.
// Anything that cannot be bounded safely is kept.
__device__ __forceinline__ uint64_t splat_touch_mask64(const float4 q0, const float4 q1, const float4 q2, float rho_max,
                                                       float tx0, float ty0)
{
    const Gs2dFootprint fp = gs2d_footprint(q0, q1, q2, rho_max);  // disc radius + validity, shared with the preprocess kernel
    if (fp.kind == 0) return 0ull;
    if (fp.kind == 2) return ~0ull;
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
    if (!(Fxx > 0.f) || !(Fyy > 0.f)) return ~0ull;  // not the convex (ellipse) case after rounding: keep
    // rounding margin: 1e-4 of the largest magnitude the terms of F can reach on the tile
    const float Px = (fabsf(Ax) + fabsf(Bx)) * 7.5f + fabsf(Cx), Py = (fabsf(Ay) + fabsf(By)) * 7.5f + fabsf(Cy);
    const float Pz = (fabsf(Az) + fabsf(Bz)) * 7.5f + fabsf(Cz);
    const float margin = 1e-4f * (Px * Px + Py * Py + c * (Pz * Pz));
    if (!(margin < 1e30f)) return ~0ull;  // overflow / NaN in the coefficients: keep (below this every term of F is finite)
    // low-pass disc {rho2d <= rho_max}: bounding box of the disc against the group columns / rows
    const uint32_t lp_cols = interval_groups(q0.w - fp.rl, q0.w + fp.rl, tx0);
    const uint32_t lp_rows = interval_groups(q1.w - fp.rl, q1.w + fp.rl, ty0);
    // pixel row e (local y): Fxx t^2 + hb t + hc <= margin  <=>  t in tc +- sqrt(hb^2 - 4 Fxx (hc - margin)) / (2 Fxx)
    const float inv2 = 0.5f * cull_rcp(Fxx), F4 = 4.f * Fxx, F0m = F0 - margin;
    uint64_t m = 0ull;
#pragma unroll
    for (int GY = 0; GY < 8; GY++) {
        float lo = 1e30f, hi = -1e30f;  // local pixel coordinates (-7.5 .. 7.5) reached on the two rows of the group row
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const float e = -7.5f + (float)(2 * GY + r);
            const float hb = 2.f * fmaf(Fxy, e, Fx), hc = fmaf(fmaf(Fyy, e, 2.f * Fy), e, F0m);
            const float disc = fmaf(hb, hb, -(F4 * hc));
            const float rad = cull_sqrt(fmaxf(disc, 0.f)) * inv2, tc = -hb * inv2;
            const bool hit = disc >= 0.f;  // (all finite: margin guard above)
            lo = hit ? fminf(lo, tc - rad) : lo;
            hi = hit ? fmaxf(hi, tc + rad) : hi;
        }
        // pixel index = local coordinate + 7.5; group column G holds pixels 2G, 2G+1
        const float fa = ceilf((lo + (7.5f - 1.f - 1e-3f)) * 0.5f), fb = floorf((hi + (7.5f + 1e-3f)) * 0.5f);
        const int a = (int)fminf(fmaxf(fa, 0.f), 8.f);
        const int b = (int)fminf(fmaxf(fb, -1.f), 7.f);
        uint32_t cm = (b < 0) ? 0u : (((2u << b) - 1u) & ~((1u << a) - 1u));
        cm |= ((lp_rows >> GY) & 1u) ? lp_cols : 0u;
        m |= place_group_row(cm, GY);
    }
    return m;
}

// Four row bits of a quadrant (the backward's 4x4 sub-blocks, one per 16-lane DPP row) from its sixteen group bits: row
// r = 2 ry + rx covers group columns 2rx, 2rx+1 and group rows 2ry, 2ry+1 (bit 4 gy + gx).
__device__ __forceinline__ uint32_t rows_from_groups(uint32_t g16)
{
    return ((g16 & 0x0033u) ? 1u : 0u) | ((g16 & 0x00CCu) ? 2u : 0u) | ((g16 & 0x3300u) ? 4u : 0u) | ((g16 & 0xCC00u) ? 8u : 0u);
}
// EXPERIMENT builds with eight backward queues per wave (-DGS2D_BWD_GROUPS=8 / -DGS2D_BWD_POSE_GROUPS=8, see gs2d_blend.hip and
// profiles/bwd_queue_ab_r04.txt) store eight HALF-ROW bits per quadrant instead: from its sixteen group bits (bit 4 gy + gx).  The backward's 4x4 sub-block r = 2 ry + rx
// (one per 16-lane DPP row) covers group columns 2rx, 2rx+1 and group rows 2ry, 2ry+1; its half h (4x2 pixels, the 8 lanes of
// half a DPP row) is group row 2ry + h.  Bit 2r + h.  The four ROW bits of the 4-queue backward are the ORs of the pairs
// (rows_from_halfrows).
__device__ __forceinline__ uint32_t halfrows_from_groups(uint32_t g16)
{
    return ((g16 & 0x0003u) ? 0x01u : 0u) | ((g16 & 0x0030u) ? 0x02u : 0u) | ((g16 & 0x000Cu) ? 0x04u : 0u) | ((g16 & 0x00C0u) ? 0x08u : 0u) |
           ((g16 & 0x0300u) ? 0x10u : 0u) | ((g16 & 0x3000u) ? 0x20u : 0u) | ((g16 & 0x0C00u) ? 0x40u : 0u) | ((g16 & 0xC000u) ? 0x80u : 0u);
}
// The same eight bits by bit tricks (13 operations instead of ~24): OR the adjacent bit pairs, compress the eight pair bits
// (pair j = 4 ry + 2 h + rx) to a byte, swap the two middle positions of each nibble (-> bit 4 ry + 2 rx + h = 2 r + h).
// Used by the eight-queue pose-only backward, which derives its half-row bits from the forward's group bits itself.
__device__ __forceinline__ uint32_t halfrows_from_groups_fast(uint32_t g16)
{
    uint32_t x = (g16 | (g16 >> 1)) & 0x5555u;
    x = (x | (x >> 1)) & 0x3333u;
    x = (x | (x >> 2)) & 0x0F0Fu;
    x = (x | (x >> 4)) & 0x00FFu;
    return (x & 0x99u) | ((x & 0x22u) << 1) | ((x & 0x44u) >> 1);
}
__device__ __forceinline__ uint32_t rows_from_halfrows(uint32_t h8)
{
    const uint32_t t = h8 | (h8 >> 1);  // bit 2r: row r
    return (t & 1u) | ((t >> 1) & 2u) | ((t >> 2) & 4u) | ((t >> 3) & 8u);
}

#ifndef GS2D_CULL_T
#define GS2D_CULL_T 256
#endif
// The tile's sorted segment walked by 256 threads, thread t takes instances t, t+256, ...; the two dependent gathers of
// the NEXT instance (id, then 52 bytes of its record) are issued before the ~700 instructions of the current one, so only
// a wave's first gather is exposed.  hits[i] = the 64 group bits of instance i (16 bits per quadrant).  (Tried and dropped: 1024-thread workgroups, -25 %; a flat one-instance-per-thread grid fed by
// per-instance tile ids from the depth sort, -30 %: every wave then pays both gather latencies for one instance.)
__device__ __forceinline__ void cull_tile_list(const uint2 range, float tx0, float ty0, const uint32_t* __restrict__ point_list,
                                               const float4* __restrict__ rec, uint64_t* __restrict__ hits,
                                               uint32_t* __restrict__ hits4)
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
        const uint64_t m64 = splat_touch_mask64(r0, r1, r2, rho_max, tx0, ty0);
        hits[i] = m64;
#if GS2D_HALFROW_BITS
        hits4[i] = halfrows_from_groups((uint32_t)m64 & 0xFFFFu) | (halfrows_from_groups((uint32_t)(m64 >> 16) & 0xFFFFu) << 8) |
                   (halfrows_from_groups((uint32_t)(m64 >> 32) & 0xFFFFu) << 16) | (halfrows_from_groups((uint32_t)(m64 >> 48)) << 24);
#else
        hits4[i] = rows_from_groups((uint32_t)m64 & 0xFFFFu) | (rows_from_groups((uint32_t)(m64 >> 16) & 0xFFFFu) << 8) |
                   (rows_from_groups((uint32_t)(m64 >> 32) & 0xFFFFu) << 16) | (rows_from_groups((uint32_t)(m64 >> 48)) << 24);
#endif
        i += GS2D_CULL_T;
        if (i >= range.y) break;
        r0 = n0; r1 = n1; r2 = n2; rho_max = nrho; id_next = id_next2;
    }
}

}  // namespace
