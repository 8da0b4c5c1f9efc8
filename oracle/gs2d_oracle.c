/*
 * gs2d_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A scalar float32 restatement of the reference 2D-Gaussian-surfel tile
 * rasterizer (vasabi-root/gaus-slam, submodules/gaus_2dgs_rasterization,
 * abbreviated RAST/ below).  Each stage cites the reference file:line whose
 * semantics it follows.  It exists so that tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg can check the HIP kernels; nothing in the
 * shipped package may import, link or call it.
 *
 * PARITY STATUS: "parity unpinned" against the reference binary -- the
 * reference ships no tests / golden vectors and is CUDA-only,
 * so it cannot be executed here.  This restatement is pinned by (1) an
 * independent pure-PyTorch forward + autograd (oracle/torch_ref.py),
 * (2) closed-form micro-cases and (3) structural invariants (tests/).
 *
 * Conscious deviations from the reference, all below float tolerance:
 *   - quaternion normalisation uses 1/sqrtf (IEEE) instead of CUDA rsqrtf
 *     (<= 2 ulp approximate), so that oracle and HIP can agree bit-for-bit
 *     on the geometry that drives tile binning and the depth sort;
 *   - per-Gaussian gradient sums are accumulated in double (the reference
 *     uses float atomicAdd in a non-deterministic order);
 *   - the per-(pixel,splat) arithmetic of the two blend stages is written in an explicit
 *     fused-multiply-add form (fmaf) and with reciprocal-then-multiply instead of divisions.
 *     nvcc contracts the reference's expressions into FMAs in a pattern of its own choosing
 *     (--fmad=true default), so no particular contraction is "the" reference; this file fixes
 *     one and the HIP kernels use exactly the same one, which leaves only the hardware
 *     rcp/exp approximations (<= 1-2 ulp) between the two;
 *   - float->int conversions saturate and map NaN to 0 (CUDA cvt.rzi / AMD
 *     v_cvt semantics) instead of x86's INT_MIN.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 * (see oracle/Makefile).  -ffp-contract=off is REQUIRED: tile bins and the
 * depth sort are compared bit-exactly against the HIP preprocess kernel,
 * which is compiled with contraction off as well.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 16            /* RAST/cuda_rasterizer/config.h:15-17 */
#define NEAR_N 0.2f        /* auxiliary.h:37 */
#define FAR_N 100.0f       /* auxiliary.h:38 */
#define FILTER_INV_SQ 100.0f /* auxiliary.h:39 */

/* SH constants, auxiliary.h:42-59 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

/* CUDA min/max on floats are fminf/fmaxf (NaN-ignoring). */
static inline float fmin_c(float a, float b) { return fminf(a, b); }
static inline float fmax_c(float a, float b) { return fmaxf(a, b); }

/* float -> int32, truncation toward zero, saturating, NaN -> 0. */
static inline int f2i_sat(float v)
{
    if (!(v == v)) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (int)(-2147483647 - 1);
    return (int)v;
}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* rasterizer_impl.cu:35-50 */
uint32_t orc_higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* auxiliary.h:66-76 ; grid = (gx, gy) tiles. */
static void get_rect(float px, float py, int max_radius, int gx, int gy,
                     int* minx, int* miny, int* maxx, int* maxy)
{
    const float r = (float)max_radius;
    *minx = imin(gx, imax(0, f2i_sat((px - r) / (float)TILE)));
    *miny = imin(gy, imax(0, f2i_sat((py - r) / (float)TILE)));
    *maxx = imin(gx, imax(0, f2i_sat((px + r + (float)(TILE - 1)) / (float)TILE)));
    *maxy = imin(gy, imax(0, f2i_sat((py + r + (float)(TILE - 1)) / (float)TILE)));
}

/* auxiliary.h:212-234 (unit quaternion (w,x,y,z) -> R, returned as R[row][col]). */
static void quat_to_R(const float* q, float R[3][3])
{
    const float qw = q[0], qx = q[1], qy = q[2], qz = q[3];
    const float s = 1.0f / sqrtf(qz * qz + qw * qw + qx * qx + qy * qy);
    const float w = qw * s, x = qx * s, y = qy * s, z = qz * s;
    R[0][0] = 1.f - 2.f * (y * y + z * z);
    R[1][0] = 2.f * (x * y + w * z);
    R[2][0] = 2.f * (x * z - w * y);
    R[0][1] = 2.f * (x * y - w * z);
    R[1][1] = 1.f - 2.f * (x * x + z * z);
    R[2][1] = 2.f * (y * z + w * x);
    R[0][2] = 2.f * (x * z + w * y);
    R[1][2] = 2.f * (y * z - w * x);
    R[2][2] = 1.f - 2.f * (x * x + y * y);
}

/* Shared by forward (forward.cu:75-115) and backward (backward.cu:503-528):
 * T rows Tu,Tv,Tw (pixel_homog = T * (u,v,1)) and the view-space normal.
 * pm/vm are column-major 4x4 as passed by render/render_2dgs.py:10-15. */
static void compute_transmat(const float* p, float sx, float sy, const float* quat,
                             const float* pm, const float* vm, int W, int H,
                             float T[9], float normal[3], float Rout[3][3])
{
    float R[3][3];
    quat_to_R(quat, R);
    if (Rout) memcpy(Rout, R, sizeof(R));
    const float halfW = (float)W * 0.5f, halfWm = (float)(W - 1) * 0.5f;
    const float halfH = (float)H * 0.5f, halfHm = (float)(H - 1) * 0.5f;
    for (int i = 0; i < 3; i++) {
        float hx, hy, hz;
        if (i == 0) { hx = R[0][0] * sx; hy = R[1][0] * sx; hz = R[2][0] * sx; }
        else if (i == 1) { hx = R[0][1] * sy; hy = R[1][1] * sy; hz = R[2][1] * sy; }
        else { hx = p[0]; hy = p[1]; hz = p[2]; }
        float q0 = (pm[0] * hx + pm[4] * hy) + pm[8] * hz;
        float q1 = (pm[1] * hx + pm[5] * hy) + pm[9] * hz;
        float q3 = (pm[3] * hx + pm[7] * hy) + pm[11] * hz;
        if (i == 2) { q0 = q0 + pm[12]; q1 = q1 + pm[13]; q3 = q3 + pm[15]; }
        T[0 + i] = q0 * halfW + q3 * halfWm; /* Tu[i] */
        T[3 + i] = q1 * halfH + q3 * halfHm; /* Tv[i] */
        T[6 + i] = q3;                       /* Tw[i] */
    }
    /* normal = V_3x3 * R[:,2]  (auxiliary.h:99-107) */
    const float lx = R[0][2], ly = R[1][2], lz = R[2][2];
    normal[0] = (vm[0] * lx + vm[4] * ly) + vm[8] * lz;
    normal[1] = (vm[1] * lx + vm[5] * ly) + vm[9] * lz;
    normal[2] = (vm[2] * lx + vm[6] * ly) + vm[10] * lz;
}

/* forward.cu:20-71 */
static void sh_to_rgb(int idx, int deg, int M, const float* means, const float* campos,
                      const float* shs, uint8_t* clamped, float out[3])
{
    float dx = means[3 * idx] - campos[0], dy = means[3 * idx + 1] - campos[1], dz = means[3 * idx + 2] - campos[2];
    const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
    const float x = dx / len, y = dy / len, z = dz / len;
    const float* sh = shs + (size_t)idx * M * 3;
    for (int c = 0; c < 3; c++) {
#define SHC(k) sh[(k) * 3 + c]
        float r = SH_C0 * SHC(0);
        if (deg > 0) {
            r = r - SH_C1 * y * SHC(1) + SH_C1 * z * SHC(2) - SH_C1 * x * SHC(3);
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * SHC(4) + SH_C2[1] * yz * SHC(5) +
                    SH_C2[2] * (2.0f * zz - xx - yy) * SHC(6) + SH_C2[3] * xz * SHC(7) +
                    SH_C2[4] * (xx - yy) * SHC(8);
                if (deg > 2) {
                    r = r + SH_C3[0] * y * (3.0f * xx - yy) * SHC(9) + SH_C3[1] * xy * z * SHC(10) +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * SHC(11) +
                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SHC(12) +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * SHC(13) +
                        SH_C3[5] * z * (xx - yy) * SHC(14) + SH_C3[6] * x * (xx - 3.0f * yy) * SHC(15);
                }
            }
        }
#undef SHC
        r += 0.5f;
        clamped[3 * idx + c] = (r < 0);
        out[c] = fmax_c(r, 0.0f);
    }
}

/*
 * Stage 1: per-Gaussian preprocess + inclusive scan.
 * forward.cu:150-253 (preprocessCUDA), auxiliary.h:184-209 (in_frustum),
 * forward.cu:119-147 (compute_aabb), rasterizer_impl.cu:283 (InclusiveSum).
 * Null pointers select the alternative paths exactly as the reference does
 * (transMat_precomp / colors_precomp, rasterizer_impl.cu:327-328).
 * All outputs are zero-filled for culled Gaussians (the reference leaves them
 * uninitialised).  Returns num_rendered = point_offsets[P-1].
 */
int64_t orc_preprocess(int P, int D, int M, const float* means3D, const float* scales,
                       float scale_modifier, const float* rotations, const float* opacities,
                       const float* shs, const float* transMat_precomp, const float* colors_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* campos,
                       int W, int H,
                       int32_t* radii, float* means2D, float* depths, float* transMats, float* rgb,
                       float* normal_opacity, uint32_t* tiles_touched, uint8_t* clamped,
                       uint32_t* point_offsets)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    memset(radii, 0, sizeof(int32_t) * P);
    memset(means2D, 0, sizeof(float) * 2 * P);
    memset(depths, 0, sizeof(float) * P);
    memset(transMats, 0, sizeof(float) * 9 * P);
    memset(rgb, 0, sizeof(float) * 3 * P);
    memset(normal_opacity, 0, sizeof(float) * 4 * P);
    memset(tiles_touched, 0, sizeof(uint32_t) * P);
    memset(clamped, 0, 3 * (size_t)P);
    for (int idx = 0; idx < P; idx++) {
        const float* p = means3D + 3 * idx;
        const float* vm = viewmatrix;
        /* in_frustum: only the view-space z test is live (auxiliary.h:197-199) */
        const float pvx = ((vm[0] * p[0] + vm[4] * p[1]) + vm[8] * p[2]) + vm[12];
        const float pvy = ((vm[1] * p[0] + vm[5] * p[1]) + vm[9] * p[2]) + vm[13];
        const float pvz = ((vm[2] * p[0] + vm[6] * p[1]) + vm[10] * p[2]) + vm[14];
        if (pvz <= 0.2f) continue;
        float T[9], normal[3];
        if (transMat_precomp == NULL) {
            compute_transmat(p, scale_modifier * scales[2 * idx], scale_modifier * scales[2 * idx + 1],
                             rotations + 4 * idx, projmatrix, viewmatrix, W, H, T, normal, NULL);
            memcpy(transMats + 9 * idx, T, sizeof(T));
        } else {
            memcpy(T, transMat_precomp + 9 * idx, sizeof(T));
            normal[0] = 0.f; normal[1] = 0.f; normal[2] = 1.f;
        }
        /* dual-visible flip, forward.cu:211-216 */
        const float cosv = -((pvx * normal[0] + pvy * normal[1]) + pvz * normal[2]);
        if (cosv == 0) continue;
        const float mult = cosv > 0 ? 1.f : -1.f;
        normal[0] = mult * normal[0]; normal[1] = mult * normal[1]; normal[2] = mult * normal[2];
        /* compute_aabb, cutoff = 3 (forward.cu:222) */
        const float c2 = 3.0f * 3.0f;
        const float dist = ((T[6] * T[6]) * c2 + (T[7] * T[7]) * c2) + (T[8] * T[8]) * -1.0f;
        const float inv = 1 / dist;
        const float f0 = inv * c2, f1 = inv * c2, f2 = inv * -1.0f;
        if (dist == 0.0f) continue;
        const float cx = ((f0 * T[0]) * T[6] + (f1 * T[1]) * T[7]) + (f2 * T[2]) * T[8];
        const float cy = ((f0 * T[3]) * T[6] + (f1 * T[4]) * T[7]) + (f2 * T[5]) * T[8];
        const float tx = ((f0 * T[0]) * T[0] + (f1 * T[1]) * T[1]) + (f2 * T[2]) * T[2];
        const float ty = ((f0 * T[3]) * T[3] + (f1 * T[4]) * T[4]) + (f2 * T[5]) * T[5];
        const float ex = sqrtf(fmax_c(1e-4f, cx * cx - tx));
        const float ey = sqrtf(fmax_c(1e-4f, cy * cy - ty));
        const float radius = ceilf(fmax_c(ex, ey));
        int minx, miny, maxx, maxy;
        get_rect(cx, cy, f2i_sat(radius), gx, gy, &minx, &miny, &maxx, &maxy);
        if ((maxx - minx) * (maxy - miny) == 0) continue;
        if (colors_precomp == NULL) {
            float c[3];
            sh_to_rgb(idx, D, M, means3D, campos, shs, clamped, c);
            rgb[3 * idx] = c[0]; rgb[3 * idx + 1] = c[1]; rgb[3 * idx + 2] = c[2];
        }
        depths[idx] = pvz;
        radii[idx] = f2i_sat(radius);
        means2D[2 * idx] = cx; means2D[2 * idx + 1] = cy;
        normal_opacity[4 * idx] = normal[0]; normal_opacity[4 * idx + 1] = normal[1];
        normal_opacity[4 * idx + 2] = normal[2]; normal_opacity[4 * idx + 3] = opacities[idx];
        tiles_touched[idx] = (uint32_t)((maxy - miny) * (maxx - minx));
    }
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += tiles_touched[i]; point_offsets[i] = acc; }
    return P > 0 ? (int64_t)acc : 0;
}

/* rasterizer_impl.cu:70-111 (duplicateWithKeys) */
void orc_duplicate(int P, const float* means2D, const float* depths, const uint32_t* point_offsets,
                   const int32_t* radii, int W, int H, uint64_t* keys, uint32_t* vals)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] <= 0) continue;
        uint32_t off = idx == 0 ? 0 : point_offsets[idx - 1];
        int minx, miny, maxx, maxy;
        get_rect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, &minx, &miny, &maxx, &maxy);
        uint32_t dbits;
        memcpy(&dbits, depths + idx, 4);
        for (int y = miny; y < maxy; y++)
            for (int x = minx; x < maxx; x++) {
                uint64_t key = (uint64_t)((uint32_t)y * (uint32_t)gx + (uint32_t)x);
                key <<= 32;
                key |= dbits;
                keys[off] = key; vals[off] = (uint32_t)idx; off++;
            }
    }
}

/* rasterizer_impl.cu:309-314: stable LSD radix sort on the low `nbits` key bits
 * (cub::DeviceRadixSort::SortPairs).  Stability makes ties keep duplicate order. */
void orc_sort_pairs(int64_t R, const uint64_t* keys_in, const uint32_t* vals_in,
                    uint64_t* keys_out, uint32_t* vals_out, int nbits)
{
    if (R <= 0) return;
    uint64_t* ka = (uint64_t*)malloc(sizeof(uint64_t) * R);
    uint64_t* kb = (uint64_t*)malloc(sizeof(uint64_t) * R);
    uint32_t* va = (uint32_t*)malloc(sizeof(uint32_t) * R);
    uint32_t* vb = (uint32_t*)malloc(sizeof(uint32_t) * R);
    memcpy(ka, keys_in, sizeof(uint64_t) * R);
    memcpy(va, vals_in, sizeof(uint32_t) * R);
    for (int shift = 0; shift < nbits; shift += 8) {
        const int bits = nbits - shift < 8 ? nbits - shift : 8;
        const uint64_t mask = (1ull << bits) - 1;
        int64_t cnt[257] = {0};
        for (int64_t i = 0; i < R; i++) cnt[((ka[i] >> shift) & mask) + 1]++;
        for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
        for (int64_t i = 0; i < R; i++) {
            const int64_t dst = cnt[(ka[i] >> shift) & mask]++;
            kb[dst] = ka[i]; vb[dst] = va[i];
        }
        uint64_t* tk = ka; ka = kb; kb = tk;
        uint32_t* tv = va; va = vb; vb = tv;
    }
    memcpy(keys_out, ka, sizeof(uint64_t) * R);
    memcpy(vals_out, va, sizeof(uint32_t) * R);
    free(ka); free(kb); free(va); free(vb);
}

/* rasterizer_impl.cu:116-138,316 (memset + identifyTileRanges); ranges = [tiles][2] */
void orc_tile_ranges(int64_t R, const uint64_t* keys_sorted, int ntiles, uint32_t* ranges)
{
    memset(ranges, 0, sizeof(uint32_t) * 2 * ntiles);
    for (int64_t i = 0; i < R; i++) {
        const uint32_t cur = (uint32_t)(keys_sorted[i] >> 32);
        if (i == 0) ranges[2 * cur] = 0;
        else {
            const uint32_t prev = (uint32_t)(keys_sorted[i - 1] >> 32);
            if (cur != prev) { ranges[2 * prev + 1] = (uint32_t)i; ranges[2 * cur] = (uint32_t)i; }
        }
        if (i == R - 1) ranges[2 * cur + 1] = (uint32_t)R;
    }
}

/* diagnostic counters (pairs evaluated before saturation / pairs passing the alpha test); read via orc_get_stats */
static uint64_t orc_stats[2];
void orc_get_stats(uint64_t* out, int reset) { out[0] = orc_stats[0]; out[1] = orc_stats[1]; if (reset) { orc_stats[0] = 0; orc_stats[1] = 0; } }

static inline float relm(float a, float b) /* relative distance of a from threshold b */
{
    return fabsf(a - b) / fabsf(b);
}

/*
 * Stage: forward blend.  forward.cu:258-467 (renderCUDA), one pixel at a time.
 * `stab` (optional, may be NULL) receives per pixel the smallest relative
 * distance of any discrete decision (alpha>=1/255, T'>=1e-4, T>0.5,
 * rho3d<=rho2d, depth>=near) from its threshold: pixels with a tiny value
 * are knife-edge and may legitimately flip under 1-ulp arithmetic changes.
 * Its second plane, stab[HW + pix], receives the CONDITIONING of the depth re-weighting of use_sa (forward.cu:405-416):
 * exp_std = (D2 - 2 Dp m) / (1 - T) + m^2 is a variance formed by cancellation and conf = exp(-e^2 / (4 exp_std)) divides by
 * it; where the splats in front of a pixel lie at nearly one depth, exp_std is rounding noise of terms of size m^2 and conf
 * -- hence the depth channels -- moves by far more than the arithmetic that produced the noise.  The plane holds
 *   sum_k  w_k |e_k| conf_k (e_k^2 / (4 s_k)) (c_k / s_k),   c_k = m^2 + (|D2| + 2 |Dp m|) / (1 - T),  s_k = exp_std (clamped),
 * i.e. d(depth) / d(relative perturbation of the cancelling terms): multiplied by the relative rounding of float32 sums
 * (1e-6, a dozen ulps) it bounds what two correct float32 evaluations of the reference's formulas may differ by at that pixel.
 * 0 without use_sa.
 */
void orc_blend_fwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                   const float* means2D, const float* features, const float* transMats,
                   const float* normal_opacity, const float* bg, int use_sa,
                   float* out_color, float* out_others, float* final_T, uint32_t* n_contrib,
                   float* median_depth_out, float* depth_std_out, float* stab)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < TILE; ly++)
            for (int lx = 0; lx < TILE; lx++) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                const size_t pix = (size_t)W * py + px;
                const float pxf = (float)px, pyf = (float)py;
                float T = 1.0f, C[3] = {0, 0, 0}, N[3] = {0, 0, 0};
                float Dp = 0, M1 = 0, M2 = 0, D2 = 0, distortion = 0, median_depth = 0;
                float median_contributor = -1;
                uint32_t contributor = 0, last_contributor = 0;
                float margin = 1e30f, sa_amp = 0.f;
                uint64_t n_eval = 0, n_pass = 0;
                for (uint32_t it = r0; it < r1; it++) {
                    contributor++;
                    n_eval++;
                    const uint32_t g = point_list[it];
                    const float* Tm = transMats + 9 * (size_t)g;
                    const float Tu[3] = {Tm[0], Tm[1], Tm[2]}, Tv[3] = {Tm[3], Tm[4], Tm[5]},
                                Tw[3] = {Tm[6], Tm[7], Tm[8]};
                    /* forward.cu:360-371, in the FMA form shared with the HIP kernel (see header) */
                    const float k[3] = {fmaf(pxf, Tw[0], -Tu[0]), fmaf(pxf, Tw[1], -Tu[1]), fmaf(pxf, Tw[2], -Tu[2])};
                    const float l[3] = {fmaf(pyf, Tw[0], -Tv[0]), fmaf(pyf, Tw[1], -Tv[1]), fmaf(pyf, Tw[2], -Tv[2])};
                    const float p0 = fmaf(k[1], l[2], -(k[2] * l[1]));
                    const float p1 = fmaf(k[2], l[0], -(k[0] * l[2]));
                    const float p2 = fmaf(k[0], l[1], -(k[1] * l[0]));
                    if (p2 == 0.0f) continue;
                    const float ip = 1.0f / p2;
                    const float s0 = p0 * ip, s1 = p1 * ip;
                    const float rho3d = fmaf(s0, s0, s1 * s1);
                    const float d0 = means2D[2 * (size_t)g] - pxf, d1 = means2D[2 * (size_t)g + 1] - pyf;
                    const float rho2d = FILTER_INV_SQ * fmaf(d0, d0, d1 * d1);
                    const float rho = fmin_c(rho3d, rho2d);
                    float depth = (rho3d <= rho2d) ? fmaf(s0, Tw[0], fmaf(s1, Tw[1], Tw[2])) : Tw[2];
                    if (depth < NEAR_N) { if (stab) margin = fminf(margin, relm(depth, NEAR_N)); continue; }
                    const float* no = normal_opacity + 4 * (size_t)g;
                    const float power = -0.5f * rho;
                    if (power > 0.0f) continue;
                    const float alpha = fmin_c(0.99f, no[3] * expf(power));
                    if (stab) margin = fminf(margin, relm(alpha, 1.0f / 255.0f));
                    if (alpha < 1.0f / 255.0f) continue;
                    n_pass++;
                    if (stab) {
                        margin = fminf(margin, relm(depth, NEAR_N));
                        const float mx = fmaxf(rho3d, rho2d);
                        if (mx > 0) margin = fminf(margin, fabsf(rho3d - rho2d) / mx);
                    }
                    const float test_T = T * (1 - alpha);
                    if (stab) margin = fminf(margin, relm(test_T, 0.0001f));
                    if (test_T < 0.0001f) break; /* done = true */
                    const float w = alpha * T;
                    if (stab) margin = fminf(margin, relm(T, 0.5f));
                    if (T > 0.5f) { median_depth = depth; median_contributor = (float)contributor; }
                    if (use_sa) { /* forward.cu:405-416 */
                        if (Dp > 0) {
                            const float exp_depth = median_depth;
                            float exp_std = fmaf(fmaf(-2.0f * Dp, exp_depth, D2), 1.0f / (1 - T), exp_depth * exp_depth);
                            exp_std = fmax_c(exp_std, 1e-7f);
                            const float e = exp_depth - depth;
                            const float conf = expf(-(e * e) * (1.0f / (4 * exp_std)));
                            if (stab) {
                                const float cmag = exp_depth * exp_depth + (fabsf(D2) + 2.0f * fabsf(Dp * exp_depth)) / (1 - T);
                                sa_amp += w * fabsf(e) * conf * ((e * e) / (4 * exp_std)) * (cmag / exp_std);
                            }
                            depth = fmaf(conf, depth, (1 - conf) * exp_depth);
                        }
                        Dp = fmaf(depth, w, Dp);
                        D2 = fmaf(depth * depth, w, D2);
                    } else { /* forward.cu:417-423 */
                        const float A = 1 - T;
                        const float m = (FAR_N / (FAR_N - NEAR_N)) * (1 - NEAR_N * (1.0f / depth));
                        distortion = fmaf(fmaf(m * m, A, fmaf(-2.0f * m, M1, M2)), w, distortion);
                        Dp = fmaf(depth, w, Dp);
                        M1 = fmaf(m, w, M1);
                        M2 = fmaf(m * m, w, M2);
                    }
                    for (int ch = 0; ch < 3; ch++) N[ch] = fmaf(no[ch], w, N[ch]);
                    for (int ch = 0; ch < 3; ch++) C[ch] = fmaf(features[3 * (size_t)g + ch], w, C[ch]);
                    T = test_T;
                    last_contributor = contributor;
                }
                /* forward.cu:441-466 */
#pragma omp atomic
                orc_stats[0] += n_eval;
#pragma omp atomic
                orc_stats[1] += n_pass;
                final_T[pix] = T;
                n_contrib[pix] = last_contributor;
                for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix] = fmaf(T, bg[ch], C[ch]);
                n_contrib[pix + HW] = median_contributor < 0 ? 0u : (uint32_t)median_contributor;
                final_T[pix + HW] = M1;
                final_T[pix + 2 * HW] = M2;
                out_others[pix + 0 * HW] = Dp;
                out_others[pix + 1 * HW] = 1 - T;
                for (int ch = 0; ch < 3; ch++) out_others[pix + (2 + ch) * HW] = N[ch];
                out_others[pix + 5 * HW] = median_depth;
                median_depth_out[pix] = median_depth;
                const float dstd = fmaf(median_depth * median_depth, 1 - T, fmaf(-2.0f * median_depth, Dp, D2));
                depth_std_out[pix] = dstd;
                out_others[pix + 6 * HW] = use_sa ? dstd : distortion;
                if (stab) { stab[pix] = margin; stab[HW + pix] = sa_amp; }
            }
    }
}

/*
 * One pixel of the forward blend with selected knife-edge decisions inverted (test infrastructure for the parity
 * tests): every discrete decision of forward.cu:350-436 whose operand lies within `knife` (relative) of its threshold
 * is numbered in the order it is met; bit i of `flipmask` inverts decision i.  flipmask = 0 reproduces orc_blend_fwd
 * for that pixel exactly.  A 1-ulp difference in expf / rcp can legitimately flip such a decision, so a GPU value at a
 * knife-edge pixel has to equal ONE of these variants.  Returns the number of knife-edge decisions met.
 * out[13] = color[3], others[7], last_contributor, median_contributor, final_T.
 */
int orc_blend_fwd_pixel(int W, int H, int px, int py, const uint32_t* ranges, const uint32_t* point_list,
                        const float* means2D, const float* features, const float* transMats,
                        const float* normal_opacity, const float* bg, int use_sa, float knife, uint32_t flipmask,
                        float* out)
{
    const int gx = (W + TILE - 1) / TILE;
    (void)H;
    const int tile = (py / TILE) * gx + (px / TILE);
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    const float pxf = (float)px, pyf = (float)py;
    float T = 1.0f, C[3] = {0, 0, 0}, N[3] = {0, 0, 0};
    float Dp = 0, M1 = 0, M2 = 0, D2 = 0, distortion = 0, median_depth = 0;
    float median_contributor = -1;
    uint32_t contributor = 0, last_contributor = 0;
    int nk = 0;
#define KNIFE_DECIDE(cond, a, b)                                                           \
    ({ int d_ = (cond);                                                                     \
       if (relm((a), (b)) <= knife) { if (nk < 32 && ((flipmask >> nk) & 1u)) d_ = !d_; nk++; } \
       d_; })
    for (uint32_t it = r0; it < r1; it++) {
        contributor++;
        const uint32_t g = point_list[it];
        const float* Tm = transMats + 9 * (size_t)g;
        const float Tu[3] = {Tm[0], Tm[1], Tm[2]}, Tv[3] = {Tm[3], Tm[4], Tm[5]}, Tw[3] = {Tm[6], Tm[7], Tm[8]};
        const float k[3] = {fmaf(pxf, Tw[0], -Tu[0]), fmaf(pxf, Tw[1], -Tu[1]), fmaf(pxf, Tw[2], -Tu[2])};
        const float l[3] = {fmaf(pyf, Tw[0], -Tv[0]), fmaf(pyf, Tw[1], -Tv[1]), fmaf(pyf, Tw[2], -Tv[2])};
        const float p0 = fmaf(k[1], l[2], -(k[2] * l[1]));
        const float p1 = fmaf(k[2], l[0], -(k[0] * l[2]));
        const float p2 = fmaf(k[0], l[1], -(k[1] * l[0]));
        if (p2 == 0.0f) continue;
        const float ip = 1.0f / p2;
        const float s0 = p0 * ip, s1 = p1 * ip;
        const float rho3d = fmaf(s0, s0, s1 * s1);
        const float d0 = means2D[2 * (size_t)g] - pxf, d1 = means2D[2 * (size_t)g + 1] - pyf;
        const float rho2d = FILTER_INV_SQ * fmaf(d0, d0, d1 * d1);
        const float mx = fmaxf(rho3d, rho2d);
        int ray = rho3d <= rho2d;
        if (mx > 0 && fabsf(rho3d - rho2d) / mx <= knife) { if (nk < 32 && ((flipmask >> nk) & 1u)) ray = !ray; nk++; }
        const float rho = ray ? rho3d : rho2d;
        float depth = ray ? fmaf(s0, Tw[0], fmaf(s1, Tw[1], Tw[2])) : Tw[2];
        if (KNIFE_DECIDE(depth < NEAR_N, depth, NEAR_N)) continue;
        const float* no = normal_opacity + 4 * (size_t)g;
        const float power = -0.5f * rho;
        if (power > 0.0f) continue;
        const float alpha = fmin_c(0.99f, no[3] * expf(power));
        if (KNIFE_DECIDE(alpha < 1.0f / 255.0f, alpha, 1.0f / 255.0f)) continue;
        const float test_T = T * (1 - alpha);
        if (KNIFE_DECIDE(test_T < 0.0001f, test_T, 0.0001f)) break;
        const float w = alpha * T;
        if (KNIFE_DECIDE(T > 0.5f, T, 0.5f)) { median_depth = depth; median_contributor = (float)contributor; }
        if (use_sa) {
            if (Dp > 0) {
                const float exp_depth = median_depth;
                float exp_std = fmaf(fmaf(-2.0f * Dp, exp_depth, D2), 1.0f / (1 - T), exp_depth * exp_depth);
                exp_std = fmax_c(exp_std, 1e-7f);
                const float e = exp_depth - depth;
                const float conf = expf(-(e * e) * (1.0f / (4 * exp_std)));
                depth = fmaf(conf, depth, (1 - conf) * exp_depth);
            }
            Dp = fmaf(depth, w, Dp);
            D2 = fmaf(depth * depth, w, D2);
        } else {
            const float A = 1 - T;
            const float m = (FAR_N / (FAR_N - NEAR_N)) * (1 - NEAR_N * (1.0f / depth));
            distortion = fmaf(fmaf(m * m, A, fmaf(-2.0f * m, M1, M2)), w, distortion);
            Dp = fmaf(depth, w, Dp);
            M1 = fmaf(m, w, M1);
            M2 = fmaf(m * m, w, M2);
        }
        for (int ch = 0; ch < 3; ch++) N[ch] = fmaf(no[ch], w, N[ch]);
        for (int ch = 0; ch < 3; ch++) C[ch] = fmaf(features[3 * (size_t)g + ch], w, C[ch]);
        T = test_T;
        last_contributor = contributor;
    }
#undef KNIFE_DECIDE
    for (int ch = 0; ch < 3; ch++) out[ch] = fmaf(T, bg[ch], C[ch]);
    out[3] = Dp;
    out[4] = 1 - T;
    for (int ch = 0; ch < 3; ch++) out[5 + ch] = N[ch];
    out[8] = median_depth;
    out[9] = use_sa ? fmaf(median_depth * median_depth, 1 - T, fmaf(-2.0f * median_depth, Dp, D2)) : distortion;
    out[10] = (float)last_contributor;
    out[11] = median_contributor < 0 ? 0.f : median_contributor;
    out[12] = T;
    return nk;
}

/*
 * Stage: backward blend.  backward.cu:143-463 (renderCUDA), one pixel at a
 * time, splats back to front.  Accumulators are double (see header).
 * acc layout per Gaussian (20 doubles): [0..2] dL_dcolor, [3..5] dL_dnormal,
 * [6..14] dL_dtransMat (Tu,Tv,Tw), [15..16] dL_dmean2D.xy, [17] dL_dopacity.
 */
#define ACC_STRIDE 20
/* orc_set_float_accumulation(1): the per-Gaussian sums of orc_blend_bwd are accumulated in FLOAT32, one add per (pixel,
 * splat) contribution -- what the reference's atomicAdd(float) does (backward.cu:343,396,441-460; its order is whatever the
 * hardware makes it, here it is the thread schedule's).  Default 0: double accumulation (see the header).  Used by the
 * rounding-error tests to tell the error of float32 ACCUMULATION, which the reference has too, from the error of the
 * per-pair arithmetic. */
static int orc_float_acc = 0;
static float* orc_accf = NULL;
void orc_set_float_accumulation(int on) { orc_float_acc = on != 0; }
static void acc_add(double* acc, size_t g, int k, double v)
{
    if (orc_accf) {
        const float vf = (float)v;
#pragma omp atomic
        orc_accf[g * ACC_STRIDE + k] += vf;
        return;
    }
#pragma omp atomic
    acc[g * ACC_STRIDE + k] += v;
}

void orc_blend_bwd(int P, int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                   const float* bg, const float* means2D, const float* normal_opacity,
                   const float* transMats, const float* colors, const float* final_Ts,
                   const uint32_t* n_contrib, const float* dL_dpixels, const float* dL_depths,
                   const float* median_depth, const float* depth_std, int use_sa,
                   float* dL_dtransMat, float* dL_dmean2D /* [P,3] */, float* dL_dnormal3D,
                   float* dL_dopacity, float* dL_dcolors, const double* extra_acc /* NULL or [P][20] (orc_blend_bwd_pixel) */)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const size_t HW = (size_t)H * W;
    double* acc = (double*)calloc((size_t)P * ACC_STRIDE + 1, sizeof(double));
    if (extra_acc) memcpy(acc, extra_acc, (size_t)P * ACC_STRIDE * sizeof(double));
    orc_accf = orc_float_acc ? (float*)calloc((size_t)P * ACC_STRIDE + 1, sizeof(float)) : NULL;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < TILE; ly++)
            for (int lx = 0; lx < TILE; lx++) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                const size_t pix = (size_t)W * py + px;
                const float pxf = (float)px, pyf = (float)py;
                const float T_final = final_Ts[pix];
                float T = T_final;
                uint32_t contributor = r1 - r0;
                const uint32_t last_contributor = n_contrib[pix];
                const uint32_t median_contributor = n_contrib[pix + HW];
                float accum_rec[3] = {0, 0, 0}, dL_dpixel[3];
                const float dL_ddepth = dL_depths[0 * HW + pix];
                const float dL_daccum = dL_depths[1 * HW + pix];
                const float dL_dreg = dL_depths[6 * HW + pix];
                const float dL_dnormal2D[3] = {dL_depths[2 * HW + pix], dL_depths[3 * HW + pix],
                                               dL_depths[4 * HW + pix]};
                const float dL_dmedian_depth = dL_depths[5 * HW + pix];
                const float mm = median_depth[pix], mstd = depth_std[pix];
                float last_depth = 0, last_normal[3] = {0, 0, 0}, accum_depth_rec = 0, accum_alpha_rec = 0;
                float accum_normal_rec[3] = {0, 0, 0};
                const float final_D = final_Ts[pix + HW], final_D2 = final_Ts[pix + 2 * HW];
                const float final_A = 1 - T_final;
                float last_dL_dT = 0, last_alpha = 0, last_color[3] = {0, 0, 0};
                for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[i * HW + pix];
                const float bg_dot = fmaf(bg[2], dL_dpixel[2], fmaf(bg[1], dL_dpixel[1], bg[0] * dL_dpixel[0]));
                /* backward.cu:349: 1 / (4 max(mstd/(1-T_final), 1e-7)) is a per-pixel constant */
                const float sa_k = 1.0f / (4 * fmax_c(mstd * (1.0f / (1 - T_final)), 1e-7f));
                const float c1 = FAR_N / (FAR_N - NEAR_N);
                for (uint32_t it = r1; it-- > r0;) {
                    contributor--;
                    if (contributor >= last_contributor) continue;
                    const uint32_t g = point_list[it];
                    const float* Tm = transMats + 9 * (size_t)g;
                    const float Tu[3] = {Tm[0], Tm[1], Tm[2]}, Tv[3] = {Tm[3], Tm[4], Tm[5]},
                                Tw[3] = {Tm[6], Tm[7], Tm[8]};
                    /* same FMA form as the forward (and as the HIP kernel) */
                    const float k[3] = {fmaf(pxf, Tw[0], -Tu[0]), fmaf(pxf, Tw[1], -Tu[1]), fmaf(pxf, Tw[2], -Tu[2])};
                    const float l[3] = {fmaf(pyf, Tw[0], -Tv[0]), fmaf(pyf, Tw[1], -Tv[1]), fmaf(pyf, Tw[2], -Tv[2])};
                    const float p0 = fmaf(k[1], l[2], -(k[2] * l[1]));
                    const float p1 = fmaf(k[2], l[0], -(k[0] * l[2]));
                    const float p2 = fmaf(k[0], l[1], -(k[1] * l[0]));
                    if (p2 == 0.0f) continue;
                    const float ip = 1.0f / p2;
                    const float s0 = p0 * ip, s1 = p1 * ip;
                    const float rho3d = fmaf(s0, s0, s1 * s1);
                    const float d0 = means2D[2 * (size_t)g] - pxf, d1 = means2D[2 * (size_t)g + 1] - pyf;
                    const float rho2d = FILTER_INV_SQ * fmaf(d0, d0, d1 * d1);
                    const float rho = fmin_c(rho3d, rho2d);
                    float c_d = (rho3d <= rho2d) ? fmaf(s0, Tw[0], fmaf(s1, Tw[1], Tw[2])) : Tw[2];
                    if (c_d < NEAR_N) continue;
                    const float* no = normal_opacity + 4 * (size_t)g;
                    const float power = -0.5f * rho;
                    if (power > 0.0f) continue;
                    const float G = expf(power);
                    const float alpha = fmin_c(0.99f, no[3] * G);
                    if (alpha < 1.0f / 255.0f) continue;
                    const float ioma = 1.0f / (1.f - alpha);
                    T = T * ioma;
                    const float w = alpha * T;
                    float dL_dalpha = 0.0f;
                    for (int ch = 0; ch < 3; ch++) { /* backward.cu:331-344 */
                        const float c = colors[3 * (size_t)g + ch];
                        accum_rec[ch] = fmaf(last_alpha, last_color[ch], (1.f - last_alpha) * accum_rec[ch]);
                        last_color[ch] = c;
                        dL_dalpha = fmaf(c - accum_rec[ch], dL_dpixel[ch], dL_dalpha);
                        acc_add(acc, g, ch, (double)(w * dL_dpixel[ch]));
                    }
                    float conf = 1;
                    if (use_sa) { /* backward.cu:347-351 (float exp here; the reference promotes this one to double) */
                        if (T < 0.5f) {
                            const float dm = c_d - mm;
                            conf = expf(-(dm * dm) * sa_k);
                        }
                        c_d = fmaf(c_d, conf, mm * (1 - conf));
                    }
                    float dL_dz = 0.0f, dL_dweight;
                    if (contributor == median_contributor - 1u) dL_dz = dL_dmedian_depth;
                    if (use_sa) {
                        const float dm = c_d - mm;
                        dL_dweight = (dm * dm) * dL_dreg;
                        dL_dalpha += dL_dweight - last_dL_dT;
                        last_dL_dT = fmaf(dL_dweight, alpha, (1 - alpha) * last_dL_dT);
                        dL_dz = fmaf(conf * 2.0f * w * dm, dL_dreg, dL_dz);
                    } else {
                        const float icd = 1.0f / c_d;
                        const float m_d = c1 * (1 - NEAR_N * icd);
                        const float dmd_dd = (c1 * NEAR_N) * (icd * icd);
                        dL_dweight = fmaf(m_d * m_d, final_A, fmaf(-2.0f * m_d, final_D, final_D2)) * dL_dreg;
                        dL_dalpha += dL_dweight - last_dL_dT;
                        last_dL_dT = fmaf(dL_dweight, alpha, (1 - alpha) * last_dL_dT);
                        const float dL_dmd = 2.0f * w * fmaf(m_d, final_A, -final_D) * dL_dreg;
                        dL_dz = fmaf(dL_dmd, dmd_dd, dL_dz);
                    }
                    accum_depth_rec = fmaf(last_alpha, last_depth, (1.f - last_alpha) * accum_depth_rec);
                    last_depth = c_d;
                    dL_dalpha = fmaf(c_d - accum_depth_rec, dL_ddepth, dL_dalpha);
                    accum_alpha_rec = fmaf(1.f - last_alpha, accum_alpha_rec, last_alpha);
                    dL_dalpha = fmaf(1 - accum_alpha_rec, dL_daccum, dL_dalpha);
                    for (int ch = 0; ch < 3; ch++) { /* backward.cu:392-397 */
                        accum_normal_rec[ch] = fmaf(last_alpha, last_normal[ch], (1.f - last_alpha) * accum_normal_rec[ch]);
                        last_normal[ch] = no[ch];
                        dL_dalpha = fmaf(no[ch] - accum_normal_rec[ch], dL_dnormal2D[ch], dL_dalpha);
                        acc_add(acc, g, 3 + ch, (double)(w * dL_dnormal2D[ch]));
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha = fmaf(-T_final * ioma, bg_dot, dL_dalpha);
                    const float dL_dG = no[3] * dL_dalpha;
                    dL_dz = fmaf(conf * w, dL_ddepth, dL_dz);
                    if (rho3d <= rho2d) { /* backward.cu:419-449 */
                        const float gG = dL_dG * -G;
                        const float dL_ds0 = fmaf(gG, s0, dL_dz * Tw[0]);
                        const float dL_ds1 = fmaf(gG, s1, dL_dz * Tw[1]);
                        const float dsx = dL_ds0 * ip, dsy = dL_ds1 * ip;
                        const float dp2 = -fmaf(dsx, s0, dsy * s1);
                        const float dk[3] = {fmaf(l[1], dp2, -(l[2] * dsy)), fmaf(l[2], dsx, -(l[0] * dp2)),
                                             fmaf(l[0], dsy, -(l[1] * dsx))};
                        const float dl[3] = {fmaf(dsy, k[2], -(dp2 * k[1])), fmaf(dp2, k[0], -(dsx * k[2])),
                                             fmaf(dsx, k[1], -(dsy * k[0]))};
                        const float dz_dTw[3] = {dL_dz * s0, dL_dz * s1, dL_dz};
                        for (int i = 0; i < 3; i++) {
                            acc_add(acc, g, 6 + i, (double)(-dk[i]));
                            acc_add(acc, g, 9 + i, (double)(-dl[i]));
                            acc_add(acc, g, 12 + i, (double)fmaf(pxf, dk[i], fmaf(pyf, dl[i], dz_dTw[i])));
                        }
                    } else { /* backward.cu:450-457 */
                        const float t = dL_dG * (-G * FILTER_INV_SQ);
                        acc_add(acc, g, 15, (double)(t * d0));
                        acc_add(acc, g, 16, (double)(t * d1));
                        acc_add(acc, g, 14, (double)dL_dz);
                    }
                    acc_add(acc, g, 17, (double)(G * dL_dalpha));
                }
            }
    }
    if (orc_accf) {
        for (size_t i = 0; i < (size_t)P * ACC_STRIDE; i++) acc[i] += (double)orc_accf[i];
        free(orc_accf);
        orc_accf = NULL;
    }
    for (size_t g = 0; g < (size_t)P; g++) {
        const double* a = acc + g * ACC_STRIDE;
        for (int i = 0; i < 3; i++) dL_dcolors[3 * g + i] = (float)a[i];
        for (int i = 0; i < 3; i++) dL_dnormal3D[3 * g + i] = (float)a[3 + i];
        for (int i = 0; i < 9; i++) dL_dtransMat[9 * g + i] = (float)a[6 + i];
        dL_dmean2D[3 * g] = (float)a[15]; dL_dmean2D[3 * g + 1] = (float)a[16]; dL_dmean2D[3 * g + 2] = 0.f;
        dL_dopacity[g] = (float)a[17];
    }
    free(acc);
}

/*
 * Backward blend of ONE pixel under one outcome of its near-threshold forward decisions (the `flipmask` numbering of
 * orc_blend_fwd_pixel): the forward walk is repeated with those decisions flipped, every splat's outcome is recorded
 * (skipped / contributes, ray-splat or low-pass branch, T > 0.5), and the backward recurrence of orc_blend_bwd then runs on
 * the RECORDED outcomes instead of re-deriving them -- i.e. the gradient the reference's backward would produce had its
 * arithmetic fallen on that side of the thresholds.  Lets the parity tests give knife-edge pixels an upstream gradient
 * (tests/test_gpu_round3.py) instead of masking them out.  Contributions are ADDED to acc (20 doubles per Gaussian, the
 * layout of orc_blend_bwd).  dL_dpix: 3 floats, dL_doth: 7 floats (this pixel's upstream gradients).  Returns the number
 * of near-threshold decisions met (as orc_blend_fwd_pixel does).
 */
int orc_blend_bwd_pixel(int W, int H, int px, int py, const uint32_t* ranges, const uint32_t* point_list,
                        const float* means2D, const float* features, const float* transMats,
                        const float* normal_opacity, const float* bg, int use_sa, float knife, uint32_t flipmask,
                        const float* dL_dpix, const float* dL_doth, double* acc)
{
    const int gx = (W + TILE - 1) / TILE;
    (void)H;
    const int tile = (py / TILE) * gx + (px / TILE);
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    const uint32_t n = r1 - r0;
    const float pxf = (float)px, pyf = (float)py;
    /* per list position: 0 = skipped, 1 = contributes on the ray-splat branch, 2 = contributes on the low-pass branch */
    uint8_t* outcome = (uint8_t*)calloc(n + 1, 1);
    float T = 1.0f, Dp = 0, M1 = 0, M2 = 0, D2 = 0, median_depth = 0;
    uint32_t contributor = 0, last_contributor = 0, median_contributor = 0;
    int nk = 0;
#define KNIFE_DECIDE(cond, a, b)                                                           \
    ({ int d_ = (cond);                                                                     \
       if (relm((a), (b)) <= knife) { if (nk < 32 && ((flipmask >> nk) & 1u)) d_ = !d_; nk++; } \
       d_; })
    for (uint32_t it = r0; it < r1; it++) { /* the forward of orc_blend_fwd_pixel, recording outcomes */
        contributor++;
        const uint32_t g = point_list[it];
        const float* Tm = transMats + 9 * (size_t)g;
        const float Tu[3] = {Tm[0], Tm[1], Tm[2]}, Tv[3] = {Tm[3], Tm[4], Tm[5]}, Tw[3] = {Tm[6], Tm[7], Tm[8]};
        const float k[3] = {fmaf(pxf, Tw[0], -Tu[0]), fmaf(pxf, Tw[1], -Tu[1]), fmaf(pxf, Tw[2], -Tu[2])};
        const float l[3] = {fmaf(pyf, Tw[0], -Tv[0]), fmaf(pyf, Tw[1], -Tv[1]), fmaf(pyf, Tw[2], -Tv[2])};
        const float p0 = fmaf(k[1], l[2], -(k[2] * l[1]));
        const float p1 = fmaf(k[2], l[0], -(k[0] * l[2]));
        const float p2 = fmaf(k[0], l[1], -(k[1] * l[0]));
        if (p2 == 0.0f) continue;
        const float ip = 1.0f / p2;
        const float s0 = p0 * ip, s1 = p1 * ip;
        const float rho3d = fmaf(s0, s0, s1 * s1);
        const float d0 = means2D[2 * (size_t)g] - pxf, d1 = means2D[2 * (size_t)g + 1] - pyf;
        const float rho2d = FILTER_INV_SQ * fmaf(d0, d0, d1 * d1);
        const float mx = fmaxf(rho3d, rho2d);
        int ray = rho3d <= rho2d;
        if (mx > 0 && fabsf(rho3d - rho2d) / mx <= knife) { if (nk < 32 && ((flipmask >> nk) & 1u)) ray = !ray; nk++; }
        const float rho = ray ? rho3d : rho2d;
        float depth = ray ? fmaf(s0, Tw[0], fmaf(s1, Tw[1], Tw[2])) : Tw[2];
        if (KNIFE_DECIDE(depth < NEAR_N, depth, NEAR_N)) continue;
        const float* no = normal_opacity + 4 * (size_t)g;
        const float power = -0.5f * rho;
        if (power > 0.0f) continue;
        const float alpha = fmin_c(0.99f, no[3] * expf(power));
        if (KNIFE_DECIDE(alpha < 1.0f / 255.0f, alpha, 1.0f / 255.0f)) continue;
        const float test_T = T * (1 - alpha);
        if (KNIFE_DECIDE(test_T < 0.0001f, test_T, 0.0001f)) break;
        const float w = alpha * T;
        if (KNIFE_DECIDE(T > 0.5f, T, 0.5f)) { median_depth = depth; median_contributor = contributor; }
        if (use_sa) {
            if (Dp > 0) {
                const float exp_depth = median_depth;
                float exp_std = fmaf(fmaf(-2.0f * Dp, exp_depth, D2), 1.0f / (1 - T), exp_depth * exp_depth);
                exp_std = fmax_c(exp_std, 1e-7f);
                const float e = exp_depth - depth;
                const float conf = expf(-(e * e) * (1.0f / (4 * exp_std)));
                depth = fmaf(conf, depth, (1 - conf) * exp_depth);
            }
            Dp = fmaf(depth, w, Dp);
            D2 = fmaf(depth * depth, w, D2);
        } else {
            const float m = (FAR_N / (FAR_N - NEAR_N)) * (1 - NEAR_N * (1.0f / depth));
            Dp = fmaf(depth, w, Dp);
            M1 = fmaf(m, w, M1);
            M2 = fmaf(m * m, w, M2);
        }
        T = test_T;
        last_contributor = contributor;
        outcome[it - r0] = ray ? 1 : 2;
    }
#undef KNIFE_DECIDE
    /* the per-pixel state the forward hands to the backward (forward.cu:441-466) */
    const float T_final = T;
    const float mm = median_depth;
    const float mstd = fmaf(median_depth * median_depth, 1 - T, fmaf(-2.0f * median_depth, Dp, D2));
    const float final_D = M1, final_D2 = M2, final_A = 1 - T_final;
    const float dL_ddepth = dL_doth[0], dL_daccum = dL_doth[1], dL_dreg = dL_doth[6], dL_dmedian_depth = dL_doth[5];
    const float dL_dnormal2D[3] = {dL_doth[2], dL_doth[3], dL_doth[4]};
    float accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_normal[3] = {0, 0, 0}, accum_normal_rec[3] = {0, 0, 0};
    float last_depth = 0, accum_depth_rec = 0, accum_alpha_rec = 0, last_dL_dT = 0, last_alpha = 0;
    const float bg_dot = fmaf(bg[2], dL_dpix[2], fmaf(bg[1], dL_dpix[1], bg[0] * dL_dpix[0]));
    const float sa_k = 1.0f / (4 * fmax_c(mstd * (1.0f / (1 - T_final)), 1e-7f));
    const float c1 = FAR_N / (FAR_N - NEAR_N);
    contributor = n;
    for (uint32_t it = r1; it-- > r0;) { /* the recurrence of orc_blend_bwd on the recorded outcomes */
        contributor--;
        if (contributor >= last_contributor) continue;
        const int oc = outcome[it - r0];
        if (oc == 0) continue;
        const int ray = oc == 1;
        const uint32_t g = point_list[it];
        const float* Tm = transMats + 9 * (size_t)g;
        const float Tu[3] = {Tm[0], Tm[1], Tm[2]}, Tv[3] = {Tm[3], Tm[4], Tm[5]}, Tw[3] = {Tm[6], Tm[7], Tm[8]};
        const float k[3] = {fmaf(pxf, Tw[0], -Tu[0]), fmaf(pxf, Tw[1], -Tu[1]), fmaf(pxf, Tw[2], -Tu[2])};
        const float l[3] = {fmaf(pyf, Tw[0], -Tv[0]), fmaf(pyf, Tw[1], -Tv[1]), fmaf(pyf, Tw[2], -Tv[2])};
        const float p0 = fmaf(k[1], l[2], -(k[2] * l[1]));
        const float p1 = fmaf(k[2], l[0], -(k[0] * l[2]));
        const float p2 = fmaf(k[0], l[1], -(k[1] * l[0]));
        const float ip = 1.0f / p2;
        const float s0 = p0 * ip, s1 = p1 * ip;
        const float rho3d = fmaf(s0, s0, s1 * s1);
        const float d0 = means2D[2 * (size_t)g] - pxf, d1 = means2D[2 * (size_t)g + 1] - pyf;
        const float rho2d = FILTER_INV_SQ * fmaf(d0, d0, d1 * d1);
        const float rho = ray ? rho3d : rho2d;
        float c_d = ray ? fmaf(s0, Tw[0], fmaf(s1, Tw[1], Tw[2])) : Tw[2];
        const float* no = normal_opacity + 4 * (size_t)g;
        const float G = expf(-0.5f * rho);
        const float alpha = fmin_c(0.99f, no[3] * G);
        const float ioma = 1.0f / (1.f - alpha);
        T = T * ioma;
        const float w = alpha * T;
        float dL_dalpha = 0.0f;
        for (int ch = 0; ch < 3; ch++) {
            const float c = features[3 * (size_t)g + ch];
            accum_rec[ch] = fmaf(last_alpha, last_color[ch], (1.f - last_alpha) * accum_rec[ch]);
            last_color[ch] = c;
            dL_dalpha = fmaf(c - accum_rec[ch], dL_dpix[ch], dL_dalpha);
            acc[(size_t)g * ACC_STRIDE + ch] += (double)(w * dL_dpix[ch]);
        }
        float conf = 1;
        if (use_sa) {
            if (T < 0.5f) {
                const float dm = c_d - mm;
                conf = expf(-(dm * dm) * sa_k);
            }
            c_d = fmaf(c_d, conf, mm * (1 - conf));
        }
        float dL_dz = 0.0f, dL_dweight;
        if (contributor == median_contributor - 1u) dL_dz = dL_dmedian_depth;
        if (use_sa) {
            const float dm = c_d - mm;
            dL_dweight = (dm * dm) * dL_dreg;
            dL_dalpha += dL_dweight - last_dL_dT;
            last_dL_dT = fmaf(dL_dweight, alpha, (1 - alpha) * last_dL_dT);
            dL_dz = fmaf(conf * 2.0f * w * dm, dL_dreg, dL_dz);
        } else {
            const float icd = 1.0f / c_d;
            const float m_d = c1 * (1 - NEAR_N * icd);
            const float dmd_dd = (c1 * NEAR_N) * (icd * icd);
            dL_dweight = fmaf(m_d * m_d, final_A, fmaf(-2.0f * m_d, final_D, final_D2)) * dL_dreg;
            dL_dalpha += dL_dweight - last_dL_dT;
            last_dL_dT = fmaf(dL_dweight, alpha, (1 - alpha) * last_dL_dT);
            const float dL_dmd = 2.0f * w * fmaf(m_d, final_A, -final_D) * dL_dreg;
            dL_dz = fmaf(dL_dmd, dmd_dd, dL_dz);
        }
        accum_depth_rec = fmaf(last_alpha, last_depth, (1.f - last_alpha) * accum_depth_rec);
        last_depth = c_d;
        dL_dalpha = fmaf(c_d - accum_depth_rec, dL_ddepth, dL_dalpha);
        accum_alpha_rec = fmaf(1.f - last_alpha, accum_alpha_rec, last_alpha);
        dL_dalpha = fmaf(1 - accum_alpha_rec, dL_daccum, dL_dalpha);
        for (int ch = 0; ch < 3; ch++) {
            accum_normal_rec[ch] = fmaf(last_alpha, last_normal[ch], (1.f - last_alpha) * accum_normal_rec[ch]);
            last_normal[ch] = no[ch];
            dL_dalpha = fmaf(no[ch] - accum_normal_rec[ch], dL_dnormal2D[ch], dL_dalpha);
            acc[(size_t)g * ACC_STRIDE + 3 + ch] += (double)(w * dL_dnormal2D[ch]);
        }
        dL_dalpha *= T;
        last_alpha = alpha;
        dL_dalpha = fmaf(-T_final * ioma, bg_dot, dL_dalpha);
        const float dL_dG = no[3] * dL_dalpha;
        dL_dz = fmaf(conf * w, dL_ddepth, dL_dz);
        if (ray) {
            const float gG = dL_dG * -G;
            const float dL_ds0 = fmaf(gG, s0, dL_dz * Tw[0]);
            const float dL_ds1 = fmaf(gG, s1, dL_dz * Tw[1]);
            const float dsx = dL_ds0 * ip, dsy = dL_ds1 * ip;
            const float dp2 = -fmaf(dsx, s0, dsy * s1);
            const float dk[3] = {fmaf(l[1], dp2, -(l[2] * dsy)), fmaf(l[2], dsx, -(l[0] * dp2)), fmaf(l[0], dsy, -(l[1] * dsx))};
            const float dl[3] = {fmaf(dsy, k[2], -(dp2 * k[1])), fmaf(dp2, k[0], -(dsx * k[2])), fmaf(dsx, k[1], -(dsy * k[0]))};
            const float dz_dTw[3] = {dL_dz * s0, dL_dz * s1, dL_dz};
            for (int i = 0; i < 3; i++) {
                acc[(size_t)g * ACC_STRIDE + 6 + i] += (double)(-dk[i]);
                acc[(size_t)g * ACC_STRIDE + 9 + i] += (double)(-dl[i]);
                acc[(size_t)g * ACC_STRIDE + 12 + i] += (double)fmaf(pxf, dk[i], fmaf(pyf, dl[i], dz_dTw[i]));
            }
        } else {
            const float t = dL_dG * (-G * FILTER_INV_SQ);
            acc[(size_t)g * ACC_STRIDE + 15] += (double)(t * d0);
            acc[(size_t)g * ACC_STRIDE + 16] += (double)(t * d1);
            acc[(size_t)g * ACC_STRIDE + 14] += (double)dL_dz;
        }
        acc[(size_t)g * ACC_STRIDE + 17] += (double)(G * dL_dalpha);
    }
    free(outcome);
    return nk;
}

/* backward.cu:20-139 (SH backward); adds the view-direction term into dL_dmeans. */
static void sh_backward(int idx, int deg, int M, const float* means, const float* campos,
                        const float* shs, const uint8_t* clamped, const float* dL_dcolor,
                        float* dL_dmeans, float* dL_dshs)
{
    const float ox = means[3 * idx] - campos[0], oy = means[3 * idx + 1] - campos[1], oz = means[3 * idx + 2] - campos[2];
    const float len = sqrtf((ox * ox + oy * oy) + oz * oz);
    const float x = ox / len, y = oy / len, z = oz / len;
    const float* sh = shs + (size_t)idx * M * 3;
    float* dsh = dL_dshs + (size_t)idx * M * 3;
    float dRGB[3];
    for (int c = 0; c < 3; c++) dRGB[c] = dL_dcolor[3 * idx + c] * (clamped[3 * idx + c] ? 0.f : 1.f);
    float ddir[3] = {0, 0, 0};
    for (int c = 0; c < 3; c++) {
#define SHC(k) sh[(k) * 3 + c]
#define DSH(k) dsh[(k) * 3 + c]
        float dx = 0, dy = 0, dz = 0;
        DSH(0) = SH_C0 * dRGB[c];
        if (deg > 0) {
            DSH(1) = (-SH_C1 * y) * dRGB[c];
            DSH(2) = (SH_C1 * z) * dRGB[c];
            DSH(3) = (-SH_C1 * x) * dRGB[c];
            dx = -SH_C1 * SHC(3); dy = -SH_C1 * SHC(1); dz = SH_C1 * SHC(2);
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                DSH(4) = (SH_C2[0] * xy) * dRGB[c];
                DSH(5) = (SH_C2[1] * yz) * dRGB[c];
                DSH(6) = (SH_C2[2] * (2.f * zz - xx - yy)) * dRGB[c];
                DSH(7) = (SH_C2[3] * xz) * dRGB[c];
                DSH(8) = (SH_C2[4] * (xx - yy)) * dRGB[c];
                dx += SH_C2[0] * y * SHC(4) + SH_C2[2] * 2.f * -x * SHC(6) + SH_C2[3] * z * SHC(7) + SH_C2[4] * 2.f * x * SHC(8);
                dy += SH_C2[0] * x * SHC(4) + SH_C2[1] * z * SHC(5) + SH_C2[2] * 2.f * -y * SHC(6) + SH_C2[4] * 2.f * -y * SHC(8);
                dz += SH_C2[1] * y * SHC(5) + SH_C2[2] * 2.f * 2.f * z * SHC(6) + SH_C2[3] * x * SHC(7);
                if (deg > 2) {
                    DSH(9) = (SH_C3[0] * y * (3.f * xx - yy)) * dRGB[c];
                    DSH(10) = (SH_C3[1] * xy * z) * dRGB[c];
                    DSH(11) = (SH_C3[2] * y * (4.f * zz - xx - yy)) * dRGB[c];
                    DSH(12) = (SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * dRGB[c];
                    DSH(13) = (SH_C3[4] * x * (4.f * zz - xx - yy)) * dRGB[c];
                    DSH(14) = (SH_C3[5] * z * (xx - yy)) * dRGB[c];
                    DSH(15) = (SH_C3[6] * x * (xx - 3.f * yy)) * dRGB[c];
                    dx += (SH_C3[0] * SHC(9) * 3.f * 2.f * xy + SH_C3[1] * SHC(10) * yz + SH_C3[2] * SHC(11) * -2.f * xy +
                           SH_C3[3] * SHC(12) * -3.f * 2.f * xz + SH_C3[4] * SHC(13) * (-3.f * xx + 4.f * zz - yy) +
                           SH_C3[5] * SHC(14) * 2.f * xz + SH_C3[6] * SHC(15) * 3.f * (xx - yy));
                    dy += (SH_C3[0] * SHC(9) * 3.f * (xx - yy) + SH_C3[1] * SHC(10) * xz +
                           SH_C3[2] * SHC(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * SHC(12) * -3.f * 2.f * yz +
                           SH_C3[4] * SHC(13) * -2.f * xy + SH_C3[5] * SHC(14) * -2.f * yz +
                           SH_C3[6] * SHC(15) * -3.f * 2.f * xy);
                    dz += (SH_C3[1] * SHC(10) * xy + SH_C3[2] * SHC(11) * 4.f * 2.f * yz +
                           SH_C3[3] * SHC(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * SHC(13) * 4.f * 2.f * xz +
                           SH_C3[5] * SHC(14) * (xx - yy));
                }
            }
        }
#undef SHC
#undef DSH
        ddir[0] += dx * dRGB[c]; ddir[1] += dy * dRGB[c]; ddir[2] += dz * dRGB[c];
    }
    /* dnormvdv, auxiliary.h:127-137 */
    const float sum2 = ox * ox + oy * oy + oz * oz;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    dL_dmeans[3 * idx + 0] += ((+sum2 - ox * ox) * ddir[0] - oy * ox * ddir[1] - oz * ox * ddir[2]) * invsum32;
    dL_dmeans[3 * idx + 1] += (-ox * oy * ddir[0] + (sum2 - oy * oy) * ddir[1] - oz * oy * ddir[2]) * invsum32;
    dL_dmeans[3 * idx + 2] += (-ox * oz * ddir[0] - oy * oz * ddir[1] + (sum2 - oz * oz) * ddir[2]) * invsum32;
}

/*
 * Stage: backward preprocess.  backward.cu:466-664 (compute_transmat_aabb +
 * preprocessCUDA).  Outputs must be zero-initialised by the caller, as
 * rasterize_points.cu:192-200 does.  dL_dtransMat / dL_dmean2D are in-out.
 */
void orc_preprocess_bwd(int P, int D, int M, const float* means3D, const float* transMats,
                        const int32_t* radii, const float* shs, const uint8_t* clamped,
                        const float* scales, const float* rotations, float scale_modifier,
                        const float* viewmatrix, const float* projmatrix,
                        int width, int height, float tan_fovx, float tan_fovy, const float* campos,
                        float* dL_dtransMats, const float* dL_dnormal3Ds, float* dL_dcolors,
                        float* dL_dshs, float* dL_dmean2Ds, float* dL_dmean3Ds, float* dL_dscales,
                        float* dL_drots)
{
    (void)scale_modifier; /* backward.cu:504 uses scale_to_mat(scale, 1.0f) */
    /* rasterizer_impl.cu:396-397 + backward.cu:641-642: W,H rebuilt in float */
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);
    const int W = f2i_sat(focal_x * tan_fovx * 2);
    const int H = f2i_sat(focal_y * tan_fovy * 2);
    const float* pm = projmatrix;
    const float* vm = viewmatrix;
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const int precomp = (scales == NULL);
        float T[9], normal[3] = {0, 0, 0}, R[3][3];
        float Pm[4][3]; /* P(a,j) = sum_b Proj(b,a) N(b,j) */
        float sx = 0, sy = 0;
        const float* p = means3D + 3 * idx;
        if (precomp) memcpy(T, transMats + 9 * idx, sizeof(T));
        else {
            sx = scales[2 * idx]; sy = scales[2 * idx + 1];
            compute_transmat(p, sx, sy, rotations + 4 * idx, pm, vm, W, H, T, normal, R);
            const float halfW = (float)W * 0.5f, halfWm = (float)(W - 1) * 0.5f;
            const float halfH = (float)H * 0.5f, halfHm = (float)(H - 1) * 0.5f;
            for (int a = 0; a < 4; a++) {
                Pm[a][0] = pm[4 * a] * halfW + pm[4 * a + 3] * halfWm;
                Pm[a][1] = pm[4 * a + 1] * halfH + pm[4 * a + 3] * halfHm;
                Pm[a][2] = pm[4 * a + 3];
            }
        }
        float dT[9];
        memcpy(dT, dL_dtransMats + 9 * idx, sizeof(dT));
        const float dmx = dL_dmean2Ds[3 * idx], dmy = dL_dmean2Ds[3 * idx + 1];
        int early = 0;
        if (dmx != 0 || dmy != 0) { /* backward.cu:538-577; note: no cutoff^2 here */
            const float distance = T[6] * T[6] + T[7] * T[7] - T[8] * T[8];
            const float f = 1 / distance;
            dT[0] += dmx * (f * T[6]);
            dT[1] += dmx * (f * T[7]);
            dT[2] += dmx * (-f * T[8]);
            dT[3] += dmy * (f * T[6]);
            dT[4] += dmy * (f * T[7]);
            dT[5] += dmy * (-f * T[8]);
            dT[6] += dmx * (T[0] * (f - 2 * f * f * T[6] * T[6])) + dmy * (T[3] * (f - 2 * f * f * T[6] * T[6]));
            dT[7] += dmx * (T[1] * (f - 2 * f * f * T[7] * T[7])) + dmy * (T[4] * (f - 2 * f * f * T[7] * T[7]));
            dT[8] += dmx * (-T[2] * (f + 2 * f * f * T[8] * T[8])) + dmy * (-T[5] * (f + 2 * f * f * T[8] * T[8]));
            if (precomp) { memcpy(dL_dtransMats + 9 * idx, dT, sizeof(dT)); early = 1; }
        }
        if (!precomp && !early) {
            /* dL_dh_i[a] = P(a,0) dTu[i] + P(a,1) dTv[i] + P(a,2) dTw[i]  (backward.cu:582) */
            float dh[3][3];
            for (int i = 0; i < 3; i++)
                for (int a = 0; a < 3; a++)
                    dh[i][a] = (Pm[a][0] * dT[i] + Pm[a][1] * dT[3 + i]) + Pm[a][2] * dT[6 + i];
            const float* dn = dL_dnormal3Ds + 3 * idx;
            float dtn[3] = {(vm[0] * dn[0] + vm[1] * dn[1]) + vm[2] * dn[2],
                            (vm[4] * dn[0] + vm[5] * dn[1]) + vm[6] * dn[2],
                            (vm[8] * dn[0] + vm[9] * dn[1]) + vm[10] * dn[2]};
            const float pvx = ((vm[0] * p[0] + vm[4] * p[1]) + vm[8] * p[2]) + vm[12];
            const float pvy = ((vm[1] * p[0] + vm[5] * p[1]) + vm[9] * p[2]) + vm[13];
            const float pvz = ((vm[2] * p[0] + vm[6] * p[1]) + vm[10] * p[2]) + vm[14];
            const float cosv = -((pvx * normal[0] + pvy * normal[1]) + pvz * normal[2]);
            const float mult = cosv > 0 ? 1.f : -1.f;
            for (int a = 0; a < 3; a++) dtn[a] = mult * dtn[a];
            /* v(r,c): gradient w.r.t. R(r,c); columns dRS0*sx, dRS1*sy, dtn  (backward.cu:596-599) */
            float v[3][3];
            for (int r = 0; r < 3; r++) { v[r][0] = dh[0][r] * sx; v[r][1] = dh[1][r] * sy; v[r][2] = dtn[r]; }
            /* quat_to_rotmat_vjp, auxiliary.h:237-281 */
            const float* q = rotations + 4 * idx;
            const float s = 1.0f / sqrtf(q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
            const float w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
            dL_drots[4 * idx + 0] = 2.f * (x * (v[2][1] - v[1][2]) + y * (v[0][2] - v[2][0]) + z * (v[1][0] - v[0][1]));
            dL_drots[4 * idx + 1] = 2.f * (-2.f * x * (v[1][1] + v[2][2]) + y * (v[1][0] + v[0][1]) +
                                           z * (v[2][0] + v[0][2]) + w * (v[2][1] - v[1][2]));
            dL_drots[4 * idx + 2] = 2.f * (x * (v[1][0] + v[0][1]) - 2.f * y * (v[0][0] + v[2][2]) +
                                           z * (v[2][1] + v[1][2]) + w * (v[0][2] - v[2][0]));
            dL_drots[4 * idx + 3] = 2.f * (x * (v[2][0] + v[0][2]) + y * (v[2][1] + v[1][2]) -
                                           2.f * z * (v[0][0] + v[1][1]) + w * (v[1][0] - v[0][1]));
            dL_dscales[2 * idx + 0] = (dh[0][0] * R[0][0] + dh[0][1] * R[1][0]) + dh[0][2] * R[2][0];
            dL_dscales[2 * idx + 1] = (dh[1][0] * R[0][1] + dh[1][1] * R[1][1]) + dh[1][2] * R[2][1];
            dL_dmean3Ds[3 * idx + 0] = dh[2][0];
            dL_dmean3Ds[3 * idx + 1] = dh[2][1];
            dL_dmean3Ds[3 * idx + 2] = dh[2][2];
        }
        if (shs != NULL) sh_backward(idx, D, M, means3D, campos, shs, clamped, dL_dcolors, dL_dmean3Ds, dL_dshs);
        /* densification hack, backward.cu:660-663 (double arithmetic as written there) */
        const float depth = transMats[9 * idx + 8];
        dL_dmean2Ds[3 * idx + 0] = (float)((double)(dL_dtransMats[9 * idx + 2] * depth) * 0.5 * (double)(float)W);
        dL_dmean2Ds[3 * idx + 1] = (float)((double)(dL_dtransMats[9 * idx + 5] * depth) * 0.5 * (double)(float)H);
    }
}

/* rasterizer_impl.cu:54-66 (checkFrustum / markVisible) */
void orc_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present)
{
    const float* vm = viewmatrix;
    for (int i = 0; i < P; i++) {
        const float* p = means3D + 3 * i;
        const float pvz = ((vm[2] * p[0] + vm[6] * p[1]) + vm[10] * p[2]) + vm[14];
        present[i] = pvz > 0.2f;
    }
}

/*
 * simple-knn distCUDA2 semantics (third-party module, absent from the mounted
 * reference: .gitmodules:1-3; call sites scene/Gaussians.py:77,218): for each
 * point the mean of the squared distances to its 3 nearest other points.
 * Brute force O(N^2); parity unpinned (no reference fixture exists).
 */
void orc_dist2_knn3(int N, const float* pts, float* out)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        float b0 = 3.402823466e+38f, b1 = b0, b2 = b0; /* FLT_MAX, as upstream */
        const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        for (int j = 0; j < N; j++) {
            if (j == i) continue;
            const float dx = pts[3 * j] - x, dy = pts[3 * j + 1] - y, dz = pts[3 * j + 2] - z;
            const float d = (dx * dx + dy * dy) + dz * dz;
            if (d < b2) {
                if (d < b1) { b2 = b1; if (d < b0) { b1 = b0; b0 = d; } else b1 = d; }
                else b2 = d;
            }
        }
        out[i] = ((b0 + b1) + b2) / 3.0f;
    }
}

void orc_set_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}
