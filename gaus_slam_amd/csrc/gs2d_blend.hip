// Per-tile alpha compositing, forward and backward (the hot kernels).
//
// Semantics: RAST/cuda_rasterizer/forward.cu:258-467 and backward.cu:143-463 of the reference.
// Design (gfx950, wave64):
//   * one workgroup per 16x16 tile, 4 waves, each wave owns an 8x8 pixel quadrant (compact footprint ->
//     coherent per-wave skip / early-out decisions);
//   * the tile's depth-sorted splat list is staged through LDS in batches of packed 80-byte splat
//     records (5 x ds_write_b128 per splat), then read back as wave-uniform broadcasts;
//   * forward: the 4 waves run independently (no workgroup barrier in the loop; a wave leaves as soon as
//     its 64 pixels are saturated);
//   * backward: per-(pixel,splat) gradients are summed over the 64 lanes with DPP row/bcast adds, combined
//     across the 4 waves in LDS, and flushed with ONE global atomic per (tile, splat, component) --
//     the reference issues one per (pixel, splat, component).
// Arithmetic follows the oracle's expression order; the file is compiled with -ffp-contract=off so the
// per-pixel recurrences reproduce the CPU oracle up to the ulp-level difference of expf.
#include "gs2d_common.h"

namespace {

__device__ __forceinline__ void wave_lds_fence()
{
    // LDS operations of one wave execute in order; this only stops the compiler from reordering them.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------- forward
template <bool USE_SA>
__global__ void __launch_bounds__(256)
blend_fwd_kernel(int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                 const float4* __restrict__ rec, const float* __restrict__ bg, float* __restrict__ out_color,
                 float* __restrict__ out_others, float* __restrict__ pix_state, size_t plane)
{
    __shared__ float4 sm[4][GS2D_REC_F4][64];
    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lx = (wave & 1) * 8 + (lane & 7), ly = (wave >> 1) * 8 + (lane >> 3);
    const int px = tx * GS2D_TILE + lx, py = ty * GS2D_TILE + ly;
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
    float Dp = 0.f, M1 = 0.f, M2 = 0.f, D2 = 0.f, distortion = 0.f, median_depth = 0.f;
    uint32_t median_contributor = 0;  // the reference keeps a float initialised to -1 and stores (uint)-> 0
    uint32_t last_contributor = 0;
    bool done = !inside;

    for (uint32_t base = range.x; base < range.y; base += 64) {
        if (__ballot(!done) == 0) break;
        const int n = min(64, (int)(range.y - base));
        if (lane < n) {
            const uint32_t id = point_list[base + lane];
            const float4* rp = rec + (size_t)id * GS2D_REC_F4;
#pragma unroll
            for (int k = 0; k < GS2D_REC_F4; k++) sm[wave][k][lane] = rp[k];
        }
        wave_lds_fence();
        for (int j = 0; j < n; j++) {
            if (__ballot(!done) == 0) break;
            if (!done) {
                const uint32_t contributor = (base - range.x) + (uint32_t)j + 1u;
                const float4 q0 = sm[wave][0][j], q1 = sm[wave][1][j], q2 = sm[wave][2][j];
                // forward.cu:360-371
                const float k0 = pxf * q2.x - q0.x, k1 = pxf * q2.y - q0.y, k2 = pxf * q2.z - q0.z;
                const float l0 = pyf * q2.x - q1.x, l1 = pyf * q2.y - q1.y, l2 = pyf * q2.z - q1.z;
                const float p0 = k1 * l2 - k2 * l1;
                const float p1 = k2 * l0 - k0 * l2;
                const float p2 = k0 * l1 - k1 * l0;
                if (p2 == 0.0f) continue;
                const float s0 = p0 / p2, s1 = p1 / p2;
                const float rho3d = s0 * s0 + s1 * s1;
                const float d0 = q0.w - pxf, d1 = q1.w - pyf;
                const float rho2d = GS2D_FILTER_INV_SQ * (d0 * d0 + d1 * d1);
                const float rho = fminf(rho3d, rho2d);
                float depth = (rho3d <= rho2d) ? (s0 * q2.x + s1 * q2.y) + q2.z : q2.z;
                if (depth < GS2D_NEAR_N) continue;
                const float power = -0.5f * rho;
                if (power > 0.0f) continue;
                const float alpha = fminf(0.99f, q2.w * expf(power));
                if (alpha < 1.0f / 255.0f) continue;
                const float test_T = T * (1 - alpha);
                if (test_T < 0.0001f) { done = true; continue; }
                const float w = alpha * T;
                if (T > 0.5f) { median_depth = depth; median_contributor = contributor; }
                if (USE_SA) {  // forward.cu:405-416
                    if (Dp > 0) {
                        const float exp_depth = median_depth;
                        float exp_std = (D2 - 2 * Dp * exp_depth) / (1 - T) + exp_depth * exp_depth;
                        exp_std = fmaxf(exp_std, 1e-7f);
                        const float error = (exp_depth - depth) * (exp_depth - depth);
                        const float conf = expf(-error / (4 * exp_std));
                        depth = conf * depth + (1 - conf) * exp_depth;
                    }
                    Dp += depth * w;
                    D2 += depth * depth * w;
                } else {  // forward.cu:417-423
                    const float A = 1 - T;
                    const float m = GS2D_FAR_N / (GS2D_FAR_N - GS2D_NEAR_N) * (1 - GS2D_NEAR_N / depth);
                    distortion += (m * m * A + M2 - 2 * m * M1) * w;
                    Dp += depth * w;
                    M1 += m * w;
                    M2 += m * m * w;
                }
                const float4 q3 = sm[wave][3][j], q4 = sm[wave][4][j];
                N0 += q3.x * w; N1 += q3.y * w; N2 += q3.z * w;
                C0 += q3.w * w; C1 += q4.x * w; C2 += q4.y * w;
                T = test_T;
                last_contributor = contributor;
            }
        }
        wave_lds_fence();
    }
    if (inside) {  // forward.cu:441-466
        const size_t HW = (size_t)H * W;
        const size_t pix = (size_t)W * py + px;
        out_color[pix] = C0 + T * bg0;
        out_color[HW + pix] = C1 + T * bg1;
        out_color[2 * HW + pix] = C2 + T * bg2;
        const float dstd = D2 - 2 * median_depth * Dp + median_depth * median_depth * (1 - T);
        out_others[pix] = Dp;
        out_others[HW + pix] = 1 - T;
        out_others[2 * HW + pix] = N0;
        out_others[3 * HW + pix] = N1;
        out_others[4 * HW + pix] = N2;
        out_others[5 * HW + pix] = median_depth;
        out_others[6 * HW + pix] = USE_SA ? D2 - 2 * median_depth * Dp + (1 - T) * median_depth * median_depth : distortion;
        const size_t si = (size_t)tile * GS2D_TILE_PIX + threadIdx.x;  // wave-major: coalesced 256-B rows
        pix_state[PS_TFINAL * plane + si] = T;
        pix_state[PS_M1 * plane + si] = M1;
        pix_state[PS_M2 * plane + si] = M2;
        pix_state[PS_MEDIAN * plane + si] = median_depth;
        pix_state[PS_STD * plane + si] = dstd;
        reinterpret_cast<uint32_t*>(pix_state)[PS_LAST * plane + si] = last_contributor;
        reinterpret_cast<uint32_t*>(pix_state)[PS_MEDC * plane + si] = median_contributor;
    }
}

// ------------------------------------------------------------------------------------------ backward
// Sum over the 64 lanes of a wave; the total is valid in lane 63.
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    // inclusive prefix inside each row of 16 (row_shr 1,2,4,8 with zero fill) ...
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    // ... then row_bcast15 into rows 1,3 and row_bcast31 into rows 2,3
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));
    return v;
}

constexpr int BWD_BATCH = 128;
constexpr int ACC_STRIDE = 20;

template <bool USE_SA>
__global__ void __launch_bounds__(256)
blend_bwd_kernel(int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                 const float4* __restrict__ rec, const float* __restrict__ bg, const float* __restrict__ pix_state,
                 size_t plane, const float* __restrict__ dL_dpix, const float* __restrict__ dL_dothers,
                 float* __restrict__ grad_rec)
{
    __shared__ float4 sm[GS2D_REC_F4][BWD_BATCH];
    __shared__ uint32_t sm_id[BWD_BATCH];
    __shared__ float acc[BWD_BATCH * ACC_STRIDE];
    __shared__ uint32_t s_max_last;

    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lx = (wave & 1) * 8 + (lane & 7), ly = (wave >> 1) * 8 + (lane >> 3);
    const int px = tx * GS2D_TILE + lx, py = ty * GS2D_TILE + ly;
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const uint32_t total = range.y - range.x;
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)W * py + px;
    const size_t si = (size_t)tile * GS2D_TILE_PIX + threadIdx.x;  // wave-major: coalesced 256-B rows

    if (threadIdx.x == 0) s_max_last = 0;
    __syncthreads();

    // backward.cu:197-248
    const float T_final = inside ? pix_state[PS_TFINAL * plane + si] : 0.f;
    float T = T_final;
    const uint32_t last_contributor = inside ? reinterpret_cast<const uint32_t*>(pix_state)[PS_LAST * plane + si] : 0u;
    const uint32_t median_contributor = inside ? reinterpret_cast<const uint32_t*>(pix_state)[PS_MEDC * plane + si] : 0u;
    float dpx0 = 0.f, dpx1 = 0.f, dpx2 = 0.f, dL_dreg = 0.f, dL_ddepth = 0.f, dL_daccum = 0.f;
    float dn0 = 0.f, dn1 = 0.f, dn2 = 0.f, dL_dmedian_depth = 0.f, mm = 0.f, mstd = 0.f, final_D = 0.f, final_D2 = 0.f;
    if (inside) {
        dpx0 = dL_dpix[pix]; dpx1 = dL_dpix[HW + pix]; dpx2 = dL_dpix[2 * HW + pix];
        dL_ddepth = dL_dothers[pix];
        dL_daccum = dL_dothers[HW + pix];
        dn0 = dL_dothers[2 * HW + pix]; dn1 = dL_dothers[3 * HW + pix]; dn2 = dL_dothers[4 * HW + pix];
        dL_dmedian_depth = dL_dothers[5 * HW + pix];
        dL_dreg = dL_dothers[6 * HW + pix];
        mm = pix_state[PS_MEDIAN * plane + si];
        mstd = pix_state[PS_STD * plane + si];
        final_D = pix_state[PS_M1 * plane + si];
        final_D2 = pix_state[PS_M2 * plane + si];
    }
    const float final_A = 1 - T_final;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const float bg_dot_dpixel = (bg0 * dpx0 + bg1 * dpx1) + bg2 * dpx2;
    float ar0 = 0.f, ar1 = 0.f, ar2 = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;
    float last_depth = 0.f, ln0 = 0.f, ln1 = 0.f, ln2 = 0.f, accum_depth_rec = 0.f, accum_alpha_rec = 0.f;
    float an0 = 0.f, an1 = 0.f, an2 = 0.f, last_dL_dT = 0.f, last_alpha = 0.f;

    // Nothing behind the deepest contributor of the whole tile can receive a gradient: start there.
    {
        uint32_t m = last_contributor;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
        if (lane == 0) atomicMax(&s_max_last, m);
    }
    __syncthreads();
    const uint32_t max_last = s_max_last;  // contributors are 1-based; splats with 0-based index >= max_last are dead

    // walk batches back to front; batch b covers 0-based splat indices [b*BATCH, b*BATCH+n)
    const int nbatches = (int)((max_last + BWD_BATCH - 1) / BWD_BATCH);
    for (int b = nbatches - 1; b >= 0; b--) {
        const uint32_t b0 = (uint32_t)b * BWD_BATCH;
        const int n = (int)min((uint32_t)BWD_BATCH, max_last - b0);
        __syncthreads();  // previous flush finished
        for (int i = threadIdx.x; i < n * ACC_STRIDE; i += 256) acc[i] = 0.f;
        if ((int)threadIdx.x < n) {
            const uint32_t id = point_list[range.x + b0 + threadIdx.x];
            sm_id[threadIdx.x] = id;
            const float4* rp = rec + (size_t)id * GS2D_REC_F4;
#pragma unroll
            for (int k = 0; k < GS2D_REC_F4; k++) sm[k][threadIdx.x] = rp[k];
        }
        __syncthreads();
        for (int j = n - 1; j >= 0; j--) {
            const uint32_t contributor = b0 + (uint32_t)j;  // 0-based, as in backward.cu:285
            bool active = inside && contributor < last_contributor;
            if (__ballot(active) == 0) continue;
            float g_c0 = 0.f, g_c1 = 0.f, g_c2 = 0.f, g_n0 = 0.f, g_n1 = 0.f, g_n2 = 0.f, g_op = 0.f;
            float g_T0 = 0.f, g_T1 = 0.f, g_T2 = 0.f, g_T3 = 0.f, g_T4 = 0.f, g_T5 = 0.f, g_T6 = 0.f, g_T7 = 0.f, g_T8 = 0.f;
            float g_mx = 0.f, g_my = 0.f;
            bool lowpass = false;
            if (active) {
                const float4 q0 = sm[0][j], q1 = sm[1][j], q2 = sm[2][j];
                const float k0 = pxf * q2.x - q0.x, k1 = pxf * q2.y - q0.y, k2 = pxf * q2.z - q0.z;
                const float l0 = pyf * q2.x - q1.x, l1 = pyf * q2.y - q1.y, l2 = pyf * q2.z - q1.z;
                const float p0 = k1 * l2 - k2 * l1;
                const float p1 = k2 * l0 - k0 * l2;
                const float p2 = k0 * l1 - k1 * l0;
                active = !(p2 == 0.0f);
                const float s0 = p0 / p2, s1 = p1 / p2;
                const float rho3d = s0 * s0 + s1 * s1;
                const float d0 = q0.w - pxf, d1 = q1.w - pyf;
                const float rho2d = GS2D_FILTER_INV_SQ * (d0 * d0 + d1 * d1);
                const float rho = fminf(rho3d, rho2d);
                float c_d = (rho3d <= rho2d) ? (s0 * q2.x + s1 * q2.y) + q2.z : q2.z;
                active = active && !(c_d < GS2D_NEAR_N);
                const float power = -0.5f * rho;
                active = active && !(power > 0.0f);
                const float G = expf(power);
                const float alpha = fminf(0.99f, q2.w * G);
                active = active && !(alpha < 1.0f / 255.0f);
                if (active) {
                    const float4 q3 = sm[3][j], q4 = sm[4][j];
                    T = T / (1.f - alpha);
                    const float w = alpha * T;
                    float dL_dalpha = 0.0f;
                    // backward.cu:331-344
                    ar0 = last_alpha * lc0 + (1.f - last_alpha) * ar0; lc0 = q3.w;
                    dL_dalpha += (q3.w - ar0) * dpx0; g_c0 = w * dpx0;
                    ar1 = last_alpha * lc1 + (1.f - last_alpha) * ar1; lc1 = q4.x;
                    dL_dalpha += (q4.x - ar1) * dpx1; g_c1 = w * dpx1;
                    ar2 = last_alpha * lc2 + (1.f - last_alpha) * ar2; lc2 = q4.y;
                    dL_dalpha += (q4.y - ar2) * dpx2; g_c2 = w * dpx2;
                    float conf = 1.f;
                    if (USE_SA) {  // backward.cu:347-351 (the reference evaluates this exp in double)
                        conf = T < 0.5f ? expf(-(c_d - mm) * (c_d - mm) / (4 * fmaxf(mstd / (1 - T_final), 1e-7f))) : 1.f;
                        c_d = c_d * conf + mm * (1 - conf);
                    }
                    float dL_dz = 0.0f, dL_dweight = 0.f;
                    const float m_d = GS2D_FAR_N / (GS2D_FAR_N - GS2D_NEAR_N) * (1 - GS2D_NEAR_N / c_d);
                    const float dmd_dd = (GS2D_FAR_N * GS2D_NEAR_N) / ((GS2D_FAR_N - GS2D_NEAR_N) * c_d * c_d);
                    if (contributor == median_contributor - 1u) dL_dz += dL_dmedian_depth;
                    if (USE_SA) dL_dweight += ((c_d - mm) * (c_d - mm)) * dL_dreg;
                    else dL_dweight += (final_D2 + m_d * m_d * final_A - 2 * m_d * final_D) * dL_dreg;
                    dL_dalpha += dL_dweight - last_dL_dT;
                    last_dL_dT = dL_dweight * alpha + (1 - alpha) * last_dL_dT;
                    if (USE_SA) dL_dz += conf * 2.0f * w * (c_d - mm) * dL_dreg;
                    else {
                        const float dL_dmd = 2.0f * (T * alpha) * (m_d * final_A - final_D) * dL_dreg;
                        dL_dz += dL_dmd * dmd_dd;
                    }
                    accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
                    last_depth = c_d;
                    dL_dalpha += (c_d - accum_depth_rec) * dL_ddepth;
                    accum_alpha_rec = last_alpha + (1.f - last_alpha) * accum_alpha_rec;
                    dL_dalpha += (1 - accum_alpha_rec) * dL_daccum;
                    // backward.cu:392-397
                    an0 = last_alpha * ln0 + (1.f - last_alpha) * an0; ln0 = q3.x;
                    dL_dalpha += (q3.x - an0) * dn0; g_n0 = alpha * T * dn0;
                    an1 = last_alpha * ln1 + (1.f - last_alpha) * an1; ln1 = q3.y;
                    dL_dalpha += (q3.y - an1) * dn1; g_n1 = alpha * T * dn1;
                    an2 = last_alpha * ln2 + (1.f - last_alpha) * an2; ln2 = q3.z;
                    dL_dalpha += (q3.z - an2) * dn2; g_n2 = alpha * T * dn2;
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                    const float dL_dG = q2.w * dL_dalpha;
                    dL_dz += conf * alpha * T * dL_ddepth;
                    if (rho3d <= rho2d) {  // backward.cu:419-449
                        const float dL_ds0 = dL_dG * -G * s0 + dL_dz * q2.x;
                        const float dL_ds1 = dL_dG * -G * s1 + dL_dz * q2.y;
                        const float dsx = dL_ds0 / p2, dsy = dL_ds1 / p2;
                        const float dp0 = dsx, dp1 = dsy, dp2 = -(dsx * s0 + dsy * s1);
                        const float dk0 = l1 * dp2 - l2 * dp1, dk1 = l2 * dp0 - l0 * dp2, dk2 = l0 * dp1 - l1 * dp0;
                        const float dl0 = dp1 * k2 - dp2 * k1, dl1 = dp2 * k0 - dp0 * k2, dl2 = dp0 * k1 - dp1 * k0;
                        g_T0 = -dk0; g_T1 = -dk1; g_T2 = -dk2;
                        g_T3 = -dl0; g_T4 = -dl1; g_T5 = -dl2;
                        g_T6 = pxf * dk0 + pyf * dl0 + dL_dz * s0;
                        g_T7 = pxf * dk1 + pyf * dl1 + dL_dz * s1;
                        g_T8 = pxf * dk2 + pyf * dl2 + dL_dz * 1.0f;
                    } else {  // backward.cu:450-457
                        g_mx = dL_dG * (-G * GS2D_FILTER_INV_SQ * d0);
                        g_my = dL_dG * (-G * GS2D_FILTER_INV_SQ * d1);
                        g_T8 = dL_dz;
                        lowpass = true;
                    }
                    g_op = G * dL_dalpha;
                }
            }
            if (__ballot(active) == 0) continue;
            // 64-lane sums (DPP), then lane 63 folds them into the tile accumulator in LDS
            const bool any_lp = __ballot(lowpass) != 0;
            g_c0 = wave_sum_to_lane63(g_c0); g_c1 = wave_sum_to_lane63(g_c1); g_c2 = wave_sum_to_lane63(g_c2);
            g_n0 = wave_sum_to_lane63(g_n0); g_n1 = wave_sum_to_lane63(g_n1); g_n2 = wave_sum_to_lane63(g_n2);
            g_T0 = wave_sum_to_lane63(g_T0); g_T1 = wave_sum_to_lane63(g_T1); g_T2 = wave_sum_to_lane63(g_T2);
            g_T3 = wave_sum_to_lane63(g_T3); g_T4 = wave_sum_to_lane63(g_T4); g_T5 = wave_sum_to_lane63(g_T5);
            g_T6 = wave_sum_to_lane63(g_T6); g_T7 = wave_sum_to_lane63(g_T7); g_T8 = wave_sum_to_lane63(g_T8);
            g_op = wave_sum_to_lane63(g_op);
            if (any_lp) { g_mx = wave_sum_to_lane63(g_mx); g_my = wave_sum_to_lane63(g_my); }
            if (lane == 63) {
                float* a = acc + j * ACC_STRIDE;
                atomicAdd(a + 0, g_c0); atomicAdd(a + 1, g_c1); atomicAdd(a + 2, g_c2);
                atomicAdd(a + 3, g_n0); atomicAdd(a + 4, g_n1); atomicAdd(a + 5, g_n2);
                atomicAdd(a + 6, g_T0); atomicAdd(a + 7, g_T1); atomicAdd(a + 8, g_T2);
                atomicAdd(a + 9, g_T3); atomicAdd(a + 10, g_T4); atomicAdd(a + 11, g_T5);
                atomicAdd(a + 12, g_T6); atomicAdd(a + 13, g_T7); atomicAdd(a + 14, g_T8);
                if (any_lp) { atomicAdd(a + 15, g_mx); atomicAdd(a + 16, g_my); }
                atomicAdd(a + 17, g_op);
            }
        }
        __syncthreads();
        // flush: 18 consecutive lanes add one splat's 72 contiguous bytes
        for (int e = threadIdx.x; e < n * 18; e += 256) {
            const int j = e / 18, k = e - j * 18;
            const float v = acc[j * ACC_STRIDE + k];
            if (v != 0.f) atomicAdd(grad_rec + (size_t)sm_id[j] * GS2D_GRAD_FLOATS + k, v);
        }
    }
}

}  // namespace

namespace gs2d {

void launch_blend_fwd(int W, int H, const uint2* ranges, const uint32_t* point_list, const float4* rec,
                      const float* bg, float* out_color, float* out_others, float* pix_state, int use_sa,
                      hipStream_t s)
{
    const int gx = (W + GS2D_TILE - 1) / GS2D_TILE, gy = (H + GS2D_TILE - 1) / GS2D_TILE;
    const size_t plane = (size_t)gx * gy * GS2D_TILE_PIX;
    if (use_sa)
        hipLaunchKernelGGL(blend_fwd_kernel<true>, dim3(gx * gy), dim3(256), 0, s, W, H, gx, ranges, point_list, rec,
                           bg, out_color, out_others, pix_state, plane);
    else
        hipLaunchKernelGGL(blend_fwd_kernel<false>, dim3(gx * gy), dim3(256), 0, s, W, H, gx, ranges, point_list, rec,
                           bg, out_color, out_others, pix_state, plane);
}

void launch_blend_bwd(int W, int H, const uint2* ranges, const uint32_t* point_list, const float4* rec,
                      const float* bg, const float* pix_state, const float* dL_dpix, const float* dL_dothers,
                      float* grad_rec, int use_sa, hipStream_t s)
{
    const int gx = (W + GS2D_TILE - 1) / GS2D_TILE, gy = (H + GS2D_TILE - 1) / GS2D_TILE;
    const size_t plane = (size_t)gx * gy * GS2D_TILE_PIX;
    if (use_sa)
        hipLaunchKernelGGL(blend_bwd_kernel<true>, dim3(gx * gy), dim3(256), 0, s, W, H, gx, ranges, point_list, rec,
                           bg, pix_state, plane, dL_dpix, dL_dothers, grad_rec);
    else
        hipLaunchKernelGGL(blend_bwd_kernel<false>, dim3(gx * gy), dim3(256), 0, s, W, H, gx, ranges, point_list, rec,
                           bg, pix_state, plane, dL_dpix, dL_dothers, grad_rec);
}

}  // namespace gs2d
