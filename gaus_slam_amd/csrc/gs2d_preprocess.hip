// Per-Gaussian kernels: forward preprocess (projection, bounding, colour), backward preprocess
// (dL/dT -> means/scales/rotations, SH backward), frustum marking.
//
// Semantics follow RAST/cuda_rasterizer/forward.cu:20-253, backward.cu:20-139,466-664 and
// auxiliary.h:61-291 of the reference; the code is written for gfx950 with plain float math
// (no glm), one thread per Gaussian, SoA inputs read coalesced, one packed 80-byte record out.
//
// This file MUST be compiled with -ffp-contract=off: tile rectangles, radii and depth keys are
// compared bit-exactly with the CPU oracle, and the expression order below is part of that contract.
#include "gs2d_common.h"

namespace {

__constant__ float SH_C0 = 0.28209479177387814f;
__constant__ float SH_C1 = 0.4886025119029199f;
__constant__ float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
__constant__ float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

// float -> int32: truncation toward zero, saturating, NaN -> 0 (what v_cvt_i32_f32 does, spelled out
// so the optimiser cannot treat out-of-range inputs as poison).
__device__ __forceinline__ int f2i_sat(float v)
{
    if (!(v == v)) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (int)(-2147483647 - 1);
    return (int)v;
}

// auxiliary.h:66-76
__device__ __forceinline__ void get_rect(float px, float py, int max_radius, int gx, int gy, int& minx, int& miny,
                                         int& maxx, int& maxy)
{
    const float r = (float)max_radius;
    minx = min(gx, max(0, f2i_sat((px - r) / (float)GS2D_TILE)));
    miny = min(gy, max(0, f2i_sat((py - r) / (float)GS2D_TILE)));
    maxx = min(gx, max(0, f2i_sat((px + r + (float)(GS2D_TILE - 1)) / (float)GS2D_TILE)));
    maxy = min(gy, max(0, f2i_sat((py + r + (float)(GS2D_TILE - 1)) / (float)GS2D_TILE)));
}

struct Mat3 {
    float m[3][3];  // m[row][col]
};

// auxiliary.h:212-234; IEEE 1/sqrt instead of the approximate rsqrt so the result is reproducible on the host.
__device__ __forceinline__ Mat3 quat_to_R(const float4 q /* w,x,y,z */, float& w, float& x, float& y, float& z)
{
    const float s = 1.0f / sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    w = q.x * s; x = q.y * s; y = q.z * s; z = q.w * s;
    Mat3 R;
    R.m[0][0] = 1.f - 2.f * (y * y + z * z);
    R.m[1][0] = 2.f * (x * y + w * z);
    R.m[2][0] = 2.f * (x * z - w * y);
    R.m[0][1] = 2.f * (x * y - w * z);
    R.m[1][1] = 1.f - 2.f * (x * x + z * z);
    R.m[2][1] = 2.f * (y * z + w * x);
    R.m[0][2] = 2.f * (x * z + w * y);
    R.m[1][2] = 2.f * (y * z - w * x);
    R.m[2][2] = 1.f - 2.f * (x * x + y * y);
    return R;
}

// forward.cu:75-115 / backward.cu:503-528.  T rows: Tu (T[0..2]), Tv (T[3..5]), Tw (T[6..8]).
__device__ __forceinline__ void compute_transmat(const float px, const float py, const float pz, const float sx,
                                                 const float sy, const Mat3& R, const float* pm, const float* vm,
                                                 int W, int H, float T[9], float normal[3])
{
    const float halfW = (float)W * 0.5f, halfWm = (float)(W - 1) * 0.5f;
    const float halfH = (float)H * 0.5f, halfHm = (float)(H - 1) * 0.5f;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float hx, hy, hz;
        if (i == 0) { hx = R.m[0][0] * sx; hy = R.m[1][0] * sx; hz = R.m[2][0] * sx; }
        else if (i == 1) { hx = R.m[0][1] * sy; hy = R.m[1][1] * sy; hz = R.m[2][1] * sy; }
        else { hx = px; hy = py; hz = pz; }
        float q0 = (pm[0] * hx + pm[4] * hy) + pm[8] * hz;
        float q1 = (pm[1] * hx + pm[5] * hy) + pm[9] * hz;
        float q3 = (pm[3] * hx + pm[7] * hy) + pm[11] * hz;
        if (i == 2) { q0 = q0 + pm[12]; q1 = q1 + pm[13]; q3 = q3 + pm[15]; }
        T[0 + i] = q0 * halfW + q3 * halfWm;
        T[3 + i] = q1 * halfH + q3 * halfHm;
        T[6 + i] = q3;
    }
    const float lx = R.m[0][2], ly = R.m[1][2], lz = R.m[2][2];
    normal[0] = (vm[0] * lx + vm[4] * ly) + vm[8] * lz;
    normal[1] = (vm[1] * lx + vm[5] * ly) + vm[9] * lz;
    normal[2] = (vm[2] * lx + vm[6] * ly) + vm[10] * lz;
}


// Optional rigid pose applied inside the preprocess kernels (tracking regime of the reference,
// render/__init__.py:31-36: means3D_cam = R x + t, rotations = q_cam (x) q, rendered with an identity view).
// pose_Rt: 12 floats, row-major [R | t]; pose_q: q_cam as (w,x,y,z).  Expression order is part of the
// bit-exact contract with oracle/gs2d_oracle.py::compose_pose.
__device__ __forceinline__ void pose_point(const float* Rt, float& x, float& y, float& z)
{
    const float nx = ((Rt[0] * x + Rt[1] * y) + Rt[2] * z) + Rt[3];
    const float ny = ((Rt[4] * x + Rt[5] * y) + Rt[6] * z) + Rt[7];
    const float nz = ((Rt[8] * x + Rt[9] * y) + Rt[10] * z) + Rt[11];
    x = nx; y = ny; z = nz;
}
// pytorch3d quaternion_multiply(a, b) = standardize(raw_multiply(a, b)), real part first
__device__ __forceinline__ float4 pose_quat(const float* a, const float4 b, float& sign)
{
    const float aw = a[0], ax = a[1], ay = a[2], az = a[3];
    float ow = ((aw * b.x - ax * b.y) - ay * b.z) - az * b.w;
    float ox = ((aw * b.y + ax * b.x) + ay * b.w) - az * b.z;
    float oy = ((aw * b.z - ax * b.w) + ay * b.x) + az * b.y;
    float oz = ((aw * b.w + ax * b.z) - ay * b.y) + az * b.x;
    sign = ow < 0.f ? -1.f : 1.f;
    if (ow < 0.f) { ow = -ow; ox = -ox; oy = -oy; oz = -oz; }
    return make_float4(ow, ox, oy, oz);
}

// forward.cu:20-71
__device__ void sh_to_rgb(int idx, int deg, int M, float posx, float posy, float posz, const float* campos, const float* shs,
                          uint8_t* clamped, float out[3])
{
    const float dx = posx - campos[0], dy = posy - campos[1], dz = posz - campos[2];
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
                r = r + SH_C2[0] * xy * SHC(4) + SH_C2[1] * yz * SHC(5) + SH_C2[2] * (2.0f * zz - xx - yy) * SHC(6) +
                    SH_C2[3] * xz * SHC(7) + SH_C2[4] * (xx - yy) * SHC(8);
                if (deg > 2) {
                    r = r + SH_C3[0] * y * (3.0f * xx - yy) * SHC(9) + SH_C3[1] * xy * z * SHC(10) +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * SHC(11) +
                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SHC(12) +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * SHC(13) + SH_C3[5] * z * (xx - yy) * SHC(14) +
                        SH_C3[6] * x * (xx - 3.0f * yy) * SHC(15);
                }
            }
        }
#undef SHC
        r += 0.5f;
        clamped[3 * idx + c] = (r < 0);
        out[c] = fmaxf(r, 0.0f);
    }
}

// Rotation matrix -> unit quaternion (w,x,y,z), w >= 0: the published algorithm of pytorch3d's matrix_to_quaternion
// (four candidates from the diagonal, the best-conditioned one wins, first maximum on ties), in the same float32
// operation order as gaus_slam_amd/tracking.py::matrix_to_quaternion.  One thread: it exists so that a tracking
// iteration needs neither a host sync (argmax -> index) nor ~20 tiny PyTorch launches for four numbers.
__device__ __forceinline__ void pose_quat_from_Rt(const float* __restrict__ Rt, float q_out[4])
{
    const float m00 = Rt[0], m01 = Rt[1], m02 = Rt[2], m10 = Rt[4], m11 = Rt[5], m12 = Rt[6], m20 = Rt[8], m21 = Rt[9], m22 = Rt[10];
    const float qa[4] = {sqrtf(fmaxf(((1.0f + m00) + m11) + m22, 0.f)), sqrtf(fmaxf(((1.0f + m00) - m11) - m22, 0.f)),
                         sqrtf(fmaxf(((1.0f - m00) + m11) - m22, 0.f)), sqrtf(fmaxf(((1.0f - m00) - m11) + m22, 0.f))};
    const float cand[4][4] = {{qa[0] * qa[0], m21 - m12, m02 - m20, m10 - m01},
                              {m21 - m12, qa[1] * qa[1], m10 + m01, m02 + m20},
                              {m02 - m20, m10 + m01, qa[2] * qa[2], m12 + m21},
                              {m10 - m01, m20 + m02, m21 + m12, qa[3] * qa[3]}};
    int best = 0;
    for (int i = 1; i < 4; i++)
        if (qa[i] > qa[best]) best = i;
    const float den = 2.0f * fmaxf(qa[best], 0.1f);
    float q[4];
    for (int i = 0; i < 4; i++) q[i] = cand[best][i] / den;
    const bool neg = q[0] < 0.f;
    for (int i = 0; i < 4; i++) q_out[i] = neg ? -q[i] : q[i];
}
__global__ void pose_quat_kernel(const float* __restrict__ Rt, float* __restrict__ q_out)
{
    if (threadIdx.x != 0) return;
    float q[4];
    pose_quat_from_Rt(Rt, q);
    for (int i = 0; i < 4; i++) q_out[i] = q[i];
}
// q_cam of a posed call: the caller's (gs2d_pose_quat's output, or its own), or -- pose_q == NULL -- derived here from the
// rotation block, the same code gs2d_pose_quat runs (every thread for itself: fifty operations, no kernel and no dependent
// dispatch in front of the preprocess of every tracking iteration)
__device__ __forceinline__ void load_pose_quat(const float* __restrict__ pose_Rt, const float* __restrict__ pose_q, float q[4])
{
    if (pose_q != nullptr) { q[0] = pose_q[0]; q[1] = pose_q[1]; q[2] = pose_q[2]; q[3] = pose_q[3]; }
    else pose_quat_from_Rt(pose_Rt, q);
}

__device__ __forceinline__ void
preprocess_fwd_body(int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ scales,
                      float scale_modifier, const float* __restrict__ rotations, const float* __restrict__ opacities,
                      const float* __restrict__ shs, const float* __restrict__ transMat_precomp,
                      const float* __restrict__ colors_precomp, const CamParams cam, int* __restrict__ radii,
                      float* __restrict__ depths, float4* __restrict__ rec, uint32_t* __restrict__ tiles_touched,
                      ushort4* __restrict__ rect, uint8_t* __restrict__ clamped, const float* __restrict__ pose_Rt,
                      const float* __restrict__ pose_q, uint32_t* __restrict__ block_sums)
{
    __shared__ uint32_t wave_tiles[4];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const bool valid = idx < P;
    // forward.cu:183-184: invisible unless proven otherwise
    int out_radius = 0;
    uint32_t out_tiles = 0;
    ushort4 out_rect = make_ushort4(0, 0, 0, 0);
    float out_depth = 0.f;
    float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0, r4 = r0;

    float px = 0.f, py = 0.f, pz = 0.f;
    if (valid) { px = means3D[3 * idx]; py = means3D[3 * idx + 1]; pz = means3D[3 * idx + 2]; }
    if (pose_Rt != nullptr) pose_point(pose_Rt, px, py, pz);
    const float* vm = cam.vm;
    const float pvx = ((vm[0] * px + vm[4] * py) + vm[8] * pz) + vm[12];
    const float pvy = ((vm[1] * px + vm[5] * py) + vm[9] * pz) + vm[13];
    const float pvz = ((vm[2] * px + vm[6] * py) + vm[10] * pz) + vm[14];
    do {
        if (!valid) break;
        if (pvz <= 0.2f) break;  // auxiliary.h:199
        float T[9], normal[3];
        if (transMat_precomp == nullptr) {
            float4 q = reinterpret_cast<const float4*>(rotations)[idx];
            if (pose_Rt != nullptr) { float sgn, qc[4]; load_pose_quat(pose_Rt, pose_q, qc); q = pose_quat(qc, q, sgn); }
            const float2 sc = reinterpret_cast<const float2*>(scales)[idx];
            float w, x, y, z;
            const Mat3 R = quat_to_R(q, w, x, y, z);
            compute_transmat(px, py, pz, scale_modifier * sc.x, scale_modifier * sc.y, R, cam.pm, cam.vm, cam.W,
                             cam.H, T, normal);
        } else {
#pragma unroll
            for (int i = 0; i < 9; i++) T[i] = transMat_precomp[9 * (size_t)idx + i];
            normal[0] = 0.f; normal[1] = 0.f; normal[2] = 1.f;
        }
        // forward.cu:211-216
        const float cosv = -((pvx * normal[0] + pvy * normal[1]) + pvz * normal[2]);
        if (cosv == 0) break;
        const float mult = cosv > 0 ? 1.f : -1.f;
        normal[0] = mult * normal[0]; normal[1] = mult * normal[1]; normal[2] = mult * normal[2];
        // forward.cu:119-147, cutoff 3
        const float c2 = 3.0f * 3.0f;
        const float dist = ((T[6] * T[6]) * c2 + (T[7] * T[7]) * c2) + (T[8] * T[8]) * -1.0f;
        const float inv = 1 / dist;
        const float f0 = inv * c2, f1 = inv * c2, f2 = inv * -1.0f;
        if (dist == 0.0f) break;
        const float cx = ((f0 * T[0]) * T[6] + (f1 * T[1]) * T[7]) + (f2 * T[2]) * T[8];
        const float cy = ((f0 * T[3]) * T[6] + (f1 * T[4]) * T[7]) + (f2 * T[5]) * T[8];
        const float tx = ((f0 * T[0]) * T[0] + (f1 * T[1]) * T[1]) + (f2 * T[2]) * T[2];
        const float ty = ((f0 * T[3]) * T[3] + (f1 * T[4]) * T[4]) + (f2 * T[5]) * T[5];
        const float ex = sqrtf(fmaxf(1e-4f, cx * cx - tx));
        const float ey = sqrtf(fmaxf(1e-4f, cy * cy - ty));
        const float radius = ceilf(fmaxf(ex, ey));
        int minx, miny, maxx, maxy;
        get_rect(cx, cy, f2i_sat(radius), cam.gx, cam.gy, minx, miny, maxx, maxy);
        if ((maxx - minx) * (maxy - miny) == 0) break;
        float col[3];
        if (colors_precomp == nullptr) sh_to_rgb(idx, D, M, px, py, pz, cam.campos, shs, clamped, col);
        else { col[0] = colors_precomp[3 * idx]; col[1] = colors_precomp[3 * idx + 1]; col[2] = colors_precomp[3 * idx + 2]; }
        out_depth = pvz;
        out_radius = f2i_sat(radius);
        r0 = make_float4(T[0], T[1], T[2], cx);
        r1 = make_float4(T[3], T[4], T[5], cy);
        r2 = make_float4(T[6], T[7], T[8], opacities[idx]);
        r3 = make_float4(normal[0], normal[1], normal[2], col[0]);
        // q4.z: rho_max = 2 ln(255 opacity) with a safety margin -- the largest min(rho3d,rho2d) at which
        // opacity*exp(-rho/2) can still reach 1/255.  Only used to CULL (never to decide) in the blend kernels.
        const float opa = opacities[idx];
        const float rho_max = opa > 0.f ? 2.0f * logf(255.0f * opa) * 1.0001f + 1e-3f : (opa == opa ? -1.f : 1e30f);
        r4 = make_float4(col[1], col[2], rho_max, 0.f);
        // Tile rectangle.  Reference: the square of the 3-sigma radius (rasterizer_impl.cu:70-111 re-derives it from
        // radii).  The pixels the splat can actually reach with alpha >= 1/255 lie inside its footprint bound, which is
        // much smaller for the usual opacities (opacity 0.1 ends at 2.5 sigma, and the bound follows the ellipse instead
        // of its circumscribed square): with cam.tight only the tiles of [reference rectangle] x [footprint bound] become
        // instances -- every dropped (Gaussian, tile) pair is one whose pixels the reference would all `continue` past
        // (forward.cu:385-387), so no output changes; the same bound, per sub-block, is what the cull kernel applies later.
        if (cam.tight) {
            const Gs2dFootprint fp = gs2d_footprint(r0, r1, r2, rho_max);
            if (fp.kind == 0) { maxx = minx; maxy = miny; }
            else if (fp.kind == 1) {
                // 0.05 px of slack over the cull kernel's own intervals (it evaluates them per tile with other roundings)
                const float lox = fminf(cx - fp.rl, fp.cx - fp.ex - fp.mx) - 0.05f, hix = fmaxf(cx + fp.rl, fp.cx + fp.ex + fp.mx) + 0.05f;
                const float loy = fminf(cy - fp.rl, fp.cy - fp.ey - fp.my) - 0.05f, hiy = fmaxf(cy + fp.rl, fp.cy + fp.ey + fp.my) + 0.05f;
                if (lox == lox && hix == hix && loy == loy && hiy == hiy) {
                    // tile column t holds pixels 16 t .. 16 t + 15: met iff 16 t + 15 >= lo and 16 t <= hi
                    minx = max(minx, f2i_sat(ceilf((lox - 15.f) * (1.0f / GS2D_TILE))));
                    miny = max(miny, f2i_sat(ceilf((loy - 15.f) * (1.0f / GS2D_TILE))));
                    maxx = min(maxx, min(f2i_sat(floorf(hix * (1.0f / GS2D_TILE))), cam.gx) + 1);
                    maxy = min(maxy, min(f2i_sat(floorf(hiy * (1.0f / GS2D_TILE))), cam.gy) + 1);
                    if (maxx <= minx || maxy <= miny) { maxx = minx; maxy = miny; }
                }
            }
        }
        out_tiles = (uint32_t)((maxy - miny) * (maxx - minx));
        out_rect = make_ushort4((unsigned short)minx, (unsigned short)miny, (unsigned short)maxx, (unsigned short)maxy);
    } while (0);
    if (valid) {
        radii[idx] = out_radius;
        tiles_touched[idx] = out_tiles;
        rect[idx] = out_rect;
        depths[idx] = out_depth;
    }
    {
        // the 80-byte records of the workgroup's 256 Gaussians are one contiguous 20-KB block: staged through LDS so that
        // every store instruction writes 1 KB of consecutive bytes (a lane-per-record store touches 64 separate lines:
        // 22.8 -> 17.7 us at 500k Gaussians)
        __shared__ float4 stage[GS2D_REC_F4][256];
        stage[0][threadIdx.x] = r0; stage[1][threadIdx.x] = r1; stage[2][threadIdx.x] = r2;
        stage[3][threadIdx.x] = r3; stage[4][threadIdx.x] = r4;
        __syncthreads();
        const int nrec = min(256, P - (int)blockIdx.x * 256);
        float4* rb = rec + (size_t)blockIdx.x * 256 * GS2D_REC_F4;
#pragma unroll
        for (int k = 0; k < GS2D_REC_F4; k++) {
            const int j = k * 256 + threadIdx.x;  // float4 number j of the block = component j % 5 of record j / 5
            if (j < nrec * GS2D_REC_F4) rb[j] = stage[j % GS2D_REC_F4][j / GS2D_REC_F4];
        }
    }
    // first step of the prefix sum over tiles_touched: this workgroup's total (see launch_duplicate)
    uint32_t tsum = out_tiles;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tsum += (uint32_t)__shfl_xor((int)tsum, d, 64);
    if ((threadIdx.x & 63) == 0) wave_tiles[threadIdx.x >> 6] = tsum;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = (wave_tiles[0] + wave_tiles[1]) + (wave_tiles[2] + wave_tiles[3]);
}

__global__ void __launch_bounds__(256)
preprocess_fwd_kernel(int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ scales,
                      float scale_modifier, const float* __restrict__ rotations, const float* __restrict__ opacities,
                      const float* __restrict__ shs, const float* __restrict__ transMat_precomp,
                      const float* __restrict__ colors_precomp, const CamParams cam, int* __restrict__ radii,
                      float* __restrict__ depths, float4* __restrict__ rec, uint32_t* __restrict__ tiles_touched,
                      ushort4* __restrict__ rect, uint8_t* __restrict__ clamped, const float* __restrict__ pose_Rt,
                      const float* __restrict__ pose_q, uint32_t* __restrict__ block_sums)
{
    preprocess_fwd_body(P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, transMat_precomp, colors_precomp, cam,
                        radii, depths, rec, tiles_touched, rect, clamped, pose_Rt, pose_q, block_sums);
}

// Batched form (gs2d_forward_batch): blockIdx.y = frame; the Gaussians are shared, camera and outputs come from a by-value table
__global__ void __launch_bounds__(256)
preprocess_fwd_batch_kernel(int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ scales,
                            float scale_modifier, const float* __restrict__ rotations, const float* __restrict__ opacities,
                            const float* __restrict__ shs, const float* __restrict__ transMat_precomp,
                            const float* __restrict__ colors_precomp, const CamParams cam0, const gs2d::PreFwdFrames tab)
{
    const gs2d::PreFwdFrame& f = tab.f[blockIdx.y];
    CamParams cam = cam0;
    cam.vm = f.vm; cam.pm = f.pm; cam.campos = f.campos;
    preprocess_fwd_body(P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, transMat_precomp, colors_precomp, cam,
                        f.radii, f.depths, f.rec, f.tiles_touched, f.rect, f.clamped, nullptr, nullptr, f.block_sums);
}

// backward.cu:20-139
__device__ void sh_backward(int idx, int deg, int M, float posx, float posy, float posz, const float* campos, const float* shs,
                            const uint8_t* clamped, const float dL_dcolor[3], float dL_dmean[3], float* dL_dshs)
{
    const float ox = posx - campos[0], oy = posy - campos[1], oz = posz - campos[2];
    const float len = sqrtf((ox * ox + oy * oy) + oz * oz);
    const float x = ox / len, y = oy / len, z = oz / len;
    const float* sh = shs + (size_t)idx * M * 3;
    float* dsh = dL_dshs + (size_t)idx * M * 3;
    for (int i = 3 * (deg + 1) * (deg + 1); i < 3 * M; i++) dsh[i] = 0.f;  // coefficients above the active degree
    float dRGB[3];
    for (int c = 0; c < 3; c++) dRGB[c] = dL_dcolor[c] * (clamped[3 * idx + c] ? 0.f : 1.f);
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
    // auxiliary.h:127-137
    const float sum2 = ox * ox + oy * oy + oz * oz;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    dL_dmean[0] += ((+sum2 - ox * ox) * ddir[0] - oy * ox * ddir[1] - oz * ox * ddir[2]) * invsum32;
    dL_dmean[1] += (-ox * oy * ddir[0] + (sum2 - oy * oy) * ddir[1] - oz * oy * ddir[2]) * invsum32;
    dL_dmean[2] += (-ox * oz * ddir[0] - oy * oz * ddir[1] + (sum2 - oz * oz) * ddir[2]) * invsum32;
}

// backward.cu:466-664 for one Gaussian.  Reads the packed gradient record written by the backward blend and writes
// every element of the public gradient tensors (zeros for Gaussians culled by the forward).  With a pose
// (tracking regime) means/rotations are transformed exactly as in the forward, dL_dmean3D / dL_drot are mapped back
// to the untransformed parameters and pg[12] accumulates this Gaussian's share of dL/d[R|t].
__device__ __forceinline__ void
preprocess_bwd_one(int idx, int P, int D, int M, const float* __restrict__ means3D, const float4* __restrict__ rec,
                   const int* __restrict__ radii, const float* __restrict__ shs, const uint8_t* __restrict__ clamped,
                   const float* __restrict__ scales, const float* __restrict__ rotations, const CamParams& cam,
                   const float* __restrict__ grad_rec, float* __restrict__ dL_dtransMat, float* __restrict__ dL_dnormal,
                   float* __restrict__ dL_dcolor, float* __restrict__ dL_dopacity, float* __restrict__ dL_dsh,
                   float* __restrict__ dL_dmean2D, float* __restrict__ dL_dmean3D, float* __restrict__ dL_dscale,
                   float* __restrict__ dL_drot, const float* __restrict__ pose_Rt, const float* __restrict__ pose_q, float pg[12])
{
    // (Staging the 80-byte gradient / geometry records of a workgroup's 256 Gaussians through LDS with coalesced loads,
    // as the forward does for its stores, measured slower here: 29.3 vs 26.6 us.)
    (void)P;
    // pose-only call (tracking with every Gaussian parameter detached, render/__init__.py:31-36): the per-Gaussian
    // tensors are not wanted, only the pose gradient reduced from dL/dmean -- all six pointers are NULL then
    const bool wr = dL_dmean3D != nullptr;
    if (!(radii[idx] > 0)) {
        if (!wr) return;
        // culled in the forward: every gradient is exactly zero (the reference gets this from torch::zeros,
        // rasterize_points.cu:192-200; here the kernel writes it so the caller can hand in uninitialised memory)
#pragma unroll
        for (int i = 0; i < 3; i++) { dL_dcolor[3 * idx + i] = 0.f; dL_dmean2D[3 * idx + i] = 0.f; dL_dmean3D[3 * idx + i] = 0.f; }
        if (dL_dnormal) for (int i = 0; i < 3; i++) dL_dnormal[3 * idx + i] = 0.f;
        dL_dopacity[idx] = 0.f;
        if (dL_dtransMat) {
#pragma unroll
            for (int i = 0; i < 9; i++) dL_dtransMat[9 * (size_t)idx + i] = 0.f;
        }
        dL_dscale[2 * idx] = 0.f; dL_dscale[2 * idx + 1] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++) dL_drot[4 * idx + i] = 0.f;
        if (shs != nullptr)
            for (int i = 0; i < 3 * M; i++) dL_dsh[(size_t)idx * M * 3 + i] = 0.f;
        return;
    }
    const float4* gr4 = reinterpret_cast<const float4*>(grad_rec) + (size_t)idx * (GS2D_GRAD_FLOATS / 4);
    float g[GS2D_GRAD_FLOATS];
#pragma unroll
    for (int i = 0; i < GS2D_GRAD_FLOATS / 4; i++) {
        const float4 v = gr4[i];
        g[4 * i] = v.x; g[4 * i + 1] = v.y; g[4 * i + 2] = v.z; g[4 * i + 3] = v.w;
    }
    // unpack the blend-stage gradients into the public tensors
    float dcol[3] = {g[0], g[1], g[2]};
    if (wr) { dL_dcolor[3 * idx] = dcol[0]; dL_dcolor[3 * idx + 1] = dcol[1]; dL_dcolor[3 * idx + 2] = dcol[2]; }
    if (dL_dnormal) { dL_dnormal[3 * idx] = g[3]; dL_dnormal[3 * idx + 1] = g[4]; dL_dnormal[3 * idx + 2] = g[5]; }
    if (wr) dL_dopacity[idx] = g[15];
    float dT[9];
#pragma unroll
    for (int i = 0; i < 9; i++) dT[i] = g[6 + i];
    float dTout[9];
#pragma unroll
    for (int i = 0; i < 9; i++) dTout[i] = dT[i];  // what the reference leaves in dL_dtransMat
    const float dmx = g[16], dmy = g[17];

    const bool precomp = (scales == nullptr);
    const float wx = means3D[3 * idx], wy = means3D[3 * idx + 1], wz = means3D[3 * idx + 2];  // untransformed
    float px = wx, py = wy, pz = wz;
    if (pose_Rt != nullptr) pose_point(pose_Rt, px, py, pz);
    const float* pm = cam.pm;
    const float* vm = cam.vm;
    float T[9], normal[3] = {0.f, 0.f, 0.f};
    float Pm[4][3];
    Mat3 R;
    float sx = 0.f, sy = 0.f, w = 0.f, x = 0.f, y = 0.f, z = 0.f, qsign = 1.f;
    float qcam[4] = {1.f, 0.f, 0.f, 0.f};
    // rec == nullptr (launch_preprocess_bwd): Tw.z, the one word of the record the usual path needs, is recomputed below bit
    // for bit, so the 80-byte-strided record reads (a whole line per Gaussian for four useful bytes) are not issued
    float4 q2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rec != nullptr) q2 = rec[(size_t)idx * GS2D_REC_F4 + 2];
    if (precomp) {
        const float4* rp = rec + (size_t)idx * GS2D_REC_F4;
        const float4 q0 = rp[0], q1 = rp[1];
        T[0] = q0.x; T[1] = q0.y; T[2] = q0.z; T[3] = q1.x; T[4] = q1.y; T[5] = q1.z; T[6] = q2.x; T[7] = q2.y; T[8] = q2.z;
    } else {
        float4 q = reinterpret_cast<const float4*>(rotations)[idx];
        if (pose_Rt != nullptr) { load_pose_quat(pose_Rt, pose_q, qcam); q = pose_quat(qcam, q, qsign); }
        const float2 sc = reinterpret_cast<const float2*>(scales)[idx];
        sx = sc.x; sy = sc.y;  // backward.cu:504: scale_modifier is ignored here
        R = quat_to_R(q, w, x, y, z);
        compute_transmat(px, py, pz, sx, sy, R, pm, vm, cam.W, cam.H, T, normal);
        const float halfW = (float)cam.W * 0.5f, halfWm = (float)(cam.W - 1) * 0.5f;
        const float halfH = (float)cam.H * 0.5f, halfHm = (float)(cam.H - 1) * 0.5f;
#pragma unroll
        for (int a = 0; a < 4; a++) {
            Pm[a][0] = pm[4 * a] * halfW + pm[4 * a + 3] * halfWm;
            Pm[a][1] = pm[4 * a + 1] * halfH + pm[4 * a + 3] * halfHm;
            Pm[a][2] = pm[4 * a + 3];
        }
    }
    bool early = false;
    if (dmx != 0 || dmy != 0) {  // backward.cu:538-577
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
        if (precomp) {
#pragma unroll
            for (int i = 0; i < 9; i++) dTout[i] = dT[i];
            early = true;
        }
    }
    float dmean[3] = {0.f, 0.f, 0.f};
    bool have_mean = false;
    if (!precomp && !early) {
        float dh[3][3];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int a = 0; a < 3; a++) dh[i][a] = (Pm[a][0] * dT[i] + Pm[a][1] * dT[3 + i]) + Pm[a][2] * dT[6 + i];
        float dtn[3] = {(vm[0] * g[3] + vm[1] * g[4]) + vm[2] * g[5], (vm[4] * g[3] + vm[5] * g[4]) + vm[6] * g[5],
                        (vm[8] * g[3] + vm[9] * g[4]) + vm[10] * g[5]};
        const float pvx = ((vm[0] * px + vm[4] * py) + vm[8] * pz) + vm[12];
        const float pvy = ((vm[1] * px + vm[5] * py) + vm[9] * pz) + vm[13];
        const float pvz = ((vm[2] * px + vm[6] * py) + vm[10] * pz) + vm[14];
        const float cosv = -((pvx * normal[0] + pvy * normal[1]) + pvz * normal[2]);
        const float mult = cosv > 0 ? 1.f : -1.f;
#pragma unroll
        for (int a = 0; a < 3; a++) dtn[a] = mult * dtn[a];
        float v[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++) { v[r][0] = dh[0][r] * sx; v[r][1] = dh[1][r] * sy; v[r][2] = dtn[r]; }
        // auxiliary.h:237-281
        float4 dq;
        dq.x = 2.f * (x * (v[2][1] - v[1][2]) + y * (v[0][2] - v[2][0]) + z * (v[1][0] - v[0][1]));
        dq.y = 2.f * (-2.f * x * (v[1][1] + v[2][2]) + y * (v[1][0] + v[0][1]) + z * (v[2][0] + v[0][2]) + w * (v[2][1] - v[1][2]));
        dq.z = 2.f * (x * (v[1][0] + v[0][1]) - 2.f * y * (v[0][0] + v[2][2]) + z * (v[2][1] + v[1][2]) + w * (v[0][2] - v[2][0]));
        dq.w = 2.f * (x * (v[2][0] + v[0][2]) + y * (v[2][1] + v[1][2]) - 2.f * z * (v[0][0] + v[1][1]) + w * (v[1][0] - v[0][1]));
        if (pose_Rt != nullptr) {  // dL/dq = sign * L(q_cam)^T dL/dq'
            const float aw = qcam[0], ax = qcam[1], ay = qcam[2], az = qcam[3];
            const float4 t = dq;
            dq.x = qsign * (((aw * t.x + ax * t.y) + ay * t.z) + az * t.w);
            dq.y = qsign * (((-ax * t.x + aw * t.y) + az * t.z) - ay * t.w);
            dq.z = qsign * (((-ay * t.x - az * t.y) + aw * t.z) + ax * t.w);
            dq.w = qsign * (((-az * t.x + ay * t.y) - ax * t.z) + aw * t.w);
        }
        if (wr) reinterpret_cast<float4*>(dL_drot)[idx] = dq;
        float2 ds;
        ds.x = (dh[0][0] * R.m[0][0] + dh[0][1] * R.m[1][0]) + dh[0][2] * R.m[2][0];
        ds.y = (dh[1][0] * R.m[0][1] + dh[1][1] * R.m[1][1]) + dh[1][2] * R.m[2][1];
        if (wr) reinterpret_cast<float2*>(dL_dscale)[idx] = ds;
        dmean[0] = dh[2][0]; dmean[1] = dh[2][1]; dmean[2] = dh[2][2];
        have_mean = true;
    }
    if (shs != nullptr) {
        sh_backward(idx, D, M, px, py, pz, cam.campos, shs, clamped, dcol, dmean, dL_dsh);
        have_mean = true;
    }
    (void)have_mean;
    if (pose_Rt != nullptr) {
        // x_cam = R x + t:  dL/dR = g (x) x, dL/dt = g, dL/dx = R^T g
        const float g0 = dmean[0], g1 = dmean[1], g2 = dmean[2];
        pg[0] += g0 * wx; pg[1] += g0 * wy; pg[2] += g0 * wz; pg[3] += g0;  // (+=: a thread may own several Gaussians)
        pg[4] += g1 * wx; pg[5] += g1 * wy; pg[6] += g1 * wz; pg[7] += g1;
        pg[8] += g2 * wx; pg[9] += g2 * wy; pg[10] += g2 * wz; pg[11] += g2;
        dmean[0] = (pose_Rt[0] * g0 + pose_Rt[4] * g1) + pose_Rt[8] * g2;
        dmean[1] = (pose_Rt[1] * g0 + pose_Rt[5] * g1) + pose_Rt[9] * g2;
        dmean[2] = (pose_Rt[2] * g0 + pose_Rt[6] * g1) + pose_Rt[10] * g2;
    }
    if (!wr) return;
    dL_dmean3D[3 * idx] = dmean[0]; dL_dmean3D[3 * idx + 1] = dmean[1]; dL_dmean3D[3 * idx + 2] = dmean[2];
    if (precomp || early) {  // no scale / rotation gradient on these paths (backward.cu:565-579)
        dL_dscale[2 * idx] = 0.f; dL_dscale[2 * idx + 1] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++) dL_drot[4 * idx + i] = 0.f;
    }
    if (dL_dtransMat) {
#pragma unroll
        for (int i = 0; i < 9; i++) dL_dtransMat[9 * (size_t)idx + i] = dTout[i];
    }
    // densification hack, backward.cu:660-663 (double arithmetic as written in the reference)
    const float depth = rec != nullptr ? q2.z : T[8];
    dL_dmean2D[3 * idx + 0] = (float)((double)(dTout[2] * depth) * 0.5 * (double)(float)cam.W);
    dL_dmean2D[3 * idx + 1] = (float)((double)(dTout[5] * depth) * 0.5 * (double)(float)cam.H);
    dL_dmean2D[3 * idx + 2] = 0.f;
}

__global__ void __launch_bounds__(256)
preprocess_bwd_kernel(int first, int P, int D, int M, const float* __restrict__ means3D, const float4* __restrict__ rec,
                      const int* __restrict__ radii, const float* __restrict__ shs,
                      const uint8_t* __restrict__ clamped, const float* __restrict__ scales,
                      const float* __restrict__ rotations, const CamParams cam, const float* __restrict__ grad_rec,
                      float* __restrict__ dL_dtransMat, float* __restrict__ dL_dnormal, float* __restrict__ dL_dcolor,
                      float* __restrict__ dL_dopacity, float* __restrict__ dL_dsh, float* __restrict__ dL_dmean2D,
                      float* __restrict__ dL_dmean3D, float* __restrict__ dL_dscale, float* __restrict__ dL_drot,
                      const float* __restrict__ pose_Rt, const float* __restrict__ pose_q, float* __restrict__ dL_dpose,
                      float* __restrict__ pose_partials)
{
    // P = one past the last Gaussian of this launch's range.  One Gaussian per thread; with a pose the grid is capped and
    // the threads stride over the range, so that fewer workgroups queue up on the 12 pose-gradient words.
    // pose gradient: sum_i g_i (x) x_i and sum_i g_i over the Gaussians of this workgroup, then 12 atomics
    float pg[12];
#pragma unroll
    for (int i = 0; i < 12; i++) pg[i] = 0.f;
    for (int idx = first + blockIdx.x * 256 + threadIdx.x; idx < P; idx += gridDim.x * 256)
        preprocess_bwd_one(idx, P, D, M, means3D, rec, radii, shs, clamped, scales, rotations, cam, grad_rec, dL_dtransMat,
                           dL_dnormal, dL_dcolor, dL_dopacity, dL_dsh, dL_dmean2D, dL_dmean3D, dL_dscale, dL_drot, pose_Rt,
                           pose_q, pg);
    if (dL_dpose != nullptr) {
        __shared__ float red[4][12];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            float v = pg[i];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
            if (lane == 0) red[wave][i] = v;
        }
        __syncthreads();
        if (threadIdx.x < 12) {
            const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
            // deterministic mode: one partial per workgroup, summed in a fixed order by pose_reduce_kernel
            if (pose_partials != nullptr) pose_partials[blockIdx.x * 12 + threadIdx.x] = v;
            else if (v != 0.f) atomicAdd(dL_dpose + threadIdx.x, v);
        }
    }
}

// Pose-only per-Gaussian stage (tracking with every Gaussian parameter detached, render/__init__.py:31-36): the blend stage
// ran in its POSE instantiation and left, densely, dense_T[g] = (dT[2], dT[5], dT[8], -) and dense_m2d[g] = dL_dmean2D.xy.
// dL/dmean = row 2 of dL_dM = Pm dL_dT^T (backward.cu:583-590, 637-663) needs nothing else -- no rotation, no scale, no
// transform -- unless the rare low-pass pair is non-zero, whose centre-formula terms (backward.cu:538-563) need T.
// Same expression order as preprocess_bwd_one, so both paths produce the same per-Gaussian dmean bit for bit.
// 40 B per visible Gaussian instead of ~150.
__global__ void __launch_bounds__(256)
preprocess_bwd_pose_kernel(int P, const float* __restrict__ means3D, const int* __restrict__ radii,
                           const float* __restrict__ scales, const float* __restrict__ rotations, const CamParams cam,
                           const float4* __restrict__ dense_T, const float2* __restrict__ dense_m2d,
                           const float* __restrict__ pose_Rt, const float* __restrict__ pose_q, float* __restrict__ dL_dpose)
{
    const float* pm = cam.pm;
    const float halfW = (float)cam.W * 0.5f, halfWm = (float)(cam.W - 1) * 0.5f;
    const float halfH = (float)cam.H * 0.5f, halfHm = (float)(cam.H - 1) * 0.5f;
    float Pm[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        Pm[a][0] = pm[4 * a] * halfW + pm[4 * a + 3] * halfWm;
        Pm[a][1] = pm[4 * a + 1] * halfH + pm[4 * a + 3] * halfHm;
        Pm[a][2] = pm[4 * a + 3];
    }
    float pg[12];
#pragma unroll
    for (int i = 0; i < 12; i++) pg[i] = 0.f;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < P; idx += gridDim.x * 256) {
        if (!(radii[idx] > 0)) continue;
        const float4 gT = dense_T[idx];
        const float2 gm = dense_m2d[idx];
        float d2 = gT.x, d5 = gT.y, d8 = gT.z;
        const float wx = means3D[3 * idx], wy = means3D[3 * idx + 1], wz = means3D[3 * idx + 2];
        if (gm.x != 0 || gm.y != 0) {  // backward.cu:538-563 (rare: a pixel took the low-pass branch of this splat)
            float px = wx, py = wy, pz = wz;
            pose_point(pose_Rt, px, py, pz);
            float qcam[4], qsign, w, x, y, z, T[9], normal[3];
            load_pose_quat(pose_Rt, pose_q, qcam);
            const float4 q = pose_quat(qcam, reinterpret_cast<const float4*>(rotations)[idx], qsign);
            const float2 sc = reinterpret_cast<const float2*>(scales)[idx];
            const Mat3 R = quat_to_R(q, w, x, y, z);
            compute_transmat(px, py, pz, sc.x, sc.y, R, pm, cam.vm, cam.W, cam.H, T, normal);
            const float distance = T[6] * T[6] + T[7] * T[7] - T[8] * T[8];
            const float f = 1 / distance;
            d2 += gm.x * (-f * T[8]);
            d5 += gm.y * (-f * T[8]);
            d8 += gm.x * (-T[2] * (f + 2 * f * f * T[8] * T[8])) + gm.y * (-T[5] * (f + 2 * f * f * T[8] * T[8]));
        }
        const float g0 = (Pm[0][0] * d2 + Pm[0][1] * d5) + Pm[0][2] * d8;
        const float g1 = (Pm[1][0] * d2 + Pm[1][1] * d5) + Pm[1][2] * d8;
        const float g2 = (Pm[2][0] * d2 + Pm[2][1] * d5) + Pm[2][2] * d8;
        pg[0] += g0 * wx; pg[1] += g0 * wy; pg[2] += g0 * wz; pg[3] += g0;
        pg[4] += g1 * wx; pg[5] += g1 * wy; pg[6] += g1 * wz; pg[7] += g1;
        pg[8] += g2 * wx; pg[9] += g2 * wy; pg[10] += g2 * wz; pg[11] += g2;
    }
    __shared__ float red[4][12];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        float v = pg[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (v != 0.f) atomicAdd(dL_dpose + threadIdx.x, v);
    }
}

// Batched form (gs2d_backward_batch): blockIdx.y = frame.  No pose, no deterministic variant.
__global__ void __launch_bounds__(256)
preprocess_bwd_batch_kernel(int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ shs,
                            const float* __restrict__ scales, const float* __restrict__ rotations, int use_rec,
                            const gs2d::PreBwdFrames tab)
{
    const gs2d::PreBwdFrame& f = tab.f[blockIdx.y];
    CamParams cam;
    cam.vm = f.vm; cam.pm = f.pm; cam.campos = f.campos; cam.W = f.W; cam.H = f.H; cam.gx = 0; cam.gy = 0; cam.tight = 0;
    float pg[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // (pose sums: unused without a pose)
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < P)
        preprocess_bwd_one(idx, P, D, M, means3D, use_rec ? f.rec : nullptr, f.radii, shs, f.clamped, scales, rotations, cam, f.grad_rec,
                           f.dL_dtransMat, f.dL_dnormal, f.dL_dcolor, f.dL_dopacity, f.dL_dsh, f.dL_dmean2D, f.dL_dmean3D, f.dL_dscale,
                           f.dL_drot, nullptr, nullptr, pg);
}

// gs2d_backward_batch(accumulate = 1): frame 0's PARAMETER gradients += frame 1's + ... + frame K-1's, added in frame order (the
// sums K separate backwards followed by tensor additions produce, bit for bit) -- one launch instead of 5 (K - 1) elementwise kernels
__global__ void __launch_bounds__(256) sum_frames_kernel(int P, int K, int M, const gs2d::PreBwdFrames tab)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= P) return;
#define GS2D_SUM_FIELD(F, N)                                                                                   \
    if (tab.f[0].F != nullptr) {                                                                                \
        float v[N];                                                                                             \
        _Pragma("unroll") for (int i = 0; i < N; i++) v[i] = tab.f[0].F[(size_t)g * N + i];                     \
        for (int k = 1; k < K; k++) {                                                                           \
            const float* __restrict__ sp = tab.f[k].F;                                                          \
            _Pragma("unroll") for (int i = 0; i < N; i++) v[i] += sp[(size_t)g * N + i];                        \
        }                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < N; i++) tab.f[0].F[(size_t)g * N + i] = v[i];                     \
    }
    // (not dL_dmean2D: the screen-space gradient is a per-VIEW quantity -- the reference accumulates its norm view by view for
    // densification, scene/Gaussians.py:58-62 -- so every frame keeps its own, frame 0 included)
    GS2D_SUM_FIELD(dL_dmean3D, 3)
    GS2D_SUM_FIELD(dL_dcolor, 3)
    GS2D_SUM_FIELD(dL_dopacity, 1)
    GS2D_SUM_FIELD(dL_dscale, 2)
    GS2D_SUM_FIELD(dL_drot, 4)
    GS2D_SUM_FIELD(dL_dnormal, 3)
    GS2D_SUM_FIELD(dL_dtransMat, 9)
#undef GS2D_SUM_FIELD
    if (tab.f[0].dL_dsh != nullptr && M > 0) {
        for (int i = 0; i < 3 * M; i++) {
            float v = tab.f[0].dL_dsh[(size_t)g * 3 * M + i];
            for (int k = 1; k < K; k++) v += tab.f[k].dL_dsh[(size_t)g * 3 * M + i];
            tab.f[0].dL_dsh[(size_t)g * 3 * M + i] = v;
        }
    }
}

// Deterministic pose gradient: dL_dpose[c] += sum over the workgroups' partials, always in the same order (lane l adds
// partials l, l + 64, ... one after the other, then a fixed butterfly over the 64 lanes).
__global__ void __launch_bounds__(64) pose_reduce_kernel(int n, const float* __restrict__ partials, float* __restrict__ dL_dpose)
{
    for (int c = 0; c < 12; c++) {
        float v = 0.f;
        for (int i = threadIdx.x; i < n; i += 64) v += partials[i * 12 + c];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if (threadIdx.x == 0) dL_dpose[c] += v;
    }
}


// rasterizer_impl.cu:54-66
__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ vm,
                                    uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float px = means3D[3 * idx], py = means3D[3 * idx + 1], pz = means3D[3 * idx + 2];
    const float pvz = ((vm[2] * px + vm[6] * py) + vm[10] * pz) + vm[14];
    present[idx] = pvz > 0.2f;
}

}  // namespace

namespace gs2d {

void launch_preprocess_fwd(int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                           const float* rotations, const float* opacities, const float* shs,
                           const float* transMat_precomp, const float* colors_precomp, const CamParams& cam,
                           int* radii, float* depths, float4* rec, uint32_t* tiles_touched, ushort4* rect, uint8_t* clamped,
                           const float* pose_Rt, const float* pose_q, uint32_t* block_sums, hipStream_t s)
{
    hipLaunchKernelGGL(preprocess_fwd_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, D, M, means3D, scales,
                       scale_modifier, rotations, opacities, shs, transMat_precomp, colors_precomp, cam, radii, depths,
                       rec, tiles_touched, rect, clamped, pose_Rt, pose_q, block_sums);
}

void launch_preprocess_fwd_batch(int P, int K, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                                 const float* rotations, const float* opacities, const float* shs, const float* transMat_precomp,
                                 const float* colors_precomp, const CamParams& cam0, const PreFwdFrames& tab, hipStream_t s)
{
    hipLaunchKernelGGL(preprocess_fwd_batch_kernel, dim3((P + 255) / 256, K), dim3(256), 0, s, P, D, M, means3D, scales, scale_modifier,
                       rotations, opacities, shs, transMat_precomp, colors_precomp, cam0, tab);
}

void launch_preprocess_bwd_batch(int P, int K, int D, int M, const float* means3D, const float* shs, const float* scales,
                                 const float* rotations, int need_record, const PreBwdFrames& tab, hipStream_t s)
{
    if (P <= 0) return;
    // (rec is read only when Tw.z cannot be recomputed: precomputed transforms or scale_modifier != 1, see launch_preprocess_bwd)
    const int use_rec = (scales == nullptr || need_record) ? 1 : 0;
    hipLaunchKernelGGL(preprocess_bwd_batch_kernel, dim3((P + 255) / 256, K), dim3(256), 0, s, P, D, M, means3D, shs, scales, rotations,
                       use_rec, tab);
}

void launch_sum_frames(int P, int K, int M, const PreBwdFrames& tab, hipStream_t s)
{
    if (P <= 0 || K <= 1) return;
    hipLaunchKernelGGL(sum_frames_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, K, M, tab);
}

void launch_preprocess_bwd(int first, int P, int D, int M, const float* means3D, const float4* rec, const int* radii,
                           const float* shs, const uint8_t* clamped, const float* scales, const float* rotations,
                           const CamParams& cam, const float* grad_rec, float* dL_dtransMat, float* dL_dnormal,
                           float* dL_dcolor, float* dL_dopacity, float* dL_dsh, float* dL_dmean2D,
                           float* dL_dmean3D, float* dL_dscale, float* dL_drot, const float* pose_Rt, const float* pose_q,
                           float* dL_dpose, int need_record, float* pose_partials, hipStream_t s)
{
    if (P <= first) return;
    // The per-Gaussian code needs the forward's Tw.z (record word 10).  With scales / rotations and scale_modifier == 1 it
    // recomputes the whole transform from the same inputs with the same code (-ffp-contract=off: same IEEE operations), Tw.z
    // included (it does not depend on the image size the backward re-derives), so the records stay unread: -25 % traffic.
    if (scales != nullptr && !need_record) rec = nullptr;
    int grid = (P - first + 255) / 256;
    // every workgroup ends with 12 atomics on the same pose-gradient words (1954 workgroups at 500k Gaussians: +14 us);
    // two Gaussians per thread halve that without starving the memory system of waves
#ifndef GS2D_POSE_GPT
#define GS2D_POSE_GPT 2  // Gaussians per thread with a pose
#endif
    if (dL_dpose != nullptr && grid > 1024) grid = max(1024 / (GS2D_POSE_GPT / 2), (grid + GS2D_POSE_GPT - 1) / GS2D_POSE_GPT);
    hipLaunchKernelGGL(preprocess_bwd_kernel, dim3(grid), dim3(256), 0, s, first, P, D, M, means3D, rec, radii, shs,
                       clamped, scales, rotations, cam, grad_rec, dL_dtransMat, dL_dnormal, dL_dcolor, dL_dopacity,
                       dL_dsh, dL_dmean2D, dL_dmean3D, dL_dscale, dL_drot, pose_Rt, pose_q, dL_dpose,
                       dL_dpose != nullptr ? pose_partials : nullptr);
    if (dL_dpose != nullptr && pose_partials != nullptr)
        hipLaunchKernelGGL(pose_reduce_kernel, dim3(1), dim3(64), 0, s, grid, pose_partials, dL_dpose);
}

void launch_preprocess_bwd_pose(int P, const float* means3D, const int* radii, const float* scales, const float* rotations,
                                const CamParams& cam, const float* dense_T, const float* dense_m2d, const float* pose_Rt,
                                const float* pose_q, float* dL_dpose, hipStream_t s)
{
    if (P <= 0) return;
    // a streaming kernel of 40 B per Gaussian: enough workgroups to fill the chip, few enough that the 12 atomics each of them
    // ends with do not queue up on the pose-gradient words
    int grid = (P + 255) / 256;
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(preprocess_bwd_pose_kernel, dim3(grid), dim3(256), 0, s, P, means3D, radii, scales, rotations, cam,
                       reinterpret_cast<const float4*>(dense_T), reinterpret_cast<const float2*>(dense_m2d), pose_Rt, pose_q, dL_dpose);
}

void launch_pose_quat(const float* pose_Rt, float* q_out, hipStream_t s)
{
    hipLaunchKernelGGL(pose_quat_kernel, dim3(1), dim3(64), 0, s, pose_Rt, q_out);
}

void launch_mark_visible(int P, const float* means3D, const float* vm, uint8_t* present, hipStream_t s)
{
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, vm, present);
}

}  // namespace gs2d
