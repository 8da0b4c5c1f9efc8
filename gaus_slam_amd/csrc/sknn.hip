// simple_knn.distCUDA2 for gfx950: mean squared distance of every point to its 3 nearest other points.
//
// The reference imports this from the third-party `simple-knn` submodule (scene/Gaussians.py:8, call sites
// :77 and :218), whose source is NOT vendored in the mounted tree (.gitmodules:1-3).  Semantics restated from
// its published behaviour: exact 3-NN over all other points (self excluded by index), best[] initialised to
// FLT_MAX, result (b0+b1+b2)/3.  Algorithm here: 30-bit Morton codes -> radix sort (the rasterizer's own
// wave64 sort) -> 64-point boxes with bounds (plus bounds of their runs of 8 and of groups of 64 boxes) -> one WAVE per box
// of queries (four independent waves per workgroup, no barrier): the wave scans its own box and its two neighbours on the
// curve, then walks the groups of 64 boxes outwards along the curve; a group whose AABB is beyond every run's 3-NN radius
// is skipped whole, otherwise 64 boxes are tested at once (one per lane, against the 8 runs of queries), each wanted
// candidate's 64 points are staged in wave-private LDS with one coalesced load and every lane scans them as LDS broadcasts
// behind its own pruning
// test.  The result is the exact k-NN set, so it equals a brute-force evaluation bit for bit (squared distances use
// the same expression order; the three smallest distances do not depend on the evaluation order).
#include "gs2d_common.h"
#include "../../include/gs2d_rasterizer.h"

#include <float.h>

namespace {

constexpr int BOX = 64;  // points per box = queries per wave

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}

// bounds[0..2] = min xyz, bounds[3..5] = max xyz. One workgroup per `chunk` of points, partials combined by atomics
// on order-preserving integer encodings.
__device__ __forceinline__ uint32_t enc(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec(uint32_t e)
{
    const uint32_t u = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
    return __uint_as_float(u);
}

__global__ void __launch_bounds__(256) bounds_init_kernel(uint32_t* b)
{
    if (threadIdx.x < 3) b[threadIdx.x] = 0xffffffffu;
    else if (threadIdx.x < 6) b[threadIdx.x] = 0u;
}

__global__ void __launch_bounds__(256) bounds_kernel(int N, const float* __restrict__ pts, uint32_t* __restrict__ b)
{
    __shared__ float red[4][6];
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float v = pts[3 * (size_t)i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lo = wave_min(mn[a]), hi = wave_max(mx[a]);
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][a] = lo; red[threadIdx.x >> 6][3 + a] = hi; }
    }
    __syncthreads();
    // six atomics per WORKGROUP on the six result words (one per wave saturated them: ~90 atomics/us on one address)
    if (threadIdx.x < 3)
        atomicMin(&b[threadIdx.x], enc(fminf(fminf(red[0][threadIdx.x], red[1][threadIdx.x]), fminf(red[2][threadIdx.x], red[3][threadIdx.x]))));
    else if (threadIdx.x < 6)
        atomicMax(&b[threadIdx.x], enc(fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]), fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]))));
}

__device__ __forceinline__ uint32_t spread10(uint32_t x)
{
    x &= 0x3ffu;
    x = (x | (x << 16)) & 0x030000ffu;
    x = (x | (x << 8)) & 0x0300f00fu;
    x = (x | (x << 4)) & 0x030c30c3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ void __launch_bounds__(256)
morton_kernel(int N, const float* __restrict__ pts, const uint32_t* __restrict__ b, uint64_t* __restrict__ keys,
              uint32_t* __restrict__ vals)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    uint32_t code = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lo = dec(b[a]), hi = dec(b[3 + a]);
        const float ext = hi - lo;
        float t = ext > 0.f ? (pts[3 * (size_t)i + a] - lo) / ext : 0.f;
        t = fminf(fmaxf(t, 0.f), 1.f);
        const uint32_t q = (uint32_t)(t * 1023.f);
        code |= spread10(q) << a;
    }
    keys[i] = code;
    vals[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(256)
gather_kernel(int N, const float* __restrict__ pts, const uint32_t* __restrict__ order, float4* __restrict__ sorted)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint32_t o = order[i];
    sorted[i] = make_float4(pts[3 * (size_t)o], pts[3 * (size_t)o + 1], pts[3 * (size_t)o + 2], 0.f);
}

// one wave per 64-point box, four boxes per workgroup.  boxes[b] = the box's AABB; sub[b][s] = AABB of its s-th run of 8
// consecutive points: 64 Morton-consecutive points sometimes straddle a jump of the curve and then have a huge AABB that
// prunes nothing, their runs of 8 are (almost) always compact.
constexpr int SUB = 8;
__global__ void __launch_bounds__(256)
box_bounds_kernel(int N, const float4* __restrict__ sorted, float* __restrict__ boxes, float* __restrict__ sub, int nboxes)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= nboxes) return;
    const int i = b * BOX + lane;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (i < N) {
        const float4 p = sorted[i];
        mn[0] = p.x; mn[1] = p.y; mn[2] = p.z; mx[0] = p.x; mx[1] = p.y; mx[2] = p.z;
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float lo = mn[a], hi = mx[a];
#pragma unroll
        for (int d = 1; d < SUB; d <<= 1) { lo = fminf(lo, __shfl_xor(lo, d, 64)); hi = fmaxf(hi, __shfl_xor(hi, d, 64)); }
        if ((lane & (SUB - 1)) == 0) {
            float* sb = sub + ((size_t)b * (BOX / SUB) + lane / SUB) * 6;
            sb[a] = lo; sb[3 + a] = hi;  // an empty run keeps (+max, -max): its distance test is never passed
        }
        lo = wave_min(mn[a]); hi = wave_max(mx[a]);
        if (lane == 0) { boxes[6 * b + a] = lo; boxes[6 * b + 3 + a] = hi; }
    }
}

// super[g] = AABB of the 64 boxes of group g (one wave per group)
__global__ void __launch_bounds__(64) super_bounds_kernel(const float* __restrict__ boxes, float* __restrict__ super, int nboxes)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lo = wave_min(b < nboxes ? boxes[6 * b + a] : FLT_MAX);
        const float hi = wave_max(b < nboxes ? boxes[6 * b + 3 + a] : -FLT_MAX);
        if (threadIdx.x == 0) { super[6 * blockIdx.x + a] = lo; super[6 * blockIdx.x + 3 + a] = hi; }
    }
}

__device__ __forceinline__ void update3(const float4 ref, const float4 p, float& b0, float& b1, float& b2)
{
    const float dx = p.x - ref.x, dy = p.y - ref.y, dz = p.z - ref.z;
    const float d = (dx * dx + dy * dy) + dz * dz;
    if (d < b2) {
        if (d < b1) {
            b2 = b1;
            if (d < b0) { b1 = b0; b0 = d; } else b1 = d;
        } else b2 = d;
    }
}

__device__ __forceinline__ float box_dist2(const float4 p, const float* bx)
{
    float dist = 0.f;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float v = a == 0 ? p.x : (a == 1 ? p.y : p.z);
        const float d = v < bx[a] ? bx[a] - v : (v > bx[3 + a] ? v - bx[3 + a] : 0.f);
        dist += d * d;
    }
    return dist;
}

// One wave = the 64 consecutive (Morton-sorted) points of one box as queries; waves are independent.
__global__ void __launch_bounds__(256)
knn_kernel(int N, const float4* __restrict__ sorted, const uint32_t* __restrict__ order, const float* __restrict__ boxes,
           const float* __restrict__ sub, const float* __restrict__ super, int nboxes, float* __restrict__ out)
{
    __shared__ float4 stage[4][BOX];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qb = blockIdx.x * 4 + wave;
    if (qb >= nboxes) return;
    float4* pts = stage[wave];
    const int i = qb * BOX + lane;
    const bool valid = i < N;
    const float inf = __builtin_inff();
    const float4 far = make_float4(inf, inf, inf, 0.f);  // padding point: distance +inf, never accepted
    const float4 ref = valid ? sorted[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    // wave-private LDS: operations of one wave execute in order; the fences only stop compiler reordering
#define WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                             __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
    // the query box itself
    pts[lane] = valid ? ref : far;
    WAVE_LDS_SYNC();
    for (int j = 0; j < BOX; j++)
        if (j != lane) update3(ref, pts[j], b0, b1, b2);
    // The two Morton neighbours of the query box first, unpruned: a box that straddles a jump of the curve leaves the few
    // queries on one side of the jump without three neighbours in their own box (3-NN radius = the length of the jump),
    // their true neighbours precede / follow them on the curve.  Without this, one such wave walked 1700 candidates.
    for (int nb = qb - 1; nb <= qb + 1; nb += 2) {
        if (nb < 0 || nb >= nboxes) continue;
        const int j = nb * BOX + lane;
        WAVE_LDS_SYNC();
        pts[lane] = j < N ? sorted[j] : far;
        WAVE_LDS_SYNC();
        if (valid)
            for (int t = 0; t < BOX; t++) update3(ref, pts[t], b0, b1, b2);
    }
    // Candidate boxes are selected per RUN of 8 queries, not for the whole wave: a straddling box has a huge AABB whose gap
    // to almost every other box is 0, its runs of 8 are compact.  qsub[r] = (AABB of run r, largest 3-NN radius among its
    // queries); the radius is refreshed after every group of 64 candidate boxes that contributed points.
    __shared__ float qsub_all[4][BOX / SUB][8];
    float (*qsub)[8] = qsub_all[wave];
    if ((lane & (SUB - 1)) == 0) {
        const float* sb = sub + ((size_t)qb * (BOX / SUB) + lane / SUB) * 6;
#pragma unroll
        for (int a = 0; a < 6; a++) qsub[lane / SUB][a] = sb[a];
    }
    const int ngroups = (nboxes + 63) >> 6, g0 = qb >> 6;
    bool refresh = true;
    // groups of 64 boxes, outwards from the query's own group along the curve: the near ones shrink the radii first
    for (int k = 0; k < 2 * ngroups; k++) {
        const int g = __builtin_amdgcn_readfirstlane((k & 1) ? g0 + ((k + 1) >> 1) : g0 - (k >> 1));
        if (g < 0 || g >= ngroups) continue;
        if (refresh) {
            float rmax = valid ? b2 : 0.f;
#pragma unroll
            for (int d = 1; d < SUB; d <<= 1) rmax = fmaxf(rmax, __shfl_xor(rmax, d, 64));
            WAVE_LDS_SYNC();
            if ((lane & (SUB - 1)) == 0) qsub[lane / SUB][6] = rmax;
            WAVE_LDS_SYNC();
            refresh = false;
        }
        {   // the whole group at once: lane r tests run r against the group's AABB
            const float* gx = super + 6 * g;
            const int r = lane & (BOX / SUB - 1);
            const float d0 = fmaxf(0.f, fmaxf(gx[0] - qsub[r][3], qsub[r][0] - gx[3]));
            const float d1 = fmaxf(0.f, fmaxf(gx[1] - qsub[r][4], qsub[r][1] - gx[4]));
            const float d2 = fmaxf(0.f, fmaxf(gx[2] - qsub[r][5], qsub[r][2] - gx[5]));
            const float dist = (d0 * d0 + d1 * d1) + d2 * d2;
            if (__ballot(!(dist * 0.999999f > qsub[r][6])) == 0) continue;
        }
        const int base = g << 6;
        const int c = base + lane;
        bool want = false;
        if (c < nboxes && (c < qb - 1 || c > qb + 1)) {
            const float* bx = boxes + 6 * c;
            const float c0 = bx[0], c1 = bx[1], c2 = bx[2], C0 = bx[3], C1 = bx[4], C2 = bx[5];
#pragma unroll
            for (int r = 0; r < BOX / SUB; r++) {
                const float d0 = fmaxf(0.f, fmaxf(c0 - qsub[r][3], qsub[r][0] - C0));
                const float d1 = fmaxf(0.f, fmaxf(c1 - qsub[r][4], qsub[r][1] - C1));
                const float d2 = fmaxf(0.f, fmaxf(c2 - qsub[r][5], qsub[r][2] - C2));
                const float dist = (d0 * d0 + d1 * d1) + d2 * d2;  // squared gap between run r and the candidate box
                // conservative pruning; 0.999999f guards the different rounding of box distances vs point distances
                // (an empty run has an inverted AABB: infinite gap, never wanted)
                want = want || !(dist * 0.999999f > qsub[r][6]);
            }
        }
        uint64_t todo = __ballot(want);
        refresh = todo != 0;
        // the next candidate's points are fetched while the current one is scanned
        int cb = todo ? base + __builtin_ctzll(todo) : 0;
        float4 nextp = far;
        if (todo) { const int j = cb * BOX + lane; if (j < N) nextp = sorted[j]; }
        while (todo) {
            todo &= todo - 1;
            const int cur = cb;
            WAVE_LDS_SYNC();  // previous candidate fully scanned before pts is overwritten
            pts[lane] = nextp;
            WAVE_LDS_SYNC();
            if (todo) {
                cb = base + __builtin_ctzll(todo);
                const int j = cb * BOX + lane;
                nextp = j < N ? sorted[j] : far;
            }
            if (valid && !(box_dist2(ref, boxes + 6 * cur) * 0.999999f > b2)) {
                const float* sb = sub + (size_t)cur * (BOX / SUB) * 6;
                for (int r = 0; r < BOX / SUB; r++) {  // run by run: a straddling box is pruned piecewise
                    if (box_dist2(ref, sb + 6 * r) * 0.999999f > b2) continue;
#pragma unroll
                    for (int t = 0; t < SUB; t++) update3(ref, pts[r * SUB + t], b0, b1, b2);
                }
            }
        }
    }
#undef WAVE_LDS_SYNC
    if (valid) out[order[i]] = ((b0 + b1) + b2) / 3.0f;
}

}  // namespace

extern "C" int sknn_dist2(int N, const float* points, float* out, gs2d_alloc_fn ws_alloc, void* ws_user, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (N <= 0) return 0;
    if (!ws_alloc) return -1;
    const int nboxes = (N + BOX - 1) / BOX;
    const BinLayout BL = bin_layout(N);
    size_t o = BL.total;
    const size_t off_sorted = o; o = gs2d_align_up(o + sizeof(float4) * (size_t)N, 256);
    const size_t off_boxes = o; o = gs2d_align_up(o + sizeof(float) * 6 * (size_t)nboxes, 256);
    const size_t off_sub = o; o = gs2d_align_up(o + sizeof(float) * 6 * (BOX / SUB) * (size_t)nboxes, 256);
    const int ngroups = (nboxes + 63) / 64;
    const size_t off_super = o; o = gs2d_align_up(o + sizeof(float) * 6 * (size_t)ngroups, 256);
    const size_t off_bounds = o; o = gs2d_align_up(o + 64, 256);
    char* ws = (char*)ws_alloc(ws_user, o);
    if (!ws) return -1;
    uint32_t* order = (uint32_t*)(ws + BL.point_list);
    uint64_t* keys = (uint64_t*)(ws + BL.keys);
    uint32_t* vals_alt = (uint32_t*)(ws + BL.vals_alt);
    uint64_t* keys_alt = (uint64_t*)(ws + BL.keys_alt);
    uint32_t* hist = (uint32_t*)(ws + BL.hist);
    float4* sorted = (float4*)(ws + off_sorted);
    float* boxes = (float*)(ws + off_boxes);
    float* sub = (float*)(ws + off_sub);
    float* super = (float*)(ws + off_super);
    uint32_t* bounds = (uint32_t*)(ws + off_bounds);

    hipLaunchKernelGGL(bounds_init_kernel, dim3(1), dim3(256), 0, s, bounds);
    const int rb = min(256, (N + 255) / 256);
    hipLaunchKernelGGL(bounds_kernel, dim3(rb), dim3(256), 0, s, N, points, bounds);
    const int end_bit = 30, passes = (end_bit + 7) / 8;
    uint64_t* k_unsorted = (passes & 1) ? keys_alt : keys;
    uint32_t* v_unsorted = (passes & 1) ? vals_alt : order;
    hipLaunchKernelGGL(morton_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, points, bounds, k_unsorted, v_unsorted);
    gs2d::launch_sort_pairs(N, keys, order, keys_alt, vals_alt, 0, end_bit, hist, BL.hist_elems, s);
    hipLaunchKernelGGL(gather_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, points, order, sorted);
    hipLaunchKernelGGL(box_bounds_kernel, dim3((nboxes + 3) / 4), dim3(256), 0, s, N, sorted, boxes, sub, nboxes);
    hipLaunchKernelGGL(super_bounds_kernel, dim3(ngroups), dim3(64), 0, s, boxes, super, nboxes);
    hipLaunchKernelGGL(knn_kernel, dim3((nboxes + 3) / 4), dim3(256), 0, s, N, sorted, order, boxes, sub, super, nboxes, out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
