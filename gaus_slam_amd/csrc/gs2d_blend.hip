// Per-tile alpha compositing, forward and backward (the hot kernels).
//
// Semantics: RAST/cuda_rasterizer/forward.cu:258-467 and backward.cu:143-463 of the reference.
// Design (gfx950, wave64):
//   * one workgroup per 16x16 tile, 4 independent waves (no workgroup barrier in the blend loops), each wave owns an 8x8
//     pixel quadrant; a forward wave splits it into sixteen 2x2 pixel GROUPS (one per DPP quad), a backward wave into four
//     4x4 sub-blocks (one per 16-lane DPP row) -- finer groups mean fewer loop trips, but every (group, splat) pair of the
//     backward ends in 13 LDS float atomics, which sixteen groups would serialise (measured: 0.59 vs 0.31 ms);
//   * which splats can touch which group is decided once per instance, in phase 0 of the forward kernel (the four
//     waves cull their tile's list together, gs2d_cull.h; one workgroup barrier, then they part ways); both kernels
//     read those bits, fetch only the touching splats' packed records (one lane per record), stage them compacted in
//     wave-private LDS in batches and give every group its own depth-ordered queue of slot numbers (a byte list in
//     LDS): each loop trip the groups composite different splats, records read back as per-group LDS broadcasts,
//     prefetched one trip ahead;
//   * backward: per-(pixel,splat) gradients are summed over the 16 lanes of a row with a 16-value DPP butterfly and added
//     into per-splat accumulators in wave-private LDS -- a plain read / add / store on trips whose four rows hold four
//     different splats, ds_add_f32 only when two rows meet on a splat -- and every touched splat of a batch is flushed to
//     its Gaussian's gradient record once per (quadrant, batch), one global float atomic per component (the reference
//     issues 16-18 scalar global atomics per (pixel, splat)).
// Arithmetic follows the oracle's expression order; the file is compiled with -ffp-contract=off so the
// per-pixel recurrences reproduce the CPU oracle up to the ulp-level difference of v_exp_f32 / v_rcp_f32.
#include "gs2d_common.h"
#ifndef GS2D_BWD_GROUPS
#define GS2D_BWD_GROUPS 4       // queues per wave of the full backward: 4 (one per 4x4 sub-block = DPP row) or 8 (one per 4x2 half-row)
#endif
#ifndef GS2D_BWD_POSE_GROUPS
#define GS2D_BWD_POSE_GROUPS 8  // ... of the pose-only backward (its accumulators are per row: no LDS atomics to eat the trips
                                // eight queues save; it derives its half-row cull bits from the forward's group bits itself)
#endif
// eight queues in the FULL backward (experiment builds): the forward's phase 0 stores eight half-row cull bits per (instance,
// quadrant) instead of four row bits
#define GS2D_HALFROW_BITS (GS2D_BWD_GROUPS == 8)
#ifndef GS2D_BWD_LANE_CLASH
#define GS2D_BWD_LANE_CLASH 1   // four queues, full backward: the LDS-atomic fallback of the accumulate is decided per LANE (a flag in
                                // bit 6 of the queue entry) instead of per trip: 0.2385 -> 0.2331 ms (profiles/bwd_queue_ab_r04.txt);
                                // the pose-only kernel keeps the per-trip test (per lane measured 3 % slower there)
#endif
#include "gs2d_cull.h"
#include "gs2d_tile_sort.h"
#include "gs2d_blend_dev.h"  // dev-only probes (wave profile, ingredient pricing): all pass-through in the product build

#include <type_traits>

namespace {

// ------------------------------------------------------------------------------------------- helpers
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }          // v_rcp_f32, 1 ulp
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }         // v_sqrt_f32, 1 ulp
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }  // v_exp_f32
// exp(-0.5 r) and exp(-q/4) with the power-of-two factors folded into the log2(e) constant: scaling by 2^k commutes
// with rounding, so these return exactly fast_exp(-0.5f * r) / fast_exp(-(q * 0.25f)) with one multiply less.
__device__ __forceinline__ float fast_exp_neg_half(float r) { return __builtin_amdgcn_exp2f(r * -0.72134752044448170368f); }
__device__ __forceinline__ float fast_exp_neg_quarter(float q) { return __builtin_amdgcn_exp2f(q * -0.36067376022224085184f); }

// Lane mask of a predicate.  The ballot builtin takes the predicate as a bool, so a test like `ballot64(x) != 0` compiles to a
// scalar compare of the mask the v_cmp already produced; hip's __ballot(int) goes through an integer compare of a materialised
// 0/1 (v_cndmask + v_cmp on the vector unit in every trip of both blend loops).
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Part A of the per-(pixel, splat) work: ray-splat intersection and alpha (forward.cu:360-387; FMA form identical
// to oracle/gs2d_oracle.c).  Branch-free; `ok_m` (a wave mask) folds the reference's skip tests in their original order
// (p.z == 0, depth < near, power > 0, alpha < 1/255).
__device__ __forceinline__ void fwd_eval(const float4 q0, const float4 q1, const float4 q2, float pxf, float pyf,
                                         float& alpha, float& depth, uint64_t& ok_m)
{
    const float k0 = fmaf(pxf, q2.x, -q0.x), k1 = fmaf(pxf, q2.y, -q0.y), k2 = fmaf(pxf, q2.z, -q0.z);
    const float l0 = fmaf(pyf, q2.x, -q1.x), l1 = fmaf(pyf, q2.y, -q1.y), l2 = fmaf(pyf, q2.z, -q1.z);
    const float p0 = fmaf(k1, l2, -(k2 * l1));
    const float p1 = fmaf(k2, l0, -(k0 * l2));
    const float p2 = fmaf(k0, l1, -(k1 * l0));
    const float ip = fast_rcp(p2);
    const float s0 = p0 * ip, s1 = p1 * ip;
    const float rho3d = fmaf(s0, s0, s1 * s1);
    const float d0 = q0.w - pxf, d1 = q1.w - pyf;
    const float rho2d = GS2D_FILTER_INV_SQ * fmaf(d0, d0, d1 * d1);
    const float rho = fminf(rho3d, rho2d);
    depth = (rho3d <= rho2d) ? fmaf(s0, q2.x, fmaf(s1, q2.y, q2.z)) : q2.z;
    alpha = fminf(0.99f, q2.w * fast_exp_neg_half(rho));
    // power = -0.5 rho > 0  <=>  rho < 0
    // (one ballot per comparison, combined on the scalar unit: the ballot of a compound predicate goes through v_cndmask 0/1 +
    // v_cmp_ne, the ballot of a single comparison IS the comparison)
    ok_m = ballot64(!(p2 == 0.0f)) & ballot64(!(depth < GS2D_NEAR_N)) & ballot64(!(rho < 0.0f)) & ballot64(!(alpha < 1.0f / 255.0f));
}

// Wave-private LDS: operations of one wave execute in order; the fence only stops the compiler from reordering them.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef GS2D_WAVES_PER_EU
#define GS2D_WAVES_PER_EU 5  // 96 VGPRs: all 1200 tiles of a 640x480 frame resident at once (5 workgroups per CU)
#endif
#ifndef GS2D_FWD_WAVES_PER_EU
#define GS2D_FWD_WAVES_PER_EU GS2D_WAVES_PER_EU
#endif
// ------------------------------------------------------------------------------------------- queue helpers
__device__ __forceinline__ int row_select(int row8, int j0, int j1, int j2, int j3)
{
    // j0..j3 are wave-uniform (<= 64): pack them into one scalar and let every lane extract its row's byte with a single
    // v_bfe_u32 (row8 = 8 * row) instead of a chain of selects
    const uint32_t packed = (uint32_t)j0 | ((uint32_t)j1 << 8) | ((uint32_t)j2 << 16) | ((uint32_t)j3 << 24);
    return (int)((packed >> row8) & 0xffu);
}
// pops the lowest set bit of a queue; 64 = empty
__device__ __forceinline__ int pop_front(uint64_t& m)
{
    const int j = m ? __builtin_ctzll(m) : 64;
    m &= m - 1;  // no-op on 0
    return j;
}
// pops the highest set bit (back-to-front walks); 64 = empty
__device__ __forceinline__ int pop_back(uint64_t& m)
{
    if (!m) return 64;
    const int j = 63 - __builtin_clzll(m);
    m &= ~(1ull << j);
    return j;
}

// ------------------------------------------------------------------------------------------- forward
// One wave per 8x8 pixel quadrant, 4 independent waves per 16x16 tile; inside the wave each DPP quad (4 lanes) owns
// one 2x2 pixel group.  The tile's depth-sorted list is read in 64-instance chunks; the cull bits of this quadrant
// (phase 0 below, gs2d_cull.h) say which splats touch which group, so only touching splats are fetched and staged,
// compacted, in wave-private LDS until the 64 slots of a BATCH are full.  Each group then gets its own
// depth-ordered QUEUE of slot numbers -- a byte list in LDS, built once per batch with ballot + mbcnt -- and every loop
// trip group g reads the next entry of ITS queue and that splat's record (per-group address, 5 x ds_read_b128,
// software-pipelined one trip ahead; the queue entries two trips ahead), so the sixteen groups composite up to sixteen
// different splats at once: per-pixel order is untouched (a splat that touches several groups sits in several queues), the
// trip count is the LONGEST queue instead of the whole list, the VALU executes only per-pixel math and the scalar unit
// only the loop counter (round 1 popped 64-bit bit-queues with ~65 scalar instructions per trip).
struct FwdBatch {
    float4 qf[GS2D_REC_F4 * 64];  // staged records, SoA by quarter (q(k)[slot])
    float4 dead;                  // = q(4)[64], the last quarter of the end marker's "record": its fourth word (list position + 1) is 0
    uint8_t ql[16][64];           // per-group queues: slot numbers in depth order, 64 = end.  The trip loop indexes records with
                                  // the marker unmasked: an exhausted group reads the next quarter's slot 0 (the last quarter:
                                  // `dead`) -- inside this struct, never used: contributor number 0 says "not live"
    __device__ __forceinline__ float4* q(int k) { return qf + k * 64; }
    uint32_t tail[4];           // the pipeline reads up to two entries past a full queue (values unused)
    uint16_t tm[64];            // group bits of the staged splats
    uint32_t sid[64];           // Gaussian id and list position of the staged splats: a batch's slots are handed out chunk by
    uint32_t spos[64];          // chunk, its records are gathered in ONE round trip afterwards
};

// XCD-aware workgroup -> tile mapping.  Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its own
// L2.  Handing XCD x the x-th contiguous band of tiles (row-major) keeps the tiles a Gaussian touches on ONE XCD in all
// but the band-border cases, so record gathers hit in that L2.
// Purely a placement heuristic: any dispatch order gives the same results.
#define GS2D_XCDS 8
__device__ __forceinline__ int xcd_tile(int block, int ntiles)
{
    const int chunk = (ntiles + GS2D_XCDS - 1) / GS2D_XCDS;
    const int k = block / GS2D_XCDS;
    const int tile = (block % GS2D_XCDS) * chunk + k;
    return tile < ntiles ? tile : -1;  // the last band may be short
}

// Issue priority from the share of the wave's list that is still ahead of it.  The SIMD arbiter serves the oldest wave
// first, so without this the waves of a SIMD finish one after the other and the youngest runs its last third alone, at
// single-wave issue rate and with every latency exposed (scripts/dev/wave_profile.py: the last 40 % of the kernel ran
// with less than half of the waves).  Priority by remaining work makes the laggard the favourite: the waves of a SIMD
// stay together and finish together.  Placement only -- results do not depend on it.
// Level boundaries, SHIFT > 0: geometric, at remaining = total >> (SHIFT * k), k = 1..3 (1: 1/2, 1/4, 1/8) -- only the END
// of the lists is synchronised, and the last level (where the arbiter is back to oldest-first) is short; SHIFT == 0:
// uniform quarters.  Measured on the shipped kernels: the backward wants the geometric levels (0.314 vs 0.320 ms with
// quarters; 1/4, 1/16, 1/64: 0.333), the forward -- whose waves start staggered by their cull phase -- the quarters
// (0.151 vs 0.156 ms); no priorities at all: 0.173 / 0.351 ms.
#ifndef GS2D_FWD_PRIO_SHIFT
#define GS2D_FWD_PRIO_SHIFT 0
#endif
#ifndef GS2D_BWD_PRIO_SHIFT
#define GS2D_BWD_PRIO_SHIFT 1
#endif
#ifndef GS2D_NO_SETPRIO
template <int SHIFT>
__device__ __forceinline__ void prio_by_remaining(uint32_t remaining, uint32_t total)
{
    if (SHIFT == 0) {
        const uint32_t q = remaining * 4u;
        if (q > total * 3u) __builtin_amdgcn_s_setprio(3);
        else if (q > total * 2u) __builtin_amdgcn_s_setprio(2);
        else if (q > total) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    } else {
        if (remaining > (total >> SHIFT)) __builtin_amdgcn_s_setprio(3);
        else if (remaining > (total >> (2 * SHIFT))) __builtin_amdgcn_s_setprio(2);
        else if (remaining > (total >> (3 * SHIFT))) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
}
#else
template <int SHIFT>
__device__ __forceinline__ void prio_by_remaining(uint32_t, uint32_t) {}
#endif

// lane's position among the set bits of a ballot below it
__device__ __forceinline__ int rank_below(uint64_t b)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
}

// Side job of the forward blend: clear the backward's gradient accumulator with this kernel's idle store path (fire-and-
// forget stores as every wave finishes), so that the backward starts without a memset on its critical path.  Every
// thread of the grid takes its share, the padding workgroups too.
__device__ __forceinline__ void clear_share(float4* __restrict__ zero, size_t zero_n, int block, int nblocks)
{
    for (size_t z = (size_t)block * 256 + threadIdx.x; z < zero_n; z += (size_t)nblocks * 256)
        zero[z] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// BATCH: one grid over the tiles of K frames (gs2d_forward_batch): frame = blockIdx.x / blocks_per_frame, the per-frame
// pointers come from a by-value table in the kernel arguments (one scalar load), everything else is shared.  Workgroups
// are dispatched in blockIdx order, so the frames run one after the other and the next frame's tiles fill the SIMDs the
// previous frame's last waves leave idle.  The single-frame instantiation is the same code with the table's first entry.
template <bool USE_SA, bool BATCH>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GS2D_FWD_WAVES_PER_EU, GS2D_FWD_WAVES_PER_EU)))
blend_fwd_kernel(int W, int H, int gx, int ntiles, int blocks_per_frame, const float* __restrict__ bg, size_t plane, size_t zero_n,
                 int sort_cap, int write_keys, const std::conditional_t<BATCH, gs2d::BlendFwdBatch, gs2d::BlendFwdFrame> args)
{
    // ONE dynamic LDS region, used twice: by the depth sort of phase -1 (4 x sort_cap words, or 2 x sort_cap for the index sort, + 4 x 256 digit counters) and then by
    // the four waves' staging batches (launch_blend_fwd sizes it for the larger of the two)
    extern __shared__ uint32_t dyn_lds[];
    FwdBatch* batches = reinterpret_cast<FwdBatch*>(dyn_lds);
    int local_block = blockIdx.x;
    const gs2d::BlendFwdFrame* fa;
    if constexpr (BATCH) {
        const int frame = blockIdx.x / blocks_per_frame;
        local_block = blockIdx.x - frame * blocks_per_frame;
        fa = &args.f[frame];
    } else {
        fa = &args;
    }
    const uint2* __restrict__ ranges = fa->ranges;
    const uint32_t* __restrict__ point_list = fa->point_list;
    const float4* __restrict__ rec = fa->rec;
    float* __restrict__ out_color = fa->out_color;
    float* __restrict__ out_others = fa->out_others;
    float* __restrict__ pix_state = fa->pix_state;
    uint8_t* hits = fa->hits;    /* written by phase 0: neither const nor restrict */
    uint8_t* hits4 = fa->hits4;
    float4* __restrict__ zero = fa->zero;
    uint64_t* sort_keys = fa->keys; uint64_t* sort_keys_alt = fa->keys_alt; uint32_t* sort_vals_alt = fa->vals_alt;
    uint32_t* sort_list = fa->point_list;
    if (fa->dev.base != nullptr) {
        // launched before the host knew num_rendered (DevBin, gs2d_common.h): the count is on the device, the binning chunk's
        // arrays follow from it (scalar loads and a little scalar arithmetic).  A count beyond the chunk's capacity: nothing
        // is touched -- the host sees the same number and runs the stages again, in stream order, in a chunk that fits.
        const uint32_t Rd = *fa->dev.R_dev;
        if (Rd > fa->dev.cap) return;
        const BinLayout BL = bin_layout((int)Rd, fa->dev.det != 0, (int)fa->dev.cap);
        char* bb = fa->dev.base;
        sort_list = (uint32_t*)(bb + BL.point_list); point_list = sort_list;
        hits = (uint8_t*)(bb + BL.hits); hits4 = (uint8_t*)(bb + BL.hits4);
        sort_keys = (uint64_t*)(bb + BL.keys); sort_keys_alt = (uint64_t*)(bb + BL.keys_alt); sort_vals_alt = (uint32_t*)(bb + BL.vals_alt);
    }
    const int tile = xcd_tile(local_block, ntiles);
    if (tile < 0) { clear_share(zero, zero_n, local_block, blocks_per_frame); return; }
    // phase -1 (sort_cap > 0): this tile's list, binned by the counting sort in Gaussian order, is sorted by depth here --
    // the per-tile LDS radix sort that used to be a kernel of its own (22 us + a dependent dispatch at 640x480 / 500k), with
    // the same workgroup-per-tile shape; now tiles that sort overlap with tiles that already blend.  The sorted ids go to
    // point_list in global memory (the backward and the cull phase read them); workgroup scope is enough for the waves of
    // this workgroup to see them (see phase 0).
    if (sort_cap > 0) {
        if (sort_cap > GS2D_FUSED_SORT_CAP)  // (lists of up to 3072 in the same LDS: the index sort, gs2d_tile_sort.h)
            tile_depth_sort_idx_body(tile, dyn_lds, reinterpret_cast<uint32_t (*)[256]>(dyn_lds + 2 * sort_cap), fa->ranges, sort_keys,
                                     sort_list, sort_keys_alt, sort_vals_alt, sort_cap, write_keys);
        else
            tile_depth_sort_body(tile, dyn_lds, reinterpret_cast<uint32_t (*)[256]>(dyn_lds + 4 * sort_cap), fa->ranges, sort_keys,
                                 sort_list, sort_keys_alt, sort_vals_alt, sort_cap, /*packed=*/1, write_keys);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    const int tx = tile % gx, ty = tile / gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    FwdBatch& wb = batches[wave];
    const int qx0 = tx * GS2D_TILE + (wave & 1) * 8, qy0 = ty * GS2D_TILE + (wave >> 1) * 8;
    const int row = lane >> 2, li = lane & 3;                        // DPP quad = 2x2 pixel group, 16 groups per quadrant
    const int lx = (row & 3) * 2 + (li & 1), ly = (row >> 2) * 2 + (li >> 1);  // position inside the quadrant
    const int px = qx0 + lx, py = qy0 + ly;
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const uint16_t* hits16 = reinterpret_cast<const uint16_t*>(hits);  // [instance][quadrant]: 16 group bits
    // phase 0: the sub-block cull bits of this tile's list, by all four waves.  Workgroup scope is enough: the waves of a
    // workgroup share the CU's write-through vector cache, so after the release / barrier / acquire the plain loads
    // below see the stores (an agent-scope fence would write back the XCD's whole L2 from every workgroup)
    cull_tile_list(range, (float)(tx * GS2D_TILE), (float)(ty * GS2D_TILE), point_list, rec,
                   reinterpret_cast<uint64_t*>(hits), reinterpret_cast<uint32_t*>(hits4));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const uint8_t* qrow = wb.ql[row];
    // the bytes behind the queues (the pipeline reads up to two entries past a full queue): wave-private, written once (the
    // LDS was the depth sort's until the barrier above)
    if (lane < 4) wb.tail[lane] = 0x40404040u;
    if (lane == 4) wb.dead = make_float4(0.f, 0.f, 0.f, 0.f);

    float T = 1.0f, C0_ = 0.f, C1_ = 0.f, C2_ = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
    float Dp = 0.f, M1 = 0.f, M2 = 0.f, D2 = 0.f, distortion = 0.f, median_depth = 0.f;
    uint32_t median_contributor = 0;  // the reference keeps a float initialised to -1 and stores (uint) -> 0
    uint32_t last_contributor = 0;
    // "this pixel is finished" (outside the image, or saturated) as a WAVE MASK in a scalar register pair: a loop-carried bool
    // is kept by the compiler as 0/1 in a vector register and turned back into a mask with v_cmp in every trip
    uint64_t done_m = ballot64(!inside);
    GS2D_PROF_BEGIN();

    // Batches are COMPACTED: the list is read in 64-instance chunks, but only the splats whose cull bits touch this
    // quadrant are fetched and staged, and chunks keep being added until the 64 LDS slots are full.  A chunk that does not
    // fit is split: the lanes left over keep their cull bits (carry_tm) and are staged first in the next batch.
    uint32_t next_chunk = range.x;  // absolute index of the next chunk to read
    uint32_t carry_base = 0, carry_tm = 0u, carry_id = 0u;
    bool carry = false;
    // Staging a batch costs ONE exposed memory round trip: the cull bits and Gaussian ids of the next TWO chunks are always in
    // flight (unconditional loads, index clamped to the list), the chunk loop only hands out slots (id, position, bits -> LDS),
    // and when the 64 slots are taken -- or the list ends -- lane l gathers the record of slot l.  (Until round 3 every chunk
    // gathered its own records before the next chunk was looked at: about three dependent gathers per batch, 20 % of a
    // forward wave's lifetime, scripts/dev/wave_profile.py.)
    const uint32_t last_i = range.y > range.x ? range.y - 1u : range.x;
    uint32_t pf_tm = hits16[(size_t)min(next_chunk + lane, last_i) * 4 + wave];
    uint32_t pf_id = point_list[min(next_chunk + lane, last_i)];
    uint32_t pg_tm = hits16[(size_t)min(next_chunk + 64u + lane, last_i) * 4 + wave];
    uint32_t pg_id = point_list[min(next_chunk + 64u + lane, last_i)];
    for (;;) {
        if (~done_m == 0ull) break;
        prio_by_remaining<GS2D_FWD_PRIO_SHIFT>(range.y - min(next_chunk, range.y), range.y - range.x);
        int fill = 0;
        GS2D_PROF_STAGE_BEGIN();
        wave_lds_sync();  // previous batch fully consumed before it is overwritten
        for (;;) {
            uint32_t cbase, tm, id;
            if (carry) { cbase = carry_base; tm = carry_tm; id = carry_id; carry = false; }
            else {
                if (next_chunk >= range.y) break;
                cbase = next_chunk; next_chunk += 64;
                tm = cbase + lane < range.y ? pf_tm : 0u;
                id = pf_id;
                pf_tm = pg_tm; pf_id = pg_id;
                pg_tm = hits16[(size_t)min(next_chunk + 64u + lane, last_i) * 4 + wave];
                pg_id = point_list[min(next_chunk + 64u + lane, last_i)];
            }
            const uint64_t tb = ballot64(tm != 0u);
            const int c = __popcll(tb);
            if (c == 0) continue;
            const int slot = fill + rank_below(tb);
            const bool take = tm != 0u && slot < 64;
            if (take) {
                wb.sid[slot] = id;
                wb.spos[slot] = cbase - range.x + lane + 1u;  // list position + 1 (the contributor number; 0: end marker) rides in the record's free word
                wb.tm[slot] = (uint16_t)tm;
            }
            if (fill + c > 64) { carry_tm = take ? 0u : tm; carry_id = id; carry_base = cbase; carry = true; fill = 64; break; }
            fill += c;
            if (fill == 64) break;
        }
        if (fill == 0) break;  // list exhausted
        wave_lds_sync();
        if (lane < fill) {
            const float4* rp = rec + (size_t)wb.sid[lane] * GS2D_REC_F4;
            const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
            float4 r4 = rp[4];
            r4.w = __uint_as_float(wb.spos[lane]);
            wb.q(0)[lane] = r0; wb.q(1)[lane] = r1; wb.q(2)[lane] = r2; wb.q(3)[lane] = r3; wb.q(4)[lane] = r4;
        }
        if (fill == 0) break;  // list exhausted
        wave_lds_sync();
        // the sixteen group queues: slot numbers of the splats whose bit r is set, in slot (= depth) order, ended by 255
        reinterpret_cast<uint4*>(&wb.ql[0][0])[lane] = make_uint4(0x40404040u, 0x40404040u, 0x40404040u, 0x40404040u);
        // groups whose four pixels are all finished (saturated, or outside the image) take no more splats: their queues stay
        // empty from this batch on, so the batch's trip count is the longest queue among the groups still at work
        // (scripts/dev/group_trips.c: 719k -> 686k trips per frame on the bench scene, where every pixel saturates)
        uint64_t dq = done_m;
        dq &= dq >> 1; dq &= dq >> 2; dq &= 0x1111111111111111ull;            // bit 4g: group g is finished
        dq = (dq | (dq >> 3)) & 0x0303030303030303ull;                         // gather the sixteen bits ...
        dq = (dq | (dq >> 6)) & 0x000F000F000F000Full;
        dq = (dq | (dq >> 12)) & 0x000000FF000000FFull;
        const uint32_t done16 = (uint32_t)(dq | (dq >> 24)) & 0xFFFFu;         // ... bit g: group g is finished
#ifdef GS2D_NO_DONE16  // dev A/B switch (scripts/dev/variants.sh)
        const uint32_t nib = lane < fill ? (uint32_t)wb.tm[lane] : 0u; (void)done16;
#else
        const uint32_t nib = lane < fill ? (uint32_t)wb.tm[lane] & ~done16 : 0u;
#endif
        int trips = 0;  // the longest queue (0: everything staged belongs to finished groups -- the trip loop then runs one idle step)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const bool in_r = ((nib >> r) & 1u) != 0u;
            const uint64_t mr = ballot64(in_r);
            if (in_r) wb.ql[r][rank_below(mr)] = (uint8_t)lane;
            trips = max(trips, (int)__popcll(mr));
        }
        wave_lds_sync();
        GS2D_PROF_STAGE_END();
        // software pipeline: the records of the NEXT trip are fetched from LDS while the current ones are evaluated (queue
        // entries one trip further ahead); unrolled by two with the two register sets swapping roles (no copies).
        int t = 0;
        uint32_t ja = qrow[0], jb = qrow[1];
        float4 a0 = wb.q(0)[ja], a1 = wb.q(1)[ja], a2 = wb.q(2)[ja], a3 = wb.q(3)[ja], a4 = wb.q(4)[ja];
        float4 b0, b1, b2, b3, b4;
#define GS2D_FWD_STEP(C0, C1, C2, C3, C4, N0_, N1_, N2_, N3_, N4_, JN, JNN)                                          \
        {                                                                                                            \
            GS2D_PROF_TRIP();                                                                                        \
            const uint32_t contributor = __float_as_uint(C4.w); /* list position + 1; 0: this group's queue is exhausted (end marker) */ \
            const bool live_ = contributor != 0u;                                                                    \
            N0_ = wb.q(0)[JN]; N1_ = wb.q(1)[JN]; N2_ = wb.q(2)[JN]; N3_ = wb.q(3)[JN];                              \
            N4_ = wb.q(4)[JN];                                                                                       \
            JNN = qrow[t + 2];                                                                                       \
            asm volatile("" : "+v"(JNN)); /* a full 32-bit value from here on: ds_read_u8 zero-extends, no v_and 0xff later */ \
            float alpha, depth;                                                                                      \
            uint64_t ok_m;                                                                                           \
            fwd_eval(C0, C1, C2, pxf, pyf, alpha, depth, ok_m);                                                      \
            const float test_T = T * (1 - alpha);                                                                    \
            const uint64_t pass_m = ok_m & ballot64(live_) & ~done_m;                                                \
            const uint64_t stop_m = pass_m & ballot64(test_T < 0.0001f);                                             \
            done_m |= stop_m;                                                                                        \
            if (__builtin_amdgcn_inverse_ballot_w64(pass_m & ~stop_m)) {                                             \
                const float w = alpha * T;                                                                           \
                const uint64_t front_m = ballot64(T > 0.5f);                                                         \
                if (__builtin_amdgcn_inverse_ballot_w64(front_m)) { median_depth = depth; median_contributor = contributor; } \
                if (USE_SA) { /* forward.cu:405-416 */                                                               \
                    /* While T > 0.5 the median IS this depth: e = 0, conf = 1, the depth stays as it is (bit-exact), so  \
                       those lanes skip the block.  The block itself keeps the oracle's operation order to the bit: the    \
                       running variance is a difference of nearly equal sums -- rounding noise where the blended depths   \
                       agree -- and conf depends on that noise at first order, so an algebraically equal form (one         \
                       reciprocal instead of two was tried) moves mid-magnitude gradients by 1e-3 */                      \
                    if (__builtin_amdgcn_inverse_ballot_w64(ballot64(Dp > 0) & ~front_m)) {                         \
                        const float exp_depth = median_depth;                                                        \
                        float exp_std = fmaf(fmaf(-2.0f * Dp, exp_depth, D2), fast_rcp(1 - T), exp_depth * exp_depth); \
                        exp_std = fmaxf(exp_std, 1e-7f);                                                             \
                        const float e = exp_depth - depth;                                                           \
                        const float conf = fast_exp_neg_quarter((e * e) * fast_rcp(exp_std));                        \
                        depth = fmaf(conf, depth, (1 - conf) * exp_depth);                                           \
                    }                                                                                                \
                    Dp = fmaf(depth, w, Dp);                                                                         \
                    D2 = fmaf(depth * depth, w, D2);                                                                 \
                } else { /* forward.cu:417-423 */                                                                    \
                    const float A = 1 - T;                                                                           \
                    const float m = (GS2D_FAR_N / (GS2D_FAR_N - GS2D_NEAR_N)) * (1 - GS2D_NEAR_N * fast_rcp(depth)); \
                    distortion = fmaf(fmaf(m * m, A, fmaf(-2.0f * m, M1, M2)), w, distortion);                       \
                    Dp = fmaf(depth, w, Dp);                                                                         \
                    M1 = fmaf(m, w, M1);                                                                             \
                    M2 = fmaf(m * m, w, M2);                                                                         \
                }                                                                                                    \
                N0 = fmaf(C3.x, w, N0); N1 = fmaf(C3.y, w, N1); N2 = fmaf(C3.z, w, N2);                              \
                C0_ = fmaf(C3.w, w, C0_); C1_ = fmaf(C4.x, w, C1_); C2_ = fmaf(C4.y, w, C2_);                        \
                T = test_T;                                                                                          \
                last_contributor = contributor;                                                                      \
            }                                                                                                        \
            if (++t >= trips || ~done_m == 0ull) break;                                                              \
        }
        for (;;) {
            GS2D_FWD_STEP(a0, a1, a2, a3, a4, b0, b1, b2, b3, b4, jb, ja)
            GS2D_FWD_STEP(b0, b1, b2, b3, b4, a0, a1, a2, a3, a4, ja, jb)
        }
#undef GS2D_FWD_STEP
    }
    GS2D_PROF_END(0)
    if (inside) {  // forward.cu:441-466
        const size_t HW = (size_t)H * W;
        const size_t pix = (size_t)W * py + px;
        out_color[pix] = fmaf(T, bg0, C0_);
        out_color[HW + pix] = fmaf(T, bg1, C1_);
        out_color[2 * HW + pix] = fmaf(T, bg2, C2_);
        const float dstd = fmaf(median_depth * median_depth, 1 - T, fmaf(-2.0f * median_depth, Dp, D2));
        out_others[pix] = Dp;
        out_others[HW + pix] = 1 - T;
        out_others[2 * HW + pix] = N0;
        out_others[3 * HW + pix] = N1;
        out_others[4 * HW + pix] = N2;
        out_others[5 * HW + pix] = median_depth;
        out_others[6 * HW + pix] = USE_SA ? dstd : distortion;
        const size_t si = (size_t)tile * GS2D_TILE_PIX + threadIdx.x;  // wave-major: coalesced 256-B rows
        pix_state[PS_TFINAL * plane + si] = T;
        pix_state[PS_M1 * plane + si] = M1;
        pix_state[PS_M2 * plane + si] = M2;
        pix_state[PS_MEDIAN * plane + si] = median_depth;
        pix_state[PS_STD * plane + si] = dstd;
        reinterpret_cast<uint32_t*>(pix_state)[PS_LAST * plane + si] = last_contributor;
        reinterpret_cast<uint32_t*>(pix_state)[PS_MEDC * plane + si] = median_contributor;
    }
    clear_share(zero, zero_n, local_block, blocks_per_frame);
}

// The backward keeps one queue per 4x4 sub-block (16-lane DPP row): finer queues would cut its trips too, but every
// (group, splat) pair costs 13 LDS float atomics and those run at about half a lane per clock per CU -- with sixteen
// 2x2 groups the kernel measured 0.59 ms instead of 0.31 ms.  Its four row bits per quadrant are the ORs of the forward's
// group bits, written next to them by phase 0 of the forward (hits4, rows_from_groups in gs2d_cull.h).

// ------------------------------------------------------------------------------------------ backward
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// pick(hi) ? b : a   kept by this lane, the other one is what the partner lane needs
template <int CTRL>
__device__ __forceinline__ float seladd(float a, float b, bool hi)
{
    const float keep = hi ? b : a, give = hi ? a : b;
    return keep + dpp_get<CTRL>(give);
}
// --- 16-value butterfly inside ONE 16-lane DPP row: 16 values summed over the row's 16 lanes in 45 VALU ops; lane l of
// every row ends up with value number 8*bit0(l) + 4*bit1(l) + 2*bit2(l) + bit3(l) of ITS row, so the four rows of a wave
// reduce four different splats at once (scripts/dev/reduce16_row_probe.hip).
__device__ __forceinline__ float reduce16_row(const float v[16], int lane)
{
    // Levels 1 and 2 pair lanes across DPP banks (l <-> 15-l flips bit 3, l <-> l^7 flips bit 2), so "lanes with the bit
    // set keep the odd value" is a bank mask: sum the even value on all lanes, then overwrite banks {2,3} (resp. {1,3})
    // with the sum of the odd value -- two DPP adds per output and no selects.  Same operands and order as
    // keep + partner(give), so the sums are bit-identical.  (s_nop at the start: VALU write -> DPP read of the same register
    // needs 2 wait states and the compiler does not look inside asm blocks.  None is needed between or behind the blocks: a
    // level's last two writes are read by the next level's third instruction or later, and what follows the second block are
    // plain VALU reads, for which the compiler keeps its own hazard bookkeeping.)
    float e0, e1, e2, e3, e4, e5, e6, e7, f0, f1, f2, f3;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %8, %8 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %10, %10 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %12, %12 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %14, %14 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %16, %16 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %18, %18 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %20, %20 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %7, %22, %22 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %9, %9 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %11, %11 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %13, %13 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %15, %15 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %17, %17 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %5, %19, %19 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %6, %21, %21 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %7, %23, %23 row_mirror row_mask:0xf bank_mask:0xc"
        : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3), "=&v"(e4), "=&v"(e5), "=&v"(e6), "=&v"(e7)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]),
          "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
    asm("v_add_f32_dpp %0, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %10, %10 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %9, %9 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %3, %11, %11 row_half_mirror row_mask:0xf bank_mask:0xa"
        : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3)
        : "v"(e0), "v"(e1), "v"(e2), "v"(e3), "v"(e4), "v"(e5), "v"(e6), "v"(e7));
    const bool b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
    const float g0 = seladd<0x4E>(f0, f1, b1), g1 = seladd<0x4E>(f2, f3, b1);  // quad_perm [2,3,0,1]
    return seladd<0xB1>(g0, g1, b0);                                            // quad_perm [1,0,3,2]
}
// reduce16_row for inputs whose values 12, 13 and 14 are exact zeros on every lane (no upstream gradient on the normal
// channels): the (12,13) pair and half of the (14,15) subtree vanish.  Lanes that would hold 12..14 end with +0.
__device__ __forceinline__ float reduce16_row_z(const float v[16], int lane)
{
    float e0, e1, e2, e3, e4, e5, e7, f0, f1, f2, f3;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %7, %7 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %9, %9 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %11, %11 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %13, %13 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %15, %15 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %17, %17 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %19, %19 row_mirror row_mask:0xf bank_mask:0xf\n\t"   /* value 15; only banks 2,3 are used */
        "v_add_f32_dpp %0, %8, %8 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %10, %10 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %12, %12 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %14, %14 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %16, %16 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %5, %18, %18 row_mirror row_mask:0xf bank_mask:0xc"
        : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3), "=&v"(e4), "=&v"(e5), "=&v"(e7)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]),
          "v"(v[10]), "v"(v[11]), "v"(v[15]));
    // (f3's zero is written inside the block: as an input it would pin a register holding 0.0 through the whole trip loop)
    asm("v_mov_b32 %3, 0\n\t"
        "v_add_f32_dpp %0, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %10, %10 row_half_mirror row_mask:0xf bank_mask:0x8\n\t"  /* lanes 12-15 <- value 15 */
        "v_add_f32_dpp %0, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %9, %9 row_half_mirror row_mask:0xf bank_mask:0xa"
        : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3)
        : "v"(e0), "v"(e1), "v"(e2), "v"(e3), "v"(e4), "v"(e5), "v"(e7));
    const bool b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
    const float g0 = seladd<0x4E>(f0, f1, b1), g1 = seladd<0x4E>(f2, f3, b1);
    return seladd<0xB1>(g0, g1, b0);
}
__device__ __forceinline__ int reduce16_row_index(int lane)
{
    return 8 * (lane & 1) + 4 * ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1) + ((lane >> 3) & 1);
}
// sum over the 16 lanes of each row, valid in lane 15 of the row (rare low-pass branch)
__device__ __forceinline__ float row_sum_to_lane15(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
}

// --- 3-value reduction inside ONE 16-lane DPP row (pose-only backward): the same mirror / half-mirror levels with bank-masked
// overwrites, then two full quad levels: 7 DPP adds.  Every lane of quad 0 (lanes 0-3) of a row ends with the row's total of
// v0, quad 2 (lanes 8-11) with that of v1, quad 1 (lanes 4-7) with that of v2; lanes 12-15 hold nothing of use.
__device__ __forceinline__ float reduce3_row(float v0, float v1, float v2)
{
    float e0, e1, f0;
    // (s_nop: VALU write -> DPP read of the same register needs 2 wait states and the compiler does not look inside asm blocks;
    // e0's last write is followed by the write of e1 and one s_nop 0 before the half-mirror level reads it)
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %4, %4 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %5, %5 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %2, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa"
        : "=&v"(e0), "=&v"(e1), "=&v"(f0)
        : "v"(v0), "v"(v1), "v"(v2));
    const float g = f0 + dpp_get<0x4E>(f0);  // quad_perm [2,3,0,1]
    return g + dpp_get<0xB1>(g);             // quad_perm [1,0,3,2]
}

// --- the same for EIGHT queues per wave (one per 8-lane half of a DPP row = 4x2 pixels): 16 values summed over the 8 lanes of a
// half, two results per lane.  Levels: half-mirror (l <-> 7-l, bank-masked, no selects), then the two quad levels as
// select-adds.  Lane l of a half (b2 b1 b0 = its low three bits) ends with value 4 b0 + 2 b1 + b2 in `lo` and 8 + that in `hi`.
__device__ __forceinline__ void reduce16_half(const float v[16], int lane, float& lo, float& hi)
{
    float e0, e1, e2, e3, e4, e5, e6, e7;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %10, %10 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %12, %12 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %14, %14 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %16, %16 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %18, %18 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %20, %20 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %7, %22, %22 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %9, %9 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %11, %11 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %13, %13 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %3, %15, %15 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %4, %17, %17 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %5, %19, %19 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %6, %21, %21 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %7, %23, %23 row_half_mirror row_mask:0xf bank_mask:0xa"
        : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3), "=&v"(e4), "=&v"(e5), "=&v"(e6), "=&v"(e7)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]),
          "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
    const bool b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
    const float f0 = seladd<0x4E>(e0, e1, b1), f1 = seladd<0x4E>(e2, e3, b1);
    const float f2 = seladd<0x4E>(e4, e5, b1), f3 = seladd<0x4E>(e6, e7, b1);
    lo = seladd<0xB1>(f0, f1, b0);
    hi = seladd<0xB1>(f2, f3, b0);
}
__device__ __forceinline__ int reduce16_half_index(int lane)  // the value `lo` holds (hi: + 8)
{
    return 4 * (lane & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 2) & 1);
}
// three values over the 8 lanes of a half (pose-only backward with eight queues): lanes 0-3 of the half end with the total of
// v0 in `a`, lanes 4-7 with that of v1; every lane has the total of v2 in `b`
__device__ __forceinline__ void reduce3_half(float v0, float v1, float v2, float& a, float& b)
{
    float e0, e1;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa"
        : "=&v"(e0), "=&v"(e1) : "v"(v0), "v"(v1), "v"(v2));
    const float g0 = e0 + dpp_get<0x4E>(e0), g1 = e1 + dpp_get<0x4E>(e1);
    a = g0 + dpp_get<0xB1>(g0);
    b = g1 + dpp_get<0xB1>(g1);
}
// the value of the lane 8 further on in the same 16-lane row (row_ror:8): what the OTHER half of the row holds
__device__ __forceinline__ float dpp_other_half(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
}
__device__ __forceinline__ uint32_t dpp_other_half(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);
}
// Eight queues: the two halves of a row often hold the SAME splat in a trip (a splat that covers both halves of a 4x4 sub-block
// sits at about the same place of both queues).  Left alone, their totals meet on one accumulator -- the LDS-atomic path, which
// cost the eight-queue kernel all it had gained (0.166 vs 0.147 ms with the atomics compiled out, pose-only backward).  So the
// twins are merged in registers: the lower half adds the upper half's total to its own, the upper half drops its own.
// twin_m: lanes whose row's halves hold the same slot; lo_m / hi_m: the lanes of the lower / upper halves (wave masks).
__device__ __forceinline__ float merge_twin_halves(float tot, uint64_t twin_lo_m, uint64_t twin_hi_m)
{
    const float other = dpp_other_half(tot);
    tot = __builtin_amdgcn_inverse_ballot_w64(twin_lo_m) ? tot + other : tot;
    return __builtin_amdgcn_inverse_ballot_w64(twin_hi_m) ? 0.f : tot;
}
// sum over the 8 lanes of each half of a row, valid in lanes 7 and 15 of the row (rare low-pass branch, eight queues)
__device__ __forceinline__ float half_sum_to_lane7(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    return v;
}

// Gradient record (GS2D_GRAD_FLOATS = 20 floats per Gaussian):
//   [0..2] dL_dcolor  [3..5] dL_dnormal  [6..14] dL_dT (Tu,Tv,Tw)  [15] dL_dopacity  [16,17] dL_dmean2D.xy
//
// One wave per 8x8 quadrant, waves independent (wave-private LDS staging, no workgroup barrier).  Per contributing
// (row, splat) pair the 16 main components are reduced over the row's 16 lanes with the butterfly above, added into the
// splat's LDS accumulators (plain store or LDS atomic, see GS2D_BWD_ACC_PRE below), and flushed to the Gaussian's record
// once per (quadrant, batch): per flush pass one global-atomic instruction per touched component, its lanes covering four
// records (the reference issues 16-18 scalar atomics per (pixel, splat) pair).
// Wave-private LDS of the backward: a 64-byte staged record per splat of the batch (Tu|cx, Tv|cy, Tw|opacity,
// r,g,b,id -- the normal is fetched from global memory only by waves that carry normal gradients) and 13 gradient
// accumulators per splat (3 colour, 9 dT, 1 opacity).  The gradient atomics are the backward's scarcest resource
// (scripts/dev/atomic_bench.hip: the kernel's former atomic stream alone takes ~360 us on the whole chip), so every
// (row, splat) contribution is first summed in LDS (GS2D_BWD_LDS_ACCUM below) and each touched splat of the batch is flushed to its
// global record ONCE per (quadrant, batch) -- about half as many global atomics, three quarters of a workgroup's LDS
// budget (29.6 KB, 5 workgroups per CU).
//
// DETERMINISTIC variant (DET, opt-in through gs2d_set_deterministic; the reference has none -- its float atomics make
// gradients run-to-run non-deterministic, backward.cu:441-460): no global atomic at all.  Every (instance, quadrant) pair
// is accumulated by exactly one wave in exactly one batch, so that wave STORES its 18 sums (the 13 above plus the normal
// and the low-pass mean2D components) into a slot of its own, det_slots[(instance * 4 + quadrant) * 20 ...], LDS adds are
// plain read-add-writes issued row by row (no two lanes of one instruction meet on an address), and det_reduce_kernel sums a Gaussian's slots
// in a fixed order (its tiles row-major, quadrants 0..3).  Two runs give bit-identical gradients.
// Adding a row's 13 totals to its splat's accumulators.  An LDS float atomic is the slowest instruction of the kernel:
// scripts/dev/lds_atomic_rate.hip measures ds_add_f32 at ~137 LDS clocks per wave instruction with 52 active lanes
// (ds_add_u32: 4.4, ds_write_b32: 7.1), and the LDS index unit was busy 81 % of the kernel (SQ_LDS_IDX_ACTIVE).  So a
// trip READS its accumulator first thing (the latency hides behind the trip's ~250 instructions) and, when the four rows
// hold four different splats (wave-uniform test on four v_readlane), finishes with a plain store of old + total; only
// trips in which two rows meet on a splat fall back to the atomic.  0.329 -> 0.306 ms.  Measured and dropped: reading
// the accumulator at the end of the trip (0.338 ms, the latency enters the dependency chain), merging twin rows' totals
// with v_permlane16/32_swap so that no trip needs the atomic (0.330 ms, +4 spills), and a per-row instead of
// wave-uniform fallback (no change).
#define GS2D_BWD_ACC_PRE(JJ, FJ_)                                                                                        \
    const int ai_ = (int)__umul24((JJ), (uint32_t)NACC) + (acc_comp < 0 ? 0 : acc_comp); /* v_mul_lo_u32 is quarter rate */ \
    const float acc_old_ = wb.acc()[ai_];                                                                                 \
    /* eight queues, full kernel: the lane's second value (reduce16_half) has an accumulator of its own */               \
    const int ai2_ = ai_ - (acc_comp < 0 ? 0 : acc_comp) + (acc_comp_hi < 0 ? 0 : acc_comp_hi);                         \
    float acc_old2_ = 0.f;                                                                                              \
    if (NG == 8 && !POSE) acc_old2_ = wb.acc()[ai2_];                                                                    \
    /* do two rows hold the same splat in this trip?  Decided for all trips of the batch at once when the queues are    \
       built (clash_mask, one bit per trip): one scalar bit test here instead of four v_readlane + a scalar compare     \
       chain per trip */                                                                                                \
    /* eight queues: per LANE instead (the flag rides in bit 6 of the queue entry, see the queue build): with eight queues   \
       some two of them meet in a good share of the trips, and an LDS float atomic costs by the lane */               \
    const bool clash_ = GS2D_DEV_CLASH(ROW_ACC ? false : LANE_CLASH ? __builtin_amdgcn_inverse_ballot_w64(FJ_) : ((clash_mask >> t) & 1ull) != 0ull);
#ifndef GS2D_BWD_LDS_ACCUM  // (gs2d_blend_dev.h may have replaced it in an experiment build)
#define GS2D_BWD_LDS_ACCUM(JJ, V)                                                                                       \
    if ((V) != 0.f) {                                                                                                   \
        if (clash_) atomicAdd(&wb.acc()[ai_], V);                                                                         \
        else wb.acc()[ai_] = acc_old_ + (V);                                                                              \
    }
#define GS2D_BWD_LDS_ACCUM2(JJ, V)                                                                                      \
    if ((V) != 0.f) {                                                                                                   \
        if (clash_) atomicAdd(&wb.acc()[ai2_], V);                                                                        \
        else wb.acc()[ai2_] = acc_old2_ + (V);                                                                            \
    }
#endif
// dev A/B switches (scripts/dev/variants.sh) for the wave-level "does any lane contribute?" shortcuts of the trip step.
// Measured in round 3 (one session, two runs each): skipping the whole step when no lane can be active (a test on the freshly
// loaded list position, BEFORE the geometry) costs more than it saves -- every trip waits for v_cmp -> SGPR -> s_cbranch
// before its first FMA: 0.2745 -> 0.2665 ms without it (default now); the second shortcut, in front of the gradient expansion
// and the butterfly, pays: 0.279 without it.  (Also tried, no gain: letting every live row store on clash-free trips so that
// the store's lane mask does not depend on the butterfly's result, 0.267; announcing the rare low-pass branch by an early
// ballot, 0.269.)
#ifdef GS2D_BWD_EARLY_SKIP
#define GS2D_BWD_SKIP1(X) (X)
#else
#define GS2D_BWD_SKIP1(X) true
#endif
#ifdef GS2D_BWD_NO_SKIP2
#define GS2D_BWD_SKIP2(X) true
#else
#define GS2D_BWD_SKIP2(X) (X)
#endif
#define GS2D_ACC 13
#define GS2D_ACC_DET 18
#define GS2D_ACC_POSE 3   // pose-only backward: dT[2], dT[5], dT[8] -- all dL/dmean needs (gs2d_preprocess.hip, backward.cu:637-663)
#define GS2D_ACC_POSE_ROWS (4 * GS2D_ACC_POSE)  // ... kept once per ROW of the wave (see ROW_ACC in blend_bwd_kernel)
template <int NACC, int NQ = 4>
struct BwdBatchT {
    // Entries past a queue's end hold slot 63 -- the deepest staged splat, always a real record -- and a row is live while the
    // trip number is below its queue length: an exhausted row evaluates a real (finite) record with all its lanes inactive,
    // so its sums are exact zeros and nothing is accumulated, and no access masks the slot number.  (Tried on the way: a
    // marker of 64 read unmasked.  With a zeroed dummy 65th slot the extra 512 B of LDS per workgroup cost a fifth workgroup
    // per CU, 0.385 instead of 0.273 ms; without one the row "blends" whatever words follow the arrays, 0 x Inf = NaN passes the
    // `!= 0` test of the accumulate and lands behind the accumulator array -- a corrupted Gaussian id and a memory fault in
    // the flush, caught by tests/test_gpu_batch.py.)
    float raw[4 * 64 * 4 + 64 * NACC];  // q: 4 quarters x 64 slots x float4 (Tu|cx, Tv|cy, Tw|opacity, r g b position), then acc
    __device__ __forceinline__ float4* q(int k) { return reinterpret_cast<float4*>(raw) + k * 64; }
    __device__ __forceinline__ float* acc() { return raw + 4 * 64 * 4; }
    uint32_t pn[64];    // Gaussian id | four cull bits << 28 of the staged splat (read when queues are built and at the flush only;
                        // with eight queues the other four bits ride in the top of the position word until the queues are built)
    uint8_t ql[NQ][64]; // per-group queues: slot numbers, deepest first, then 63s
    uint32_t tail[NQ == 4 ? 4 : 1];  // four queues: the pipeline reads up to two entries past a full queue (slot 63 again); eight
                                     // queues have no LDS left for that (32 768 B is the last size with five workgroups per CU,
                                     // scripts/dev/lds_occupancy_probe) and clamp the read-ahead index instead
    static constexpr int BYTES = 4 * (4 * 64 * 4 + 64 * NACC) + 256 + 64 * NQ + (NQ == 4 ? 16 : 0);  // what the kernel allocates
};
static_assert(BwdBatchT<GS2D_ACC, 4>::BYTES == 7952 && BwdBatchT<GS2D_ACC, 8>::BYTES == 8192, "blend_bwd LDS budget: 31 808 / 32 768 B per workgroup");
static_assert(sizeof(BwdBatchT<GS2D_ACC, 4>) == 7952, "BwdBatchT<13, 4>: 4 x 7952 = 31 808 B, five workgroups per CU");

// POSE (tracking with every Gaussian parameter detached, render/__init__.py:31-36; chosen by gs2d_backward_staged when all six
// per-Gaussian outputs are NULL): the pose gradient needs dL/dmean = Pm^T (dT[2], dT[5], dT[8]) only (backward.cu:637-663; plus
// the rare low-pass dL_dmean2D pair, which feeds those three through the centre formula), so the trip forms only -dk.z, -dl.z and
// the Tw.z component (no dp2, no colour / opacity / x,y terms: the atomics of backward.cu:343,396,441-449,460 that tracking
// throws away), reduces them with the 3-value row reduction, keeps 3 accumulators per staged splat and flushes them into a
// DENSE float4-per-Gaussian array laid over the head of the (forward-cleared) gradient records; dL_dmean2D goes to
// dense_m2d[2 g].  Single frame, non-deterministic mode only.
// NG: queues per wave.  4: one per 16-lane DPP row (4x4 pixels), the 16 gradient components of a (row, splat) pair end as one
// value per lane.  8: one per 8-lane half-row (4x2 pixels): fewer loop trips (a trip ends when the LONGEST queue has moved on;
// scripts/dev/group_trips.c: -13.5 % at the bench size), but two values per lane to accumulate (reduce16_half).
template <bool USE_SA, bool DET, bool BATCH, bool POSE = false, int NG = (DET ? 4 : (POSE ? GS2D_BWD_POSE_GROUPS : GS2D_BWD_GROUPS))>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DET ? 4 : GS2D_WAVES_PER_EU, DET ? 4 : GS2D_WAVES_PER_EU)))
blend_bwd_kernel(int W, int H, int gx, int ntiles, int blocks_per_frame, const float* __restrict__ bg, size_t plane,
                 float* __restrict__ clear12, int clear_n,
                 const std::conditional_t<BATCH, gs2d::BlendBwdBatch, gs2d::BlendBwdFrame> args)
{
    if (clear12 != nullptr && blockIdx.x == 0 && (int)threadIdx.x < clear_n) clear12[threadIdx.x] = 0.f;
    static_assert(!(POSE && (DET || BATCH)), "the pose-only backward is single-frame and non-deterministic");
    static_assert(NG == 4 || (NG == 8 && !DET), "queues per wave: 4, or 8 in the non-deterministic kernels");
    // the LDS-atomic fallback of the accumulate decided per LANE (flag in bit 6 of the queue entry) instead of per trip

    // POSE: three sums per (row, splat) are so few that every ROW of the wave can keep accumulators of its own (4 x 3 per staged
    // splat, less LDS than the full kernel's 13): two rows never meet on an address, so the accumulate is a plain read / add /
    // store in every trip -- no LDS float atomic anywhere in the kernel; the flush adds a splat's four row sums up.
    constexpr bool ROW_ACC = POSE;
    constexpr int NACC = DET ? GS2D_ACC_DET : (POSE ? GS2D_ACC_POSE_ROWS : GS2D_ACC);
    typedef BwdBatchT<NACC, NG> BwdBatch;
    constexpr bool LANE_CLASH = !ROW_ACC && (NG == 8 || (GS2D_BWD_LANE_CLASH && !DET));
    // (a byte array of exactly 4 x BYTES: the eight-queue struct is declared with a one-word `tail` it never touches, the
    // allocation ends with its last queue)
    __shared__ __attribute__((aligned(16))) unsigned char batch_mem[4 * BwdBatch::BYTES + GS2D_DEV_LDS_PAD];
    int local_block = blockIdx.x;
    const gs2d::BlendBwdFrame* fa;
    if constexpr (BATCH) {  // see blend_fwd_kernel
        const int frame = blockIdx.x / blocks_per_frame;
        local_block = blockIdx.x - frame * blocks_per_frame;
        fa = &args.f[frame];
    } else {
        fa = &args;
    }
    const uint2* __restrict__ ranges = fa->ranges;
    const uint32_t* __restrict__ point_list = fa->point_list;
    const float4* __restrict__ rec = fa->rec;
    const float* __restrict__ pix_state = fa->pix_state;
    const uint8_t* __restrict__ hits = fa->hits;
    const uint16_t* __restrict__ hits16 = fa->hits16;
    const float* __restrict__ dL_dpix = fa->dL_dpix;
    const float* __restrict__ dL_dothers = fa->dL_dothers;
    float* __restrict__ grad_rec = fa->grad_rec;
    float* __restrict__ det_slots = fa->det_slots;
    float* __restrict__ dense_m2d = fa->dense_m2d;
    const int tile = xcd_tile(local_block, ntiles);
    if (tile < 0) return;
    const int tx = tile % gx, ty = tile / gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    BwdBatch& wb = *reinterpret_cast<BwdBatch*>(batch_mem + wave * BwdBatch::BYTES);
    const int qx0 = tx * GS2D_TILE + (wave & 1) * 8, qy0 = ty * GS2D_TILE + (wave >> 1) * 8;
    const int row = lane >> 4, li = lane & 15;                       // DPP row = 4x4 sub-block (same mapping as the forward)
    const int grp = NG == 8 ? lane >> 3 : row;                       // the queue this lane follows (NG == 8: lanes 0-7 of a row =
    const int grp8 = grp * 8;                                        // its pixel rows 0-1, lanes 8-15 = pixel rows 2-3)
    // the cull bits of list position `at` for this quadrant, in the form this instantiation's queues want
    auto cull_bits = [&](uint32_t at) -> uint32_t {
        if (NG == 8 && !GS2D_HALFROW_BITS) return halfrows_from_groups_fast(hits16[(size_t)at * 4 + wave]);
        return hits[(size_t)at * 4 + wave];
    };
    const uint8_t* qrow = wb.ql[grp];
    const int px = qx0 + (row & 1) * 4 + (li & 3), py = qy0 + (row >> 1) * 4 + (li >> 2);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)W * py + px;
    // the forward keeps its per-pixel state in ITS lane order (2x2 pixel groups, blend_fwd_kernel)
    const int flx = (row & 1) * 4 + (li & 3), fly = (row >> 1) * 4 + (li >> 2);
    const size_t si = (size_t)tile * GS2D_TILE_PIX + wave * 64 + ((fly >> 1) * 4 + (flx >> 1)) * 4 + (fly & 1) * 2 + (flx & 1);

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
    // Every gradient this wave produces is linear in its pixels' upstream gradients: a quadrant whose 64 pixels all
    // carry exact zeros (masked-out regions of the SLAM losses) contributes exactly nothing.
    if (ballot64(dpx0 != 0.f || dpx1 != 0.f || dpx2 != 0.f || dL_ddepth != 0.f || dL_daccum != 0.f || dn0 != 0.f ||
                 dn1 != 0.f || dn2 != 0.f || dL_dmedian_depth != 0.f || dL_dreg != 0.f) == 0)
        return;
    const float final_A = 1 - T_final;
    // background term of dL_dalpha, -T_final/(1-alpha) * dot(bg, dL_dpixel) (backward.cu:407-410): the two per-pixel factors
    // are folded into one register (one rounding apart from the oracle's (-T_final * ioma) * bg_dot; exactly 0 for bg = 0)
    // ... and so is the opacity-map term (backward.cu:386-389): the reference keeps accum_alpha_rec = 1 - prod(1 - alpha_j) over the
    // splats behind the current one and adds (1 - accum_alpha_rec) * dL_daccum before the multiplication by T; that product is
    // T_final / (T (1 - alpha)), so the term equals +T_final/(1-alpha) * dL_daccum -- the background term's form.  Both ride in
    // one register: a recurrence, a per-pixel constant and three operations per trip less (rounding-level deviation from the
    // oracle's recurrence: the closed form is the more accurate of the two where 1 - accum_alpha_rec cancels).
    const float tf_bg = T_final * (fmaf(bg[2], dpx2, fmaf(bg[1], dpx1, bg[0] * dpx0)) - dL_daccum);
    const float sa_k = 1.0f / (4 * fmaxf(mstd * (1.0f / (1 - T_final)), 1e-7f));  // per-pixel constant (IEEE, as the oracle)
    const float c1f = GS2D_FAR_N / (GS2D_FAR_N - GS2D_NEAR_N);
    // The reference keeps, per blended channel x (r, g, b, depth, normal xyz), the back-to-front blend accum_x of the values
    // behind the current splat and adds (x - accum_x) * dL_dx to dL_dalpha (backward.cu:331-344, 380-397).  All channels are
    // blended with the same weights, so the SUM over the channels obeys one scalar recurrence:
    //     D = sum_x x * dL_dx  (this splat),   dL_dalpha += D - S,   then  S <- alpha * D + (1 - alpha) * S
    // (the reference defers the update of S to the next splat -- last_alpha, last_color; done at once it needs neither)
    // and the regulariser's  dL_dweight - last_dL_dT  (backward.cu:353-373: last_dL_dT <- dL_dweight alpha + (1 - alpha)
    // last_dL_dT, i.e. the same blend taken one splat later) is one more such channel with unit upstream gradient.
    // One register (S) and six operations per trip instead of nine registers (fifteen with normals) and two dozen
    // operations; the same sums in a different order (rounding-level deviation from the oracle's per-channel form)
    float blend_S = 0.f;

    // nothing behind the deepest contributor of this quadrant can receive a gradient from it
    uint32_t max_last = last_contributor;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) max_last = max(max_last, (uint32_t)__shfl_xor((int)max_last, d, 64));
    // ... and nothing behind the deepest contributor of a 4x4 ROW can receive a gradient from that row: its queue leaves those
    // splats out (the quadrant-level bound above only decides where the walk starts; 954k -> 933k trips per frame)
    uint32_t row_last = last_contributor;
#pragma unroll
    for (int d = (NG == 8 ? 4 : 8); d >= 1; d >>= 1) row_last = max(row_last, (uint32_t)__shfl_xor((int)row_last, d, 64));
    uint32_t grp_last[NG];  // (wave-uniform: scalar registers)
#pragma unroll
    for (int g = 0; g < NG; g++) grp_last[g] = (uint32_t)__builtin_amdgcn_readlane((int)row_last, g * (64 / NG));
    max_last = (uint32_t)__builtin_amdgcn_readfirstlane((int)max_last);

    // butterfly slot held by this lane -> offset in the gradient record, and the sign of that component.  Slots:
    // 0-2 colour | 3-8 Tu,Tv (-dk, -dl) | 9-11 Tw | 12-14 normal | 15 opacity
    const int slot = reduce16_row_index(lane);
    // LDS accumulator of this lane's slot; -1: normal component, added straight to the global record (not in DET: 13..15)
    // (POSE: reduce3_row leaves the totals of -dk.z / -dl.z / Tw.z in quads 0 / 2 / 1 of the row; one lane of each adds them up)
    // NG == 8: reduce16_half leaves values h and 8 + h in every lane (h = reduce16_half_index): the first is always an LDS
    // accumulator (0-7), the second one of 8-11, a normal component (12-14) or the opacity (15 -> accumulator 12);
    // POSE: reduce3_half leaves -dk.z in lanes 0-3 of a half, -dl.z in lanes 4-7, Tw.z everywhere: lanes 0, 4 and 2 add them up
    const int half_idx = reduce16_half_index(lane);
    const int pose_comp = NG == 8 ? ((li & 7) == 0 ? 0 : ((li & 7) == 4 ? 1 : ((li & 7) == 2 ? 2 : -1)))
                                  : (li == 0 ? 0 : (li == 8 ? 1 : (li == 4 ? 2 : -1)));
    const int acc_comp = POSE ? (pose_comp < 0 ? -1 : row * GS2D_ACC_POSE + pose_comp)  // (this row's own three accumulators)
                       : NG == 8 ? half_idx
                              : (slot < 12 ? slot : (slot == 15 ? 12 : (DET ? slot + 1 : -1)));
    const int acc_comp_hi = half_idx < 4 ? 8 + half_idx : (half_idx == 7 ? 12 : -1);  // NG == 8, full kernel: the second value
    const uint64_t lo_half_m = ballot64((li & 8) == 0);                               // lanes of the lower halves of the rows
#pragma unroll
    for (int i = 0; i < NACC; i++) wb.acc()[i * 64 + lane] = 0.f;
    if (NG == 4 && lane < 4) wb.tail[lane] = 0x3F3F3F3Fu;
    // Wave-uniform data-dependent shortcut: when no pixel of this quadrant carries an upstream gradient on the
    // normal channels (SLAM's losses never touch them unless use_normal_loss), everything that only feeds
    // dL_dnormal / the normal term of dL_dalpha is exactly zero and is skipped.  Results are unchanged.
    const bool any_dn = ballot64(dn0 != 0.f || dn1 != 0.f || dn2 != 0.f) != 0;
    // Batches are COMPACTED: the list is read in 64-instance chunks (back to front), but only the splats whose cull bits
    // (stored by the forward) touch this quadrant are staged, and chunks keep being added until all 64 LDS slots are in
    // use (a chunk that does not fit is split: its shallower part is carried into the next batch).  Compared with
    // "one chunk = one batch" the four row queues are ~3.5x longer per batch, so waiting for the longest queue costs
    // relatively less, untouched records are never loaded, and there are fewer batch prologues and flushes.
    // Slots are handed out from 63 downwards, deepest splat first, so a queue popped from its highest bit walks the
    // splats back to front exactly as before.
    // The body is instantiated twice, with and without gradients on the normal channels, and the wave picks one: in the
    // usual case (no normal loss) nine per-pixel registers (dn, last normal, normal accumulators) are never live, which is
    // what keeps the loop inside the 96-VGPR budget of five waves per SIMD.
    GS2D_PROF_BEGIN();
    auto run = [&](auto dn_tag) __attribute__((always_inline)) {
    constexpr bool ANY_DN = decltype(dn_tag)::value;
    if (max_last == 0u) return;  // no pixel of this quadrant has a contributor
    int chunk = (int)((max_last + 63) / 64) - 1;  // next chunk to read
    int carry_chunk = -1;                           // chunk whose shallower part is still waiting
    uint32_t carry_tm = 0u, carry_id = 0u;
    // Staging a batch costs TWO exposed memory round trips: the chunk loop only hands out slots (Gaussian id + cull bits -> pn,
    // list position -> the record's free word) while the cull bits and ids of the NEXT chunk are already in flight, and when
    // the 64 slots are taken -- or the list ends -- lane l gathers the record of slot l.  (Until round 3 every chunk loaded its
    // bits and ids, waited, gathered its records, waited: about six dependent round trips per batch.)  The prefetch registers
    // live inside the staging loop only -- kept across the trip loop they cost the loop a register it does not have.
    const uint32_t last_i = range.x + max_last - 1u;
    for (;;) {
        int fill = 0;
        prio_by_remaining<GS2D_BWD_PRIO_SHIFT>((uint32_t)(chunk + 1), (max_last + 63u) / 64u);
        GS2D_PROF_STAGE_BEGIN();
        wave_lds_sync();  // previous batch fully consumed before it is overwritten
        uint32_t pf_tm = 0u, pf_id = 0u;
        if (chunk >= 0) {
            const uint32_t at = min(range.x + (uint32_t)chunk * 64u + lane, last_i);
            pf_tm = cull_bits(at);
            pf_id = point_list[at];  // unconditional and coalesced: issued together with the cull bits
        }
        for (;;) {
            uint32_t tm, my_id;
            int cb;
            if (carry_chunk >= 0) { tm = carry_tm; my_id = carry_id; cb = carry_chunk; carry_chunk = -1; }
            else {
                if (chunk < 0) break;
                cb = chunk--;
                const int n = (int)min(64u, max_last - (uint32_t)cb * 64u);
                // queues come from the cull bits of this (instance, quadrant): eight half-row bits, ORed to four row bits here
                // when the wave keeps four queues
                tm = lane < n ? ((NG == 8 || !GS2D_HALFROW_BITS) ? pf_tm : rows_from_halfrows(pf_tm)) : 0u;
                my_id = pf_id;
                const uint32_t nb = min(range.x + (uint32_t)max(chunk, 0) * 64u + lane, last_i);
                pf_tm = cull_bits(nb);
                pf_id = point_list[nb];
            }
            const uint64_t tb = ballot64(tm != 0u);
            const int c = __popcll(tb);
            if (c == 0) continue;
            // deepest touched lane -> highest free slot: touched lanes above me = c - 1 - (touched lanes below me)
            const int slot = 64 - fill - c + rank_below(tb);
            const bool take = tm != 0u && slot >= 0;
            if (take) {
                // the list position (< 2^28: a Gaussian is in a tile's list at most once); with eight queues the upper four
                // cull bits ride on top of it until the queues are built
                wb.q(3)[slot].w = __uint_as_float(((uint32_t)cb * 64u + lane) | (NG == 8 ? (tm >> 4) << 28 : 0u));
                wb.pn[slot] = my_id | ((tm & 15u) << 28);                     // Gaussian id (< 2^28, checked by the API) + cull bits
            }
            if (fill + c > 64) { carry_tm = take ? 0u : tm; carry_id = my_id; carry_chunk = cb; fill = 64; break; }
            fill += c;
            if (fill == 64) break;
        }
        if (fill == 0) break;  // list exhausted
        wave_lds_sync();
        if (lane >= 64 - fill) {
            const float4* rp = rec + (size_t)(wb.pn[lane] & 0x0FFFFFFFu) * GS2D_REC_F4;
            const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
            const float red = rp[3].w;
            const float4 r4 = rp[4];
            wb.q(0)[lane] = r0; wb.q(1)[lane] = r1; wb.q(2)[lane] = r2;
            float* cw = reinterpret_cast<float*>(&wb.q(3)[lane]);
            cw[0] = red; cw[1] = r4.x; cw[2] = r4.y;  // colour; the fourth word already holds the list position
        }
        if (fill == 0) break;  // list exhausted
        wave_lds_sync();
        // four depth-ordered queues, one per 4x4 sub-block (= DPP row): byte lists of slot numbers, deepest (= highest
        // slot) first, so walking a list front to back visits the row's splats back to front
        uint32_t nib = lane >= 64 - fill ? wb.pn[lane] >> 28 : 0u;
        {
            const uint32_t posw = __float_as_uint(wb.q(3)[lane].w);
            const uint32_t pos = posw & 0x0FFFFFFFu;
            if (NG == 8) {  // the upper four bits come off the position word, which the trip loop reads as a plain number
                if (lane >= 64 - fill) { nib |= (posw >> 28) << 4; wb.q(3)[lane].w = __uint_as_float(pos); }
            }
#ifndef GS2D_NO_ROW_LAST  // dev A/B switch (scripts/dev/variants.sh)
            // group g keeps the splat only if it lies in front of the group's deepest contributor (pos < grp_last[g])
            uint32_t keep = 0u;
#pragma unroll
            for (int g = 0; g < NG; g++) keep |= pos < grp_last[g] ? (1u << g) : 0u;
            nib &= keep;
#endif
        }
        uint64_t gm[NG];
        int glen[NG];
        int trips = 0;  // (0: nothing staged lies in front of its groups' deepest contributors -- one idle step)
        uint64_t lens_packed = 0ull;
#pragma unroll
        for (int g = 0; g < NG; g++) {
            gm[g] = ballot64((nib >> g) & 1u);
            glen[g] = __popcll(gm[g]);
            trips = max(trips, glen[g]);
            lens_packed |= (uint64_t)(uint32_t)glen[g] << (8 * g);
        }
        // every entry past a queue's end reads slot 63 (see BwdBatchT)
#pragma unroll
        for (int w = 0; w < NG / 4; w++) reinterpret_cast<uint32_t*>(wb.ql)[lane + 64 * w] = 0x3F3F3F3Fu;
        const int mylen = (int)((lens_packed >> grp8) & 0xffull);  // this lane's group is live in trips 0 .. mylen - 1
#pragma unroll
        for (int g = 0; g < NG; g++)
            if ((nib >> g) & 1u) wb.ql[g][glen[g] - 1 - rank_below(gm[g])] = (uint8_t)lane;
        wave_lds_sync();
        // bit t: in trip t two LIVE groups hold the same splat: their totals then meet on one accumulator -- LDS atomic instead
        // of the plain store, GS2D_BWD_LDS_ACCUM
        uint64_t clash_mask;
        {
            uint64_t seen = 0ull;
            bool clash = false;
            uint32_t prev_e = 0u;
            bool prev_live = false;
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const bool live = (int)lane < glen[g];
                const uint32_t e = wb.ql[g][lane];
                // (eight queues: the upper half of a row holding its lower half's splat is merged in registers, merge_twin_halves)
                const bool twin = NG == 8 && (g & 1) && live && prev_live && e == prev_e;
                const uint64_t bit = live && !twin ? 1ull << e : 0ull;
                clash = clash || (seen & bit) != 0ull;
                seen |= bit;
                prev_e = e; prev_live = live;
            }
            clash_mask = ballot64(clash);
            if (LANE_CLASH) {
                // which GROUPS meet on a splat in this trip: their queue entries get bit 6 set (slot numbers are 0-63); the
                // trip loop takes the flag off when it loads an entry and sends only those lanes through the LDS atomic
                // The WRITERS of a trip, exactly as the trip loop will have them: the two halves of a row whose entries are equal
                // (exhausted halves read the sentinel, slot 63) are merged in registers and written by the LOWER half's lanes
                // -- even when only the upper half is live; otherwise every live half writes for itself.
                uint64_t once = 0ull, twice = 0ull;
                uint32_t es[NG];
                bool ws[NG];
#pragma unroll
                for (int r = 0; r < NG / 2; r++) {
                    const bool l_lo = (int)lane < glen[2 * r], l_hi = (int)lane < glen[2 * r + 1];
                    const uint32_t e_lo = wb.ql[2 * r][lane], e_hi = wb.ql[2 * r + 1][lane];
                    const bool merged = NG == 8 && e_lo == e_hi;  // (four queues: no merging, every live row writes for itself)
                    es[2 * r] = e_lo; es[2 * r + 1] = e_hi;
                    ws[2 * r] = merged ? (l_lo || l_hi) : l_lo;
                    ws[2 * r + 1] = merged ? false : l_hi;
                }
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    const uint64_t bit = ws[g] ? 1ull << es[g] : 0ull;
                    twice |= once & bit;
                    once |= bit;
                }
                if (twice != 0ull) {
#pragma unroll
                    for (int g = 0; g < NG; g++)
                        if (ws[g] && ((twice >> es[g]) & 1ull)) wb.ql[g][lane] = (uint8_t)(es[g] | 0x40u);
                }
                wave_lds_sync();
            }
        }
        GS2D_PROF_STAGE_END();
        // software pipeline: queue entries are read two trips ahead, records one trip ahead
        int t = 0;
        uint32_t j = qrow[0], jx = qrow[1];
        uint64_t fj = 0ull, fjx = 0ull;  // eight queues: "this lane's group meets another one on its splat" for the entries in j, jx
        if (LANE_CLASH) { fj = ballot64(j > 63u); fjx = ballot64(jx > 63u); j &= 63u; jx &= 63u; }
        float4 ga0 = wb.q(0)[j], ga1 = wb.q(1)[j], ga2 = wb.q(2)[j];
        float4 gb0, gb1, gb2;
#define GS2D_BWD_STEP(G0, G1, G2, J, N0_, N1_, N2_, JN, FJ_)                                                             \
        {                                                                                                             \
            GS2D_PROF_TRIP();                                                                                         \
            uint32_t jnn = qrow[NG == 8 ? min(t + 2, 63) : t + 2]; /* (eight queues: nothing to over-read into, see BwdBatchT) */ \
            asm volatile("" : "+v"(jnn)); /* a full 32-bit value from here on: ds_read_u8 zero-extends, no v_and 0xff later */ \
            uint64_t fnn_ = 0ull;                                                                                     \
            if (LANE_CLASH) { fnn_ = ballot64(jnn > 63u); jnn &= 63u; }                                               \
            N0_ = wb.q(0)[JN]; N1_ = wb.q(1)[JN]; N2_ = wb.q(2)[JN];                                                  \
            const float4 cc = wb.q(3)[J]; /* r, g, b, list position */                                                \
            const uint32_t contributor = __float_as_uint(cc.w); /* 0-based, as in backward.cu:285 */                  \
            /* "this lane contributes" as a wave mask in scalar registers (tests on it are scalar compares, not v_cndmask + v_cmp) */ \
            uint64_t am = ballot64(t < mylen) & ballot64(contributor < last_contributor); /* past its queue's end the row idles; outside: last = 0 */ \
            if (GS2D_BWD_SKIP1(am != 0ull)) {                                                                          \
                GS2D_BWD_ACC_PRE(J, FJ_)                                                                              \
                /* eight queues: do the two halves of my row hold the same splat in this trip? (an exhausted half holds slot \
                   63 with an all-zero total: merging it is harmless) */                                              \
                uint64_t twin_m_ = 0ull;                                                                              \
                if (NG == 8) twin_m_ = ballot64(dpp_other_half((uint32_t)(J)) == (uint32_t)(J));                      \
                /* Part A (all lanes): same geometry / alpha as the forward */                                        \
                const float k0 = fmaf(pxf, G2.x, -G0.x), k1 = fmaf(pxf, G2.y, -G0.y), k2 = fmaf(pxf, G2.z, -G0.z);    \
                const float l0 = fmaf(pyf, G2.x, -G1.x), l1 = fmaf(pyf, G2.y, -G1.y), l2 = fmaf(pyf, G2.z, -G1.z);    \
                const float p0 = fmaf(k1, l2, -(k2 * l1));                                                            \
                const float p1 = fmaf(k2, l0, -(k0 * l2));                                                            \
                const float p2 = fmaf(k0, l1, -(k1 * l0));                                                            \
                const float ip = p2 == 0.0f ? 0.0f : fast_rcp(p2); /* 0 keeps the skipped lanes finite */             \
                const float s0 = p0 * ip, s1 = p1 * ip;                                                               \
                const float rho3d = fmaf(s0, s0, s1 * s1);                                                            \
                const float d0 = G0.w - pxf, d1 = G1.w - pyf;                                                         \
                const float rho2d = GS2D_FILTER_INV_SQ * fmaf(d0, d0, d1 * d1);                                       \
                const float rho = fminf(rho3d, rho2d);                                                                \
                const bool ray = rho3d <= rho2d;                                                                      \
                float c_d = ray ? fmaf(s0, G2.x, fmaf(s1, G2.y, G2.z)) : G2.z;                                        \
                const float G = fast_exp_neg_half(rho);                                                               \
                const float alpha = fminf(0.99f, G2.w * G);                                                           \
                am &= ballot64(!(p2 == 0.0f)) & ballot64(!(c_d < GS2D_NEAR_N)) & ballot64(!(rho < 0.0f)) & ballot64(!(alpha < 1.0f / 255.0f)); \
                const bool active = __builtin_amdgcn_inverse_ballot_w64(am);                                          \
                /* Part B (contributing lanes only): state recurrences; it leaves four "drivers" from which every      \
                   gradient component follows linearly -- all zero for the lanes that do not contribute. */           \
                float d_w = 0.f, d_x = 0.f, d_z = 0.f, d_op = 0.f;                                                    \
                if (active) {                                                                                         \
                    const float ioma = fast_rcp(1.f - alpha);                                                         \
                    T = T * ioma;                                                                                     \
                    const float w = alpha * T;                                                                        \
                    /* backward.cu:331-344: the colour part of D (see blend_S above) */                               \
                    float D_ = fmaf(cc.z, dpx2, fmaf(cc.y, dpx1, cc.x * dpx0));                                       \
                    float conf = 1.f;                                                                                 \
                    float dL_dz = 0.0f, dL_dweight;                                                                   \
                    if (contributor == median_contributor - 1u) dL_dz = dL_dmedian_depth;                             \
                    if (USE_SA) { /* backward.cu:347-351, 353-360 */                                                  \
                        const float dm0 = c_d - mm;                                                                   \
                        if (T < 0.5f) conf = fast_exp(-(dm0 * dm0) * sa_k);                                           \
                        /* the re-weighted depth c_d conf + mm (1 - conf) = mm + conf (c_d - mm), and its distance from \
                           the median, conf (c_d - mm), without forming (1 - conf) and without the cancellation of     \
                           subtracting mm again (rounding-level deviation from the oracle's form) */                  \
                        c_d = fmaf(conf, dm0, mm);                                                                    \
                        const float dm = conf * dm0;                                                                  \
                        dL_dweight = (dm * dm) * dL_dreg; /* enters D below: dL_dweight - last_dL_dT obeys the same recurrence */ \
                        const float cw2 = conf * w;                                                                   \
                        dL_dz = fmaf((cw2 + cw2) * dm, dL_dreg, dL_dz);                                               \
                    } else {                                                                                          \
                        const float icd = fast_rcp(c_d);                                                              \
                        const float m_d = c1f * (1 - GS2D_NEAR_N * icd);                                              \
                        const float dmd_dd = (c1f * GS2D_NEAR_N) * (icd * icd);                                       \
                        dL_dweight = fmaf(m_d * m_d, final_A, fmaf(-2.0f * m_d, final_D, final_D2)) * dL_dreg;        \
                        const float dL_dmd = 2.0f * w * fmaf(m_d, final_A, -final_D) * dL_dreg;                       \
                        dL_dz = fmaf(dL_dmd, dmd_dd, dL_dz);                                                          \
                    }                                                                                                 \
                    D_ = fmaf(c_d, dL_ddepth, D_) + dL_dweight; /* backward.cu:380-385: the depth channel; :353-373: the regulariser */ \
                    if (ANY_DN) { /* backward.cu:392-397; the normal is not staged: rare path, read it from the record */ \
                        const float4 nn = rec[(size_t)(wb.pn[J] & 0x0FFFFFFFu) * GS2D_REC_F4 + 3];                    \
                        D_ = fmaf(nn.z, dn2, fmaf(nn.y, dn1, fmaf(nn.x, dn0, D_)));                                   \
                    }                                                                                                 \
                    const float DmS = D_ - blend_S;                                                                   \
                    const float dL_dalpha = fmaf(-ioma, tf_bg, DmS * T);                                              \
                    blend_S = fmaf(alpha, DmS, blend_S); /* S <- alpha D + (1 - alpha) S: what the next splat in front sees */ \
                    d_w = w;                                                                                          \
                    d_op = G * dL_dalpha;                                                                             \
                    d_x = (G2.w * dL_dalpha) * -G;                                                                    \
                    d_z = fmaf(conf * w, dL_ddepth, dL_dz);                                                           \
                }                                                                                                     \
                /* ray-plane intersection / low-pass fall-back (backward.cu:419-449 / 450-457) */                     \
                const float d_gG = ray ? d_x : 0.f, d_t = ray ? 0.f : d_x * GS2D_FILTER_INV_SQ;                       \
                const float d_zr = ray ? d_z : 0.f, d_zl = ray ? 0.f : d_z;                                           \
                if (GS2D_BWD_SKIP2(am != 0ull)) {                                                                      \
                    const float dL_ds0 = fmaf(d_gG, s0, d_zr * G2.x);                                                 \
                    const float dL_ds1 = fmaf(d_gG, s1, d_zr * G2.y);                                                 \
                    const float dsx = dL_ds0 * ip, dsy = dL_ds1 * ip;                                                 \
                    float tot;                                                                                        \
                    if (POSE) { /* the z components only: the same expressions as g[5], g[8], g[11] below */          \
                        const float nk2 = fmaf(l1, dsx, -(l0 * dsy));                                                 \
                        const float nl2 = fmaf(dsy, k0, -(dsx * k1));                                                 \
                        if (NG == 8) {                                                                                \
                            float ta_, tb_;                                                                           \
                            reduce3_half(nk2, nl2, fmaf(-pxf, nk2, fmaf(-pyf, nl2, d_zr)) + d_zl, ta_, tb_);          \
                            tot = merge_twin_halves((li & 7) == 2 ? tb_ : ta_, twin_m_ & lo_half_m, twin_m_ & ~lo_half_m); \
                        } else                                                                                        \
                            tot = reduce3_row(nk2, nl2, fmaf(-pxf, nk2, fmaf(-pyf, nl2, d_zr)) + d_zl);               \
                        if (acc_comp >= 0) {                                                                          \
                            GS2D_BWD_LDS_ACCUM(J, tot)                                                                \
                        }                                                                                             \
                    } else {                                                                                          \
                    float g[16];                                                                                      \
                    g[0] = d_w * dpx0; g[1] = d_w * dpx1; g[2] = d_w * dpx2;                                          \
                    const float dp2 = -fmaf(dsx, s0, dsy * s1);                                                       \
                    /* the record holds -dk, -dl: formed directly (operands of the cross products swapped; the sign of  \
                       the Tw terms rides on the FMAs' source modifiers) -- exact, and no sign flip after the reduction */ \
                    const float nk0 = fmaf(l2, dsy, -(l1 * dp2)), nk1 = fmaf(l0, dp2, -(l2 * dsx)), nk2 = fmaf(l1, dsx, -(l0 * dsy)); \
                    const float nl0 = fmaf(dp2, k1, -(dsy * k2)), nl1 = fmaf(dsx, k2, -(dp2 * k0)), nl2 = fmaf(dsy, k0, -(dsx * k1)); \
                    g[3] = nk0; g[4] = nk1; g[5] = nk2;                                                               \
                    g[6] = nl0; g[7] = nl1; g[8] = nl2;                                                               \
                    g[9] = fmaf(-pxf, nk0, fmaf(-pyf, nl0, d_zr * s0));                                               \
                    g[10] = fmaf(-pxf, nk1, fmaf(-pyf, nl1, d_zr * s1));                                              \
                    g[11] = fmaf(-pxf, nk2, fmaf(-pyf, nl2, d_zr)) + d_zl;                                            \
                    g[15] = d_op;                                                                                     \
                    /* each row reduces ITS splat into the splat's LDS accumulators */                                \
                    GS2D_EXP_BUTTERFLY                                                                                \
                    if (NG == 8) { /* two values per lane: component half_idx (always an accumulator) and 8 + half_idx */ \
                        float tot2;                                                                                   \
                        if (ANY_DN) { g[12] = d_w * dn0; g[13] = d_w * dn1; g[14] = d_w * dn2; }                      \
                        else { g[12] = 0.f; g[13] = 0.f; g[14] = 0.f; }                                               \
                        reduce16_half(g, lane, tot, tot2);                                                            \
                        tot = merge_twin_halves(tot, twin_m_ & lo_half_m, twin_m_ & ~lo_half_m);                      \
                        tot2 = merge_twin_halves(tot2, twin_m_ & lo_half_m, twin_m_ & ~lo_half_m);                    \
                        GS2D_BWD_LDS_ACCUM(J, tot)                                                                    \
                        if (acc_comp_hi >= 0) {                                                                       \
                            GS2D_BWD_LDS_ACCUM2(J, tot2)                                                              \
                        } else if (ANY_DN && tot2 != 0.f) /* components 12-14 = the normal: record words 3-5 */       \
                            atomicAdd(grad_rec + (size_t)(wb.pn[J] & 0x0FFFFFFFu) * GS2D_GRAD_FLOATS + (half_idx - 1), tot2); \
                    } else {                                                                                          \
                    if (ANY_DN) {                                                                                     \
                        g[12] = d_w * dn0; g[13] = d_w * dn1; g[14] = d_w * dn2;                                      \
                        tot = reduce16_row(g, lane);                                                                  \
                    } else {                                                                                          \
                        g[12] = 0.f; g[13] = 0.f; g[14] = 0.f;                                                        \
                        tot = reduce16_row_z(g, lane);                                                                \
                    }                                                                                                 \
                    /* LDS float atomics run at a few lanes per clock: rows/components that sum to +-0 skip them */    \
                    if (DET) { /* one row at a time: lanes of different rows may hold the same (splat, component) */ \
                        _Pragma("unroll") for (int r_ = 0; r_ < 4; r_++)                                              \
                            if (row == r_ && acc_comp >= 0 && tot != 0.f) wb.acc()[J * NACC + acc_comp] += tot;        \
                    } else if (acc_comp >= 0) {                                                                       \
                        GS2D_BWD_LDS_ACCUM(J, tot)                                                                    \
                    } else if (ANY_DN && tot != 0.f)                                                                  \
                        atomicAdd(grad_rec + (size_t)(wb.pn[J] & 0x0FFFFFFFu) * GS2D_GRAD_FLOATS + (slot - 9), tot); \
                    }                                                                                                 \
                    }                                                                                                 \
                    if (ballot64(d_t != 0.f) != 0) {                                                                  \
                        const float g_mx = NG == 8 ? half_sum_to_lane7(d_t * d0) : row_sum_to_lane15(d_t * d0);       \
                        const float g_my = NG == 8 ? half_sum_to_lane7(d_t * d1) : row_sum_to_lane15(d_t * d1);       \
                        if (DET) {                                                                                    \
                            _Pragma("unroll") for (int r_ = 0; r_ < 4; r_++)                                          \
                                if (row == r_ && li == 15 && (g_mx != 0.f || g_my != 0.f)) {                          \
                                    wb.acc()[J * NACC + 16] += g_mx; wb.acc()[J * NACC + 17] += g_my;                   \
                                }                                                                                     \
                        } else if ((NG == 8 ? (li & 7) == 7 : li == 15) && (g_mx != 0.f || g_my != 0.f)) {            \
                            float* dst = POSE ? dense_m2d + (size_t)(wb.pn[J] & 0x0FFFFFFFu) * 2                      \
                                              : grad_rec + (size_t)(wb.pn[J] & 0x0FFFFFFFu) * GS2D_GRAD_FLOATS + 16;  \
                            atomicAdd(dst, g_mx); atomicAdd(dst + 1, g_my);                                           \
                        }                                                                                             \
                    }                                                                                                 \
                }                                                                                                     \
            }                                                                                                         \
            J = jnn;                                                                                                  \
            FJ_ = fnn_;                                                                                               \
            if (++t >= trips) break;                                                                                  \
        }
        for (;;) {
            GS2D_BWD_STEP(ga0, ga1, ga2, j, gb0, gb1, gb2, jx, fj)
            GS2D_BWD_STEP(gb0, gb1, gb2, jx, ga0, ga1, ga2, j, fjx)
        }
#undef GS2D_BWD_STEP
        // flush: every touched splat of the batch goes to its global record once, four splats (one per row) per pass
        // two passes (eight splats) per iteration so the LDS round trips of one pass hide behind the other
        if (DET) {
            // every staged (instance, quadrant) pair is stored to its own slot, zeros included, exactly once
            for (int f0 = 64 - fill; f0 < 64; f0 += 4) {
                const int fa = f0 + row;
                if (fa < 64) {
                    float* dst = det_slots + ((size_t)(range.x + __float_as_uint(wb.q(3)[fa].w)) * 4 + wave) * GS2D_GRAD_FLOATS;
#pragma unroll
                    for (int c = li; c < NACC; c += 16) {
                        // accumulator c -> offset in the record: 0-2 colour, 3-11 dT, 12 opacity, 13-15 normal, 16-17 mean2D
                        const int off = c < 3 ? c : (c < 12 ? c + 3 : (c == 12 ? 15 : (c < 16 ? c - 10 : c)));
                        dst[off] = wb.acc()[fa * NACC + c];
                        wb.acc()[fa * NACC + c] = 0.f;
                    }
                }
            }
        } else if (POSE) {
            // sixteen splats per pass: lane l adds up the four rows' accumulators (l & 3) of slot f0 + l / 4 and flushes the sum
            // into the dense float4 of its Gaussian
            for (int f0 = 64 - fill; f0 < 64; f0 += 16) {
                const int fs = f0 + (lane >> 2), c = lane & 3;
                if (fs < 64 && c < GS2D_ACC_POSE) {
                    float* pa = &wb.acc()[fs * GS2D_ACC_POSE_ROWS + c];
                    const float v0 = pa[0], v1 = pa[GS2D_ACC_POSE], v2 = pa[2 * GS2D_ACC_POSE], v3 = pa[3 * GS2D_ACC_POSE];
                    const float va = (v0 + v1) + (v2 + v3);
                    if (v0 != 0.f || v1 != 0.f || v2 != 0.f || v3 != 0.f) {
                        pa[0] = 0.f; pa[GS2D_ACC_POSE] = 0.f; pa[2 * GS2D_ACC_POSE] = 0.f; pa[3 * GS2D_ACC_POSE] = 0.f;
                        if (va != 0.f) atomicAdd(grad_rec + (size_t)(wb.pn[fs] & 0x0FFFFFFFu) * 4 + c, va);
                    }
                }
            }
        } else {
        const int flush_off = li < 3 ? li : (li < 12 ? li + 3 : 15);  // accumulator li -> offset in the gradient record
        const bool flush_lane = li < GS2D_ACC;
        for (int f0 = 64 - fill; GS2D_EXP_FLUSH(f0 < 64); f0 += 8) {
            const int fa = f0 + row, fb = f0 + 4 + row;  // slots >= 64 do not exist
            float* pa = &wb.acc()[(fa & 63) * GS2D_ACC + (flush_lane ? li : 0)];
            float* pb = &wb.acc()[(fb & 63) * GS2D_ACC + (flush_lane ? li : 0)];
            const float va = *pa, vb = *pb;
            const uint32_t ida = wb.pn[fa & 63] & 0x0FFFFFFFu, idb = wb.pn[fb & 63] & 0x0FFFFFFFu;
            if (fa < 64 && flush_lane && va != 0.f) {
                *pa = 0.f;
                GS2D_EXP_ATOMIC(atomicAdd(grad_rec + (size_t)ida * GS2D_GRAD_FLOATS + flush_off, va);)
            }
            if (fb < 64 && flush_lane && vb != 0.f) {
                *pb = 0.f;
                GS2D_EXP_ATOMIC(atomicAdd(grad_rec + (size_t)idb * GS2D_GRAD_FLOATS + flush_off, vb);)
            }
        }
        }
    }
    };
    if (any_dn) run(std::true_type{}); else run(std::false_type{});
    GS2D_PROF_END(1)
}

}  // namespace

namespace gs2d {

void launch_blend_fwd(int W, int H, int K, const BlendFwdFrame* frames, const float* bg, int use_sa, size_t zero_n, int sort_cap,
                      int write_keys, hipStream_t s)
{
    const int gx = (W + GS2D_TILE - 1) / GS2D_TILE, gy = (H + GS2D_TILE - 1) / GS2D_TILE;
    const size_t plane = (size_t)gx * gy * GS2D_TILE_PIX;
    const int bpf = GS2D_XCDS * ((gx * gy + GS2D_XCDS - 1) / GS2D_XCDS);  // a multiple of 8: blockIdx % 8 picks the same XCD in every frame
    size_t lds = 4 * sizeof(FwdBatch);
    if (sort_cap > 0 && gs2d_fused_sort_lds(sort_cap) > lds) lds = gs2d_fused_sort_lds(sort_cap);  // (28 KB at 1536 / 3072: still 5 workgroups per CU)
    if (K == 1) {
        if (use_sa)
            hipLaunchKernelGGL((blend_fwd_kernel<true, false>), dim3(bpf), dim3(256), lds, s, W, H, gx, gx * gy, bpf, bg, plane, zero_n,
                               sort_cap, write_keys, frames[0]);
        else
            hipLaunchKernelGGL((blend_fwd_kernel<false, false>), dim3(bpf), dim3(256), lds, s, W, H, gx, gx * gy, bpf, bg, plane, zero_n,
                               sort_cap, write_keys, frames[0]);
        return;
    }
    BlendFwdBatch b;
    for (int k = 0; k < K; k++) b.f[k] = frames[k];
    for (int k = K; k < GS2D_MAX_BATCH; k++) b.f[k] = frames[0];
    if (use_sa)
        hipLaunchKernelGGL((blend_fwd_kernel<true, true>), dim3(bpf * K), dim3(256), lds, s, W, H, gx, gx * gy, bpf, bg, plane, zero_n,
                           sort_cap, write_keys, b);
    else
        hipLaunchKernelGGL((blend_fwd_kernel<false, true>), dim3(bpf * K), dim3(256), lds, s, W, H, gx, gx * gy, bpf, bg, plane, zero_n,
                           sort_cap, write_keys, b);
}

void launch_blend_bwd(int W, int H, int K, const BlendBwdFrame* frames, const float* bg, int use_sa, float* clear12, int clear_n,
                      hipStream_t s)
{
    const int gx = (W + GS2D_TILE - 1) / GS2D_TILE, gy = (H + GS2D_TILE - 1) / GS2D_TILE;
    const size_t plane = (size_t)gx * gy * GS2D_TILE_PIX;
    const int bpf = GS2D_XCDS * ((gx * gy + GS2D_XCDS - 1) / GS2D_XCDS);
    const bool det = frames[0].det_slots != nullptr;
    if (K == 1 && !det && frames[0].dense_m2d != nullptr) {  // pose-only instantiation
        if (use_sa)
            hipLaunchKernelGGL((blend_bwd_kernel<true, false, false, true>), dim3(bpf), dim3(256), 0, s, W, H, gx, gx * gy, bpf, bg,
                               plane, clear12, clear_n, frames[0]);
        else
            hipLaunchKernelGGL((blend_bwd_kernel<false, false, false, true>), dim3(bpf), dim3(256), 0, s, W, H, gx, gx * gy, bpf, bg,
                               plane, clear12, clear_n, frames[0]);
        return;
    }
#define GS2D_LAUNCH_BWD(SA, DET, BATCH, GRID, ARGS)                                                                       \
    hipLaunchKernelGGL((blend_bwd_kernel<SA, DET, BATCH>), dim3(GRID), dim3(256), 0, s, W, H, gx, gx * gy, bpf, bg, plane,  \
                       clear12, clear_n, ARGS)
    if (K == 1) {
        if (det) { if (use_sa) GS2D_LAUNCH_BWD(true, true, false, bpf, frames[0]); else GS2D_LAUNCH_BWD(false, true, false, bpf, frames[0]); }
        else { if (use_sa) GS2D_LAUNCH_BWD(true, false, false, bpf, frames[0]); else GS2D_LAUNCH_BWD(false, false, false, bpf, frames[0]); }
        return;
    }
    BlendBwdBatch b;
    for (int k = 0; k < K; k++) b.f[k] = frames[k];
    for (int k = K; k < GS2D_MAX_BATCH; k++) b.f[k] = frames[0];
    // (the deterministic variant is single-frame only: gs2d_backward_batch refuses it)
    if (use_sa) GS2D_LAUNCH_BWD(true, false, true, bpf * K, b); else GS2D_LAUNCH_BWD(false, false, true, bpf * K, b);
#undef GS2D_LAUNCH_BWD
}

}  // namespace gs2d

#ifdef GS2D_PROFILE_WAVES
extern "C" int gs2d_debug_read_wave_profile(int kernel, unsigned long long* host_out, size_t n_words)
{
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wave_prof), n_words * 8, (size_t)kernel * 4 * 8192 * 4 * 8,
                               hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
