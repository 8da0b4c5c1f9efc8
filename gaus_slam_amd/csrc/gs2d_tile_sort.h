// Per-tile depth sort, shared by tile_depth_sort_kernel (gs2d_binning.hip) and the forward blend kernel, whose workgroups
// sort their own tile's list as their first phase when it fits the LDS they hold anyway (gs2d_blend.hip).
#pragma once
#include "gs2d_common.h"
#include "gs2d_scan.h"

namespace {

// ---------------------------------------------------------------- per-tile depth sort (LDS)
// After the global passes have binned the pairs by tile id (stable, so each tile's segment is still in Gaussian
// order), one workgroup per tile sorts its segment by the 32 depth bits with a stable 4-pass LSD radix sort that
// lives entirely in LDS ("LDS-staged per-tile splat lists").  The final order is identical to a global stable
// sort on (tile | depth): LSD radix = sort by the low key first, then stably by the high key; here the high-key
// pass simply ran first because the two keys are independent and the segment boundaries are known.
// Segments longer than the LDS capacity take the same code path on global ping-pong buffers (flat pointers).
__device__ void sort_segment_by_depth(uint32_t* ka, uint32_t* va, uint32_t* kb, uint32_t* vb, int n,
                                      uint32_t (*wcnt)[256])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = ((n + 255) / 256) * 64;  // contiguous elements per wave, multiple of 64
    const int beg = wave * chunk, end = min(n, beg + chunk);
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 8 * pass;
        for (int i = threadIdx.x; i < 4 * 256; i += 256) (&wcnt[0][0])[i] = 0;
        __syncthreads();
        for (int i = beg + lane; i < end; i += 64) atomicAdd(&wcnt[wave][(ka[i] >> shift) & 255u], 1u);
        __syncthreads();
        {
            const int d = threadIdx.x;
            const uint32_t c0 = wcnt[0][d], c1 = wcnt[1][d], c2 = wcnt[2][d], c3 = wcnt[3][d];
            const uint32_t tot = c0 + c1 + c2 + c3;
            const uint32_t excl = block_incl_scan(tot, nullptr) - tot;
            wcnt[0][d] = excl; wcnt[1][d] = excl + c0; wcnt[2][d] = excl + c0 + c1; wcnt[3][d] = excl + c0 + c1 + c2;
        }
        __syncthreads();
        for (int i0 = beg; i0 < end; i0 += 64) {
            const int i = i0 + lane;
            const bool valid = i < end;
            const uint32_t k = valid ? ka[i] : 0u;
            const uint32_t v = valid ? va[i] : 0u;
            const uint32_t d = (k >> shift) & 255u;
            uint64_t peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const uint64_t vote = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? vote : ~vote;
            }
            const uint32_t before = wcnt[wave][d];
            __builtin_amdgcn_wave_barrier();
            if (valid && (peers & lt_mask) == 0) wcnt[wave][d] = before + (uint32_t)__popcll(peers);
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                const uint32_t dst = before + (uint32_t)__popcll(peers & lt_mask);
                kb[dst] = k;
                vb[dst] = v;
            }
        }
        __syncthreads();
        uint32_t* t = ka; ka = kb; kb = t;
        t = va; va = vb; vb = t;
    }
}

// The same sort on INDICES: the depth bits stay where they are (ka), what moves through the four passes is a uint16 position
// (ia -> ib -> ia ...), 8 bytes of LDS per element instead of 16.  The forward blend kernel's workgroups hold the LDS for 1536
// elements of the pair sort above; with this one they sort lists of up to 3072 themselves (phase -1), which covers the scenes
// whose lists are too long for the pair sort but not rare: at 876x584 / 2M Gaussians (mean list 1965) the stand-alone sort
// kernel that used to take them cost 77 us.  One more dependent LDS read per element and pass (the key behind the index).
__device__ void sort_segment_by_depth_idx(const uint32_t* ka, uint16_t* ia, uint16_t* ib, int n, uint32_t (*wcnt)[256])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = ((n + 255) / 256) * 64;
    const int beg = wave * chunk, end = min(n, beg + chunk);
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 8 * pass;
        for (int i = threadIdx.x; i < 4 * 256; i += 256) (&wcnt[0][0])[i] = 0;
        __syncthreads();
        for (int i = beg + lane; i < end; i += 64) atomicAdd(&wcnt[wave][(ka[ia[i]] >> shift) & 255u], 1u);
        __syncthreads();
        {
            const int d = threadIdx.x;
            const uint32_t c0 = wcnt[0][d], c1 = wcnt[1][d], c2 = wcnt[2][d], c3 = wcnt[3][d];
            const uint32_t tot = c0 + c1 + c2 + c3;
            const uint32_t excl = block_incl_scan(tot, nullptr) - tot;
            wcnt[0][d] = excl; wcnt[1][d] = excl + c0; wcnt[2][d] = excl + c0 + c1; wcnt[3][d] = excl + c0 + c1 + c2;
        }
        __syncthreads();
        for (int i0 = beg; i0 < end; i0 += 64) {
            const int i = i0 + lane;
            const bool valid = i < end;
            const uint32_t id = valid ? ia[i] : 0u;
            const uint32_t d = valid ? (ka[id] >> shift) & 255u : 0u;
            uint64_t peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const uint64_t vote = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? vote : ~vote;
            }
            const uint32_t before = wcnt[wave][d];
            __builtin_amdgcn_wave_barrier();
            if (valid && (peers & lt_mask) == 0) wcnt[wave][d] = before + (uint32_t)__popcll(peers);
            __builtin_amdgcn_wave_barrier();
            if (valid) ib[before + (uint32_t)__popcll(peers & lt_mask)] = (uint16_t)id;
        }
        __syncthreads();
        uint16_t* t = ia; ia = ib; ib = t;
    }
}

// packed != 0: the segment holds (depth bits, id) pairs in the 8-byte key slots (output of bin_scatter_kernel);
// packed == 0: 64-bit keys + separate ids (output of the generic radix passes).  The sorted ids always land in `vals`
// (the point list); the full 64-bit keys are materialised only when write_keys != 0 (debug / parity tests).
// dyn: 4 * cap words of LDS (ka, va, kb, vb); wcnt: 4 x 256 words of LDS; tile: the tile this workgroup sorts
__device__ __forceinline__ void
tile_depth_sort_body(int tile, uint32_t* dyn, uint32_t (*wcnt)[256], const uint2* __restrict__ ranges, uint64_t* __restrict__ keys,
                     uint32_t* __restrict__ vals, uint64_t* __restrict__ keys_alt, uint32_t* __restrict__ vals_alt, int cap, int packed,
                     int write_keys)
{
    const uint2 r = ranges[tile];
    const int n = (int)(r.y - r.x);
    if (n <= 0) return;
    uint64_t* kseg = keys + r.x;
    uint32_t* vseg = vals + r.x;
    const uint2* pseg = reinterpret_cast<const uint2*>(kseg);
    const uint64_t hi = (uint64_t)(uint32_t)tile << 32;
    if (n <= cap) {
        uint32_t *ka = dyn, *va = dyn + cap, *kb = dyn + 2 * cap, *vb = dyn + 3 * cap;
        for (int i = threadIdx.x; i < n; i += 256) {
            if (packed) { const uint2 p = pseg[i]; ka[i] = p.x; va[i] = p.y; }
            else { ka[i] = (uint32_t)kseg[i]; va[i] = vseg[i]; }
        }
        __syncthreads();
        if (n > 1) sort_segment_by_depth(ka, va, kb, vb, n, wcnt);  // 4 passes: result back in ka / va
        for (int i = threadIdx.x; i < n; i += 256) {
            vseg[i] = va[i];
            if (write_keys) kseg[i] = hi | ka[i];
        }
    } else {
        // oversized list: same algorithm on global ping-pong arrays carved from the segment's own scratch slots:
        // ka, va = the two halves of the segment's keys_alt slots, kb = its vals_alt slots, vb = its point-list slots
        uint32_t* ka = reinterpret_cast<uint32_t*>(keys_alt + r.x);
        uint32_t* va = ka + n;
        uint32_t* kb = vals_alt + r.x;
        uint32_t* vb = vseg;
        if (packed) {
            for (int i = threadIdx.x; i < n; i += 256) { const uint2 p = pseg[i]; ka[i] = p.x; va[i] = p.y; }
        } else {
            // vseg doubles as vb, so the ids are copied out first
            for (int i = threadIdx.x; i < n; i += 256) { ka[i] = (uint32_t)kseg[i]; va[i] = vseg[i]; }
        }
        __syncthreads();
        sort_segment_by_depth(ka, va, kb, vb, n, wcnt);  // even number of passes: result in ka / va
        for (int i = threadIdx.x; i < n; i += 256) {
            vseg[i] = va[i];
            if (write_keys) kseg[i] = hi | ka[i];
        }
    }
}

// tile_depth_sort_body for the packed pairs of the single-pass binning with the index sort: lists of up to `cap` elements in
// 8 * cap bytes of LDS (dyn: cap words of depth bits, then two uint16 arrays of cap positions), longer ones through global scratch
// (tile_depth_sort_body's second half).  cap <= 4096 (the positions are uint16).
__device__ __forceinline__ void
tile_depth_sort_idx_body(int tile, uint32_t* dyn, uint32_t (*wcnt)[256], const uint2* __restrict__ ranges, uint64_t* __restrict__ keys,
                         uint32_t* __restrict__ vals, uint64_t* __restrict__ keys_alt, uint32_t* __restrict__ vals_alt, int cap,
                         int write_keys)
{
    const uint2 r = ranges[tile];
    const int n = (int)(r.y - r.x);
    if (n <= 0) return;
    if (n > cap) {
        tile_depth_sort_body(tile, dyn, wcnt, ranges, keys, vals, keys_alt, vals_alt, /*cap=*/0, /*packed=*/1, write_keys);
        return;
    }
    uint64_t* kseg = keys + r.x;
    uint32_t* vseg = vals + r.x;
    const uint2* pseg = reinterpret_cast<const uint2*>(kseg);
    uint32_t* ka = dyn;
    uint16_t* ia = reinterpret_cast<uint16_t*>(dyn + cap);
    uint16_t* ib = ia + cap;
    for (int i = threadIdx.x; i < n; i += 256) { ka[i] = pseg[i].x; ia[i] = (uint16_t)i; }
    __syncthreads();
    if (n > 1) sort_segment_by_depth_idx(ka, ia, ib, n, wcnt);  // four passes: the sorted positions are back in ia
    // the ids are still where the binning left them: gathered from the segment's own pairs (just read: cache-warm)
    for (int i = threadIdx.x; i < n; i += 256) vseg[i] = pseg[ia[i]].y;
    if (write_keys) {  // (debug / parity tests: the full sorted keys, over the pairs -- after every id has been read)
        __syncthreads();
        const uint64_t hi = (uint64_t)(uint32_t)tile << 32;
        for (int i = threadIdx.x; i < n; i += 256) kseg[i] = hi | ka[ia[i]];
    }
}

}  // namespace
