// Workgroup-level scan primitives of the binning kernels (a header of their own since the per-tile depth sort, which uses
// them, is also compiled into the forward blend kernel, gs2d_tile_sort.h).
// Tried and dropped in round 3: letting the LAST workgroup of preprocess_fwd_kernel run scan_blocksums_body (ticket taken
// with an atomic after an agent-scope release fence) instead of a single-workgroup kernel of its own.  On this chip an
// agent-scope release writes the XCD's dirty L2 lines back, and 1954 workgroups that have just written 76 MB of records each
// paid for it: preprocess 17 -> 84 us.
#pragma once
#include "gs2d_common.h"

namespace {

// ---------------------------------------------------------------- device-wide inclusive scan (u32)
// 3 kernels: per-block reduce, single-block scan of block sums, per-block scan + offset.
constexpr int SCAN_T = 256;
constexpr int SCAN_PER_T = GS2D_SCAN_ITEMS / SCAN_T;  // 4

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}

// block-wide inclusive scan of one value per thread (256 threads); returns inclusive value, total in *total.
__device__ __forceinline__ uint32_t block_incl_scan(uint32_t v, uint32_t* total)
{
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan(v, lane);
    __syncthreads();  // protect wsum reuse across calls
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < 4; w++)
        if (w < wave) base += wsum[w];
    if (total) *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return inc + base;
}

// exclusive scan of block sums in place (single workgroup); writes the grand total.  Thread i owns the K = ceil(n/256)
// consecutive sums [i K, (i+1) K): one round of loads, one workgroup scan of the 256 partial sums, one round of stores
// (the chunk-by-chunk loop this replaces paid a load -> scan -> store -> barrier chain per 256 sums: 6.5 us for the 1954
// sums of 500k Gaussians, on the path to the host's num_rendered).
__device__ __forceinline__ void scan_blocksums_body(uint32_t* __restrict__ block_sums, int nblocks, uint32_t* __restrict__ total_out,
                                                    uint32_t* __restrict__ total_host)
{
    const int K = (nblocks + SCAN_T - 1) / SCAN_T;
    const int i0 = min(nblocks, (int)threadIdx.x * K), i1 = min(nblocks, i0 + K);
    constexpr int KR = 8;  // sums kept in registers between the two rounds (more: re-read, they are cache hits)
    uint32_t v[KR];
    uint32_t mysum = 0;
#pragma unroll
    for (int j = 0; j < KR; j++) { v[j] = i0 + j < i1 ? block_sums[i0 + j] : 0u; mysum += v[j]; }
    for (int i = i0 + KR; i < i1; i++) mysum += block_sums[i];
    uint32_t total;
    uint32_t running = block_incl_scan(mysum, &total) - mysum;
    // pinned host word polled by the caller: published before the prefix is written back
    if (threadIdx.x == 0 && total_host) __hip_atomic_store(total_host, total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0 && total_out) *total_out = total;
#pragma unroll
    for (int j = 0; j < KR; j++)
        if (i0 + j < i1) { block_sums[i0 + j] = running; running += v[j]; }
    for (int i = i0 + KR; i < i1; i++) { const uint32_t x = block_sums[i]; block_sums[i] = running; running += x; }
}

}  // namespace
