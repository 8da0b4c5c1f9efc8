// Tile binning: prefix sum of per-Gaussian tile counts, (tile|depth) key generation, stable LSD
// radix sort of the 64-bit keys, per-tile ranges.
//
// Result contract (bit-exact with the reference pipeline): point_list ordered by
// (tile id, float bits of view depth, Gaussian index) -- rasterizer_impl.cu:70-138,283-323 --
// i.e. what cub::DeviceRadixSort::SortPairs (stable) yields on keys (tile << 32) | depth_bits whose
// unsorted order is Gaussian-major.  The implementation is a hand-written wave64 radix sort:
// 8-bit digits, per-workgroup digit histograms -> device scan -> stable scatter using ballot-based
// match-any ranking (64-wide), no CUB/rocPRIM.
#include "gs2d_common.h"
#include "gs2d_scan.h"
#include "gs2d_tile_sort.h"

#include <stdlib.h>

namespace {

__global__ void __launch_bounds__(SCAN_T) scan_reduce_kernel(const uint32_t* __restrict__ in, int n, uint32_t* __restrict__ block_sums)
{
    const int base = blockIdx.x * GS2D_SCAN_ITEMS + threadIdx.x * SCAN_PER_T;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_PER_T; i++)
        if (base + i < n) s += in[base + i];
    uint32_t total;
    block_incl_scan(s, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(SCAN_T) scan_blocksums_kernel(uint32_t* __restrict__ block_sums, int nblocks, uint32_t* __restrict__ total_out,
                                                                uint32_t* __restrict__ total_host)
{
    scan_blocksums_body(block_sums, nblocks, total_out, total_host);
}
// Batched forms (gs2d_forward_batch): blockIdx.y = frame, per-frame pointers from a by-value table in the kernel arguments;
// the bodies are the single-frame kernels' own.

__global__ void __launch_bounds__(SCAN_T) scan_apply_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int n,
                                                            const uint32_t* __restrict__ block_offsets)
{
    const int base = blockIdx.x * GS2D_SCAN_ITEMS + threadIdx.x * SCAN_PER_T;
    uint32_t v[SCAN_PER_T];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_PER_T; i++) {
        v[i] = base + i < n ? in[base + i] : 0;
        s += v[i];
    }
    const uint32_t inc = block_incl_scan(s, nullptr);
    uint32_t run = block_offsets[blockIdx.x] + inc - s;
#pragma unroll
    for (int i = 0; i < SCAN_PER_T; i++) {
        run += v[i];
        if (base + i < n) out[base + i] = run;
    }
}

// ---------------------------------------------------------------- key generation
__device__ __forceinline__ int f2i_sat(float v)
{
    if (!(v == v)) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (int)(-2147483647 - 1);
    return (int)v;
}

// rasterizer_impl.cu:70-111; one thread per Gaussian, tiles emitted row-major (y outer, x inner).  The rectangle comes
// from the preprocess kernel (the reference's, or its intersection with the footprint bound), not re-derived from radii.
__device__ __forceinline__ void
duplicate_body(int P, const ushort4* __restrict__ rect, const float* __restrict__ depths,
               const uint32_t* __restrict__ tiles_touched, const uint32_t* __restrict__ block_sums, int scanned,
               uint32_t* __restrict__ point_offsets, int gx, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
               uint32_t capacity, uint32_t* __restrict__ total_host, uint32_t* __restrict__ total_dev = nullptr)
{
    // The prefix sum of tiles_touched (rasterizer_impl.cu:283) is finished here: the preprocess kernel left one sum per
    // 256 Gaussians; this workgroup (the same 256 Gaussians) adds up the sums of the blocks before it -- at most a few
    // thousand cached words, eight per thread at 500k Gaussians -- and its own inclusive scan on top.  No scan kernel (until
    // round 3 a single-workgroup kernel scanned the block sums first and stored the total for the host: 4-7 us plus a
    // dependent dispatch on the critical path of every forward; it remains for the callers that need the total up front,
    // `scanned`).  The last block's offset plus its own sum IS the total: it goes to the host from here.
    uint32_t before = 0;
    if (!scanned)
        for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) before += block_sums[b];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int ci = min(idx, P - 1);
    const uint32_t mine = idx < P ? tiles_touched[ci] : 0u;
    // everything the emit loop needs is requested before the workgroup scan, whose two barriers then hide the latency
    const ushort4 r = rect[ci];
    const uint32_t dbits = __float_as_uint(depths[ci]);
    uint32_t block_offset, own_total;
    if (scanned) block_offset = block_sums[blockIdx.x];
    else (void)block_incl_scan(before, &block_offset);
    const uint32_t incl = block_offset + block_incl_scan(mine, &own_total);
    if (!scanned && total_host != nullptr && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        // a relaxed system-scope store into coherent pinned memory: the host polls the word (0xFFFFFFFF = not yet)
        const uint32_t total = block_offset + own_total;
        if (total_dev != nullptr) *total_dev = total;  // for the kernels behind this one (DevBin): visible at the kernel boundary
        __hip_atomic_store(total_host, total == 0xFFFFFFFFu ? 0xFFFFFFFEu : total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (idx >= P) return;
    point_offsets[idx] = incl;  // inclusive offsets, as the reference's InclusiveSum leaves them
    if (mine == 0u) return;
    uint32_t off = incl - mine;
    if (incl > capacity) {  // the caller's guess was too small: it will launch again with room for everything (rare)
        for (uint32_t y = r.y; y < r.w; y++)
            for (uint32_t x = r.x; x < r.z; x++) {
                if (off < capacity) { keys[off] = ((uint64_t)(y * (uint32_t)gx + x) << 32) | dbits; vals[off] = (uint32_t)idx; }
                off++;
            }
        return;
    }
    for (uint32_t y = r.y; y < r.w; y++)
        for (uint32_t x = r.x; x < r.z; x++) {
            const uint64_t key = ((uint64_t)(y * (uint32_t)gx + x) << 32) | dbits;
            keys[off] = key;
            vals[off] = (uint32_t)idx;
            off++;
        }
}
__global__ void __launch_bounds__(256)
duplicate_kernel(int P, const ushort4* __restrict__ rect, const float* __restrict__ depths,
                 const uint32_t* __restrict__ tiles_touched, const uint32_t* __restrict__ block_sums, int scanned,
                 uint32_t* __restrict__ point_offsets, int gx, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                 uint32_t capacity, uint32_t* __restrict__ total_host, uint32_t* __restrict__ total_dev)
{
    duplicate_body(P, rect, depths, tiles_touched, block_sums, scanned, point_offsets, gx, keys, vals, capacity, total_host, total_dev);
}
__global__ void __launch_bounds__(256) duplicate_batch_kernel(int P, int gx, const gs2d::BinFrames tab)
{
    const gs2d::BinFrame& f = tab.f[blockIdx.y];
    duplicate_body(P, f.rect, f.depths, f.tiles_touched, f.block_sums, 0, f.point_offsets, gx, f.keys_unsorted, f.vals_unsorted,
                   f.capacity, f.total_host);
}

// ---------------------------------------------------------------- radix sort pass (8-bit digit)
constexpr int SORT_T = 256;
constexpr int SORT_PER_T = GS2D_SORT_ITEMS / SORT_T;  // 8 rounds of 64 per wave
constexpr int WAVE_ITEMS = GS2D_SORT_ITEMS / 4;       // 512 consecutive elements per wave

// element index handled by (wave, round, lane): consecutive within a wave-round, so that
// (wave, round, lane) lexicographic order == element order (needed for stability).
__device__ __forceinline__ int sort_elem(int block, int wave, int round, int lane)
{
    return block * GS2D_SORT_ITEMS + wave * WAVE_ITEMS + round * 64 + lane;
}

// hist[digit * nblocks + block] = number of keys of this block with that digit
__global__ void __launch_bounds__(SORT_T)
radix_hist_kernel(const uint64_t* __restrict__ keys, int n, int shift, uint32_t* __restrict__ hist, int nblocks)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < SORT_PER_T; r++) {
        const int i = sort_elem(blockIdx.x, wave, r, lane);
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// `offs` = exclusive scan of hist (same indexing).  Stable scatter.
__global__ void __launch_bounds__(SORT_T)
radix_scatter_kernel(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint64_t* __restrict__ keys_out,
                     uint32_t* __restrict__ vals_out, int n, int shift, const uint32_t* __restrict__ offs_incl, int nblocks)
{
    __shared__ uint32_t wcnt[4][256];  // per-wave running digit counts, later exclusive bases
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 256; i += SORT_T) (&wcnt[0][0])[i] = 0;
    __syncthreads();

    uint64_t k[SORT_PER_T];
    uint32_t rank[SORT_PER_T];  // rank among same-digit keys within this wave (stable)
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < SORT_PER_T; r++) {
        const int i = sort_elem(blockIdx.x, wave, r, lane);
        const bool valid = i < n;
        k[r] = valid ? keys_in[i] : 0;
        const uint32_t d = (uint32_t)(k[r] >> shift) & 255u;
        // match-any over the 8 digit bits: peers = lanes holding the same digit (valid lanes only)
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint64_t vote = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? vote : ~vote;
        }
        const uint32_t before = wcnt[wave][d];  // same-wave LDS ops are ordered: read happens before the leader's write
        rank[r] = before + (uint32_t)__popcll(peers & lt_mask);
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lt_mask) == 0) wcnt[wave][d] = before + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // turn per-wave totals into exclusive bases across waves and add the global offset of (digit, block)
    {
        const int d = threadIdx.x;
        const size_t hidx = (size_t)d * nblocks + blockIdx.x;
        // offs_incl is the INCLUSIVE scan of hist; exclusive start = inclusive - own count
        const uint32_t c0 = wcnt[0][d], c1 = wcnt[1][d], c2 = wcnt[2][d], c3 = wcnt[3][d];
        const uint32_t start = offs_incl[hidx] - (c0 + c1 + c2 + c3);
        wcnt[0][d] = start;
        wcnt[1][d] = start + c0;
        wcnt[2][d] = start + c0 + c1;
        wcnt[3][d] = start + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_PER_T; r++) {
        const int i = sort_elem(blockIdx.x, wave, r, lane);
        if (i < n) {
            const uint32_t d = (uint32_t)(k[r] >> shift) & 255u;
            const uint32_t dst = wcnt[wave][d] + rank[r];
            keys_out[dst] = k[r];
            vals_out[dst] = vals_in[i];
        }
    }
}

// ---------------------------------------------------------------- single-pass binning by tile id
// Counting sort on the whole tile id (up to GS2D_BIN_MAX_TILES bins) in ONE pass: per-workgroup tile histogram ->
// scan -> stable scatter.  Replaces ceil(bits/8) passes of the generic 8-bit sort and yields the tile ranges for free
// (they are the scanned histogram's tile boundaries).  The scan of the tiles x blocks counters is split the way the
// data lies: one wave per tile scans that tile's row of block counts in place (bin_row_scan_kernel, coalesced, no
// cross-workgroup step), and every scatter workgroup derives the tile bases from the <= 4096 row totals itself -- one
// short launch instead of the three of the generic device scan (15 us for 325k counters at 640x480 / 500k).
constexpr int BIN_T = 256;
constexpr int BIN_ITEMS = GS2D_BIN_ITEMS;   // instances per workgroup
constexpr int BIN_WAVE_ITEMS = BIN_ITEMS / 4;
#ifndef GS2D_DEV_BIN  // timing experiments only (wrong lists; always with 2): 1 no counter-table read, 2 linear stores, 4 no ranking, 8 no counting
#define GS2D_DEV_BIN 0
#endif

// Which block of pairs a binning workgroup takes.  Workgroups go to the eight XCDs round robin and each XCD has its own L2; block
// b's counter of tile t sits right behind block b - 1's, and so do their pairs of the tile in the output.  XCD x takes the blocks
// [x per, (x + 1) per): the pieces of a cache line go through ONE L2 at about the same time.  Measured at 1168x876 / 2M:
// bin_hist 17.5 -> 15.5 us (its 4-byte counters leave as lines); bin_scatter unchanged (an XCD's share of the output is as large
// as its L2).  Returns -1 for the workgroups beyond the last block (the grid is rounded up to 8 per: bin_grid).
__device__ __forceinline__ int bin_block_of(int wg, int nblocks)
{
    const int per = (nblocks + 7) >> 3;
    const int b = (wg & 7) * per + (wg >> 3);
    return (wg >> 3) < per && b < nblocks ? b : -1;
}

// hist[tile * nblocks + block].  ITEMS = instances per workgroup: BIN_ITEMS (4096) or a multiple of it.  Every workgroup
// pays for the whole tile table (clearing, writing and later scanning `ntiles` counters), so with thousands of tiles a
// 4096-pair workgroup spends more time on the table than on its pairs (ScanNet++ shape, 4015 tiles, round 3's kernels: 40 + 13 + 108 us
// for the three passes): images of many tiles take larger workgroups (bin_items_for), a quarter of the counters and of the rows to scan.
template <int ITEMS>
__device__ __forceinline__ void
bin_hist_body(const uint64_t* __restrict__ keys, int n, int ntiles, uint32_t* __restrict__ hist, int nblocks)
{
    const int blk = bin_block_of(blockIdx.x, nblocks);
    if (blk < 0) return;
    extern __shared__ uint32_t lds[];  // [ntiles]
    for (int t = threadIdx.x; t < ntiles; t += BIN_T) lds[t] = 0;
    __syncthreads();
#pragma unroll 1
    for (int sub = 0; sub < ITEMS / BIN_ITEMS; sub++) {
        const int base = blk * ITEMS + sub * BIN_ITEMS;
        const int end = min(n, base + BIN_ITEMS);
        if (base >= end) break;
        // all loads in flight before the first LDS atomic: the kernel is latency-bound, not bandwidth-bound
        uint32_t tl[BIN_ITEMS / BIN_T];
#pragma unroll
        for (int r = 0; r < BIN_ITEMS / BIN_T; r++) {
            const int i = base + r * BIN_T + threadIdx.x;
            tl[r] = i < end ? (uint32_t)(keys[i] >> 32) : 0xffffffffu;
        }
#pragma unroll
        for (int r = 0; r < BIN_ITEMS / BIN_T; r++)
            if (tl[r] != 0xffffffffu) atomicAdd(&lds[tl[r]], 1u);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ntiles; t += BIN_T) hist[(size_t)t * nblocks + blk] = lds[t];
}
template <int ITEMS>
__global__ void __launch_bounds__(BIN_T)
bin_hist_kernel(const uint64_t* __restrict__ keys, int n, int ntiles, uint32_t* __restrict__ hist, int nblocks)
{
    bin_hist_body<ITEMS>(keys, n, ntiles, hist, nblocks);
}
__global__ void __launch_bounds__(BIN_T) bin_hist_batch_kernel(int ntiles, const gs2d::BinFrames tab)
{
    const gs2d::BinFrame& f = tab.f[blockIdx.y];
    bin_hist_body<BIN_ITEMS>(f.keys_unsorted, f.R, ntiles, f.hist, f.nblocks);
}

// hist[tile][0..nblocks) -> exclusive scan along the blocks, in place; tile_total[tile] = the row's sum.  One wave per tile.
__device__ __forceinline__ void
bin_row_scan_body(uint32_t* __restrict__ hist, int ntiles, int nblocks, uint32_t* __restrict__ tile_total)
{
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= ntiles) return;
    uint32_t* row = hist + (size_t)tile * nblocks;
    uint32_t carry = 0;
    for (int c = 0; c < nblocks; c += 64) {
        const int b = c + lane;
        const uint32_t v = b < nblocks ? row[b] : 0u;
        const uint32_t incl = wave_incl_scan(v, lane);
        if (b < nblocks) row[b] = carry + incl - v;
        carry += (uint32_t)__shfl((int)incl, 63, 64);
    }
    if (lane == 0) tile_total[tile] = carry;
}
__global__ void __launch_bounds__(256)
bin_row_scan_kernel(uint32_t* __restrict__ hist, int ntiles, int nblocks, uint32_t* __restrict__ tile_total)
{
    bin_row_scan_body(hist, ntiles, nblocks, tile_total);
}
__global__ void __launch_bounds__(256) bin_row_scan_batch_kernel(int ntiles, const gs2d::BinFrames tab)
{
    const gs2d::BinFrame& f = tab.f[blockIdx.y];
    if (f.nblocks > 0) bin_row_scan_body(f.hist, ntiles, f.nblocks, f.hist + (size_t)ntiles * f.nblocks);
}

// offs_excl[tile][block] = instances of the tile in earlier blocks (bin_row_scan_kernel), tile_total[tile] = all of them.
// Stable: element order inside a tile is preserved.
// Output: packed (depth bits, id) pairs in keys_out's 8-byte slots (vals_out is not written).
// ITEMS == BIN_ITEMS: each wave's 1024 pairs stay in registers between the counting and the scatter; larger ITEMS (images of
// many tiles, see bin_hist_body): each wave owns ITEMS / 4 CONSECUTIVE pairs (so that (wave, position) order is element
// order, which stability needs), counts them in rounds of 1024 and reads them a second time (cache-warm) to scatter.
//
// LDS: base[tile] (uint32: where this workgroup's pairs of the tile start) + one uint16 counter per (wave, tile) -- a wave owns
// at most 4096 pairs -- two to a word: 12 bytes per tile (until round 4: 16, four uint32 running destinations; at 4015 tiles
// that was 64 KB and two workgroups per CU).  The counter first counts the wave's pairs of the tile, then holds how many
// pairs of the tile the waves before it own, then runs.
// Rank of a pair among its wave's pairs of the same tile (round 4): the LDS atomic's return value.  Lanes of one instruction
// that name the same tile are served in an order the hardware does not promise, so they are found (the counter moved by more
// than one) and re-ranked in lane order, one ballot per distinct tile they share -- 1.6 such tiles per 64 pairs at 1200 tiles,
// fewer at more.  Until round 4 every instruction paid a match-any over all tile bits (11-12 ballots, each a 64-bit select per
// lane): 29 of bin_scatter's 93 us at 1168x876 / 2M (scripts/dev/binexp.sh).
template <int ITEMS>
__device__ __forceinline__ void
bin_scatter_body(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint64_t* __restrict__ keys_out,
                 uint32_t* __restrict__ vals_out, int n, int ntiles, int nbits, const uint32_t* __restrict__ offs_excl,
                 const uint32_t* __restrict__ tile_total, int nblocks, uint2* __restrict__ ranges)
{
    extern __shared__ uint32_t lds[];  // base[ntiles], then [4][nt2 / 2] words of uint16 counter pairs
    // (no instances: the grid is one workgroup, which writes the empty tile ranges)
    const int blk = nblocks > 0 ? bin_block_of(blockIdx.x, nblocks) : (blockIdx.x == 0 ? 0 : -1);
    if (blk < 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nt2 = ntiles + (ntiles & 1);
    uint32_t* const cnt_words = lds + ntiles;
    for (int t = threadIdx.x; t < 2 * nt2; t += BIN_T) cnt_words[t] = 0;
    __syncthreads();
    constexpr int WAVE_SPAN = ITEMS / 4;           // consecutive pairs per wave
    static_assert(WAVE_SPAN < 65536 && ITEMS < 65536, "uint16 counters");
    constexpr int ROUNDS = WAVE_SPAN / BIN_WAVE_ITEMS;  // rounds of 1024
    const int wbeg0 = blk * ITEMS + wave * WAVE_SPAN;
    const int wend0 = min(n, wbeg0 + WAVE_SPAN);
    uint32_t* const mine = cnt_words + wave * (nt2 >> 1);  // counter of tile d: half (d & 1) of mine[d >> 1]
    constexpr int PER_LANE = BIN_WAVE_ITEMS / 64;  // 16
    uint64_t k[PER_LANE];
    uint32_t v[PER_LANE];
    int wbeg = wbeg0, wend = min(wend0, wbeg0 + BIN_WAVE_ITEMS);
#pragma unroll 1
    for (int rd = 0; rd < ROUNDS; rd++) {
        wbeg = wbeg0 + rd * BIN_WAVE_ITEMS;
        wend = min(wend0, wbeg + BIN_WAVE_ITEMS);
        if (rd > 0 && wbeg >= wend) break;
        // one round of loads, issued back to back (ROUNDS == 1: the pairs stay in registers for the scatter below)
#pragma unroll
        for (int r = 0; r < PER_LANE; r++) {
            const int i = wbeg + r * 64 + lane;
            k[r] = i < wend ? keys_in[i] : ~0ull;
            if (ROUNDS == 1) v[r] = i < wend ? vals_in[i] : 0u;
        }
#pragma unroll
        for (int r = 0; r < PER_LANE; r++) {
            const uint32_t d = (uint32_t)(k[r] >> 32);
            if (!(GS2D_DEV_BIN & 8) && wbeg + r * 64 + lane < wend) atomicAdd(&mine[d >> 1], 1u << (16 * (d & 1u)));
        }
    }
    __syncthreads();
    {
        // tile bases = exclusive scan of the row totals: thread i owns the tiles [i K, (i+1) K), sums them, the 256 sums are
        // scanned across the workgroup, then the thread walks its tiles with a running base
        const int K = (ntiles + BIN_T - 1) / BIN_T;
        const int t0 = min(ntiles, (int)threadIdx.x * K), t1 = min(ntiles, t0 + K);
        uint32_t mysum = 0;
        for (int t = t0; t < t1; t++) mysum += tile_total[t];
        uint32_t running = block_incl_scan(mysum, nullptr) - mysum;
        uint16_t* const c16 = reinterpret_cast<uint16_t*>(cnt_words);
        for (int t = t0; t < t1; t++) {
            const uint32_t tot = tile_total[t];
            const uint32_t c0 = c16[t], c1 = c16[nt2 + t], c2 = c16[2 * nt2 + t];
            lds[t] = running + (nblocks > 0 && !(GS2D_DEV_BIN & 1) ? offs_excl[(size_t)t * nblocks + blk] : 0u);
            c16[t] = 0; c16[nt2 + t] = (uint16_t)c0; c16[2 * nt2 + t] = (uint16_t)(c0 + c1); c16[3 * nt2 + t] = (uint16_t)(c0 + c1 + c2);
            // tile ranges = boundaries of the scanned histogram (rasterizer_impl.cu:116-138 semantics)
            if (blk == 0) ranges[t] = tot ? make_uint2(running, running + tot) : make_uint2(0u, 0u);
            running += tot;
        }
    }
    __syncthreads();
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    (void)nbits;
#pragma unroll 1
    for (int rd = 0; rd < ROUNDS; rd++) {
        if (ROUNDS > 1) {  // second look at this round's pairs (the counting pass brought them into the caches)
            wbeg = wbeg0 + rd * BIN_WAVE_ITEMS;
            wend = min(wend0, wbeg + BIN_WAVE_ITEMS);
            if (wbeg >= wend) break;
#pragma unroll
            for (int r = 0; r < PER_LANE; r++) {
                const int i = wbeg + r * 64 + lane;
                k[r] = i < wend ? keys_in[i] : ~0ull;
                v[r] = i < wend ? vals_in[i] : 0u;
            }
        }
        // all sixteen (atomic, re-read, base) triples are issued before the first result is looked at: the LDS serves a
        // wave's instructions in order, so step r's re-read sees exactly the counter after step r's atomics (the order of
        // these instructions in the binary is what the argument rests on: relaxed atomics, checked in the ISA, and
        // tests/test_gpu_parity.py's list comparisons would show a swap)
        uint32_t rank[PER_LANE], now[PER_LANE], dstb[PER_LANE];
#pragma unroll
        for (int r = 0; r < PER_LANE; r++) {
            const bool valid = wbeg + r * 64 + lane < wend;
            const uint32_t d = valid ? (uint32_t)(k[r] >> 32) : 0u;
            const uint32_t sh = 16 * (d & 1u);
            rank[r] = 0; now[r] = 1; dstb[r] = 0;
            if (valid && !(GS2D_DEV_BIN & 4)) {
                rank[r] = __hip_atomic_fetch_add(&mine[d >> 1], 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                now[r] = __hip_atomic_load(&mine[d >> 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (valid) dstb[r] = lds[d];
        }
#pragma unroll
        for (int r = 0; r < PER_LANE; r++) {
            if (wbeg + r * 64 >= wend) break;  // wave-uniform
            const bool valid = wbeg + r * 64 + lane < wend;
            const uint32_t d = valid ? (uint32_t)(k[r] >> 32) : 0u;
            const uint32_t sh = 16 * (d & 1u);
            uint32_t rk = rank[r], nw = now[r];
            if (!(GS2D_DEV_BIN & 4)) { rk = (rk >> sh) & 0xffffu; nw = (nw >> sh) & 0xffffu; }
            // lanes of this instruction that share a tile: the counter moved by more than one for all of them but the one
            // served last; re-rank every such tile's lanes in lane order (= element order)
            uint64_t clash = __ballot(valid && nw != rk + 1u);
            while (clash) {
                const int leader = __builtin_ctzll(clash);
                const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)d, leader);
                const uint32_t nowl = (uint32_t)__builtin_amdgcn_readlane((int)nw, leader);
                const uint64_t m = __ballot(valid && d == dl);
                if (valid && d == dl) rk = nowl - (uint32_t)__popcll(m) + (uint32_t)__popcll(m & lt_mask);
                clash &= ~m;
            }
            if (valid) {
                const uint32_t dst = dstb[r] + rk;
                // one 8-byte store per pair: (depth bits, Gaussian id); the tile id is implied by the position
                // (a streaming / non-temporal store here doubles the kernel's time: the L2 is what joins the runs into lines)
                reinterpret_cast<uint2*>(keys_out)[(GS2D_DEV_BIN & 2) ? (uint32_t)(wbeg + r * 64 + lane) : dst] = make_uint2((uint32_t)k[r], v[r]);
            }
        }
    }
}
template <int ITEMS>
__global__ void __launch_bounds__(BIN_T)
bin_scatter_kernel(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint64_t* __restrict__ keys_out,
                   uint32_t* __restrict__ vals_out, int n, int ntiles, int nbits, const uint32_t* __restrict__ offs_excl,
                   const uint32_t* __restrict__ tile_total, int nblocks, uint2* __restrict__ ranges)
{
    bin_scatter_body<ITEMS>(keys_in, vals_in, keys_out, vals_out, n, ntiles, nbits, offs_excl, tile_total, nblocks, ranges);
}
__global__ void __launch_bounds__(BIN_T) bin_scatter_batch_kernel(int ntiles, int nbits, const gs2d::BinFrames tab)
{
    const gs2d::BinFrame& f = tab.f[blockIdx.y];
    if (f.nblocks == 0) return;
    bin_scatter_body<BIN_ITEMS>(f.keys_unsorted, f.vals_unsorted, f.keys, f.point_list, f.R, ntiles, nbits, f.hist,
                                f.hist + (size_t)ntiles * f.nblocks, f.nblocks, f.ranges);
}

// The single-frame kernels with the count read on the device (DevBin, gs2d_common.h): same bodies, pointers from the layout of
// the count duplicate_kernel stored; a count beyond the chunk's capacity means "do nothing" (the host will come back).
struct DevBinPtrs {
    int R, nblocks;
    const uint64_t* keys_unsorted; const uint32_t* vals_unsorted;
    uint64_t* keys; uint32_t* point_list; uint64_t* keys_alt; uint32_t* vals_alt; uint32_t* hist;
};
__device__ __forceinline__ bool dev_bin_ptrs(const gs2d::DevBin& db, DevBinPtrs* o, int items)
{
    const uint32_t R = *db.R_dev;
    if (R > db.cap) return false;
    const BinLayout L = bin_layout((int)R, db.det != 0, (int)db.cap);
    o->R = (int)R;
    o->nblocks = ((int)R + items - 1) / items;
    o->keys = (uint64_t*)(db.base + L.keys); o->point_list = (uint32_t*)(db.base + L.point_list);
    o->keys_alt = (uint64_t*)(db.base + L.keys_alt); o->vals_alt = (uint32_t*)(db.base + L.vals_alt);
    o->keys_unsorted = o->keys_alt; o->vals_unsorted = o->vals_alt;  // one pass: the unsorted pairs sit in the "alt" buffers
    o->hist = (uint32_t*)(db.base + L.hist);
    return true;
}
template <int ITEMS>
__global__ void __launch_bounds__(BIN_T) bin_hist_dev_kernel(const gs2d::DevBin db, int ntiles)
{
    DevBinPtrs p;
    if (!dev_bin_ptrs(db, &p, ITEMS)) return;
    bin_hist_body<ITEMS>(p.keys_unsorted, p.R, ntiles, p.hist, p.nblocks);
}
__global__ void __launch_bounds__(256) bin_row_scan_dev_kernel(const gs2d::DevBin db, int ntiles, int items)
{
    DevBinPtrs p;
    if (!dev_bin_ptrs(db, &p, items)) return;
    bin_row_scan_body(p.hist, ntiles, p.nblocks, p.hist + (size_t)ntiles * p.nblocks);  // (no instances: every row total is 0)
}
template <int ITEMS>
__global__ void __launch_bounds__(BIN_T) bin_scatter_dev_kernel(const gs2d::DevBin db, int ntiles, int nbits, uint2* __restrict__ ranges)
{
    DevBinPtrs p;
    if (!dev_bin_ptrs(db, &p, ITEMS)) return;
    // (workgroup 0 always has work: it writes the tile ranges, all empty when there are no instances)
    bin_scatter_body<ITEMS>(p.keys_unsorted, p.vals_unsorted, p.keys, p.point_list, p.R, ntiles, nbits, p.hist,
                     p.hist + (size_t)ntiles * p.nblocks, p.nblocks, ranges);
}
__global__ void __launch_bounds__(256)
tile_depth_sort_dev_kernel(const gs2d::DevBin db, const uint2* __restrict__ ranges, int cap, int write_keys)
{
    extern __shared__ uint32_t dyn[];
    __shared__ uint32_t wcnt[4][256];
    DevBinPtrs p;
    if (!dev_bin_ptrs(db, &p, BIN_ITEMS) || p.R == 0) return;
    tile_depth_sort_body(blockIdx.x, dyn, wcnt, ranges, p.keys, p.point_list, p.keys_alt, p.vals_alt, cap, 1, write_keys);
}

__global__ void __launch_bounds__(256)
tile_depth_sort_kernel(const uint2* __restrict__ ranges, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                       uint64_t* __restrict__ keys_alt, uint32_t* __restrict__ vals_alt, int cap, int packed, int write_keys)
{
    extern __shared__ uint32_t dyn[];  // [4][cap]: ka, va, kb, vb
    __shared__ uint32_t wcnt[4][256];
    tile_depth_sort_body(blockIdx.x, dyn, wcnt, ranges, keys, vals, keys_alt, vals_alt, cap, packed, write_keys);
}
__global__ void __launch_bounds__(256) tile_depth_sort_batch_kernel(int cap, int write_keys, const gs2d::BinFrames tab)
{
    const gs2d::BinFrame& f = tab.f[blockIdx.y];
    extern __shared__ uint32_t dyn[];
    __shared__ uint32_t wcnt[4][256];
    if (f.R > 0) tile_depth_sort_body(blockIdx.x, dyn, wcnt, f.ranges, f.keys, f.point_list, f.keys_alt, f.vals_alt, cap, 1, write_keys);
}

// rasterizer_impl.cu:116-138
__global__ void __launch_bounds__(256) tile_ranges_kernel(int L, const uint64_t* __restrict__ keys, uint2* __restrict__ ranges)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= L) return;
    const uint32_t cur = (uint32_t)(keys[idx] >> 32);
    if (idx == 0) ranges[cur].x = 0;
    else {
        const uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
        if (cur != prev) { ranges[prev].y = idx; ranges[cur].x = idx; }
    }
    if (idx == L - 1) ranges[cur].y = L;
}

}  // namespace

namespace gs2d {

// Instances per binning workgroup.  Every workgroup clears, writes and reads a whole tile table, so its share of pairs should
// not be small against the number of tiles -- but small workgroups keep their pairs in registers and fill the chip.  With round
// 4's bin_scatter (12 B of LDS per tile, ranks from the LDS atomic) the sort stage measures, 4096 / 8192 / 16384 pairs per
// workgroup: 41 / 38 / 47 us at 1200x680 / 600k, 157 / 165 / 177 us at 876x584 / 2M (77 us of it the stand-alone depth sort),
// 107 / 105 / 109 us at 1168x876 / 2M (profiles/ab_bin_items_r04.txt).  count: the number of instances (or the capacity of the
// chunk when the host does not know it yet); the three kernels of one pass must be given the same value.
int bin_items_for(long long count, int tiles)
{
    static const int forced = [] { const char* e = getenv("GS2D_BIN_ITEMS_FORCE"); return e ? atoi(e) : 0; }();
    if (forced == 4096 || forced == 8192 || forced == 16384) return forced;
    return tiles > 2560 && count >= 8192LL * 128 ? 8192 : BIN_ITEMS;
}

static size_t bin_lds_bytes(int tiles) { return 4 * (size_t)(tiles + 2 * (tiles + (tiles & 1))); }  // bin_scatter_body's layout
static int bin_grid(int nblocks) { return nblocks > 0 ? 8 * ((nblocks + 7) / 8) : 1; }  // see bin_block_of

#define GS2D_BIN_DISPATCH(ITEMS_, ...)                          \
    switch (ITEMS_) {                                          \
    case 16384: { constexpr int I = 16384; __VA_ARGS__; } break; \
    case 8192: { constexpr int I = 8192; __VA_ARGS__; } break;   \
    default: { constexpr int I = BIN_ITEMS; __VA_ARGS__; } break; \
    }

void launch_inclusive_scan(const uint32_t* in, uint32_t* out, int n, uint32_t* tmp, uint32_t* total_out, hipStream_t s,
                           uint32_t* total_host)
{
    if (n <= 0) return;
    const int nblocks = (n + GS2D_SCAN_ITEMS - 1) / GS2D_SCAN_ITEMS;
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nblocks), dim3(SCAN_T), 0, s, in, n, tmp);
    hipLaunchKernelGGL(scan_blocksums_kernel, dim3(1), dim3(SCAN_T), 0, s, tmp, nblocks, total_out, total_host);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nblocks), dim3(SCAN_T), 0, s, in, out, n, tmp);
}

void launch_offsets_blocksums(int P, uint32_t* block_sums, uint32_t* total_dev, uint32_t* total_host, hipStream_t s)
{
    hipLaunchKernelGGL(scan_blocksums_kernel, dim3(1), dim3(SCAN_T), 0, s, block_sums, (P + 255) / 256, total_dev, total_host);
}

void launch_duplicate(int P, const ushort4* rect, const float* depths, const uint32_t* tiles_touched,
                      const uint32_t* block_sums, int scanned, uint32_t* point_offsets, int gx, uint64_t* keys, uint32_t* vals,
                      uint32_t capacity, uint32_t* total_host, hipStream_t s, uint32_t* total_dev)
{
    hipLaunchKernelGGL(duplicate_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, rect, depths, tiles_touched, block_sums, scanned,
                       point_offsets, gx, keys, vals, capacity, total_host, total_dev);
}

void launch_sort_pairs(int R, uint64_t* keys_a, uint32_t* vals_a, uint64_t* keys_b, uint32_t* vals_b, int begin_bit,
                       int end_bit, uint32_t* hist, size_t hist_elems, hipStream_t s)
{
    if (R <= 0) return;
    const int nblocks = (R + GS2D_SORT_ITEMS - 1) / GS2D_SORT_ITEMS;
    uint32_t* scan_tmp = hist + hist_elems;
    // data starts in the "b" buffers (unsorted) when the pass count is odd, so the result always lands in "a".
    const int passes = (end_bit - begin_bit + 7) / 8;
    uint64_t* kin = (passes & 1) ? keys_b : keys_a;
    uint32_t* vin = (passes & 1) ? vals_b : vals_a;
    uint64_t* kout = (passes & 1) ? keys_a : keys_b;
    uint32_t* vout = (passes & 1) ? vals_a : vals_b;
    for (int p = 0; p < passes; p++) {
        const int shift = begin_bit + 8 * p;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(nblocks), dim3(SORT_T), 0, s, kin, R, shift, hist, nblocks);
        launch_inclusive_scan(hist, hist, (int)hist_elems, scan_tmp, nullptr, s);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblocks), dim3(SORT_T), 0, s, kin, vin, kout, vout, R, shift, hist,
                           nblocks);
        uint64_t* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }
}

bool launch_bin_by_tile(int R, int tiles, int nbits, const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out,
                        uint32_t* vals_out, uint32_t* hist, uint2* ranges, hipStream_t s)
{
    if (tiles > GS2D_BIN_MAX_TILES) return false;  // caller falls back to the 8-bit passes + tile_ranges kernel
    const int items = bin_items_for(R, tiles);
    const int nblocks = (R + items - 1) / items;
    const size_t hist_elems = (size_t)tiles * nblocks;
    uint32_t* tile_total = hist + hist_elems;  // GS2D_BIN_MAX_TILES words behind the counters (bin_layout)
    GS2D_BIN_DISPATCH(items, hipLaunchKernelGGL((bin_hist_kernel<I>), dim3(bin_grid(nblocks)), dim3(BIN_T), (size_t)tiles * 4, s, keys_in, R, tiles, hist, nblocks));
    hipLaunchKernelGGL(bin_row_scan_kernel, dim3((tiles + 3) / 4), dim3(256), 0, s, hist, tiles, nblocks, tile_total);
    GS2D_BIN_DISPATCH(items, hipLaunchKernelGGL((bin_scatter_kernel<I>), dim3(bin_grid(nblocks)), dim3(BIN_T), (size_t)bin_lds_bytes(tiles), s, keys_in, vals_in, keys_out,
                                                vals_out, R, tiles, nbits, hist, tile_total, nblocks, ranges));
    return true;
}

void launch_bin_by_tile_dev(const DevBin& db, int tiles, int nbits, uint2* ranges, hipStream_t s)
{
    const int items = bin_items_for((long long)db.cap, tiles);
    int grid = ((int)db.cap + items - 1) / items;
    if (grid < 1) grid = 1;
    GS2D_BIN_DISPATCH(items, hipLaunchKernelGGL((bin_hist_dev_kernel<I>), dim3(bin_grid(grid)), dim3(BIN_T), (size_t)tiles * 4, s, db, tiles));
    hipLaunchKernelGGL(bin_row_scan_dev_kernel, dim3((tiles + 3) / 4), dim3(256), 0, s, db, tiles, items);
    GS2D_BIN_DISPATCH(items, hipLaunchKernelGGL((bin_scatter_dev_kernel<I>), dim3(bin_grid(grid)), dim3(BIN_T), (size_t)bin_lds_bytes(tiles), s, db, tiles, nbits, ranges));
}

void launch_tile_depth_sort_dev(const DevBin& db, int tiles, const uint2* ranges, int cap_class, int write_keys, hipStream_t s)
{
    if (tiles <= 0) return;
    hipLaunchKernelGGL(tile_depth_sort_dev_kernel, dim3(tiles), dim3(256), (size_t)cap_class * 16, s, db, ranges, cap_class, write_keys);
}

int tile_sort_capacity(long long R, int tiles)
{
    if (tiles <= 0) return 1536;
    const int want = (int)((R * 13) / ((long long)tiles * 10));
    return want <= 1536 ? 1536 : (want <= 2048 ? 2048 : (want <= 3072 ? 3072 : 4096));
}

void launch_tile_depth_sort(int R, int tiles, const uint2* ranges, uint64_t* keys, uint32_t* vals, uint64_t* keys_alt,
                            uint32_t* vals_alt, int packed, int write_keys, hipStream_t s)
{
    if (R <= 0 || tiles <= 0) return;
    // LDS capacity per tile: the smallest of 1536 / 2048 / 3072 / 4096 elements (16 B each, <= 64 KB dynamic LDS) that
    // is at least 1.3x the mean list length.  Smaller capacity = more workgroups per CU (1536 -> 5 per CU, i.e. all
    // 1200 tiles of a 640x480 frame resident at once); the few tiles above the capacity take the global-memory variant.
    const int cap = tile_sort_capacity(R, tiles);
    hipLaunchKernelGGL(tile_depth_sort_kernel, dim3(tiles), dim3(256), (size_t)cap * 16, s, ranges, keys, vals, keys_alt,
                       vals_alt, cap, packed, write_keys);
}

void launch_duplicate_batch(int P, int K, int gx, const BinFrames& tab, hipStream_t s)
{
    hipLaunchKernelGGL(duplicate_batch_kernel, dim3((P + 255) / 256, K), dim3(256), 0, s, P, gx, tab);
}

// the single-pass tile binning + the per-tile depth sort for K frames (tiles <= GS2D_BIN_MAX_TILES; launch_duplicate_batch ran before)
void launch_bin_sort_batch(int P, int K, int tiles, int gx, int nbits, BinFrames& tab, int write_keys, bool depth_sort, hipStream_t s)
{
    int max_blocks = 0;
    long long max_R = 0;
    for (int k = 0; k < K; k++) {
        tab.f[k].nblocks = (tab.f[k].R + BIN_ITEMS - 1) / BIN_ITEMS;
        max_blocks = tab.f[k].nblocks > max_blocks ? tab.f[k].nblocks : max_blocks;
        max_R = tab.f[k].R > max_R ? tab.f[k].R : max_R;
        if (tab.f[k].R == 0) (void)hipMemsetAsync(tab.f[k].ranges, 0, sizeof(uint2) * (size_t)tiles, s);  // nothing will write them
    }
    (void)P; (void)gx;
    if (max_blocks == 0) return;
    hipLaunchKernelGGL(bin_hist_batch_kernel, dim3(bin_grid(max_blocks), K), dim3(BIN_T), (size_t)tiles * 4, s, tiles, tab);
    hipLaunchKernelGGL(bin_row_scan_batch_kernel, dim3((tiles + 3) / 4, K), dim3(256), 0, s, tiles, tab);
    hipLaunchKernelGGL(bin_scatter_batch_kernel, dim3(bin_grid(max_blocks), K), dim3(BIN_T), (size_t)bin_lds_bytes(tiles), s, tiles, nbits, tab);
    if (!depth_sort) return;  // the forward blend kernel sorts each tile's list itself
    const int cap = tile_sort_capacity(max_R, tiles);  // (launch_tile_depth_sort's rule, on the longest frame)
    hipLaunchKernelGGL(tile_depth_sort_batch_kernel, dim3(tiles, K), dim3(256), (size_t)cap * 16, s, cap, write_keys, tab);
}

void launch_tile_ranges(int R, const uint64_t* keys, uint2* ranges, int tiles, hipStream_t s)
{
    (void)hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)tiles, s);
    if (R > 0) hipLaunchKernelGGL(tile_ranges_kernel, dim3((R + 255) / 256), dim3(256), 0, s, R, keys, ranges);
}

}  // namespace gs2d
