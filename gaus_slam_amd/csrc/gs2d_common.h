// Internal definitions shared by the gfx950 kernels of the 2D-Gaussian-surfel rasterizer.
// Constants carried over verbatim from the reference (RAST/cuda_rasterizer/config.h:15-17,
// auxiliary.h:37-59); everything else (memory layout, kernel decomposition) is this library's own.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define GS2D_TILE 16
#define GS2D_TILE_PIX 256
#define GS2D_NEAR_N 0.2f
#define GS2D_FAR_N 100.0f
#define GS2D_FILTER_INV_SQ 100.0f

// Splat record: 5 x float4 = 80 B per Gaussian, written once by the forward preprocess and
// gathered by both blend kernels (one record = everything a pixel needs about a splat):
//   q0 = (Tu.x, Tu.y, Tu.z, center.x)   q1 = (Tv.x, Tv.y, Tv.z, center.y)
//   q2 = (Tw.x, Tw.y, Tw.z, opacity)    q3 = (n.x, n.y, n.z, red)   q4 = (green, blue, rho_max, 0)
// rho_max = 2 ln(255 opacity) (+margin): cull bound only, see splat_may_touch() in gs2d_blend.hip.
#define GS2D_REC_F4 5
#define GS2D_REC_FLOATS 20

// Gradient record accumulated by the backward blend, consumed by the backward preprocess:
//   [0..2] dL_dcolor  [3..5] dL_dnormal  [6..14] dL_dT (Tu,Tv,Tw)  [15] dL_dopacity  [16,17] dL_dmean2D.xy
#define GS2D_GRAD_FLOATS 20

// Pixel-state planes kept between forward and backward (element index = tile*256 + thread).
enum { PS_TFINAL = 0, PS_M1, PS_M2, PS_MEDIAN, PS_STD, PS_LAST, PS_MEDC, PS_PLANES };

__host__ __device__ static inline size_t gs2d_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct GeomLayout {
    size_t depths, tiles_touched, point_offsets, rec, clamped, scan_tmp, grad_rec, rect, total;
};
struct BinLayout {
    size_t point_list, hits, hits4, keys, vals_alt, keys_alt, hist, det_inv, det_slots, total;
    size_t hist_elems;
};
struct ImgLayout {
    size_t ranges, pix, total;
    int tiles;
};

#define GS2D_SCAN_ITEMS 1024  // elements per workgroup in the device scan
#define GS2D_SORT_ITEMS 2048  // elements per workgroup in one radix pass (256 threads x 8)
#ifndef GS2D_BIN_ITEMS
#define GS2D_BIN_ITEMS 4096   // elements per workgroup in the single-pass tile binning
#endif
#define GS2D_BIN_MAX_TILES 4096  // bin_scatter keeps 12 B of LDS per tile (48 KB here: three workgroups per CU)

static inline GeomLayout geom_layout(int P)
{
    GeomLayout L;
    size_t o = 0;
    const size_t p = (size_t)(P > 0 ? P : 1);
    L.depths = o; o = gs2d_align_up(o + 4 * p, 256);
    L.tiles_touched = o; o = gs2d_align_up(o + 4 * p, 256);
    L.point_offsets = o; o = gs2d_align_up(o + 4 * p, 256);
    L.rec = o; o = gs2d_align_up(o + 4 * GS2D_REC_FLOATS * p, 256);
    L.clamped = o; o = gs2d_align_up(o + 3 * p, 256);
    const size_t nblk = (p + 255) / 256;  // one tiles_touched sum per preprocess workgroup (256 Gaussians)
    L.scan_tmp = o; o = gs2d_align_up(o + 4 * (nblk + 64), 256);
    // backward-time accumulator (the reference backward has no allocator callback, so it is reserved here)
    L.grad_rec = o; o = gs2d_align_up(o + 4 * GS2D_GRAD_FLOATS * p, 256);
    // tile rectangle [minx, maxx) x [miny, maxy) of every Gaussian as four u16 (preprocess -> duplicate / det_inverse)
    L.rect = o; o = gs2d_align_up(o + 8 * p, 256);
    L.total = o;
    return L;
}

// The binning chunk.  Everything that is read after the forward's duplicate step -- by the rest of the forward, by the backward,
// by the debug readers -- sits at offsets that depend on num_rendered (R) alone.  Only the two arrays duplicate_kernel writes
// (the unsorted pairs: keys_alt, vals_alt) follow a CAPACITY C >= R: the forward launches that kernel before the host knows R
// (gs2d_api.hip, fwd_phase_b), into a chunk sized for C instances, so their offsets must not depend on R.  C < 0: C = R, the
// layout a caller who knows R computes (gs2d_binning_bytes; the backward, which never touches the two arrays).
// det: also room for the deterministic backward (inverse permutation + one partial gradient record per (instance, quadrant))
// (__host__ __device__: the kernels of the forward that run before the host knows R evaluate the layout themselves, DevBin below)
__host__ __device__ static inline BinLayout bin_layout(int R, bool det = false, int C = -1)
{
    BinLayout L;
    const size_t r = (size_t)(R > 0 ? R : 1);
    const size_t c = (C >= 0 && (size_t)C > r) ? (size_t)C : r;
    auto fixed_part = [det](size_t n, BinLayout* out) -> size_t {
        size_t o = 0;
        const size_t point_list = o; o = gs2d_align_up(o + 4 * n, 256);
        // phase 0 of blend_fwd -> both blend kernels: per instance and quadrant, the 16 group bits of the cull test (u16 each)
        const size_t hits = o; o = gs2d_align_up(o + 8 * n, 256);
        // the same bits ORed down to the four 4x4 sub-blocks of each quadrant (one byte per quadrant): what the backward's
        // four row queues are built from
        const size_t hits4 = o; o = gs2d_align_up(o + 4 * n, 256);
        const size_t keys = o; o = gs2d_align_up(o + 8 * n, 256);
        const size_t nblk = (n + GS2D_SORT_ITEMS - 1) / GS2D_SORT_ITEMS;
        const size_t hist_elems = 256 * nblk;
        // the single-pass tile binning needs tiles x ceil(R / GS2D_BIN_ITEMS) counters; size for the larger of the two
        const size_t bin_elems = (size_t)GS2D_BIN_MAX_TILES * ((n + GS2D_BIN_ITEMS - 1) / GS2D_BIN_ITEMS);
        const size_t cap_elems = hist_elems > bin_elems ? hist_elems : bin_elems;
        const size_t scan_blk = (cap_elems + GS2D_SCAN_ITEMS - 1) / GS2D_SCAN_ITEMS;
        // (+ GS2D_BIN_MAX_TILES: the per-tile totals of the binning pass sit behind its counters)
        const size_t hist = o; o = gs2d_align_up(o + 4 * (cap_elems + scan_blk + 64 + GS2D_BIN_MAX_TILES), 256);
        size_t det_inv = o, det_slots = o;
        if (det) {
            o = gs2d_align_up(o + 4 * n, 256);
            det_slots = o; o = gs2d_align_up(o + 4 * (size_t)GS2D_GRAD_FLOATS * 4 * n, 256);
        }
        if (out) {
            out->point_list = point_list; out->hits = hits; out->hits4 = hits4; out->keys = keys; out->hist = hist;
            out->hist_elems = hist_elems; out->det_inv = det_inv; out->det_slots = det_slots;
        }
        return o;
    };
    (void)fixed_part(r, &L);
    size_t o = fixed_part(c, nullptr);  // the capacity-sized arrays start where the fixed part of a FULL chunk would end
    L.vals_alt = o; o = gs2d_align_up(o + 4 * c, 256);
    L.keys_alt = o; o = gs2d_align_up(o + 8 * c, 256);
    L.total = o;
    return L;
}

static inline ImgLayout img_layout(int W, int H)
{
    ImgLayout L;
    const int gx = (W + GS2D_TILE - 1) / GS2D_TILE, gy = (H + GS2D_TILE - 1) / GS2D_TILE;
    L.tiles = gx * gy;
    size_t o = 0;
    const size_t t = (size_t)(L.tiles > 0 ? L.tiles : 1);
    L.ranges = o; o = gs2d_align_up(o + 8 * t, 256);
    L.pix = o; o = gs2d_align_up(o + 4 * (size_t)PS_PLANES * t * GS2D_TILE_PIX, 256);
    L.total = o;
    return L;
}

// Camera constants passed by value to the per-Gaussian kernels.
struct CamParams {
    const float* vm;      // device, 16 floats column-major (wave-uniform scalar loads in the kernels)
    const float* pm;      // device, 16 floats column-major
    const float* campos;  // device, 3 floats
    int W, H;      // forward: true size; backward: size rebuilt as the reference does (backward.cu:641-642)
    int gx, gy;
    int tight;     // forward: shrink every Gaussian's tile rectangle to its footprint bound (gs2d_footprint)
};

// Conservative bound of the pixels a splat can reach with alpha >= 1/255, i.e. min(rho3d, rho2d) <= rho_max
// (rho_max = 2 ln(255 opacity) + margin, record slot q4.z).  Shared by the preprocess kernel (tile rectangle of the
// Gaussian) and the cull kernel (sub-blocks of one tile) so that both use bit-identical numbers:
//  * {rho2d <= rho_max} is a disc of radius sqrt(rho_max/100) px around the stored centre (q0.w, q1.w): rl, +0.5 px margin;
//  * {rho3d <= rho_max} is the image of the surfel's disc u^2+v^2 <= rho_max.  When that disc lies safely in front of
//    the eye the image is an ellipse and its exact AABB [cx-ex, cx+ex] x [cy-ey, cy+ey] follows from the closed form
//    the reference uses for its 3-sigma box (forward.cu:119-147) with cutoff^2 = rho_max; mx, my are rounding margins.
// kind: 0 = the splat can never reach 1/255, 1 = bounded (disc + ellipse), 2 = no safe bound (only the disc part is valid).
// Cull-only math: hardware sqrt / rcp, the margins cover their ulps.
struct Gs2dFootprint {
    float rl, cx, cy, ex, ey, mx, my;
    int kind;
};
__device__ __forceinline__ Gs2dFootprint gs2d_footprint(const float4 q0, const float4 q1, const float4 q2, float rho_max)
{
    Gs2dFootprint f;
    f.rl = 0.f; f.cx = 0.f; f.cy = 0.f; f.ex = 0.f; f.ey = 0.f; f.mx = 0.f; f.my = 0.f; f.kind = 0;
    if (!(rho_max >= 0.f)) return f;  // opacity*G can never reach 1/255 (NaN opacity is encoded as +huge)
    f.rl = __builtin_amdgcn_sqrtf(rho_max * (1.0f / GS2D_FILTER_INV_SQ)) + 0.5f;
    f.kind = 2;
    const float a = rho_max * (q2.x * q2.x + q2.y * q2.y), zz = q2.z * q2.z;
    if (!(a <= 0.9f * zz) || !(q2.z > 0.f)) return f;  // disc not safely in front of the eye: no bound
    const float inv = __builtin_amdgcn_rcpf(a - zz);
    const float f0 = rho_max * inv, f2 = -inv;
    f.cx = f0 * (q0.x * q2.x + q0.y * q2.y) + f2 * (q0.z * q2.z);
    f.cy = f0 * (q1.x * q2.x + q1.y * q2.y) + f2 * (q1.z * q2.z);
    // Half extents.  The reference's form (forward.cu:139-146) is cx^2 - <Tu,Tu>/<Tw,Tw> with <a,b> = rho (a.x b.x + a.y b.y)
    // - a.z b.z: two numbers of the size of cx^2 that differ by the extent^2 -- at x = 1000 px and an extent of 1 px that
    // is a 10^6 : 1 cancellation.  The same quantity without it (Lagrange's identity for this metric):
    //   <Tu,Tw>^2 - <Tu,Tu><Tw,Tw> = rho [ c.x^2 + c.y^2 - rho c.z^2 ],  c = Tu x Tw,
    // whose terms are the 2x2 minors of (Tu, Tw), each a mild difference of products.
    const float ux = q0.y * q2.z - q0.z * q2.y, uy = q0.z * q2.x - q0.x * q2.z, uz = q0.x * q2.y - q0.y * q2.x;
    const float vx = q1.y * q2.z - q1.z * q2.y, vy = q1.z * q2.x - q1.x * q2.z, vz = q1.x * q2.y - q1.y * q2.x;
    const float k = rho_max * (inv * inv);
    const float hx = k * ((ux * ux + uy * uy) - rho_max * (uz * uz));
    const float hy = k * ((vx * vx + vy * vy) - rho_max * (vz * vz));
    if (!(hx == hx) || !(hy == hy)) return f;
    f.ex = __builtin_amdgcn_sqrtf(fmaxf(hx, 0.f)); f.ey = __builtin_amdgcn_sqrtf(fmaxf(hy, 0.f));
    f.mx = 0.1f + 0.002f * f.ex + 1e-4f * fabsf(f.cx); f.my = 0.1f + 0.002f * f.ey + 1e-4f * fabsf(f.cy);
    f.kind = 1;
    return f;
}

// Internal launchers (defined in the .hip files).
namespace gs2d {
void launch_preprocess_fwd(int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                           const float* rotations, const float* opacities, const float* shs,
                           const float* transMat_precomp, const float* colors_precomp, const CamParams& cam,
                           int* radii, float* depths, float4* rec, uint32_t* tiles_touched, ushort4* rect, uint8_t* clamped,
                           const float* pose_Rt, const float* pose_q, uint32_t* block_sums, hipStream_t s);
// Gaussians [first, P).  pose_partials (deterministic mode, else NULL): room for 12 floats per workgroup of the launch
// (<= ceil((P - first) / 256) workgroups); the pose gradient is then summed in a fixed order instead of with atomics
void launch_preprocess_bwd(int first, int P, int D, int M, const float* means3D, const float4* rec, const int* radii,
                           const float* shs, const uint8_t* clamped, const float* scales, const float* rotations,
                           const CamParams& cam, const float* grad_rec, float* dL_dtransMat, float* dL_dnormal,
                           float* dL_dcolor, float* dL_dopacity, float* dL_dsh, float* dL_dmean2D,
                           float* dL_dmean3D, float* dL_dscale, float* dL_drot, const float* pose_Rt, const float* pose_q,
                           float* dL_dpose, int need_record, float* pose_partials, hipStream_t s);
// pose-only per-Gaussian stage: reads the dense sums the POSE instantiation of blend_bwd left (BlendBwdFrame::dense_m2d)
void launch_preprocess_bwd_pose(int P, const float* means3D, const int* radii, const float* scales, const float* rotations,
                                const CamParams& cam, const float* dense_T, const float* dense_m2d, const float* pose_Rt,
                                const float* pose_q, float* dL_dpose, hipStream_t s);
void launch_mark_visible(int P, const float* means3D, const float* vm, uint8_t* present, hipStream_t s);
void launch_pose_quat(const float* pose_Rt, float* q_out, hipStream_t s);
// inclusive scan of n u32; tmp must hold ceil(n/1024)+64 u32. If total_out != nullptr the grand total is stored there.
// total_host (optional): pinned host word that receives the total with a system-scope store as soon as it is known.
void launch_inclusive_scan(const uint32_t* in, uint32_t* out, int n, uint32_t* tmp, uint32_t* total_out, hipStream_t s,
                           uint32_t* total_host = nullptr);
// prefix sum of tiles_touched: launch_preprocess_fwd leaves one sum per 256 Gaussians in block_sums.
//  * scanned == 0 (the normal path): launch_duplicate adds up the sums of the blocks before each block itself, its own scan on
//    top, writes point_offsets, and its LAST block stores the grand total (num_rendered) into the pinned host word total_host
//    (if given) as soon as it knows it -- while the kernel is still running.  No instance at or beyond `capacity` is written:
//    the caller launches this before it knows the total, into arrays sized from a guess, and repeats the launch with the exact
//    size when the guess was too small.
//  * scanned != 0: block_sums already holds exclusive block offsets (launch_offsets_blocksums, which also publishes the total):
//    the path of callers that must know the total before they can place the output.
void launch_offsets_blocksums(int P, uint32_t* block_sums, uint32_t* total_dev, uint32_t* total_host, hipStream_t s);
// total_dev (optional): device word that receives the total as well (for the kernels behind it, DevBin)
void launch_duplicate(int P, const ushort4* rect, const float* depths, const uint32_t* tiles_touched,
                      const uint32_t* block_sums, int scanned, uint32_t* point_offsets, int gx, uint64_t* keys, uint32_t* vals,
                      uint32_t capacity, uint32_t* total_host, hipStream_t s, uint32_t* total_dev = nullptr);
// stable LSD radix sort of (u64 key, u32 val) pairs on key bits [begin_bit, end_bit). Result lands in keys_a/vals_a;
// the unsorted input must sit in the "b" buffers when the pass count ceil((end-begin)/8) is odd, else in "a".
void launch_sort_pairs(int R, uint64_t* keys_a, uint32_t* vals_a, uint64_t* keys_b, uint32_t* vals_b, int begin_bit,
                       int end_bit, uint32_t* hist, size_t hist_elems, hipStream_t s);
// single-pass stable counting sort of the pairs by tile id (key >> 32) that also writes the tile ranges; the output is
// PACKED: keys_out[i] = id << 32 | depth bits (vals_out untouched);
// returns false (nothing launched) when there are more than GS2D_BIN_MAX_TILES tiles
// The binning chunk as the kernels see it while the HOST does not know num_rendered yet (gs2d_api.hip, "forward without a host
// reaction"): its base address, the device word duplicate_kernel's last workgroup stored the count in, the capacity C the chunk
// was sized for and the mode.  A kernel reads R = *R_dev, does NOTHING when R > cap (the guess was too small: the offsets of the
// layout would leave the chunk; the host notices the same number a little later and runs the stages again in a chunk of the right
// size, in stream order, before anybody can read an output), and otherwise derives its pointers from bin_layout(R, det, cap) --
// the very layout the backward computes from the R the host knows by then.
struct DevBin {
    char* base;
    const uint32_t* R_dev;
    uint32_t cap;
    int det;
};
// launch_bin_by_tile with the count read on the device: grids sized for `cap` instances, surplus workgroups leave at once
void launch_bin_by_tile_dev(const DevBin& db, int tiles, int nbits, uint2* ranges, hipStream_t s);
// launch_tile_depth_sort (packed pairs) with the count read on the device; cap_class: the LDS capacity class to use
void launch_tile_depth_sort_dev(const DevBin& db, int tiles, const uint2* ranges, int cap_class, int write_keys, hipStream_t s);
bool launch_bin_by_tile(int R, int tiles, int nbits, const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out,
                        uint32_t* vals_out, uint32_t* hist, uint2* ranges, hipStream_t s);
// one workgroup per tile: stable sort of the tile's segment by the 32 depth bits, in LDS.  packed: the segment holds
// (depth, id) pairs as written by launch_bin_by_tile (else 64-bit keys + ids); the sorted ids land in `vals`, the full
// 64-bit keys are written back only if write_keys
void launch_tile_depth_sort(int R, int tiles, const uint2* ranges, uint64_t* keys, uint32_t* vals, uint64_t* keys_alt,
                            uint32_t* vals_alt, int packed, int write_keys, hipStream_t s);
void launch_tile_ranges(int R, const uint64_t* keys, uint2* ranges, int tiles, hipStream_t s);
// Per-frame pointers of the per-Gaussian kernels in the batched calls (frames share the Gaussian SoA and the image size).
struct PreFwdFrame {
    const float *vm, *pm, *campos;
    int* radii; float* depths; float4* rec; uint32_t* tiles_touched; ushort4* rect; uint8_t* clamped; uint32_t* block_sums;
};
struct PreFwdFrames { PreFwdFrame f[8]; };
struct PreBwdFrame {
    const float *vm, *pm, *campos;
    int W, H;                      // rebuilt from focal * tan as the reference does (backward.cu:641-642)
    const float4* rec; const int* radii; const uint8_t* clamped; const float* grad_rec;
    float *dL_dtransMat, *dL_dnormal, *dL_dcolor, *dL_dopacity, *dL_dsh, *dL_dmean2D, *dL_dmean3D, *dL_dscale, *dL_drot;
};
struct PreBwdFrames { PreBwdFrame f[8]; };
void launch_preprocess_fwd_batch(int P, int K, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                                 const float* rotations, const float* opacities, const float* shs, const float* transMat_precomp,
                                 const float* colors_precomp, const CamParams& cam0, const PreFwdFrames& tab, hipStream_t s);
void launch_preprocess_bwd_batch(int P, int K, int D, int M, const float* means3D, const float* shs, const float* scales,
                                 const float* rotations, int need_record, const PreBwdFrames& tab, hipStream_t s);
// frame 0's dL_* outputs += those of frames 1 .. K-1, in frame order (fields that are NULL in frame 0 are skipped)
void launch_sum_frames(int P, int K, int M, const PreBwdFrames& tab, hipStream_t s);

// Per-frame pointers of the binning chain (duplicate, tile binning, depth sort) in the batched forward.
struct BinFrame {
    const ushort4* rect; const float* depths; const uint32_t* tiles_touched;
    const uint32_t* block_sums; // per-256-Gaussian sums of tiles_touched (preprocess)
    uint32_t* total_host;       // pinned host word for num_rendered (NULL: do not publish); capacity: instances the unsorted arrays hold
    uint32_t capacity;
    uint32_t* point_offsets;
    uint64_t* keys_unsorted; uint32_t* vals_unsorted;   // duplicate's output
    uint64_t* keys; uint32_t* point_list;                // binned (packed pairs) -> sorted ids
    uint64_t* keys_alt; uint32_t* vals_alt;
    uint32_t* hist;            // tiles x nblocks counters, the tile totals behind them
    uint2* ranges;
    int R, nblocks;
};
struct BinFrames { BinFrame f[8]; };
void launch_duplicate_batch(int P, int K, int gx, const BinFrames& tab, hipStream_t s);
// depth_sort = false: stop after the scatter (the forward blend kernel then sorts every tile's list as its first phase)
void launch_bin_sort_batch(int P, int K, int tiles, int gx, int nbits, BinFrames& tab, int write_keys, bool depth_sort, hipStream_t s);
// LDS capacity (elements) launch_tile_depth_sort picks for R instances on `tiles` tiles: 1536 / 2048 / 3072 / 4096
int tile_sort_capacity(long long R, int tiles);
#define GS2D_FUSED_SORT_CAP 1536  // the capacity at which the depth sort (16 B per element) fits the LDS a forward blend workgroup holds anyway
#define GS2D_FUSED_SORT_CAP_IDX 3072  // ... and at which the index sort (8 B per element, gs2d_tile_sort.h) does
// capacity the forward blend kernel sorts at itself for a capacity class of tile_sort_capacity (0: the stand-alone kernel)
static inline int gs2d_fused_sort_cap(int cap_class)
{
    return cap_class <= GS2D_FUSED_SORT_CAP ? GS2D_FUSED_SORT_CAP : (cap_class <= GS2D_FUSED_SORT_CAP_IDX ? GS2D_FUSED_SORT_CAP_IDX : 0);
}
// dynamic LDS (bytes) of that sort: elements + 4 x 256 digit counters
static inline size_t gs2d_fused_sort_lds(int sort_cap) { return (size_t)sort_cap * (sort_cap > GS2D_FUSED_SORT_CAP ? 8 : 16) + 4096; }

// Per-frame pointers of the blend kernels.  A launch handles K frames of the same size over the same Gaussians (K = 1: the
// plain call; K > 1: gs2d_forward_batch / gs2d_backward_batch, one grid over K x tiles).
#define GS2D_MAX_BATCH 8
struct BlendFwdFrame {
    const uint2* ranges; uint32_t* point_list; const float4* rec;
    float* out_color; float* out_others; float* pix_state;
    uint8_t* hits; uint8_t* hits4;  // written by phase 0 (cull bits, gs2d_cull.h)
    float4* zero;                   // the backward's gradient accumulator, cleared with the kernel's idle store slots
    uint64_t* keys; uint64_t* keys_alt; uint32_t* vals_alt;  // sort_cap > 0: the binned (depth, id) pairs of the tile lists + scratch
    DevBin dev;                     // dev.base != nullptr: point_list, hits, hits4, keys, keys_alt, vals_alt are derived from it
                                    // in the kernel (the host did not know num_rendered when it launched)
};
struct BlendFwdBatch { BlendFwdFrame f[GS2D_MAX_BATCH]; };
struct BlendBwdFrame {
    const uint2* ranges; const uint32_t* point_list; const float4* rec; const float* pix_state; const uint8_t* hits;
    const float* dL_dpix; const float* dL_dothers; float* grad_rec;
    float* det_slots;  // != nullptr selects the deterministic variant (single frame only): no atomics, per-(instance, quadrant)
                       // partial records (GS2D_GRAD_FLOATS floats each, R * 4 of them, zero-initialised by the caller)
    const uint16_t* hits16;  // the forward's sixteen group bits per (instance, quadrant): the eight-queue pose-only kernel derives
                             // its half-row bits from them (the other kernels read `hits`, the forward's four row bits)
    float* dense_m2d;  // != nullptr selects the POSE-ONLY variant (single frame, non-deterministic): grad_rec is then used as a
                       // dense float4[P] (dT[2], dT[5], dT[8], -) and dense_m2d as float2[P] dL_dmean2D.xy (both zero on entry)
};
struct BlendBwdBatch { BlendBwdFrame f[GS2D_MAX_BATCH]; };
// Writes hits (u16[4 * i + q] = the 2x2 pixel groups of quadrant q that instance i of the sorted list can touch, see
// gs2d_cull.h) for every instance, then blends.  Also clears zero_n float4 at `zero` (the backward's gradient accumulator)
// with its idle store slots.
// sort_cap > 0 (GS2D_FUSED_SORT_CAP: pair sort, GS2D_FUSED_SORT_CAP_IDX: index sort): every workgroup first sorts its tile's list by depth (the per-tile depth sort as phase
// -1, lists longer than sort_cap through global scratch), write_keys as in launch_tile_depth_sort
void launch_blend_fwd(int W, int H, int K, const BlendFwdFrame* frames, const float* bg, int use_sa, size_t zero_n, int sort_cap,
                      int write_keys, hipStream_t s);
// clear12 (optional): clear_n floats zeroed by the kernel (the pose-gradient sums the next stage accumulates into).
void launch_blend_bwd(int W, int H, int K, const BlendBwdFrame* frames, const float* bg, int use_sa, float* clear12, int clear_n,
                      hipStream_t s);
// deterministic mode: inv[unsorted instance] = sorted position, then grad_rec[g] = sum of g's slots in a fixed order
void launch_det_reduce(int P, int R, int W, int H, const uint2* ranges, const uint32_t* point_list, const ushort4* rect,
                       const uint32_t* tiles_touched, const uint32_t* point_offsets, const uint8_t* hits,
                       uint32_t* inv, const float* det_slots, float* grad_rec, hipStream_t s);
}  // namespace gs2d
