// Deterministic backward, second half (opt-in: gs2d_set_deterministic): the DET variant of blend_bwd_kernel leaves one
// partial gradient record per (instance, quadrant) in det_slots, written with plain stores by the single wave that owns
// that pair.  Here a Gaussian's records are summed in a FIXED order -- its tiles row-major as duplicate_kernel emits them
// (rasterizer_impl.cu:70-111), quadrants 0..3 -- so two runs give bit-identical gradients.  The reference accumulates with
// float atomics (backward.cu:343,396,441-460) and is run-to-run non-deterministic; SURVEY.md section 5 asks for this mode
// for parity work.  Costs an inverse permutation (sorted position of every unsorted instance) and ~80 B per touched pair.
#include "gs2d_common.h"

namespace {

// inv[first unsorted instance of g + (tile's index in g's rectangle)] = sorted position; one workgroup per tile
__global__ void __launch_bounds__(256)
det_inverse_kernel(int gx, int ntiles, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                   const ushort4* __restrict__ rect, const uint32_t* __restrict__ tiles_touched,
                   const uint32_t* __restrict__ point_offsets, uint32_t* __restrict__ inv)
{
    const int tile = blockIdx.x;
    if (tile >= ntiles) return;
    const uint2 range = ranges[tile];
    const int tx = tile % gx, ty = tile / gx;
    for (uint32_t i = range.x + threadIdx.x; i < range.y; i += 256) {
        const uint32_t id = point_list[i];
        const ushort4 r = rect[id];  // the rectangle duplicate_kernel (gs2d_binning.hip) walked
        const uint32_t local = (uint32_t)((ty - (int)r.y) * ((int)r.z - (int)r.x) + (tx - (int)r.x));
        inv[point_offsets[id] - tiles_touched[id] + local] = i;
    }
}

__global__ void __launch_bounds__(256)
det_reduce_kernel(int P, const uint32_t* __restrict__ tiles_touched, const uint32_t* __restrict__ point_offsets,
                  const uint32_t* __restrict__ inv, const uint32_t* __restrict__ hits, const float* __restrict__ det_slots,
                  float* __restrict__ grad_rec)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= P) return;
    float sum[18];
#pragma unroll
    for (int c = 0; c < 18; c++) sum[c] = 0.f;
    const uint32_t cnt = tiles_touched[g], start = point_offsets[g] - cnt;
    for (uint32_t u = 0; u < cnt; u++) {
        const uint32_t i = inv[start + u];
        const uint32_t h = hits[i];
        for (int q = 0; q < 4; q++) {
            if (((h >> (8 * q)) & 0xFFu) == 0u) continue;  // that quadrant never staged this instance (8 half-row bits per quadrant)
            const float* row = det_slots + ((size_t)i * 4 + q) * GS2D_GRAD_FLOATS;
#pragma unroll
            for (int c = 0; c < 18; c++) sum[c] += row[c];
        }
    }
    float* out = grad_rec + (size_t)g * GS2D_GRAD_FLOATS;
#pragma unroll
    for (int c = 0; c < 18; c++) out[c] = sum[c];
    out[18] = 0.f; out[19] = 0.f;
}

}  // namespace

namespace gs2d {

void launch_det_reduce(int P, int R, int W, int H, const uint2* ranges, const uint32_t* point_list, const ushort4* rect,
                       const uint32_t* tiles_touched, const uint32_t* point_offsets, const uint8_t* hits,
                       uint32_t* inv, const float* det_slots, float* grad_rec, hipStream_t s)
{
    const int gx = (W + GS2D_TILE - 1) / GS2D_TILE, gy = (H + GS2D_TILE - 1) / GS2D_TILE;
    if (R > 0)
        hipLaunchKernelGGL(det_inverse_kernel, dim3(gx * gy), dim3(256), 0, s, gx, gx * gy, ranges, point_list, rect,
                           tiles_touched, point_offsets, inv);
    hipLaunchKernelGGL(det_reduce_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, tiles_touched, point_offsets, inv,
                       reinterpret_cast<const uint32_t*>(hits), det_slots, grad_rec);
}

}  // namespace gs2d
