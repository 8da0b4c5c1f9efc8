// Fused post-op + loss of the SLAM iterations: the weight-normalised depth and outlier zeroing of
// render/__init__.py:46-49, the nan_to_num / masks / masked L1 sums-or-means of slam/Loss.py:22-58, and their
// gradients w.r.t. the rasterizer outputs, in two pixel-parallel kernels (reduce, then gradients) instead of ~30
// HW-sized PyTorch kernels.  Covers the reference's default loss (use_normal_loss = False, ignore_outliners = False,
// enable_exposure = False -- configs/replica/config*.py); other settings stay on the PyTorch path.
#include "gs2d_common.h"
#include "../../include/gs2d_rasterizer.h"

namespace {

struct LossCfg {
    int mode;  // 0 = tracking (masked sums, Loss.py:35-49), 1 = mapping (masked means, Loss.py:51-58)
    int use_weight_norm, use_edge_growth;
    float w_color, w_depth, w_dist, silmask_th, edge_thres, eps, depth_near, depth_far;
};

// torch.nan_to_num(x, 0, 0): nan -> 0, +inf -> 0, -inf -> lowest finite; gradient passes where x is finite
__device__ __forceinline__ float nan0(float v, bool& finite)
{
    finite = true;
    if (!(v == v)) { finite = false; return 0.f; }
    if (v == __builtin_inff()) { finite = false; return 0.f; }
    if (v == -__builtin_inff()) { finite = false; return -3.402823466e+38f; }
    return v;
}
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

struct PixelTerms {
    float c[3], d, a, dist, gtd;
    bool c_fin[3], d_fin, dist_fin, d_live;  // d_live: depth gradient reaches D/A (not an outlier, finite)
    bool depth_mask, color_mask;
    float inv_ae;  // 1 / (A + eps)
    float Draw;
};

__device__ __forceinline__ PixelTerms pixel_terms(const LossCfg& L, size_t HW, size_t pix, const float* __restrict__ color,
                                                  const float* __restrict__ allmap, const float* __restrict__ gt_depth)
{
    PixelTerms t;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) t.c[ch] = nan0(color[ch * HW + pix], t.c_fin[ch]);
    const float D = allmap[pix], A = allmap[HW + pix];
    t.Draw = D;
    float d = D;
    t.d_live = true;
    t.inv_ae = 1.f;
    if (L.use_weight_norm) {  // render/__init__.py:46-49
        t.inv_ae = 1.0f / (A + L.eps);
        d = D * t.inv_ae;
        if (d > L.depth_far || d < L.depth_near) { d = 0.f; t.d_live = false; }
    }
    bool fin;
    t.d = nan0(d, fin);
    t.d_live = t.d_live && fin;
    t.d_fin = fin;
    t.a = nan0(A, fin);
    t.dist = nan0(allmap[6 * HW + pix], t.dist_fin);
    t.gtd = gt_depth[pix];
    t.depth_mask = (t.gtd > 1e-5f) && (t.d > 1e-5f);
    if (L.mode == 0) t.color_mask = t.depth_mask && (t.a > L.silmask_th);
    else t.color_mask = L.use_edge_growth ? (t.a > L.edge_thres) : t.depth_mask;
    return t;
}

__device__ __forceinline__ double block_sum(double v, double* red)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Pass 1: per-block partial sums, partial[b*5 + k] with k = 0: sum |c - gt| over the colour mask, 1: sum |d - gt| over
// its mask, 2: sum dist over the colour mask, 3: #colour-mask pixels, 4: #depth-mask pixels  (doubles; no atomics: a few
// hundred blocks hammering five addresses cost 50 us, the per-block partials are re-reduced by every block of pass 2).
constexpr int LOSS_MAX_BLOCKS = 512;  // two workgroups per CU: the pass is latency-bound (13.8 us with 256, one wave per SIMD)

__global__ void __launch_bounds__(256)
loss_reduce_kernel(LossCfg L, int HWi, const float* __restrict__ color, const float* __restrict__ allmap,
                   const float* __restrict__ gt_color, const float* __restrict__ gt_depth, double* __restrict__ partial)
{
    __shared__ double red[4];
    const size_t HW = (size_t)HWi;
    double sc = 0, sd = 0, sdist = 0, nc = 0, nd = 0;
    for (size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x; pix < HW; pix += (size_t)gridDim.x * 256) {
        const PixelTerms t = pixel_terms(L, HW, pix, color, allmap, gt_depth);
        const bool dmask = L.mode == 0 ? t.color_mask : t.depth_mask;  // tracking uses ONE mask for both terms
        if (t.color_mask) {
            sc += (double)(fabsf(t.c[0] - gt_color[3 * pix]) + fabsf(t.c[1] - gt_color[3 * pix + 1]) +
                           fabsf(t.c[2] - gt_color[3 * pix + 2]));
            sdist += (double)t.dist;
            nc += 1.0;
        }
        if (dmask) { sd += (double)fabsf(t.d - t.gtd); nd += 1.0; }
    }
    const double v[5] = {sc, sd, sdist, nc, nd};
    for (int i = 0; i < 5; i++) {
        const double s = block_sum(v[i], red);
        if (threadIdx.x == 0) partial[blockIdx.x * 5 + i] = s;
    }
}

// Pass 2: every block first folds the (<= LOSS_MAX_BLOCKS) per-block partials into the five totals, then writes the gradients of
// its pixels; block 0 also writes the loss.
__global__ void __launch_bounds__(256)
loss_grad_kernel(LossCfg L, int HWi, int nparts, const float* __restrict__ color, const float* __restrict__ allmap,
                 const float* __restrict__ gt_color, const float* __restrict__ gt_depth, const double* __restrict__ partial,
                 float* __restrict__ loss_out, float* __restrict__ dL_dcolor, float* __restrict__ dL_dallmap,
                 const float* __restrict__ upstream)
{
    __shared__ double red[4];
    const size_t HW = (size_t)HWi;
    double acc[5];
    for (int i = 0; i < 5; i++) {
        double v = 0.0;
        for (int j = threadIdx.x; j < nparts; j += 256) v += partial[j * 5 + i];
        acc[i] = block_sum(v, red);
    }
    const double nc = acc[3], nd = acc[4];
    float gc_scale, gd_scale, gdist_scale;
    if (L.mode == 0) { gc_scale = L.w_color; gd_scale = L.w_depth; gdist_scale = 0.f; }
    else {  // masked means: colour over 3*Nc elements, depth over Nd, dist over Nc (empty mask -> NaN, as torch)
        gc_scale = (float)(L.w_color / (3.0 * nc));
        gd_scale = (float)(L.w_depth / nd);
        gdist_scale = (float)(L.w_dist / nc);
    }
    if (upstream) { const float u = upstream[0]; gc_scale *= u; gd_scale *= u; gdist_scale *= u; }  // dL/dloss from autograd
    if (loss_out && blockIdx.x == 0 && threadIdx.x == 0) {
        double loss;
        if (L.mode == 0) loss = L.w_color * acc[0] + L.w_depth * acc[1];
        else loss = L.w_color * (acc[0] / (3.0 * nc)) + L.w_depth * (acc[1] / nd) + L.w_dist * (acc[2] / nc);
        loss_out[0] = (float)loss;
        loss_out[1] = (float)acc[0]; loss_out[2] = (float)acc[1]; loss_out[3] = (float)acc[2];
        loss_out[4] = (float)nc; loss_out[5] = (float)nd;
    }
    if (!dL_dcolor) return;  // loss-only call
    for (size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x; pix < HW; pix += (size_t)gridDim.x * 256) {
        const PixelTerms t = pixel_terms(L, HW, pix, color, allmap, gt_depth);
        const bool dmask = L.mode == 0 ? t.color_mask : t.depth_mask;
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
            dL_dcolor[ch * HW + pix] = (t.color_mask && t.c_fin[ch]) ? gc_scale * sgn(t.c[ch] - gt_color[3 * pix + ch]) : 0.f;
        float gD = 0.f, gA = 0.f;
        if (dmask && t.d_live) {
            const float gd = gd_scale * sgn(t.d - t.gtd);
            if (L.use_weight_norm) { gD = gd * t.inv_ae; gA = -gd * t.Draw * t.inv_ae * t.inv_ae; }
            else gD = gd;
        }
        dL_dallmap[pix] = gD;
        dL_dallmap[HW + pix] = gA;
        dL_dallmap[2 * HW + pix] = 0.f;
        dL_dallmap[3 * HW + pix] = 0.f;
        dL_dallmap[4 * HW + pix] = 0.f;
        dL_dallmap[5 * HW + pix] = 0.f;
        dL_dallmap[6 * HW + pix] = (L.mode == 1 && t.color_mask && t.dist_fin) ? gdist_scale : 0.f;
    }
}

}  // namespace

extern "C" int gs2d_slam_loss(int mode, int width, int height, const float* color, const float* allmap,
                              const float* gt_color_hwc, const float* gt_depth, float w_color, float w_depth, float w_dist,
                              float silmask_th, float edge_thres, int use_edge_growth, int use_weight_norm, float eps,
                              float depth_near, float depth_far, double* workspace /* >= GS2D_LOSS_WS_DOUBLES doubles */, float* loss_out /* [8] */,
                              float* dL_dcolor, float* dL_dallmap, const float* upstream, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (width <= 0 || height <= 0 || (mode != 0 && mode != 1)) return -1;
    if ((dL_dcolor == nullptr) != (dL_dallmap == nullptr)) return -1;
    if (!loss_out && !dL_dcolor) return -1;
    LossCfg L;
    L.mode = mode; L.use_weight_norm = use_weight_norm; L.use_edge_growth = use_edge_growth;
    L.w_color = w_color; L.w_depth = w_depth; L.w_dist = w_dist; L.silmask_th = silmask_th; L.edge_thres = edge_thres;
    L.eps = eps; L.depth_near = depth_near; L.depth_far = depth_far;
    const int HW = width * height;
    const int blocks = (HW + 255) / 256;
    const int rgrid = blocks < LOSS_MAX_BLOCKS ? blocks : LOSS_MAX_BLOCKS;  // one partial per reduce block
    const int ggrid = blocks < 2048 ? blocks : 2048;
    static_assert(LOSS_MAX_BLOCKS * 5 <= GS2D_LOSS_WS_DOUBLES, "workspace too small");
    // loss_out != NULL: run the reduction (pass 1).  dL_d* != NULL: write gradients (pass 2, scaled by *upstream if
    // given).  A gradients-only call (loss_out == NULL) reuses the partial sums a previous loss call left in `workspace`.
    if (loss_out)
        hipLaunchKernelGGL(loss_reduce_kernel, dim3(rgrid), dim3(256), 0, s, L, HW, color, allmap, gt_color_hwc, gt_depth, workspace);
    hipLaunchKernelGGL(loss_grad_kernel, dim3(dL_dcolor ? ggrid : 1), dim3(256), 0, s, L, HW, rgrid, color, allmap, gt_color_hwc,
                       gt_depth, workspace, loss_out, dL_dcolor, dL_dallmap, upstream);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
