// Fused dense Adam over the Gaussian SoA.  The reference steps five parameter tensors with
// torch.optim.Adam(l, lr=0.0, eps=1e-15) (scene/Gaussians.py:121-137): xyz | opacity | scaling | rotation | rgb, one
// learning rate per group, no weight decay, no amsgrad.  Here the five tensors are contiguous segments of ONE flat fp32
// buffer -- the same [13*P] layout the keyframe-sharded all-reduce bucket uses (ba_shard.py) -- so a step is one
// launch that streams 16 B in / 12 B out per element (grad, param, m, v -> param, m, v): pure HBM work.
#include "gs2d_common.h"
#include "../../include/gs2d_rasterizer.h"

namespace {

struct AdamGroups {
    int n;
    unsigned long long end[GS2D_ADAM_MAX_GROUPS];  // exclusive end offset (elements) of each group, ascending
    float step_size[GS2D_ADAM_MAX_GROUPS];         // lr / (1 - beta1^t)
};

__device__ __forceinline__ float group_step(const AdamGroups& G, size_t i)
{
    float s = G.step_size[0];
#pragma unroll
    for (int g = 1; g < GS2D_ADAM_MAX_GROUPS; g++)
        if (g < G.n && i >= G.end[g - 1]) s = G.step_size[g];
    return s;
}

// torch.optim.Adam single-tensor formulas (torch/optim/adam.py _single_tensor_adam, non-capturable):
//   m += (g - m) * (1 - b1);  v = v * b2 + (1 - b2) * g * g;  p -= step_size * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__device__ __forceinline__ void adam_one(float g, float& p, float& m, float& v, float one_m_b1, float b2, float one_m_b2,
                                         float inv_bc2_sqrt, float eps, float step_size)
{
    m = m + (g - m) * one_m_b1;
    v = v * b2 + one_m_b2 * g * g;
    const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

__global__ void __launch_bounds__(256)
adam_kernel(AdamGroups G, size_t n, float one_m_b1, float b2, float one_m_b2, float inv_bc2_sqrt, float eps,
            float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v)
{
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += stride) {
        const float4 g4 = ((const float4*)grad)[q];
        float4 p4 = ((float4*)param)[q], m4 = ((float4*)m)[q], v4 = ((float4*)v)[q];
        const size_t i = 4 * q;
        adam_one(g4.x, p4.x, m4.x, v4.x, one_m_b1, b2, one_m_b2, inv_bc2_sqrt, eps, group_step(G, i));
        adam_one(g4.y, p4.y, m4.y, v4.y, one_m_b1, b2, one_m_b2, inv_bc2_sqrt, eps, group_step(G, i + 1));
        adam_one(g4.z, p4.z, m4.z, v4.z, one_m_b1, b2, one_m_b2, inv_bc2_sqrt, eps, group_step(G, i + 2));
        adam_one(g4.w, p4.w, m4.w, v4.w, one_m_b1, b2, one_m_b2, inv_bc2_sqrt, eps, group_step(G, i + 3));
        ((float4*)param)[q] = p4; ((float4*)m)[q] = m4; ((float4*)v)[q] = v4;
    }
    if (blockIdx.x == 0)  // tail (n % 4 elements)
        for (size_t i = 4 * n4 + threadIdx.x; i < n; i += 256) {
            float p = param[i], mm = m[i], vv = v[i];
            adam_one(grad[i], p, mm, vv, one_m_b1, b2, one_m_b2, inv_bc2_sqrt, eps, group_step(G, i));
            param[i] = p; m[i] = mm; v[i] = vv;
        }
}

}  // namespace

extern "C" int gs2d_adam_step(int n_groups, const unsigned long long* group_end, const float* group_lr, float beta1, float beta2,
                              float eps, int step, unsigned long long n, float* param, const float* grad, float* exp_avg,
                              float* exp_avg_sq, void* stream)
{
    if (n_groups < 1 || n_groups > GS2D_ADAM_MAX_GROUPS || step < 1) return -1;
    if (n == 0) return 0;
    if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) return -1;
    AdamGroups G;
    G.n = n_groups;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    unsigned long long prev = 0;
    for (int g = 0; g < GS2D_ADAM_MAX_GROUPS; g++) {
        if (g < n_groups) {
            if (group_end[g] < prev) return -1;
            prev = group_end[g];
            G.end[g] = group_end[g];
            G.step_size[g] = (float)((double)group_lr[g] / bc1);
        } else { G.end[g] = ~0ull; G.step_size[g] = 0.f; }
    }
    if (group_end[n_groups - 1] != n) return -1;
    const size_t n4 = n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, G, (size_t)n, 1.0f - beta1, beta2,
                       1.0f - beta2, (float)(1.0 / sqrt(bc2)), eps, param, grad, exp_avg, exp_avg_sq);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
