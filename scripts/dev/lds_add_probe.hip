// Dev probe: do LDS float atomics (ds_add_f32) resolve same-address conflicts inside one wave instruction?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(float* out, int groups)
{
    __shared__ float acc[64];
    const int lane = threadIdx.x;
    acc[lane] = 0.f;
    __syncthreads();
    // lanes of one group target the same address; `groups` distinct addresses per instruction
    for (int it = 0; it < 100; it++) atomicAdd(&acc[lane % groups], 1.0f + lane);
    __syncthreads();
    out[lane] = acc[lane];
}
int main()
{
    float* d; hipMalloc(&d, 64 * 4);
    float h[64];
    for (int groups : {1, 4, 13, 16, 64}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, groups);
        hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
        double want0 = 0; for (int l = 0; l < 64; l++) if (l % groups == 0) want0 += 100.0 * (1 + l);
        printf("groups %2d: acc[0] = %.1f (want %.1f)\n", groups, h[0], want0);
    }
    return 0;
}
