// Dev probe: cost of one LDS atomic instruction per wave, float vs integer, 52 active lanes on 52 different words (the
// backward's pattern), every SIMD of the chip busy with 5 waves.  Prints ns per wave-instruction per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KIND>
__global__ void __launch_bounds__(256) probe(float* out, int iters)
{
    __shared__ float acc[4][64 * 13];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < 64 * 13; i += 64) acc[wave][i] = 0.f;
    __syncthreads();
    const int row = lane >> 4, li = lane & 15;
    const bool on = li < 13;
    float v = 1.0f + lane;
    for (int it = 0; it < iters; it++) {
        const int slot = (it * 7 + row * 13) & 63;  // the four rows on four different splats
        if (on) {
            if (KIND == 0) atomicAdd(&acc[wave][slot * 13 + li], v);
            else if (KIND == 1) atomicAdd(reinterpret_cast<unsigned int*>(&acc[wave][slot * 13 + li]), (unsigned int)lane);
            else acc[wave][slot * 13 + li] = v;  // plain store
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[0][0];
}
int main()
{
    float* d; hipMalloc(&d, 4096 * 4);
    const int iters = 20000, grid = 256 * 5;
    const char* names[3] = {"ds_add_f32", "ds_add_u32", "ds_write_b32"};
    for (int k = 0; k < 3; k++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (k == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, 0, d, iters);
            if (k == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, 0, d, iters);
            if (k == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per CU: 5 workgroups x 4 waves x iters instructions
        printf("%-13s %8.3f ms  -> %.1f ns per wave-instruction per CU (%.1f LDS clocks at 2.1 GHz)\n", names[k], ms,
               ms * 1e6 / (20.0 * iters), ms * 1e6 / (20.0 * iters) * 2.1);
    }
    return 0;
}
