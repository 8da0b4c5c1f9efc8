// Dev probe: how many 256-thread workgroups of a 96-VGPR kernel does a CU hold as a function of the LDS per workgroup?
// (the blend kernels live at 5 workgroups per CU; DESIGN.md reports a cliff between 31 808 and 32 320 bytes)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) probe(float* out)
{
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = lds[255 - threadIdx.x];
}
int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: sharedMemPerMultiprocessor %zu, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu\n", p.name,
           (size_t)p.sharedMemPerMultiprocessor, (size_t)p.sharedMemPerBlock, (size_t)p.maxSharedMemoryPerMultiProcessor);
    for (size_t b = 30720; b <= 33792; b += 128) {
        int n = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, probe, 256, b);
        printf("dynamic LDS %6zu B -> %d workgroups per CU\n", b, n);
    }
    return 0;
}
