// Dev microbenchmark: throughput of the backward's gradient-atomic pattern in isolation.
// Each wave issues `iters` global_atomic_add_f32 instructions; in every instruction the four 16-lane rows address four
// different 20-float records (13 of 16 lanes active), records drawn from a per-tile pool (reuse as in the real kernel).
// hipcc --offload-arch=gfx950 -O3 scripts/dev/atomic_bench.hip -o scripts/dev/atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__global__ void __launch_bounds__(256) atomics(float* buf, const uint32_t* pool, int pool_per_tile, int iters, int active_lanes, int mode)
{
    const int tile = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, row = lane >> 4, li = lane & 15;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        const uint32_t h = hash32((uint32_t)(tile * 4 + wave) * 1000003u + (uint32_t)it * 4u + (uint32_t)row);
        const uint32_t id = pool[(size_t)tile * pool_per_tile + (h % (uint32_t)pool_per_tile)];
        float* dst = buf + (size_t)id * 20;
        const float v = 1.0f + (float)(it & 3);
        if (mode == 0) { if (li < active_lanes) atomicAdd(dst + li, v); }
        else if (mode == 1) { if (li < active_lanes) acc += dst[li]; }          // plain loads, same addresses
        else { if (li < active_lanes) dst[li] = v; }                               // plain stores
    }
    if (acc == 12345.f) buf[0] = acc;
}
int main(int argc, char** argv)
{
    const int P = 500000, tiles = 1200, pool_per_tile = 1100, iters = 214;
    float* buf; uint32_t* pool;
    hipMalloc(&buf, (size_t)P * 20 * 4); hipMemset(buf, 0, (size_t)P * 20 * 4);
    std::vector<uint32_t> h((size_t)tiles * pool_per_tile);
    srand(1);
    // spatially coherent pools: tile t draws ids from a window around t*P/tiles (neighbouring tiles overlap)
    for (int t = 0; t < tiles; t++)
        for (int i = 0; i < pool_per_tile; i++) {
            long base = (long)t * P / tiles + (rand() % 1600) - 800;
            if (base < 0) base += P; if (base >= P) base -= P;
            h[(size_t)t * pool_per_tile + i] = (uint32_t)base;
        }
    hipMalloc(&pool, h.size() * 4); hipMemcpy(pool, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; mode++)
        for (int active : {16, 13, 8, 4, 1}) {
            for (int w = 0; w < 3; w++) hipLaunchKernelGGL(atomics, dim3(tiles), dim3(256), 0, 0, buf, pool, pool_per_tile, iters, active, mode);
            hipEventRecord(e0);
            const int reps = 10;
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(atomics, dim3(tiles), dim3(256), 0, 0, buf, pool, pool_per_tile, iters, active, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
            const double winst = (double)tiles * 4 * iters, lanes = winst * 4 * active;
            printf("mode %d (%s) active %2d/16: %.1f us per launch, %.2f M wave-instr, %.1f G lane-ops/s, %.2f ns per wave-instr (whole GPU)\n", mode,
                   mode == 0 ? "atomic" : (mode == 1 ? "load" : "store"), active, ms * 1e3, winst / 1e6, lanes / (ms * 1e-3) / 1e9, ms * 1e6 / winst);
        }
    return 0;
}
