// Probe: per-row (16-lane) 16-value butterfly used by the backward blend kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL> __device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ float seladd(float a, float b, bool hi) {
    const float keep = hi ? b : a, give = hi ? a : b;
    return keep + dpp_get<CTRL>(give);
}
__device__ __forceinline__ float reduce16_row(const float v[16], int lane) {
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
    float e[8], f[4], g[2];
    for (int i = 0; i < 8; i++) e[i] = seladd<0x140>(v[2 * i], v[2 * i + 1], b3);
    for (int i = 0; i < 4; i++) f[i] = seladd<0x141>(e[2 * i], e[2 * i + 1], b2);
    for (int i = 0; i < 2; i++) g[i] = seladd<0x4E>(f[2 * i], f[2 * i + 1], b1);
    return seladd<0xB1>(g[0], g[1], b0);
}
__global__ void probe(float* out) {
    const int lane = threadIdx.x, row = lane >> 4;
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = (float)((i + 1) * 1000 + (lane & 15) + 100000 * row);
    out[lane] = reduce16_row(v, lane);
}
int main() {
    float* d; (void)hipMalloc(&d, 64 * 4);
    probe<<<1, 64>>>(d);
    float h[64]; (void)hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        const int row = l >> 4;
        const int pred = 8 * (l & 1) + 4 * ((l >> 1) & 1) + 2 * ((l >> 2) & 1) + ((l >> 3) & 1);
        const float expect = 16.f * ((pred + 1) * 1000 + 100000 * row) + 120.f;
        if (h[l] != expect) { bad++; printf("lane %d got %.1f expected %.1f\n", l, h[l], expect); }
    }
    printf("reduce16_row mismatches: %d\n", bad);
    return 0;
}
