// Probe: which of the 16 reduced values does each lane hold after reduce16()?
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float swapadd32(float a, float b)  // lanes 0-31: a[l]+a[l+32]; lanes 32-63: b[l-32]+b[l]
{
    // v_permlane32_swap: a.hi <-> b.lo.  Inline asm because the ROCm 7.2 builtin drops its second result
    // (scripts/dev/swap_probe.hip); hipcc adds no wait states around asm, so the nops live in the string.
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float swapadd16(float a, float b)  // even rows: a (row r + row r+1); odd rows: b
{
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
template <int CTRL> __device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ float seladd(float a, float b, bool hi) {
    const float keep = hi ? b : a, give = hi ? a : b;
    return keep + dpp_get<CTRL>(give);
}
__device__ __forceinline__ float reduce16(const float v[16], int lane) {
    float c[8], d[4], e[2];
    for (int i = 0; i < 8; i++) c[i] = swapadd32(v[2 * i], v[2 * i + 1]);
    for (int i = 0; i < 4; i++) d[i] = swapadd16(c[2 * i], c[2 * i + 1]);
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
    e[0] = seladd<0x140>(d[0], d[1], b3);
    e[1] = seladd<0x140>(d[2], d[3], b3);
    float f = seladd<0x141>(e[0], e[1], b2);
    f += dpp_get<0x4E>(f);
    f += dpp_get<0xB1>(f);
    return f;
}
__global__ void probe(float* out) {
    const int lane = threadIdx.x;
    float v[16];
    // value i in lane l = (i+1) * 1000 + l  -> total_i = 64000*(i+1) + 2016
    for (int i = 0; i < 16; i++) v[i] = (float)((i + 1) * 1000 + lane);
    out[lane] = reduce16(v, lane);
}
int main() {
    float* d; hipMalloc(&d, 64 * 4);
    probe<<<1, 64>>>(d);
    float h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++) {
        float idx = (h[l] - 2016.f) / 64000.f - 1.f;
        int pred = 8 * ((l >> 2) & 1) + 4 * ((l >> 3) & 1) + 2 * ((l >> 4) & 1) + ((l >> 5) & 1);
        printf("lane %2d total %.1f -> idx %.3f  predicted %d %s\n", l, h[l], idx, pred, (fabsf(idx - pred) < 1e-3f) ? "" : "MISMATCH");
    }
    return 0;
}
