// Probe: which of the 16 reduced values does each lane hold after reduce16()?
#include <hip/hip_runtime.h>
#include <stdio.h>
// v_permlane32_swap / v_permlane16_swap via inline asm: the ROCm 7.2 builtins drop their second result
// (scripts/dev/swap_probe.hip).  hipcc inserts no wait states around asm statements, so each level issues all of
// its (independent) swaps inside ONE asm block with the hazard nops at the block boundaries only.
__device__ __forceinline__ void swap32_x8(float v[16])  // pairs (v[2i], v[2i+1]): v[2i].hi <-> v[2i+1].lo
{
    asm("s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\t"
        "v_permlane32_swap_b32 %6, %7\n\tv_permlane32_swap_b32 %8, %9\n\tv_permlane32_swap_b32 %10, %11\n\t"
        "v_permlane32_swap_b32 %12, %13\n\tv_permlane32_swap_b32 %14, %15\n\t"
        "s_nop 1"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
          "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
}
__device__ __forceinline__ void swap16_x4(float c[8])  // pairs (c[2i], c[2i+1]): odd rows of c[2i] <-> even rows of c[2i+1]
{
    asm("s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
        "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
        "s_nop 1"
        : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
}
template <int CTRL> __device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ float seladd(float a, float b, bool hi) {
    const float keep = hi ? b : a, give = hi ? a : b;
    return keep + dpp_get<CTRL>(give);
}
__device__ __forceinline__ float reduce16(const float vin[16], int lane)
{
    float v[16], c[8], d[4], e[2];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = vin[i];
    swap32_x8(v);  // lanes 0-31 now hold both halves of v[2i], lanes 32-63 both halves of v[2i+1]
#pragma unroll
    for (int i = 0; i < 8; i++) c[i] = v[2 * i] + v[2 * i + 1];
    swap16_x4(c);
#pragma unroll
    for (int i = 0; i < 4; i++) d[i] = c[2 * i] + c[2 * i + 1];
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
    e[0] = seladd<0x140>(d[0], d[1], b3);  // row_mirror: l <-> 15-l flips bit 3
    e[1] = seladd<0x140>(d[2], d[3], b3);
    float f = seladd<0x141>(e[0], e[1], b2);  // row_half_mirror: l <-> 7-l flips bit 2
    f += dpp_get<0x4E>(f);                    // quad_perm [2,3,0,1]
    f += dpp_get<0xB1>(f);                    // quad_perm [1,0,3,2]
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
