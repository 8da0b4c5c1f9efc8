#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL> __device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__global__ void probe(float* out) {
    const int lane = threadIdx.x;
    float a = (float)lane, b = (float)(100 + lane);
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    out[lane] = __builtin_bit_cast(float, r[0]);
    out[64 + lane] = __builtin_bit_cast(float, r[1]);
    auto q = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    out[128 + lane] = __builtin_bit_cast(float, q[0]);
    out[192 + lane] = __builtin_bit_cast(float, q[1]);
    out[256 + lane] = dpp_get<0x140>(a);
    out[320 + lane] = dpp_get<0x141>(a);
    out[384 + lane] = dpp_get<0x4E>(a);
    out[448 + lane] = dpp_get<0xB1>(a);
}
int main() {
    float* d; (void)hipMalloc(&d, 512 * 4);
    probe<<<1, 64>>>(d);
    float h[512]; (void)hipMemcpy(h, d, 2048, hipMemcpyDeviceToHost);
    const char* names[8] = {"pl32 r0", "pl32 r1", "pl16 r0", "pl16 r1", "row_mirror", "half_mirror", "quad 2301", "quad 1032"};
    for (int k = 0; k < 8; k++) { printf("%-12s:", names[k]); for (int l = 0; l < 64; l++) printf(" %g", h[64 * k + l]); printf("\n"); }
    return 0;
}
