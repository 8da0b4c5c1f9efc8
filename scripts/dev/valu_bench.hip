// Microbenchmark: SIMD cycles per wave64 VALU instruction on gfx950 for the instruction kinds the blend kernels use.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2_ __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {  // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (KIND == 1) {  // 4 independent v_pk_fma_f32 (8 fmas)
            float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, mm = {m, m}, cc = {c, c};
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm), "v"(cc));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if (KIND == 2) {  // 8 dependent v_fma_f32 (one chain)
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(m), "v"(c));
        } else if (KIND == 3) {  // 8 v_exp_f32
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 4) {  // 8 v_readlane_b32 + use
            int s0, s1, s2, s3, s4, s5, s6, s7;
            asm volatile("v_readlane_b32 %0, %8, 3\n v_readlane_b32 %1, %9, 5\n v_readlane_b32 %2, %10, 7\n v_readlane_b32 %3, %11, 9\n"
                         "v_readlane_b32 %4, %12, 11\n v_readlane_b32 %5, %13, 13\n v_readlane_b32 %6, %14, 15\n v_readlane_b32 %7, %15, 17\n"
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7)
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
            a0 += __builtin_bit_cast(float, s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7) * 1e-30f;
        } else if (KIND == 5) {  // 8 v_mul_f32 with SGPR operand
            float sm = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
            asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sm));
        } else if (KIND == 6) {  // 8 v_cndmask with vcc
            asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n"
                         "v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int KIND> void run(const char* name, int wg_per_cu)
{
    float* d; (void)hipMalloc(&d, 256 * 256 * 16 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, grid = 256 * wg_per_cu;
    k<KIND><<<grid, 256>>>(d, 100, 1.f);
    hipEventRecord(e0);
    k<KIND><<<grid, 256>>>(d, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // waves per SIMD = wg_per_cu (256 threads = 4 waves = 1 per SIMD); instr per wave = iters*8
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-28s waves/SIMD %d: %.2f ms -> %.2f SIMD-cycles per wave-instruction (at 2.4 GHz)\n", name, wg_per_cu, ms,
           cyc / ((double)iters * 8 * wg_per_cu));
    (void)hipFree(d);
}
int main()
{
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0>("v_fma_f32 x8 independent", 1); run<1>("v_pk_fma_f32 (per pk instr)", 1); run<2>("v_fma_f32 dependent chain", 1); run<3>("v_exp_f32", 1); run<4>("v_readlane_b32", 1); run<5>("v_mul_f32 sgpr operand", 1); run<6>("v_cndmask vcc", 1); }
        if (w == 2) { run<0>("v_fma_f32 x8 independent", 2); run<1>("v_pk_fma_f32 (per pk instr)", 2); run<2>("v_fma_f32 dependent chain", 2); run<3>("v_exp_f32", 2); run<4>("v_readlane_b32", 2); run<5>("v_mul_f32 sgpr operand", 2); run<6>("v_cndmask vcc", 2); }
        if (w == 4) { run<0>("v_fma_f32 x8 independent", 4); run<1>("v_pk_fma_f32 (per pk instr)", 4); run<2>("v_fma_f32 dependent chain", 4); run<3>("v_exp_f32", 4); run<4>("v_readlane_b32", 4); run<5>("v_mul_f32 sgpr operand", 4); run<6>("v_cndmask vcc", 4); }
    }
    return 0;
}
