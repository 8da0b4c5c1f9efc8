// Microbenchmark (gfx950): what do compares, selects and exec-mask idioms cost?  issue_bench.hip showed a block of
// "v_cmp vcc + 7 v_cndmask vcc" running at 16 cycles per instruction at ANY occupancy; this separates the ingredients.
// Wall time per wave-instruction per SIMD is the figure of merit (8 waves per SIMD, all 1024 SIMDs busy).
//   hipcc --offload-arch=gfx950 -O3 -o select_bench select_bench.hip && ./select_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define UNROLL 8

enum { K_FMA_DEP, K_CND_VCC_ONCE, K_CMP_VCC, K_CMP_SGPR, K_CMP_CND_PAIR, K_CND_SGPR, K_MAXMIN, K_CMPX, K_SAVEEXEC, K_CMP_CLASS,
       K_CND_INDEP, K_FMA_BANK, K_FMA_NOBANK, K_MED3, K_CMP_2CND, K_CMP_FMA_CND, K_SMOV_CND, K_CND_E64_VCC, K_CMP_6FMA_CND, K_CMP_SALU_CND, K_COUNT };
static const char* kind_name[K_COUNT] = {
    "v_fma_f32 dependent x8 (reference)", "v_cndmask vcc x8, vcc set once", "v_cmp_lt_f32 vcc x8", "v_cmp_lt_f32 s[..] (VOP3) x8",
    "(v_cmp vcc + v_cndmask vcc) x4", "v_cndmask s[..] (VOP3) x8", "v_max_f32/v_min_f32 x8", "v_cmpx_lt_f32 x4 + 4 fma",
    "s_and_saveexec + 6 fma + s_mov exec", "v_cmp_class vcc x8", "v_cndmask vcc x8 independent dst/src", "v_fma 3 src same bank x8",
    "v_fma 3 src distinct banks x8", "v_med3_f32 x8", "(v_cmp vcc + 2 v_cndmask vcc + fma) x2", "(v_cmp vcc + 2 fma + v_cndmask vcc) x2",
    "s_mov vcc + 7 v_cndmask vcc", "v_cndmask_b32_e64 ... vcc x8 (vcc set once)", "v_cmp vcc + 6 fma + v_cndmask vcc", "v_cmp vcc + s_and/s_or + v_cndmask vcc, x2 + 2 fma"};
static const int kind_instr[K_COUNT] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 7, 8, 8, 6};

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0001f, c = 0.5f;
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a0), "v"(a3) : "vcc");
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (KIND == K_FMA_DEP)
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(a0) : "v"(m), "v"(c));
            if (KIND == K_CND_VCC_ONCE)
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                             "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            if (KIND == K_CND_INDEP)
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            if (KIND == K_CMP_VCC)
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %4\n"
                             "v_cmp_lt_f32 vcc, %4, %5\n v_cmp_lt_f32 vcc, %5, %6\n v_cmp_lt_f32 vcc, %6, %7\n v_cmp_lt_f32 vcc, %7, %0\n"
                             :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");
            if (KIND == K_CMP_CLASS)
                asm volatile("v_cmp_class_f32 vcc, %0, %1\n v_cmp_class_f32 vcc, %1, %2\n v_cmp_class_f32 vcc, %2, %3\n v_cmp_class_f32 vcc, %3, %4\n"
                             "v_cmp_class_f32 vcc, %4, %5\n v_cmp_class_f32 vcc, %5, %6\n v_cmp_class_f32 vcc, %6, %7\n v_cmp_class_f32 vcc, %7, %0\n"
                             :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");
            if (KIND == K_CMP_SGPR) {
                unsigned long long s0, s1, s2, s3;
                asm volatile("v_cmp_lt_f32 %0, %4, %5\n v_cmp_lt_f32 %1, %5, %6\n v_cmp_lt_f32 %2, %6, %7\n v_cmp_lt_f32 %3, %7, %8\n"
                             "v_cmp_lt_f32 %0, %8, %9\n v_cmp_lt_f32 %1, %9, %10\n v_cmp_lt_f32 %2, %10, %11\n v_cmp_lt_f32 %3, %11, %4\n"
                             : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3)
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
                if ((s0 ^ s1 ^ s2 ^ s3) == 0x1234567ull) a0 += 1.f;
            }
            if (KIND == K_CMP_CND_PAIR)
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n"
                             "v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc\n v_cmp_lt_f32 vcc, %5, %4\n v_cndmask_b32 %7, %7, %6, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");
            if (KIND == K_CND_SGPR) {
                unsigned long long sm = 0x5555aaaa5555aaaaull + blockIdx.x;
                asm volatile("v_cndmask_b32 %0, %0, %1, %8\n v_cndmask_b32 %1, %1, %2, %8\n v_cndmask_b32 %2, %2, %3, %8\n v_cndmask_b32 %3, %3, %4, %8\n"
                             "v_cndmask_b32 %4, %4, %5, %8\n v_cndmask_b32 %5, %5, %6, %8\n v_cndmask_b32 %6, %6, %7, %8\n v_cndmask_b32 %7, %7, %0, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sm));
            }
            if (KIND == K_MAXMIN)
                asm volatile("v_max_f32 %0, %0, %1\n v_min_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_min_f32 %3, %3, %4\n"
                             "v_max_f32 %4, %4, %5\n v_min_f32 %5, %5, %6\n v_max_f32 %6, %6, %7\n v_min_f32 %7, %7, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            if (KIND == K_MED3)
                asm volatile("v_med3_f32 %0, %0, %1, %2\n v_med3_f32 %1, %1, %2, %3\n v_med3_f32 %2, %2, %3, %4\n v_med3_f32 %3, %3, %4, %5\n"
                             "v_med3_f32 %4, %4, %5, %6\n v_med3_f32 %5, %5, %6, %7\n v_med3_f32 %6, %6, %7, %0\n v_med3_f32 %7, %7, %0, %1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            if (KIND == K_CMPX) {
                unsigned long long save;
                asm volatile("s_mov_b64 %8, exec\n v_cmpx_lt_f32 exec, %0, %1\n v_fma_f32 %2, %2, %9, %10\n s_mov_b64 exec, %8\n"
                             "v_cmpx_lt_f32 exec, %1, %0\n v_fma_f32 %3, %3, %9, %10\n s_mov_b64 exec, %8\n"
                             "v_cmpx_lt_f32 exec, %4, %5\n v_fma_f32 %6, %6, %9, %10\n s_mov_b64 exec, %8\n"
                             "v_cmpx_lt_f32 exec, %5, %4\n v_fma_f32 %7, %7, %9, %10\n s_mov_b64 exec, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&s"(save)
                             : "v"(m), "v"(c));
            }
            if (KIND == K_SAVEEXEC) {
                unsigned long long save;
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_and_saveexec_b64 %8, vcc\n v_fma_f32 %2, %2, %9, %10\n v_fma_f32 %3, %3, %9, %10\n"
                             "v_fma_f32 %4, %4, %9, %10\n v_fma_f32 %5, %5, %9, %10\n v_fma_f32 %6, %6, %9, %10\n v_fma_f32 %7, %7, %9, %10\n s_mov_b64 exec, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&s"(save)
                             : "v"(m), "v"(c) : "vcc");
            }
            if (KIND == K_CMP_2CND)
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %2, vcc\n v_fma_f32 %0, %0, %8, %9\n"
                             "v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %6, vcc\n v_fma_f32 %4, %4, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "vcc");
            if (KIND == K_CMP_FMA_CND)
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %3, vcc\n"
                             "v_cmp_lt_f32 vcc, %4, %5\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %7, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "vcc");
            if (KIND == K_SMOV_CND)
                asm volatile("s_mov_b64 vcc, 0x5555\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                             "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");
            if (KIND == K_CND_E64_VCC)
                asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %1, %1, %2, vcc\n v_cndmask_b32_e64 %2, %2, %3, vcc\n v_cndmask_b32_e64 %3, %3, %4, vcc\n"
                             "v_cndmask_b32_e64 %4, %4, %5, vcc\n v_cndmask_b32_e64 %5, %5, %6, vcc\n v_cndmask_b32_e64 %6, %6, %7, vcc\n v_cndmask_b32_e64 %7, %7, %0, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            if (KIND == K_CMP_6FMA_CND)
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %4, %4, %8, %9\n"
                             "v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n v_cndmask_b32 %2, %2, %3, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "vcc");
            if (KIND == K_CMP_SALU_CND) {
                unsigned long long sm = 0x5555aaaa5555aaaaull + blockIdx.x, st;
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 %8, vcc, %9\n s_or_b64 %8, %8, %9\n v_cndmask_b32 %2, %2, %3, vcc\n v_fma_f32 %0, %0, %10, %11\n"
                             "v_cmp_lt_f32 vcc, %4, %5\n s_and_b64 %8, vcc, %9\n s_or_b64 %8, %8, %9\n v_cndmask_b32 %6, %6, %7, vcc\n v_fma_f32 %4, %4, %10, %11\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&s"(st) : "s"(sm), "v"(m), "v"(c) : "vcc");
            }
            if (KIND == K_FMA_BANK)  // all three sources in one VGPR bank (register number mod 4 equal)
                asm volatile("v_fma_f32 v20, v20, v24, v28\n v_fma_f32 v21, v21, v25, v29\n v_fma_f32 v22, v22, v26, v30\n v_fma_f32 v23, v23, v27, v31\n"
                             "v_fma_f32 v20, v20, v24, v28\n v_fma_f32 v21, v21, v25, v29\n v_fma_f32 v22, v22, v26, v30\n v_fma_f32 v23, v23, v27, v31\n"
                             ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
            if (KIND == K_FMA_NOBANK)  // three sources in three different banks
                asm volatile("v_fma_f32 v20, v20, v25, v30\n v_fma_f32 v21, v21, v26, v31\n v_fma_f32 v22, v22, v27, v28\n v_fma_f32 v23, v23, v24, v29\n"
                             "v_fma_f32 v20, v20, v25, v30\n v_fma_f32 v21, v21, v26, v31\n v_fma_f32 v22, v22, v27, v28\n v_fma_f32 v23, v23, v24, v29\n"
                             ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND> void run(int wg_per_cu, float* d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1000, grid = 256 * wg_per_cu;
    k<KIND><<<grid, 256>>>(d, 20, 1.f);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
    hipEventRecord(e0);
    k<KIND><<<grid, 256>>>(d, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * UNROLL * kind_instr[KIND] * wg_per_cu;  // wave-instructions per SIMD
    printf("%-44s w/SIMD %d: %7.3f ms -> %6.3f ns per wave-instr per SIMD (= %5.2f cyc @2.4 GHz)\n", kind_name[KIND], wg_per_cu, ms,
           ms * 1e6 / n, ms * 1e6 / n * 2.4);
    hipEventDestroy(e0); hipEventDestroy(e1);
}
template <int KIND> void sweep(float* d) { for (int w : {1, 2, 5, 8}) run<KIND>(w, d); }

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    if (!getenv("SB_ONLY_NEW")) {
    sweep<K_FMA_DEP>(d); sweep<K_FMA_BANK>(d); sweep<K_FMA_NOBANK>(d); sweep<K_CND_VCC_ONCE>(d); sweep<K_CND_INDEP>(d); sweep<K_CND_SGPR>(d);
    sweep<K_CMP_VCC>(d); sweep<K_CMP_CLASS>(d); sweep<K_CMP_SGPR>(d); sweep<K_CMP_CND_PAIR>(d); sweep<K_MAXMIN>(d); sweep<K_MED3>(d);
    }
    // (K_CMPX / K_SAVEEXEC are not run by default: the exec-writing asm blocks stalled the first run on the GPU box)
    sweep<K_CMP_2CND>(d); sweep<K_CMP_FMA_CND>(d); sweep<K_CMP_6FMA_CND>(d); sweep<K_SMOV_CND>(d); sweep<K_CND_E64_VCC>(d); sweep<K_CMP_SALU_CND>(d);
    return 0;
}
