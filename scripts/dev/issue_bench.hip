// Microbenchmark (gfx950): SIMD cycles per wave64 instruction, measured IN SHADER CYCLES (s_memtime) so that the clock the
// chip holds under load does not enter, at 1..8 waves per SIMD.  Kinds cover what the blend kernels issue: plain / packed
// fp32 VALU, transcendentals, DPP adds, selects, scalar-operand VALU, SALU queue bookkeeping beside VALU, and per-row
// ds_read_b128.  Output is kept under profiles/ (it justifies bench.py's VALU ceiling).
//   hipcc --offload-arch=gfx950 -O3 -o issue_bench issue_bench.hip && ./issue_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float float2_ __attribute__((ext_vector_type(2)));
#define UNROLL 8

enum { K_FMA, K_PKFMA, K_FMA_DEP, K_EXP, K_RCP, K_CNDMASK, K_DPPADD, K_MUL_SGPR, K_BFE, K_FMA_SALU4, K_FMA_SALU8, K_FMA_SALU16,
       K_SALU_ONLY, K_LDS128, K_LDS128_FMA, K_READLANE, K_FMA_EXP, K_COUNT };
static const char* kind_name[K_COUNT] = {
    "v_fma_f32 x8 indep", "v_pk_fma_f32 x4 (per pk)", "v_fma_f32 x8 dependent", "v_exp_f32 x8", "v_rcp_f32 x8",
    "v_cndmask_b32 x8", "v_add_f32_dpp x8", "v_mul_f32 sgpr x8", "v_bfe_u32 x8", "8 v_fma + 13 salu (per vfma)",
    "8 v_fma + 26 salu (per vfma)", "8 v_fma + 53 salu (per vfma)", "salu x53 (per salu)", "ds_read_b128 x4 per-row addr (per read)",
    "4 ds_read_b128 + 8 v_fma (per vfma)", "v_readlane_b32 x8", "6 v_fma + 2 v_exp (per instr)"};
static const int kind_instr[K_COUNT] = {8, 4, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 53, 4, 8, 8, 8};

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters, float seed)
{
    __shared__ float4 lds[4][64 * 4];  // 16 KiB static
    extern __shared__ char pad_lds[];  // sized by the host so that exactly w workgroups fit on a CU
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0001f, c = 0.5f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < 64 * 4; i += 64) lds[wave][i] = make_float4(a0, a1, a2, a3);
    unsigned long long q0 = 0x123456789abcdefull + blockIdx.x, q1 = ~q0, q2 = q0 * 3, q3 = q0 * 5;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        if (KIND == K_FMA || KIND == K_FMA_SALU4 || KIND == K_FMA_SALU8 || KIND == K_FMA_SALU16 || KIND == K_LDS128_FMA) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        }
        if (KIND == K_FMA_SALU4 || KIND == K_FMA_SALU8 || KIND == K_FMA_SALU16 || KIND == K_SALU_ONLY) {
            // the queue bookkeeping of the blend loops: per rep 2 s_ff1_i32_b64, 2 s_and_b64, 3 s_add_u32 + 3 s_addc_u32, xor / or /
            // bitset = 13 SALU instructions (counted in the ISA)
            const int reps = KIND == K_FMA_SALU4 ? 1 : (KIND == K_FMA_SALU8 ? 2 : 4);
#pragma unroll
            for (int r = 0; r < reps; r++) {
                // compiler-generated SALU (values are wave-uniform): s_ff1_i32_b64, 64-bit lowest-bit clear, adds
                asm volatile("" : "+s"(q0), "+s"(q1));
                const int j0 = __builtin_ctzll(q0), j1 = __builtin_ctzll(q1);
                q0 &= q0 - 1; q1 &= q1 - 1;
                q2 += (unsigned)j0; q3 ^= (unsigned)j1;
                q0 |= 0x8000000000000000ull; q1 |= 0x4000000000000000ull;
            }
        }
        if (KIND == K_PKFMA) {
            float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, mm = {m, m}, cc = {c, c};
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm), "v"(cc));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        }
        if (KIND == K_FMA_DEP) {
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(m), "v"(c));
        }
        if (KIND == K_EXP) {
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
        if (KIND == K_RCP) {
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
        if (KIND == K_FMA_EXP) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_exp_f32 %3, %3\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        }
        if (KIND == K_CNDMASK) {
            asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n"
                         "v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");
        }
        if (KIND == K_DPPADD) {
            asm volatile("s_nop 1\n v_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %4, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %1, %0, %0 row_mirror row_mask:0xf bank_mask:0xc\n v_add_f32_dpp %3, %2, %2 row_mirror row_mask:0xf bank_mask:0xc\n"
                         "v_add_f32_dpp %5, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xa\n v_add_f32_dpp %7, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xa\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
        if (KIND == K_MUL_SGPR) {
            float sm = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
            asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sm));
        }
        if (KIND == K_BFE) {
            asm volatile("v_bfe_u32 %0, %0, 3, 29\n v_bfe_u32 %1, %1, 3, 29\n v_bfe_u32 %2, %2, 3, 29\n v_bfe_u32 %3, %3, 3, 29\n v_bfe_u32 %4, %4, 3, 29\n v_bfe_u32 %5, %5, 3, 29\n v_bfe_u32 %6, %6, 3, 29\n v_bfe_u32 %7, %7, 3, 29\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
        if (KIND == K_LDS128 || KIND == K_LDS128_FMA) {
            // per-row address (4 distinct addresses per wave), as the blend loops read their staged records
            const int j = ((lane >> 4) * 13 + i * UNROLL + u) & 63;
            const float4 r0 = lds[wave][j], r1 = lds[wave][64 + j], r2 = lds[wave][128 + j], r3 = lds[wave][192 + j];
            a0 += r0.x; a1 += r1.y; a2 += r2.z; a3 += r3.w;
        }
        if (KIND == K_READLANE) {
            int s0, s1, s2, s3, s4, s5, s6, s7;
            asm volatile("v_readlane_b32 %0, %8, 3\n v_readlane_b32 %1, %9, 5\n v_readlane_b32 %2, %10, 7\n v_readlane_b32 %3, %11, 9\n"
                         "v_readlane_b32 %4, %12, 11\n v_readlane_b32 %5, %13, 13\n v_readlane_b32 %6, %14, 15\n v_readlane_b32 %7, %15, 17\n"
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7)
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
            q2 += (unsigned)(s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7);
        }
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(q0 ^ q1 ^ q2 ^ q3);
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
    if (seed == 12345.f) pad_lds[threadIdx.x] = 1;
}

template <int KIND> void run(int wg_per_cu, float* d, unsigned long long* dc)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1000, grid = 256 * wg_per_cu;
    // 256-thread workgroups = one wave per SIMD each; dynamic LDS sized so that at most wg_per_cu fit on a CU, i.e. the
    // 256 * wg_per_cu workgroups of the grid are all resident with exactly wg_per_cu waves on every SIMD
    const long want = 163840 / wg_per_cu - 16384 - 256;
    const size_t dyn = want > 0 ? (size_t)(want / 1024) * 1024 : 0;
    if (dyn + 16384 > 163840 || (wg_per_cu + 1) * (dyn + 16384) <= 163840) { printf("bad LDS sizing for %d\n", wg_per_cu); exit(1); }
    if (hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) { printf("set attribute failed (%zu)\n", dyn); exit(1); }
    if (getenv("IB_VERBOSE")) printf("launch kind %d w %d dyn %zu\n", KIND, wg_per_cu, dyn);
    k<KIND><<<grid, 256, dyn>>>(d, dc, 20, 1.f);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
    hipEventRecord(e0);
    k<KIND><<<grid, 256, dyn>>>(d, dc, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 4);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2], mx = (double)h.back(), mn = (double)h.front();
    const double n = (double)iters * UNROLL * kind_instr[KIND];
    // a SIMD hosts wg_per_cu waves; cycles the SIMD spends per wave-instruction = per-wave elapsed / (instr * waves)
    printf("%-42s w/SIMD %d: %7.3f ms | per-wave cyc min %8.0f med %8.0f max %8.0f -> %5.2f SIMD-cyc per wave-instr (med), %5.2f (max); clock >= %.2f GHz\n",
           kind_name[KIND], wg_per_cu, ms, mn, med, mx, med / (n * wg_per_cu), mx / (n * wg_per_cu), mx / (ms * 1e-3) * 1e-9);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int KIND> void sweep(float* d, unsigned long long* dc)
{
    for (int w : {1, 2, 3, 4, 5, 6, 8}) run<KIND>(w, d, dc);
}

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    float* d; unsigned long long* dc;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    (void)hipMalloc(&dc, 256 * 8 * 4 * 8);
    sweep<K_FMA>(d, dc); sweep<K_PKFMA>(d, dc); sweep<K_FMA_DEP>(d, dc); sweep<K_EXP>(d, dc); sweep<K_RCP>(d, dc);
    sweep<K_FMA_EXP>(d, dc); sweep<K_CNDMASK>(d, dc); sweep<K_DPPADD>(d, dc); sweep<K_MUL_SGPR>(d, dc); sweep<K_BFE>(d, dc);
    sweep<K_FMA_SALU4>(d, dc); sweep<K_FMA_SALU8>(d, dc); sweep<K_FMA_SALU16>(d, dc); sweep<K_SALU_ONLY>(d, dc);
    sweep<K_LDS128>(d, dc); sweep<K_LDS128_FMA>(d, dc); sweep<K_READLANE>(d, dc);
    return 0;
}
