// Dev-only probes of the blend kernels (not part of the product): included by gs2d_blend.hip, where every macro below
// is a pass-through unless a probe build defines GS2D_EXPERIMENT=n or GS2D_PROFILE_WAVES (scripts/dev/variants.sh).
//  * GS2D_EXPERIMENT builds give WRONG results by design: each removes one ingredient of blend_bwd so that
//    scripts/dev/stage_ms.py can price it.  1: no global atomics in the flush   2: no butterfly   4: no flush loop
//    5: plain LDS store instead of the accumulate
//  * GS2D_PROFILE_WAVES records per wave [start, end, trips, hw_id] of the last launch (scripts/dev/wave_profile.py).
#pragma once
#if defined(GS2D_EXPERIMENT) && GS2D_EXPERIMENT == 1
#define GS2D_EXP_ATOMIC(X) if (va == 123.456f) grad_rec[flush_off] = vb;
#else
#define GS2D_EXP_ATOMIC(X) X
#endif
#if defined(GS2D_EXPERIMENT) && GS2D_EXPERIMENT == 4
#define GS2D_EXP_FLUSH(T) ((T) && f0 == 12345)
#else
#define GS2D_EXP_FLUSH(T) (T)
#endif
#if defined(GS2D_EXPERIMENT) && GS2D_EXPERIMENT == 5
#define GS2D_EXP_LDSADD(P, V) *(P) = (V)
#else
#define GS2D_EXP_LDSADD(P, V) atomicAdd(P, V)
#endif
#if defined(GS2D_EXPERIMENT) && GS2D_EXPERIMENT == 2
#define GS2D_EXP_BUTTERFLY tot = (g[0] + g[3]) + (g[9] + g[15]) + g[1] + g[2] + g[4] + g[5] + g[6] + g[7] + g[8] + g[10] + g[11]; if (false)
#else
#define GS2D_EXP_BUTTERFLY
#endif
#ifdef GS2D_PROFILE_WAVES
// dev-only instrumentation (scripts/dev/wave_profile.py): per wave [start, end, trips, hw_id] for the last launch
__device__ unsigned long long g_wave_prof[2][4 * 8192 * 4];
#define GS2D_PROF_BEGIN() const unsigned long long prof_t0 = wall_clock64(); unsigned int prof_trips = 0; unsigned long long prof_stage = 0, prof_s0 = 0
#define GS2D_PROF_TRIP() prof_trips++
#define GS2D_PROF_STAGE_BEGIN() prof_s0 = __builtin_amdgcn_s_memtime()
#define GS2D_PROF_STAGE_END() prof_stage += __builtin_amdgcn_s_memtime() - prof_s0
#define GS2D_PROF_END(K)                                                                                             \
    if (lane == 0 && blockIdx.x < 8192) {                                                                            \
        unsigned long long* pp = g_wave_prof[K] + ((size_t)blockIdx.x * 4 + wave) * 4;                               \
        pp[0] = prof_t0; pp[1] = wall_clock64(); pp[2] = prof_trips | (prof_stage << 32); /* staging time in shader cycles */                                                 \
        pp[3] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32); \
    }
#else
#define GS2D_PROF_BEGIN()
#define GS2D_PROF_TRIP()
#define GS2D_PROF_STAGE_BEGIN()
#define GS2D_PROF_STAGE_END()
#define GS2D_PROF_END(K)
#endif

#ifdef GS2D_EXPERIMENT
#define GS2D_BWD_LDS_ACCUM(JJ, V) if ((V) != 0.f) GS2D_EXP_LDSADD(&wb.acc()[((JJ) & 63) * NACC + acc_comp], V);
#endif

// dev A/B switches of round 4 (scripts/dev/variants.sh): GS2D_DEV_NO_CLASH = never take the LDS-atomic path of the accumulate
// (WRONG results where two groups meet on a splat: prices the atomics); GS2D_DEV_LDS_PAD = extra bytes of LDS per workgroup of
// blend_bwd (prices the occupancy cliff by itself)
#ifdef GS2D_DEV_NO_CLASH
#define GS2D_DEV_CLASH(X) false
#else
#define GS2D_DEV_CLASH(X) (X)
#endif
#ifndef GS2D_DEV_LDS_PAD
#define GS2D_DEV_LDS_PAD 0
#endif
