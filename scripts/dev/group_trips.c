// Dev tool (not product, not oracle): trips of the blend loops for different pixel-group sizes, from the oracle's forward
// state.  Model: batches of BS staged splats (those that reach the quadrant at all), trips of a batch = the longest group
// queue, a quadrant is walked up to its deepest last contributor.  With four 4x4 groups it reproduces the kernels' own trip
// counters (scripts/dev/wave_profile.py) to 0.4 %.  gcc -O2 -fopenmp -shared -fPIC -DBS=64 scripts/dev/group_trips.c -o /tmp/group_trips.so -lm
#ifndef BS
#define BS 64
#endif
#include <math.h>
#include <stdint.h>
#include <string.h>
static inline int passes(const float* T, float opa, float px, float py, float mx, float my)
{
    float kx = px * T[6] - T[0], ky = px * T[7] - T[1], kz = px * T[8] - T[2];
    float lx = py * T[6] - T[3], ly = py * T[7] - T[4], lz = py * T[8] - T[5];
    float p0 = ky * lz - kz * ly, p1 = kz * lx - kx * lz, p2 = kx * ly - ky * lx;
    if (p2 == 0.f) return 0;
    float sx = p0 / p2, sy = p1 / p2;
    float r3 = sx * sx + sy * sy, dx = mx - px, dy = my - py, r2 = 100.f * (dx * dx + dy * dy);
    float rho = fminf(r3, r2);
    float depth = (r3 <= r2) ? (sx * T[6] + sy * T[7]) + T[8] : T[8];
    if (depth < 0.2f) return 0;
    float a = fminf(0.99f, opa * expf(-0.5f * rho));
    return a >= 1.0f / 255.0f;
}
// out[0]: trips 4 groups of 4x4; out[1]: 8 groups of 4x2 (4 wide, 2 tall); out[2]: 8 groups of 2x4; out[3]: 16 groups 2x2; out[4] staged (any) count;
// out[5]: 32 groups of 2x1; out[6]: 64 groups of one pixel; out[7] / out[8]: 4 groups of 4x4 / 16 groups of 2x2 whose queues leave out
// the splats behind the GROUP's own deepest contributor (the kernels only stop at the quadrant's)
void trip8(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* means2D, const float* tm,
           const float* normal_opacity, const uint32_t* n_contrib, double* out)
{
    int gx = (W + 15) / 16, gy = (H + 15) / 16;
    double acc[9] = {0};
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < gx * gy; t++) {
        double loc[9] = {0};
        int tx = t % gx, ty = t / gx;
        uint32_t s = ranges[2 * t], e = ranges[2 * t + 1];
        for (int q = 0; q < 4; q++) {
            int qx = tx * 16 + (q & 1) * 8, qy = ty * 16 + (q >> 1) * 8;
            uint32_t qlast = 0;
            for (int i = 0; i < 64; i++) { int x = qx + (i & 7), y = qy + (i >> 3); uint32_t l = (x < W && y < H) ? n_contrib[y * W + x] : 0; if (l > qlast) qlast = l; }
            int fill = 0; int c4[4] = {0}, c8a[8] = {0}, c8b[8] = {0}, c16[16] = {0}, c32[32] = {0}, c64[64] = {0};
            int r4[4] = {0}, r16[16] = {0};                 /* queues that skip splats behind the GROUP's deepest contributor */
            uint32_t last4[4] = {0}, last16[16] = {0};
            for (int i = 0; i < 64; i++) { int x = i & 7, y = i >> 3; int xx = qx + x, yy = qy + y; uint32_t l = (xx < W && yy < H) ? n_contrib[yy * W + xx] : 0;
                int g4 = (y >> 2) * 2 + (x >> 2), g16 = (y >> 1) * 4 + (x >> 1); if (l > last4[g4]) last4[g4] = l; if (l > last16[g16]) last16[g16] = l; }
            for (uint32_t j = s; j < e && (j - s) < qlast; j++) {
                uint32_t id = point_list[j];
                uint64_t m = 0;
                for (int i = 0; i < 64; i++) {
                    int x = qx + (i & 7), y = qy + (i >> 3);
                    if (passes(tm + 9 * id, normal_opacity[4 * id + 3], (float)x, (float)y, means2D[2 * id], means2D[2 * id + 1])) m |= 1ull << i;
                }
                if (!m) continue;
                loc[4] += 1;
                for (int i = 0; i < 64; i++) if (m >> i & 1) {
                    int x = i & 7, y = i >> 3;
                    c4[(y >> 2) * 2 + (x >> 2)] |= 0x40000000; c8a[(y >> 1) * 2 + (x >> 2)] |= 0x40000000; c8b[(y >> 2) * 4 + (x >> 1)] |= 0x40000000;
                    c16[(y >> 1) * 4 + (x >> 1)] |= 0x40000000; c32[y * 4 + (x >> 1)] |= 0x40000000; c64[i] |= 0x40000000;
                }
                for (int i = 0; i < 64; i++) if (m >> i & 1) {
                    int x = i & 7, y = i >> 3; int g4 = (y >> 2) * 2 + (x >> 2), g16 = (y >> 1) * 4 + (x >> 1);
                    if ((j - s) < last4[g4]) r4[g4] |= 0x40000000;
                    if ((j - s) < last16[g16]) r16[g16] |= 0x40000000;
                }
#define COMMIT(A, N) for (int g = 0; g < N; g++) if (A[g] & 0x40000000) A[g] = (A[g] & 0x3fffffff) + 1;
                COMMIT(r4, 4) COMMIT(r16, 16)
                COMMIT(c4, 4) COMMIT(c8a, 8) COMMIT(c8b, 8) COMMIT(c16, 16) COMMIT(c32, 32) COMMIT(c64, 64)
                if (++fill == BS) {
#define CLOSE(A, N, K) { int mm = 0; for (int g = 0; g < N; g++) { if (A[g] > mm) mm = A[g]; A[g] = 0; } loc[K] += mm; }
                    CLOSE(c4, 4, 0) CLOSE(c8a, 8, 1) CLOSE(c8b, 8, 2) CLOSE(c16, 16, 3) CLOSE(c32, 32, 5) CLOSE(c64, 64, 6)
                    CLOSE(r4, 4, 7) CLOSE(r16, 16, 8)
                    fill = 0;
                }
            }
            CLOSE(c4, 4, 0) CLOSE(c8a, 8, 1) CLOSE(c8b, 8, 2) CLOSE(c16, 16, 3) CLOSE(c32, 32, 5) CLOSE(c64, 64, 6)
            CLOSE(r4, 4, 7) CLOSE(r16, 16, 8)
        }
#pragma omp critical
        for (int i = 0; i < 9; i++) acc[i] += loc[i];
    }
    memcpy(out, acc, sizeof(acc));
}
