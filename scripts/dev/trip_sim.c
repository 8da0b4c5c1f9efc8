// Dev tool (not product, not oracle): counts blend-loop trips under different sub-tile queue granularities, from the
// oracle's forward state.  gcc -O2 -fopenmp -shared -fPIC scripts/dev/trip_sim.c -o /tmp/trip_sim.so
#ifndef ROWLEVEL
#define ROWLEVEL 0
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
// out: [0] trips S4 (chunk-sync, 4 rows of 4x4), [1] trips S2 (chunk-sync, 16 groups of 2x2), [2] S4 decoupled, [3] S2 decoupled,
// [4] evaluated pairs S4, [5] evaluated pairs S2, [6] passing pairs, [7] wave-chunks, [8] S2 with 128-chunks, [9] S8 (one queue per wave)
void trip_sim(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* means2D, const float* tm,
              const float* normal_opacity, const uint32_t* n_contrib, double* out)
{
    int gx = (W + 15) / 16, gy = (H + 15) / 16;
    double acc[10] = {0}; double accx = 0, accy = 0;
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < gx * gy; t++) {
        double loc[10] = {0}; double locx = 0, locy = 0;
        int tx = t % gx, ty = t / gx;
        uint32_t s = ranges[2 * t], e = ranges[2 * t + 1];
        for (int q = 0; q < 4; q++) {
            int qx = tx * 16 + (q & 1) * 8, qy = ty * 16 + (q >> 1) * 8;
            uint32_t last[64];
            uint32_t qlast = 0;
            for (int i = 0; i < 64; i++) {
                int x = qx + (i & 7), y = qy + (i >> 3);
                last[i] = (x < W && y < H) ? n_contrib[y * W + x] : 0;
                if (last[i] > qlast) qlast = last[i];
            }
            uint32_t rowlast[4] = {0, 0, 0, 0};
            for (int i = 0; i < 64; i++) { int r = (i >> 5 << 1) | ((i & 7) >> 2); if (last[i] > rowlast[r]) rowlast[r] = last[i]; }
            double tot4[4] = {0}, tot2[16] = {0};
            int c2_128[16] = {0};
            int cb4[4] = {0}, cb_fill = 0; double cb_trips = 0;
            int fb4[4] = {0}, fb_fill = 0; double fb_trips = 0;
            for (uint32_t base = s; base < e && (base - s) < qlast; base += 64) {
                int c4[4] = {0}, c2[16] = {0}, c8 = 0;
                for (uint32_t j = base; j < e && j < base + 64; j++) {
                    uint32_t id = point_list[j];
                    uint64_t m = 0;
                    for (int i = 0; i < 64; i++) {
                        if (!ROWLEVEL && j - s >= last[i]) continue;
                        if (ROWLEVEL && j - s >= rowlast[i >> 5 << 1 | ((i & 7) >> 2)]) continue;
                        int x = qx + (i & 7), y = qy + (i >> 3);
                        if (passes(tm + 9 * id, normal_opacity[4 * id + 3], (float)x, (float)y, means2D[2 * id], means2D[2 * id + 1]))
                            m |= 1ull << i;
                    }
                    if (!m) continue;
                    c8++;
                    { /* exact 64-slot batches */
                        for (int r = 0; r < 4; r++) { uint64_t rm = 0; for (int yy = 0; yy < 4; yy++) rm |= 0xFull << (((r >> 1) * 4 + yy) * 8 + (r & 1) * 4); if (m & rm) fb4[r]++; }
                        if (++fb_fill == 64) { int mm = 0; for (int r = 0; r < 4; r++) { if (fb4[r] > mm) mm = fb4[r]; fb4[r] = 0; } fb_trips += mm; fb_fill = 0; }
                    }
                    loc[6] += __builtin_popcountll(m);
                    for (int r = 0; r < 4; r++) {
                        uint64_t rm = 0;
                        for (int yy = 0; yy < 4; yy++) rm |= 0xFull << (((r >> 1) * 4 + yy) * 8 + (r & 1) * 4);
                        if (m & rm) c4[r]++;
                    }
                    for (int g = 0; g < 16; g++) {
                        int gx2 = (g & 3) * 2, gy2 = (g >> 2) * 2;
                        uint64_t gm = (3ull << (gy2 * 8 + gx2)) | (3ull << ((gy2 + 1) * 8 + gx2));
                        if (m & gm) c2[g]++;
                    }
                }
                /* compacted batches: close the batch if this chunk's touched splats do not fit in 64 slots */
                if (cb_fill + c8 > 64) { int mm = 0; for (int r = 0; r < 4; r++) { if (cb4[r] > mm) mm = cb4[r]; cb4[r] = 0; } cb_trips += mm; cb_fill = 0; }
                cb_fill += c8; for (int r = 0; r < 4; r++) cb4[r] += c4[r];
                int m4 = 0, m2 = 0;
                for (int r = 0; r < 4; r++) { if (c4[r] > m4) m4 = c4[r]; loc[4] += 16.0 * c4[r]; tot4[r] += c4[r]; }
                for (int g = 0; g < 16; g++) { if (c2[g] > m2) m2 = c2[g]; loc[5] += 4.0 * c2[g]; tot2[g] += c2[g]; c2_128[g] += c2[g]; }
                loc[0] += m4; loc[1] += m2; loc[7] += 1; loc[9] += c8;
                if ((((base - s) / 64) & 1) == 1 || base + 64 >= e || (base + 64 - s) >= qlast) {
                    int mm = 0;
                    for (int g = 0; g < 16; g++) { if (c2_128[g] > mm) mm = c2_128[g]; c2_128[g] = 0; }
                    loc[8] += mm;
                }
            }
            { int mm = 0; for (int r = 0; r < 4; r++) if (cb4[r] > mm) mm = cb4[r]; cb_trips += mm; }
            loc[8] = loc[8] * 0 + loc[8]; loc[9] += 0; loc[5] += 0; loc[4] += 0; loc[3] += 0; loc[2] += 0; loc[1] += 0; loc[0] += 0; loc[7] += 0; loc[6] += 0;
            { int mm = 0; for (int r = 0; r < 4; r++) if (fb4[r] > mm) mm = fb4[r]; fb_trips += mm; }
            locx += cb_trips; locy += fb_trips;
            double d4 = 0, d2 = 0;
            for (int r = 0; r < 4; r++) if (tot4[r] > d4) d4 = tot4[r];
            for (int g = 0; g < 16; g++) if (tot2[g] > d2) d2 = tot2[g];
            loc[2] += d4; loc[3] += d2;
        }
#pragma omp critical
        { for (int i = 0; i < 10; i++) acc[i] += loc[i]; accx += locx; accy += locy; }
    }
    memcpy(out, acc, sizeof(acc)); out[10] = accx; out[11] = accy;
}
