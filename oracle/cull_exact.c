/* Test infrastructure (not product): brute-force "which pixel groups of a tile can a splat reach" for checking the library's
 * conservative cull bits.  passes() restates the reference's per-pixel test (forward.cu:350-390: rho3d / rho2d, depth >= near,
 * alpha >= 1/255) in plain C, per pixel, with IEEE division and libm expf. */
#include <math.h>
#include <stdint.h>
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
/* exact 16 sub-block bits per instance in the library's layout (byte q = quadrant, bit r = its 4x4 sub-block r).
 * halfrows != 0: the layout of the eight-queue experiment builds instead (bit 2 r + h = half h, pixel rows 2h and 2h+1, of
 * sub-block r; gs2d_cull.h halfrows_from_groups) */
static int g_halfrows = 0;
void exact_set_halfrows(int on) { g_halfrows = on; }
void exact_bits(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* means2D, const float* tm,
                const float* normal_opacity, uint32_t* out)
{
    int gx = (W + 15) / 16, gy = (H + 15) / 16;
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < gx * gy; t++) {
        int tx = t % gx, ty = t / gx;
        for (uint32_t j = ranges[2 * t]; j < ranges[2 * t + 1]; j++) {
            uint32_t id = point_list[j], bits = 0;
            for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) {
                int px = tx * 16 + x, py = ty * 16 + y;
                if (px >= W || py >= H) continue;
                if (passes(tm + 9 * id, normal_opacity[4 * id + 3], (float)px, (float)py, means2D[2 * id], means2D[2 * id + 1])) {
                    int q = (y >> 3) * 2 + (x >> 3), r = ((y & 7) >> 2) * 2 + ((x & 7) >> 2), h = ((y & 3) >> 1);
                    bits |= g_halfrows ? 1u << (8 * q + 2 * r + h) : 1u << (8 * q + r);
                }
            }
            out[j] = bits;
        }
    }
}
/* exact 64 group bits per instance (2x2 pixel groups; bit 16 q + 4 (gy & 3) + (gx & 3), the forward's layout) */
void exact_group_bits(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* means2D, const float* tm,
                      const float* normal_opacity, uint64_t* out)
{
    int gx = (W + 15) / 16, gy = (H + 15) / 16;
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < gx * gy; t++) {
        int tx = t % gx, ty = t / gx;
        for (uint32_t j = ranges[2 * t]; j < ranges[2 * t + 1]; j++) {
            uint32_t id = point_list[j];
            uint64_t bits = 0;
            for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) {
                int px = tx * 16 + x, py = ty * 16 + y;
                if (px >= W || py >= H) continue;
                if (passes(tm + 9 * id, normal_opacity[4 * id + 3], (float)px, (float)py, means2D[2 * id], means2D[2 * id + 1])) {
                    int q = (y >> 3) * 2 + (x >> 3), g = (((y & 7) >> 1) << 2) | ((x & 7) >> 1);
                    bits |= 1ull << (16 * q + g);
                }
            }
            out[j] = bits;
        }
    }
}
