/*
 * gs2d_oracle_f64.c -- CPU ORACLE, float64 evaluation of the backward (test infrastructure, NOT product code).
 *
 * Purpose: a yardstick for float32 rounding.  The float32 oracle (gs2d_oracle.c) and the HIP kernels evaluate the same
 * backward (RAST/cuda_rasterizer/backward.cu:143-664) in two different float32 operation orders; which of the two is closer
 * to the exact value of that function cannot be told from their difference.  The functions below evaluate it in double:
 *   - INPUTS are exactly the float32 values both float32 paths consume (the forward's stored per-pixel state, the float32
 *     splat records, the float32 upstream gradients, the float32 parameters);
 *   - every DISCRETE decision (the skips of backward.cu:301-320, ray-splat vs low-pass branch, the alpha clamp at 0.99, the
 *     T < 0.5 gate of the surface-aware confidence, backward.cu:347-351) is taken from a float32 shadow computation that
 *     repeats gs2d_oracle.c's orc_blend_bwd arithmetic, so all three paths walk the same branch of the piecewise function
 *     (pixels whose decisions are within rounding of a threshold are kept out of such comparisons by the tests);
 *   - all continuous arithmetic is double, exp() is libm's double exp.
 * err(float32 path, this) is then that path's rounding error, per entry.
 *
 * PARITY STATUS: as gs2d_oracle.c -- "parity unpinned" against the reference binary.
 * Build: part of libgs2d_oracle.so (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16
#define NEAR_N 0.2f
#define FAR_N 100.0f
#define FILTER_INV_SQ 100.0f
#define ACC_STRIDE 20 /* layout of orc_blend_bwd: [0..2] colour, [3..5] normal, [6..14] dT, [15,16] mean2D, [17] opacity */

/*
 * backward.cu:143-463 in double on float32 decisions.  Outputs (double): dL_dtransMat [P,9], dL_dmean2D [P,3] (z = 0),
 * dL_dnormal3D [P,3], dL_dopacity [P], dL_dcolors [P,3].
 */
void orc_blend_bwd_f64(int P, int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                       const float* bg, const float* means2D, const float* normal_opacity,
                       const float* transMats, const float* colors, const float* final_Ts,
                       const uint32_t* n_contrib, const float* dL_dpixels, const float* dL_depths,
                       const float* median_depth, const float* depth_std, int use_sa,
                       double* dL_dtransMat, double* dL_dmean2D, double* dL_dnormal3D,
                       double* dL_dopacity, double* dL_dcolors)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const size_t HW = (size_t)H * W;
    double* acc = (double*)calloc((size_t)P * ACC_STRIDE + 1, sizeof(double));
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < TILE; ly++)
            for (int lx = 0; lx < TILE; lx++) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                const size_t pix = (size_t)W * py + px;
                const float pxf = (float)px, pyf = (float)py;
                const double pxd = px, pyd = py;
                const float T_final = final_Ts[pix];
                float Tf = T_final;        /* float32 shadow of the transmittance: decides T < 0.5 */
                double T = (double)T_final;
                uint32_t contributor = r1 - r0;
                const uint32_t last_contributor = n_contrib[pix];
                const uint32_t median_contributor = n_contrib[pix + HW];
                const double dL_ddepth = dL_depths[0 * HW + pix], dL_daccum = dL_depths[1 * HW + pix];
                const double dL_dreg = dL_depths[6 * HW + pix], dL_dmedian_depth = dL_depths[5 * HW + pix];
                const double dL_dn[3] = {dL_depths[2 * HW + pix], dL_depths[3 * HW + pix], dL_depths[4 * HW + pix]};
                const double dpx[3] = {dL_dpixels[pix], dL_dpixels[HW + pix], dL_dpixels[2 * HW + pix]};
                const double mm = median_depth[pix], mstd = depth_std[pix];
                const double final_D = final_Ts[pix + HW], final_D2 = final_Ts[pix + 2 * HW];
                const double final_A = 1.0 - (double)T_final;
                const double bg_dot = (double)bg[0] * dpx[0] + (double)bg[1] * dpx[1] + (double)bg[2] * dpx[2];
                /* backward.cu:349: conf = exp(-(c_d - mm)^2 / (4 max(mstd / (1 - T_final), 1e-7))) */
                const double sa_k = 1.0 / (4.0 * fmax(mstd / (1.0 - (double)T_final), (double)1e-7f));
                const double c1 = (double)FAR_N / ((double)FAR_N - (double)NEAR_N);
                double accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, accum_normal_rec[3] = {0, 0, 0}, last_normal[3] = {0, 0, 0};
                double last_depth = 0, accum_depth_rec = 0, accum_alpha_rec = 0, last_dL_dT = 0, last_alpha = 0;
                for (uint32_t it = r1; it-- > r0;) {
                    contributor--;
                    if (contributor >= last_contributor) continue;
                    const uint32_t g = point_list[it];
                    const float* Tm = transMats + 9 * (size_t)g;
                    /* ---- float32 shadow: the arithmetic of orc_blend_bwd, for the decisions only */
                    const float kf[3] = {fmaf(pxf, Tm[6], -Tm[0]), fmaf(pxf, Tm[7], -Tm[1]), fmaf(pxf, Tm[8], -Tm[2])};
                    const float lf[3] = {fmaf(pyf, Tm[6], -Tm[3]), fmaf(pyf, Tm[7], -Tm[4]), fmaf(pyf, Tm[8], -Tm[5])};
                    const float p0f = fmaf(kf[1], lf[2], -(kf[2] * lf[1]));
                    const float p1f = fmaf(kf[2], lf[0], -(kf[0] * lf[2]));
                    const float p2f = fmaf(kf[0], lf[1], -(kf[1] * lf[0]));
                    if (p2f == 0.0f) continue;
                    const float ipf = 1.0f / p2f;
                    const float s0f = p0f * ipf, s1f = p1f * ipf;
                    const float rho3df = fmaf(s0f, s0f, s1f * s1f);
                    const float d0f = means2D[2 * (size_t)g] - pxf, d1f = means2D[2 * (size_t)g + 1] - pyf;
                    const float rho2df = FILTER_INV_SQ * fmaf(d0f, d0f, d1f * d1f);
                    const int ray = rho3df <= rho2df;
                    const float rhof = fminf(rho3df, rho2df);
                    const float cdf = ray ? fmaf(s0f, Tm[6], fmaf(s1f, Tm[7], Tm[8])) : Tm[8];
                    if (cdf < NEAR_N) continue;
                    const float* no = normal_opacity + 4 * (size_t)g;
                    const float powerf = -0.5f * rhof;
                    if (powerf > 0.0f) continue;
                    const float Gf = expf(powerf);
                    const int clamped = !(no[3] * Gf < 0.99f); /* fminf(0.99f, x): the clamp wins unless x < 0.99 (NaN -> 0.99) */
                    const float alphaf = fminf(0.99f, no[3] * Gf);
                    if (alphaf < 1.0f / 255.0f) continue;
                    Tf = Tf * (1.0f / (1.f - alphaf));
                    const int front = Tf < 0.5f;
                    /* ---- double evaluation on those decisions */
                    const double Tu[3] = {Tm[0], Tm[1], Tm[2]}, Tv[3] = {Tm[3], Tm[4], Tm[5]}, Tw[3] = {Tm[6], Tm[7], Tm[8]};
                    const double k[3] = {pxd * Tw[0] - Tu[0], pxd * Tw[1] - Tu[1], pxd * Tw[2] - Tu[2]};
                    const double l[3] = {pyd * Tw[0] - Tv[0], pyd * Tw[1] - Tv[1], pyd * Tw[2] - Tv[2]};
                    const double p0 = k[1] * l[2] - k[2] * l[1], p1 = k[2] * l[0] - k[0] * l[2], p2 = k[0] * l[1] - k[1] * l[0];
                    const double ip = 1.0 / p2;
                    const double s0 = p0 * ip, s1 = p1 * ip;
                    const double rho3d = s0 * s0 + s1 * s1;
                    const double d0 = (double)means2D[2 * (size_t)g] - pxd, d1 = (double)means2D[2 * (size_t)g + 1] - pyd;
                    const double rho2d = (double)FILTER_INV_SQ * (d0 * d0 + d1 * d1);
                    const double rho = ray ? rho3d : rho2d;
                    double c_d = ray ? s0 * Tw[0] + s1 * Tw[1] + Tw[2] : Tw[2];
                    const double G = exp(-0.5 * rho);
                    const double opac = no[3];
                    const double alpha = clamped ? (double)0.99f : opac * G;
                    const double ioma = 1.0 / (1.0 - alpha);
                    T = T * ioma;
                    const double w = alpha * T;
                    double dL_dalpha = 0.0;
                    for (int ch = 0; ch < 3; ch++) { /* backward.cu:331-344 */
                        const double c = colors[3 * (size_t)g + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.0 - last_alpha) * accum_rec[ch];
                        last_color[ch] = c;
                        dL_dalpha += (c - accum_rec[ch]) * dpx[ch];
#pragma omp atomic
                        acc[(size_t)g * ACC_STRIDE + ch] += w * dpx[ch];
                    }
                    double conf = 1.0;
                    if (use_sa) { /* backward.cu:347-351 */
                        if (front) {
                            const double dm = c_d - mm;
                            conf = exp(-(dm * dm) * sa_k);
                        }
                        c_d = c_d * conf + mm * (1.0 - conf);
                    }
                    double dL_dz = 0.0, dL_dweight;
                    if (contributor == median_contributor - 1u) dL_dz = dL_dmedian_depth;
                    if (use_sa) {
                        const double dm = c_d - mm;
                        dL_dweight = dm * dm * dL_dreg;
                        dL_dalpha += dL_dweight - last_dL_dT;
                        last_dL_dT = dL_dweight * alpha + (1.0 - alpha) * last_dL_dT;
                        dL_dz += conf * 2.0 * w * dm * dL_dreg;
                    } else {
                        const double m_d = c1 * (1.0 - (double)NEAR_N / c_d);
                        const double dmd_dd = c1 * (double)NEAR_N / (c_d * c_d);
                        dL_dweight = (final_D2 + m_d * m_d * final_A - 2.0 * m_d * final_D) * dL_dreg;
                        dL_dalpha += dL_dweight - last_dL_dT;
                        last_dL_dT = dL_dweight * alpha + (1.0 - alpha) * last_dL_dT;
                        dL_dz += 2.0 * w * (m_d * final_A - final_D) * dL_dreg * dmd_dd;
                    }
                    accum_depth_rec = last_alpha * last_depth + (1.0 - last_alpha) * accum_depth_rec;
                    last_depth = c_d;
                    dL_dalpha += (c_d - accum_depth_rec) * dL_ddepth;
                    accum_alpha_rec = last_alpha + (1.0 - last_alpha) * accum_alpha_rec;
                    dL_dalpha += (1.0 - accum_alpha_rec) * dL_daccum;
                    for (int ch = 0; ch < 3; ch++) { /* backward.cu:392-397 */
                        accum_normal_rec[ch] = last_alpha * last_normal[ch] + (1.0 - last_alpha) * accum_normal_rec[ch];
                        last_normal[ch] = no[ch];
                        dL_dalpha += ((double)no[ch] - accum_normal_rec[ch]) * dL_dn[ch];
#pragma omp atomic
                        acc[(size_t)g * ACC_STRIDE + 3 + ch] += w * dL_dn[ch];
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-(double)T_final * ioma) * bg_dot;
                    const double dL_dG = opac * dL_dalpha;
                    dL_dz += conf * w * dL_ddepth;
                    double add[ACC_STRIDE] = {0};
                    if (ray) { /* backward.cu:419-449 */
                        const double dL_ds0 = dL_dG * -G * s0 + dL_dz * Tw[0];
                        const double dL_ds1 = dL_dG * -G * s1 + dL_dz * Tw[1];
                        const double dsx = dL_ds0 * ip, dsy = dL_ds1 * ip;
                        const double dp2 = -(dsx * s0 + dsy * s1);
                        const double dk[3] = {l[1] * dp2 - l[2] * dsy, l[2] * dsx - l[0] * dp2, l[0] * dsy - l[1] * dsx};
                        const double dl[3] = {dsy * k[2] - dp2 * k[1], dp2 * k[0] - dsx * k[2], dsx * k[1] - dsy * k[0]};
                        const double dz_dTw[3] = {dL_dz * s0, dL_dz * s1, dL_dz};
                        for (int i = 0; i < 3; i++) {
                            add[6 + i] = -dk[i];
                            add[9 + i] = -dl[i];
                            add[12 + i] = pxd * dk[i] + pyd * dl[i] + dz_dTw[i];
                        }
                    } else { /* backward.cu:450-457 */
                        const double t = dL_dG * (-G * (double)FILTER_INV_SQ);
                        add[15] = t * d0;
                        add[16] = t * d1;
                        add[14] = dL_dz;
                    }
                    add[17] = G * dL_dalpha;
                    for (int i = 6; i < 18; i++)
                        if (add[i] != 0.0) {
#pragma omp atomic
                            acc[(size_t)g * ACC_STRIDE + i] += add[i];
                        }
                }
            }
    }
    for (size_t g = 0; g < (size_t)P; g++) {
        const double* a = acc + g * ACC_STRIDE;
        for (int i = 0; i < 3; i++) dL_dcolors[3 * g + i] = a[i];
        for (int i = 0; i < 3; i++) dL_dnormal3D[3 * g + i] = a[3 + i];
        for (int i = 0; i < 9; i++) dL_dtransMat[9 * g + i] = a[6 + i];
        dL_dmean2D[3 * g] = a[15]; dL_dmean2D[3 * g + 1] = a[16]; dL_dmean2D[3 * g + 2] = 0.0;
        dL_dopacity[g] = a[17];
    }
    free(acc);
}

/*
 * backward.cu:466-664 in double (colors_precomp path: no SH term).  Inputs: float32 parameters as the float32 paths see them,
 * the double blend-stage sums of orc_blend_bwd_f64.  dL_dtransMats is in-out for the precomputed-transform path (as in the
 * reference); dL_dmean2Ds is read (the blend-stage value) and NOT overwritten by the densification hack (the hack's value is
 * a product of two of this function's inputs; the float32 paths are compared on it separately).
 */
void orc_preprocess_bwd_f64(int P, const float* means3D, const float* transMats, const int32_t* radii,
                            const float* scales, const float* rotations, const float* viewmatrix, const float* projmatrix,
                            int width, int height, float tan_fovx, float tan_fovy,
                            double* dL_dtransMats, const double* dL_dnormal3Ds, const double* dL_dmean2Ds,
                            double* dL_dmean3Ds, double* dL_dscales, double* dL_drots)
{
    /* rasterizer_impl.cu:396-397 + backward.cu:641-642: W,H rebuilt in float32 (an integer decision: kept in float32) */
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);
    const int W = (int)(focal_x * tan_fovx * 2);
    const int H = (int)(focal_y * tan_fovy * 2);
    const float* pmf = projmatrix;
    const float* vmf = viewmatrix;
    double pm[16], vm[16];
    for (int i = 0; i < 16; i++) { pm[i] = pmf[i]; vm[i] = vmf[i]; }
    const double halfW = W * 0.5, halfWm = (W - 1) * 0.5, halfH = H * 0.5, halfHm = (H - 1) * 0.5;
    double Pm[4][3];
    for (int a = 0; a < 4; a++) {
        Pm[a][0] = pm[4 * a] * halfW + pm[4 * a + 3] * halfWm;
        Pm[a][1] = pm[4 * a + 1] * halfH + pm[4 * a + 3] * halfHm;
        Pm[a][2] = pm[4 * a + 3];
    }
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const int precomp = (scales == NULL);
        double T[9], normal[3] = {0, 0, 0}, R[3][3] = {{0}}, sx = 0, sy = 0, w = 0, x = 0, y = 0, z = 0;
        const double p[3] = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
        if (precomp) {
            for (int i = 0; i < 9; i++) T[i] = transMats[9 * idx + i];
        } else {
            sx = scales[2 * idx]; sy = scales[2 * idx + 1]; /* backward.cu:504: scale_modifier ignored */
            const float* q = rotations + 4 * idx;
            const double s = 1.0 / sqrt((double)q[3] * q[3] + (double)q[0] * q[0] + (double)q[1] * q[1] + (double)q[2] * q[2]);
            w = q[0] * s; x = q[1] * s; y = q[2] * s; z = q[3] * s;
            R[0][0] = 1 - 2 * (y * y + z * z); R[1][0] = 2 * (x * y + w * z); R[2][0] = 2 * (x * z - w * y);
            R[0][1] = 2 * (x * y - w * z); R[1][1] = 1 - 2 * (x * x + z * z); R[2][1] = 2 * (y * z + w * x);
            R[0][2] = 2 * (x * z + w * y); R[1][2] = 2 * (y * z - w * x); R[2][2] = 1 - 2 * (x * x + y * y);
            for (int i = 0; i < 3; i++) { /* forward.cu:75-115 */
                double hx, hy, hz;
                if (i == 0) { hx = R[0][0] * sx; hy = R[1][0] * sx; hz = R[2][0] * sx; }
                else if (i == 1) { hx = R[0][1] * sy; hy = R[1][1] * sy; hz = R[2][1] * sy; }
                else { hx = p[0]; hy = p[1]; hz = p[2]; }
                double q0 = pm[0] * hx + pm[4] * hy + pm[8] * hz;
                double q1 = pm[1] * hx + pm[5] * hy + pm[9] * hz;
                double q3 = pm[3] * hx + pm[7] * hy + pm[11] * hz;
                if (i == 2) { q0 += pm[12]; q1 += pm[13]; q3 += pm[15]; }
                T[0 + i] = q0 * halfW + q3 * halfWm;
                T[3 + i] = q1 * halfH + q3 * halfHm;
                T[6 + i] = q3;
            }
            normal[0] = vm[0] * R[0][2] + vm[4] * R[1][2] + vm[8] * R[2][2];
            normal[1] = vm[1] * R[0][2] + vm[5] * R[1][2] + vm[9] * R[2][2];
            normal[2] = vm[2] * R[0][2] + vm[6] * R[1][2] + vm[10] * R[2][2];
        }
        double dT[9];
        memcpy(dT, dL_dtransMats + 9 * idx, sizeof(dT));
        const double dmx = dL_dmean2Ds[3 * idx], dmy = dL_dmean2Ds[3 * idx + 1];
        int early = 0;
        if (dmx != 0 || dmy != 0) { /* backward.cu:538-577 */
            const double distance = T[6] * T[6] + T[7] * T[7] - T[8] * T[8];
            const double f = 1 / distance;
            dT[0] += dmx * (f * T[6]); dT[1] += dmx * (f * T[7]); dT[2] += dmx * (-f * T[8]);
            dT[3] += dmy * (f * T[6]); dT[4] += dmy * (f * T[7]); dT[5] += dmy * (-f * T[8]);
            dT[6] += dmx * (T[0] * (f - 2 * f * f * T[6] * T[6])) + dmy * (T[3] * (f - 2 * f * f * T[6] * T[6]));
            dT[7] += dmx * (T[1] * (f - 2 * f * f * T[7] * T[7])) + dmy * (T[4] * (f - 2 * f * f * T[7] * T[7]));
            dT[8] += dmx * (-T[2] * (f + 2 * f * f * T[8] * T[8])) + dmy * (-T[5] * (f + 2 * f * f * T[8] * T[8]));
            if (precomp) { memcpy(dL_dtransMats + 9 * idx, dT, sizeof(dT)); early = 1; }
        }
        if (!precomp && !early) {
            double dh[3][3];
            for (int i = 0; i < 3; i++)
                for (int a = 0; a < 3; a++) dh[i][a] = Pm[a][0] * dT[i] + Pm[a][1] * dT[3 + i] + Pm[a][2] * dT[6 + i];
            const double* dn = dL_dnormal3Ds + 3 * idx;
            double dtn[3] = {vm[0] * dn[0] + vm[1] * dn[1] + vm[2] * dn[2], vm[4] * dn[0] + vm[5] * dn[1] + vm[6] * dn[2],
                             vm[8] * dn[0] + vm[9] * dn[1] + vm[10] * dn[2]};
            const double pvx = vm[0] * p[0] + vm[4] * p[1] + vm[8] * p[2] + vm[12];
            const double pvy = vm[1] * p[0] + vm[5] * p[1] + vm[9] * p[2] + vm[13];
            const double pvz = vm[2] * p[0] + vm[6] * p[1] + vm[10] * p[2] + vm[14];
            const double cosv = -(pvx * normal[0] + pvy * normal[1] + pvz * normal[2]);
            const double mult = cosv > 0 ? 1.0 : -1.0;
            for (int a = 0; a < 3; a++) dtn[a] *= mult;
            double v[3][3];
            for (int r = 0; r < 3; r++) { v[r][0] = dh[0][r] * sx; v[r][1] = dh[1][r] * sy; v[r][2] = dtn[r]; }
            dL_drots[4 * idx + 0] = 2 * (x * (v[2][1] - v[1][2]) + y * (v[0][2] - v[2][0]) + z * (v[1][0] - v[0][1]));
            dL_drots[4 * idx + 1] = 2 * (-2 * x * (v[1][1] + v[2][2]) + y * (v[1][0] + v[0][1]) + z * (v[2][0] + v[0][2]) + w * (v[2][1] - v[1][2]));
            dL_drots[4 * idx + 2] = 2 * (x * (v[1][0] + v[0][1]) - 2 * y * (v[0][0] + v[2][2]) + z * (v[2][1] + v[1][2]) + w * (v[0][2] - v[2][0]));
            dL_drots[4 * idx + 3] = 2 * (x * (v[2][0] + v[0][2]) + y * (v[2][1] + v[1][2]) - 2 * z * (v[0][0] + v[1][1]) + w * (v[1][0] - v[0][1]));
            dL_dscales[2 * idx + 0] = dh[0][0] * R[0][0] + dh[0][1] * R[1][0] + dh[0][2] * R[2][0];
            dL_dscales[2 * idx + 1] = dh[1][0] * R[0][1] + dh[1][1] * R[1][1] + dh[1][2] * R[2][1];
            dL_dmean3Ds[3 * idx + 0] = dh[2][0];
            dL_dmean3Ds[3 * idx + 1] = dh[2][1];
            dL_dmean3Ds[3 * idx + 2] = dh[2][2];
        }
    }
}
