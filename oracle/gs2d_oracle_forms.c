/*
 * gs2d_oracle_forms.c -- CPU ORACLE, test infrastructure only (NOT product code).
 *
 * The float32 backward blend (RAST/cuda_rasterizer/backward.cu:281-461, as restated by orc_blend_bwd in gs2d_oracle.c) with
 * the HIP kernel's algebraic rewrites switchable ONE BY ONE, so that a CPU test can price each of them against the float64
 * evaluation (gs2d_oracle_f64.c) -- "which rewrite costs how much rounding" without a GPU in the loop.  `forms` bits:
 *    1  FORM_EXP2     exponentials as exp2(x * c) with the 2^k-folded float32 constant, the way v_exp_f32 is fed
 *                     (gs2d_blend.hip: fast_exp_neg_half, fast_exp): the argument is rounded once more, |arg| * 2^-24
 *    2  FORM_RCP      reciprocals perturbed by up to +-1 ulp (deterministic hash of the operand): v_rcp_f32's error bound
 *    4  FORM_MERGED   ONE blend recurrence S for the sum over all blended channels + the regulariser instead of one
 *                     accum_x per channel (gs2d_blend.hip, blend_S)
 *    8  FORM_CLOSED   the opacity-map term through 1 - accum_alpha_rec = T_final / (T (1 - alpha)), folded with the
 *                     background term (tf_bg)
 *   16  FORM_CONF     re-weighted depth as mm + conf (c_d - mm), its distance from the median as conf (c_d - mm)
 *   32  FORM_EXPAND   -dk, -dl formed directly, Tw components as fma(-px, nk, fma(-py, nl, ...)) (exact sign moves)
 * forms = 0 reproduces orc_blend_bwd bit for bit; forms = 63 is the kernel's operation order (up to the hardware's actual
 * rcp / exp results).  Per-Gaussian sums in double, as orc_blend_bwd.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16
#define NEAR_N 0.2f
#define FAR_N 100.0f
#define FILTER_INV_SQ 100.0f
#define ACC_STRIDE 20

enum { FORM_EXP2 = 1, FORM_RCP = 2, FORM_MERGED = 4, FORM_CLOSED = 8, FORM_CONF = 16, FORM_EXPAND = 32 };

static inline float rcp_form(float x, int forms)
{
    float r = 1.0f / x;
    if (forms & FORM_RCP) {
        uint32_t b;
        memcpy(&b, &x, 4);
        b = (b * 2654435761u) >> 30; /* 0..3: -1 ulp, 0, 0, +1 ulp */
        if (b == 0) r = nextafterf(r, -INFINITY);
        else if (b == 3) r = nextafterf(r, INFINITY);
    }
    return r;
}
static inline float exp_neg_half(float rho, int forms)
{
    return (forms & FORM_EXP2) ? exp2f(rho * -0.72134752044448170368f) : expf(-0.5f * rho);
}
static inline float exp_form(float x, int forms)
{
    return (forms & FORM_EXP2) ? exp2f(x * 1.44269504088896340736f) : expf(x);
}

void orc_blend_bwd_forms(int forms, int P, int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                         const float* bg, const float* means2D, const float* normal_opacity,
                         const float* transMats, const float* colors, const float* final_Ts,
                         const uint32_t* n_contrib, const float* dL_dpixels, const float* dL_depths,
                         const float* median_depth, const float* depth_std, int use_sa,
                         float* dL_dtransMat, float* dL_dmean2D, float* dL_dnormal3D, float* dL_dopacity, float* dL_dcolors)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const size_t HW = (size_t)H * W;
    double* acc = (double*)calloc((size_t)P * ACC_STRIDE + 1, sizeof(double));
#define ACC(G_, K_, V_) do { const double v_ = (double)(V_); _Pragma("omp atomic") acc[(size_t)(G_) * ACC_STRIDE + (K_)] += v_; } while (0)
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
                const float T_final = final_Ts[pix];
                float T = T_final;
                uint32_t contributor = r1 - r0;
                const uint32_t last_contributor = n_contrib[pix];
                const uint32_t median_contributor = n_contrib[pix + HW];
                float accum_rec[3] = {0, 0, 0}, dL_dpixel[3];
                const float dL_ddepth = dL_depths[0 * HW + pix], dL_daccum = dL_depths[1 * HW + pix], dL_dreg = dL_depths[6 * HW + pix];
                const float dn[3] = {dL_depths[2 * HW + pix], dL_depths[3 * HW + pix], dL_depths[4 * HW + pix]};
                const float dL_dmedian_depth = dL_depths[5 * HW + pix];
                const float mm = median_depth[pix], mstd = depth_std[pix];
                float last_depth = 0, last_normal[3] = {0, 0, 0}, accum_depth_rec = 0, accum_alpha_rec = 0, accum_normal_rec[3] = {0, 0, 0};
                const float final_D = final_Ts[pix + HW], final_D2 = final_Ts[pix + 2 * HW], final_A = 1 - T_final;
                float last_dL_dT = 0, last_alpha = 0, last_color[3] = {0, 0, 0};
                for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[i * HW + pix];
                const float bg_dot = fmaf(bg[2], dL_dpixel[2], fmaf(bg[1], dL_dpixel[1], bg[0] * dL_dpixel[0]));
                const float tf_bg = T_final * (bg_dot - dL_daccum); /* FORM_CLOSED */
                const float sa_k = 1.0f / (4 * fmaxf(mstd * (1.0f / (1 - T_final)), 1e-7f));
                const float c1 = FAR_N / (FAR_N - NEAR_N);
                float blend_S = 0.f; /* FORM_MERGED */
                for (uint32_t it = r1; it-- > r0;) {
                    contributor--;
                    if (contributor >= last_contributor) continue;
                    const uint32_t g = point_list[it];
                    const float* Tm = transMats + 9 * (size_t)g;
                    const float Tw[3] = {Tm[6], Tm[7], Tm[8]};
                    const float k[3] = {fmaf(pxf, Tw[0], -Tm[0]), fmaf(pxf, Tw[1], -Tm[1]), fmaf(pxf, Tw[2], -Tm[2])};
                    const float l[3] = {fmaf(pyf, Tw[0], -Tm[3]), fmaf(pyf, Tw[1], -Tm[4]), fmaf(pyf, Tw[2], -Tm[5])};
                    const float p0 = fmaf(k[1], l[2], -(k[2] * l[1]));
                    const float p1 = fmaf(k[2], l[0], -(k[0] * l[2]));
                    const float p2 = fmaf(k[0], l[1], -(k[1] * l[0]));
                    if (p2 == 0.0f) continue;
                    const float ip = rcp_form(p2, forms);
                    const float s0 = p0 * ip, s1 = p1 * ip;
                    const float rho3d = fmaf(s0, s0, s1 * s1);
                    const float d0 = means2D[2 * (size_t)g] - pxf, d1 = means2D[2 * (size_t)g + 1] - pyf;
                    const float rho2d = FILTER_INV_SQ * fmaf(d0, d0, d1 * d1);
                    const float rho = fminf(rho3d, rho2d);
                    const int ray = rho3d <= rho2d;
                    float c_d = ray ? fmaf(s0, Tw[0], fmaf(s1, Tw[1], Tw[2])) : Tw[2];
                    if (c_d < NEAR_N) continue;
                    const float* no = normal_opacity + 4 * (size_t)g;
                    if (-0.5f * rho > 0.0f) continue;
                    const float G = exp_neg_half(rho, forms);
                    const float alpha = fminf(0.99f, no[3] * G);
                    if (alpha < 1.0f / 255.0f) continue;
                    const float ioma = rcp_form(1.f - alpha, forms);
                    T = T * ioma;
                    const float w = alpha * T;
                    float dL_dalpha = 0.0f;
                    float D_ = 0.f;
                    if (forms & FORM_MERGED) D_ = fmaf(colors[3 * (size_t)g + 2], dL_dpixel[2], fmaf(colors[3 * (size_t)g + 1], dL_dpixel[1], colors[3 * (size_t)g] * dL_dpixel[0]));
                    for (int ch = 0; ch < 3; ch++) {
                        const float c = colors[3 * (size_t)g + ch];
                        if (!(forms & FORM_MERGED)) {
                            accum_rec[ch] = fmaf(last_alpha, last_color[ch], (1.f - last_alpha) * accum_rec[ch]);
                            last_color[ch] = c;
                            dL_dalpha = fmaf(c - accum_rec[ch], dL_dpixel[ch], dL_dalpha);
                        }
                        ACC(g, ch, w * dL_dpixel[ch]);
                    }
                    float conf = 1, dmc = 0.f; /* dmc: distance of the re-weighted depth from the median */
                    if (use_sa) {
                        const float dm0 = c_d - mm;
                        if (T < 0.5f) conf = exp_form(-(dm0 * dm0) * sa_k, forms);
                        if (forms & FORM_CONF) { c_d = fmaf(conf, dm0, mm); dmc = conf * dm0; }
                        else { c_d = fmaf(c_d, conf, mm * (1 - conf)); dmc = c_d - mm; }
                    }
                    float dL_dz = 0.0f, dL_dweight;
                    if (contributor == median_contributor - 1u) dL_dz = dL_dmedian_depth;
                    if (use_sa) {
                        dL_dweight = (dmc * dmc) * dL_dreg;
                        if (forms & FORM_CONF) { const float cw2 = conf * w; dL_dz = fmaf((cw2 + cw2) * dmc, dL_dreg, dL_dz); }
                        else dL_dz = fmaf(conf * 2.0f * w * dmc, dL_dreg, dL_dz);
                    } else {
                        const float icd = rcp_form(c_d, forms);
                        const float m_d = c1 * (1 - NEAR_N * icd);
                        const float dmd_dd = (c1 * NEAR_N) * (icd * icd);
                        dL_dweight = fmaf(m_d * m_d, final_A, fmaf(-2.0f * m_d, final_D, final_D2)) * dL_dreg;
                        const float dL_dmd = 2.0f * w * fmaf(m_d, final_A, -final_D) * dL_dreg;
                        dL_dz = fmaf(dL_dmd, dmd_dd, dL_dz);
                    }
                    if (forms & FORM_MERGED) {
                        D_ = fmaf(c_d, dL_ddepth, D_) + dL_dweight;
                        D_ = fmaf(no[2], dn[2], fmaf(no[1], dn[1], fmaf(no[0], dn[0], D_)));
                        for (int ch = 0; ch < 3; ch++) ACC(g, 3 + ch, w * dn[ch]);
                    } else {
                        dL_dalpha += dL_dweight - last_dL_dT;
                        last_dL_dT = fmaf(dL_dweight, alpha, (1 - alpha) * last_dL_dT);
                        accum_depth_rec = fmaf(last_alpha, last_depth, (1.f - last_alpha) * accum_depth_rec);
                        last_depth = c_d;
                        dL_dalpha = fmaf(c_d - accum_depth_rec, dL_ddepth, dL_dalpha);
                    }
                    if (!(forms & FORM_CLOSED)) {
                        accum_alpha_rec = fmaf(1.f - last_alpha, accum_alpha_rec, last_alpha);
                        if (!(forms & FORM_MERGED)) dL_dalpha = fmaf(1 - accum_alpha_rec, dL_daccum, dL_dalpha);
                    }
                    if (!(forms & FORM_MERGED)) {
                        for (int ch = 0; ch < 3; ch++) {
                            accum_normal_rec[ch] = fmaf(last_alpha, last_normal[ch], (1.f - last_alpha) * accum_normal_rec[ch]);
                            last_normal[ch] = no[ch];
                            dL_dalpha = fmaf(no[ch] - accum_normal_rec[ch], dn[ch], dL_dalpha);
                            ACC(g, 3 + ch, w * dn[ch]);
                        }
                    }
                    if (forms & FORM_MERGED) {
                        const float DmS = D_ - blend_S;
                        blend_S = fmaf(alpha, DmS, blend_S);
                        if (forms & FORM_CLOSED) dL_dalpha = fmaf(-ioma, tf_bg, DmS * T);
                        else { dL_dalpha = fmaf(1 - accum_alpha_rec, dL_daccum, DmS) * T; dL_dalpha = fmaf(-T_final * ioma, bg_dot, dL_dalpha); }
                    } else {
                        if (forms & FORM_CLOSED) dL_dalpha = fmaf(-ioma, tf_bg, dL_dalpha * T);
                        else { dL_dalpha *= T; dL_dalpha = fmaf(-T_final * ioma, bg_dot, dL_dalpha); }
                    }
                    last_alpha = alpha;
                    const float dL_dG = no[3] * dL_dalpha;
                    dL_dz = fmaf(conf * w, dL_ddepth, dL_dz);
                    if (ray) {
                        const float gG = dL_dG * -G;
                        const float dL_ds0 = fmaf(gG, s0, dL_dz * Tw[0]);
                        const float dL_ds1 = fmaf(gG, s1, dL_dz * Tw[1]);
                        const float dsx = dL_ds0 * ip, dsy = dL_ds1 * ip;
                        const float dp2 = -fmaf(dsx, s0, dsy * s1);
                        if (forms & FORM_EXPAND) {
                            const float nk[3] = {fmaf(l[2], dsy, -(l[1] * dp2)), fmaf(l[0], dp2, -(l[2] * dsx)), fmaf(l[1], dsx, -(l[0] * dsy))};
                            const float nl[3] = {fmaf(dp2, k[1], -(dsy * k[2])), fmaf(dsx, k[2], -(dp2 * k[0])), fmaf(dsy, k[0], -(dsx * k[1]))};
                            const float zz[3] = {dL_dz * s0, dL_dz * s1, dL_dz};
                            for (int i = 0; i < 3; i++) {
                                ACC(g, 6 + i, nk[i]);
                                ACC(g, 9 + i, nl[i]);
                                ACC(g, 12 + i, fmaf(-pxf, nk[i], fmaf(-pyf, nl[i], zz[i])));
                            }
                        } else {
                            const float dk[3] = {fmaf(l[1], dp2, -(l[2] * dsy)), fmaf(l[2], dsx, -(l[0] * dp2)), fmaf(l[0], dsy, -(l[1] * dsx))};
                            const float dl[3] = {fmaf(dsy, k[2], -(dp2 * k[1])), fmaf(dp2, k[0], -(dsx * k[2])), fmaf(dsx, k[1], -(dsy * k[0]))};
                            const float zz[3] = {dL_dz * s0, dL_dz * s1, dL_dz};
                            for (int i = 0; i < 3; i++) {
                                ACC(g, 6 + i, -dk[i]);
                                ACC(g, 9 + i, -dl[i]);
                                ACC(g, 12 + i, fmaf(pxf, dk[i], fmaf(pyf, dl[i], zz[i])));
                            }
                        }
                    } else {
                        const float t = dL_dG * (-G * FILTER_INV_SQ);
                        ACC(g, 15, t * d0);
                        ACC(g, 16, t * d1);
                        ACC(g, 14, dL_dz);
                    }
                    ACC(g, 17, G * dL_dalpha);
                }
            }
    }
#undef ACC
    for (size_t g = 0; g < (size_t)P; g++) {
        const double* a = acc + g * ACC_STRIDE;
        for (int i = 0; i < 3; i++) dL_dcolors[3 * g + i] = (float)a[i];
        for (int i = 0; i < 3; i++) dL_dnormal3D[3 * g + i] = (float)a[3 + i];
        for (int i = 0; i < 9; i++) dL_dtransMat[9 * g + i] = (float)a[6 + i];
        dL_dmean2D[3 * g] = (float)a[15]; dL_dmean2D[3 * g + 1] = (float)a[16]; dL_dmean2D[3 * g + 2] = 0.f;
        dL_dopacity[g] = (float)a[17];
    }
    free(acc);
}
