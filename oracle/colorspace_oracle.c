/* colorspace_oracle.c — CPU restatement of the reference's colorspace filter
 * (libhb/colorspace.c:20-207): zscale -> [format=gbrpf32le -> tonemap] -> zscale -> format,
 * i.e. matrix / range / transfer / primaries conversion with optional HDR tone mapping.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 *                    ***  PARITY UNPINNED  ***
 * The arithmetic lives in FFmpeg 9.0.1 (vf_zscale.c, vf_tonemap.c) and zimg
 * snapshot-20250624 (colorspace/, resize/, depth/), which are not part of
 * /root/reference, and the reference holds no test or golden vector for this filter.
 * What is restated here, from their published algorithms:
 *   - integer -> float: (Y - 16<<s) / (219<<s), (C - 1<<(d-1)) / (224<<s) for limited
 *     range; Y / max, (C - 1<<(d-1)) / max for full range;
 *   - chroma to 4:4:4 and back with the triangle ("bilinear", vf_zscale's default filter)
 *     kernel, left-sited horizontally and centred vertically (MPEG-2 4:2:0 siting):
 *     up   x even: c[x/2]; x odd: (c[k] + c[k+1]) / 2;  y = 2k: 1/4 c[k-1] + 3/4 c[k],
 *          y = 2k+1: 3/4 c[k] + 1/4 c[k+1];   edges repeat;
 *     down columns 2k-1, 2k, 2k+1 with 1/4 1/2 1/4; rows 2k-1 .. 2k+2 with 1/8 3/8 3/8 1/8;
 *   - Y'CbCr <-> R'G'B' from (Kr, Kb) of the matrix (YCgCo: its own fixed matrix); when neither transfer class nor
 *     primaries change, one combined 3x3 matrix and no linearisation (as zimg's
 *     operation graph does);
 *   - transfer functions as zimg's display-referred set: BT.709/601/2020 = pure 2.4 gamma
 *     (BT.1886), gamma22/28, sRGB, SMPTE 240M, linear, ST 2084 (x 10000/npl) and
 *     ARIB STD-B67 with the 1.2 OOTF (x 1000/npl), both as inputs and as outputs;
 *   - primaries conversion through XYZ with Bradford adaptation between white points;
 *   - vf_tonemap.c's operators (none, linear, gamma, clip, reinhard, hable, mobius) on the
 *     brightest component, with its parameter defaults; its desaturation step is skipped
 *     because the frame is GBR at that point (no luma coefficients => FFmpeg disables it);
 *   - float -> integer: round to nearest even (lrintf), clipped to [0, max].
 * Transfer functions are evaluated per sample, as zimg's scalar path does (gamma.cpp), on unclipped arguments:
 * values below black and above white go through (the pure power laws return 0 below 0 like zimg's rec_1886
 * pair, the piecewise ones continue their linear segment), and only the final integer conversion clips.
 * The powers / exp / log are the deterministic float routines below (range reduction + fixed polynomials in
 * IEEE single arithmetic with fused multiply-adds, no libm's pow / exp / log, about 1e-7 relative for log, 6e-6 for
 * the steepest power) so that the HIP kernel can reproduce them bit for bit - every sum of products in this file is
 * written as an explicit chain of fmaf() for the same reason (round 5; before, every product was rounded on its own);
 * tests/test_colorspace_cpu.py holds them against libm.  What still separates this file from real zimg is that
 * rounding - not modelling: no tables, no clipping of super-whites.
 * The HIP path is tested bit-for-bit against THIS file, never against FFmpeg/zimg.
 */
#include "oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

typedef struct
{
    int   need_linear, gamut, tonemap;
    float yoff_in, ymul_in, coff_in, cmul_in;
    float ymul_out, yoff_out, cmul_out, coff_out;
    float m_in[3][3], m_out[3][3], m_gamut[3][3], m_direct[3][3];
    float tm_param, tm_peak, tm_a, tm_b, tm_c;       /* operator constants, see tonemap_sig() */
    int   tc_in, tc_out;                            /* transfer classes */
    float lin_scale, gam_scale;                     /* PQ / HLG: display-light scale after the EOTF / before its inverse */
    int   vmax;
} plan_t;

/* ---- small double 3x3 algebra ------------------------------------------------------- */
static void mul3(double r[3][3], const double a[3][3], const double b[3][3])
{
    double t[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            t[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j];
    memcpy(r, t, sizeof(t));
}

static void inv3(double r[3][3], const double m[3][3])
{
    const double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1];
    const double c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2];
    const double c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
    const double det = m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02;
    double t[3][3];
    t[0][0] = c00 / det;
    t[1][0] = c01 / det;
    t[2][0] = c02 / det;
    t[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) / det;
    t[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) / det;
    t[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) / det;
    t[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) / det;
    t[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) / det;
    t[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) / det;
    memcpy(r, t, sizeof(t));
}

/* ---- colour tables (AVCOL_* numbering = HB_COLR_*, libhb/handbrake/common.h) ------------ */
static int matrix_coeffs(int id, double *kr, double *kb)
{
    switch (id)
    {
        case 1:  *kr = 0.2126; *kb = 0.0722; return 1;      /* bt709      */
        case 4:  *kr = 0.30;   *kb = 0.11;   return 1;      /* fcc        */
        case 5: case 6: *kr = 0.299; *kb = 0.114; return 1; /* bt470bg, smpte170m */
        case 7:  *kr = 0.212;  *kb = 0.087;  return 1;      /* smpte240m  */
        case 8:  *kr = 0.25;   *kb = 0.25;   return 1;      /* YCgCo: placeholders, build_plan installs its fixed matrix */
        case 9:  *kr = 0.2627; *kb = 0.0593; return 1;      /* bt2020nc   */
    }
    return 0;
}

static int primaries_xy(int id, double xy[8])
{
    static const double D65[2] = { 0.3127, 0.3290 }, C[2] = { 0.310, 0.316 }, DCI[2] = { 0.314, 0.351 };
    const double *w = D65;
    switch (id)
    {
        case 1:  xy[0] = 0.640; xy[1] = 0.330; xy[2] = 0.300; xy[3] = 0.600; xy[4] = 0.150; xy[5] = 0.060; break;
        case 4:  xy[0] = 0.670; xy[1] = 0.330; xy[2] = 0.210; xy[3] = 0.710; xy[4] = 0.140; xy[5] = 0.080; w = C; break;
        case 5:  xy[0] = 0.640; xy[1] = 0.330; xy[2] = 0.290; xy[3] = 0.600; xy[4] = 0.150; xy[5] = 0.060; break;
        case 6: case 7:
                 xy[0] = 0.630; xy[1] = 0.340; xy[2] = 0.310; xy[3] = 0.595; xy[4] = 0.155; xy[5] = 0.070; break;
        case 8:  xy[0] = 0.681; xy[1] = 0.319; xy[2] = 0.243; xy[3] = 0.692; xy[4] = 0.145; xy[5] = 0.049; w = C; break;
        case 9:  xy[0] = 0.708; xy[1] = 0.292; xy[2] = 0.170; xy[3] = 0.797; xy[4] = 0.131; xy[5] = 0.046; break;
        case 11: xy[0] = 0.680; xy[1] = 0.320; xy[2] = 0.265; xy[3] = 0.690; xy[4] = 0.150; xy[5] = 0.060; w = DCI; break;
        case 12: xy[0] = 0.680; xy[1] = 0.320; xy[2] = 0.265; xy[3] = 0.690; xy[4] = 0.150; xy[5] = 0.060; break;
        case 22: xy[0] = 0.630; xy[1] = 0.340; xy[2] = 0.295; xy[3] = 0.605; xy[4] = 0.155; xy[5] = 0.077; break;
        default: return 0;
    }
    xy[6] = w[0]; xy[7] = w[1];
    return 1;
}

static int primaries_class(int id) { return id == 7 ? 6 : id; }
static int transfer_class(int id) { return (id == 6 || id == 14 || id == 15) ? 1 : id; }

static void rgb_to_xyz(double m[3][3], const double xy[8])
{
    double p[3][3], pi[3][3];
    for (int i = 0; i < 3; i++)
    {
        p[0][i] = xy[2 * i] / xy[2 * i + 1];
        p[1][i] = 1.0;
        p[2][i] = (1.0 - xy[2 * i] - xy[2 * i + 1]) / xy[2 * i + 1];
    }
    const double w[3] = { xy[6] / xy[7], 1.0, (1.0 - xy[6] - xy[7]) / xy[7] };
    inv3(pi, p);
    for (int i = 0; i < 3; i++)
    {
        const double s = pi[i][0] * w[0] + pi[i][1] * w[1] + pi[i][2] * w[2];
        for (int r = 0; r < 3; r++)
            m[r][i] = p[r][i] * s;
    }
}

static void gamut_matrix(double g[3][3], const double in_xy[8], const double out_xy[8])
{
    double a[3][3], b[3][3], bi[3][3];
    rgb_to_xyz(a, in_xy);
    rgb_to_xyz(b, out_xy);
    inv3(bi, b);
    if (in_xy[6] != out_xy[6] || in_xy[7] != out_xy[7])
    {
        static const double br[3][3] = { { 0.8951, 0.2664, -0.1614 }, { -0.7502, 1.7135, 0.0367 }, { 0.0389, -0.0685, 1.0296 } };
        double bri[3][3], d[3][3] = { { 0 } }, t[3][3];
        const double wi[3] = { in_xy[6] / in_xy[7], 1.0, (1.0 - in_xy[6] - in_xy[7]) / in_xy[7] };
        const double wo[3] = { out_xy[6] / out_xy[7], 1.0, (1.0 - out_xy[6] - out_xy[7]) / out_xy[7] };
        inv3(bri, br);
        for (int i = 0; i < 3; i++)
        {
            const double ci = br[i][0] * wi[0] + br[i][1] * wi[1] + br[i][2] * wi[2];
            const double co = br[i][0] * wo[0] + br[i][1] * wo[1] + br[i][2] * wo[2];
            d[i][i] = co / ci;
        }
        mul3(t, d, br);
        mul3(t, bri, t);
        mul3(a, t, a);
    }
    mul3(g, bi, a);
}

/* ---- deterministic float math (IEEE single +, -, *, / and fused multiply-add only - each correctly rounded, so the
 * HIP kernel carries the same sequence and gets the same bits) ---- */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* log2 of a positive normal float: x = m 2^e with m in [sqrt(1/2), sqrt(2)); ln m = f - f^2 / 2 + f^3 P(f), f = m - 1,
 * with the degree-8 polynomial of the Cephes single-precision logarithm (Moshier), evaluated by Horner's rule in fused
 * multiply-adds; no division */
static inline float det_log2f(float x)
{
    const uint32_t bits = f2u(x);
    int e = (int)(bits >> 23) - 127;
    float m = u2f((bits & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f;
    const float z = f * f;
    float p = 7.0376836292e-2f;
    p = fmaf(p, f, -1.1514610310e-1f);
    p = fmaf(p, f, 1.1676998740e-1f);
    p = fmaf(p, f, -1.2420140846e-1f);
    p = fmaf(p, f, 1.4249322787e-1f);
    p = fmaf(p, f, -1.6668057665e-1f);
    p = fmaf(p, f, 2.0000714765e-1f);
    p = fmaf(p, f, -2.4999993993e-1f);
    p = fmaf(p, f, 3.3333331174e-1f);
    float y = (f * z) * p;
    y = fmaf(-0.5f, z, y);
    return fmaf(f + y, 1.44269502f, (float)e);
}

/* 2^y: y = i + f with |f| <= 1/2, e^(f ln 2) by its Taylor polynomial of degree 7 (Horner, fused), scaled by the
 * exponent bits */
static inline float det_exp2f(float y)
{
    if (!(y >= -126.0f)) return 0.0f;                  /* also what a NaN becomes */
    if (y > 127.0f) y = 127.0f;
    const int i = (int)(y + (y < 0.0f ? -0.5f : 0.5f));
    const float z = (y - (float)i) * 0.693147182f;
    float p = 0.000198412701f;
    p = fmaf(p, z, 0.00138888892f);
    p = fmaf(p, z, 0.00833333377f);
    p = fmaf(p, z, 0.0416666679f);
    p = fmaf(p, z, 0.166666672f);
    p = fmaf(p, z, 0.5f);
    p = fmaf(p, z, 1.0f);
    p = fmaf(p, z, 1.0f);
    return p * u2f((uint32_t)(i + 127) << 23);
}

static inline float det_powf(float x, float y) { return x <= 0.0f ? 0.0f : det_exp2f(y * det_log2f(x)); }
static inline float det_expf(float x) { return det_exp2f(x * 1.44269502f); }
static inline float det_logf(float x) { return det_log2f(x) * 0.693147182f; }          /* x > 0 */

/* test entry points (tests/test_colorspace_cpu.py holds them against libm) */
float orc_det_powf(float x, float y) { return det_powf(x, y); }
float orc_det_expf(float x) { return det_expf(x); }
float orc_det_logf(float x) { return det_logf(x); }

/* ---- transfer functions, per sample, display referred as zimg's set (gamma.cpp) ------------ */
static int transfer_known(int cls, int as_output)
{
    (void)as_output;
    return cls == 1 || cls == 4 || cls == 5 || cls == 7 || (cls >= 8 && cls <= 11) || cls == 13 || cls == 16 || cls == 17 || cls == 18;
}

/* 9 / 10: zimg's log100 / log316 pair (gamma.cpp: 1 + log10(x) / 2 above 0.01, resp. 1 + log10(x) / 2.5 above
 * sqrt(10) / 1000, 0 below; inverse 10^(2 (v - 1)) resp. 10^(2.5 (v - 1)), the threshold itself at or below 0).
 * 11: IEC 61966-2-4 (xvYCC), display referred like class 1: the 2.4 power law carried to negative values by its sign.
 * 17: SMPTE ST 428-1 (gamma.cpp: st_428_eotf / inverse): L = v^2.6 * 52.37 / 48, v = (48 L / 52.37)^(1 / 2.6), 0 at or below 0. */
static inline float to_linear(int cls, float v)
{
    switch (cls)
    {
        case 9:  return v <= 0.0f ? 0.01f : det_exp2f((2.0f * (v - 1.0f)) * 3.32192802f);
        case 10: return v <= 0.0f ? 0.00316227766f : det_exp2f((2.5f * (v - 1.0f)) * 3.32192802f);
        case 11: return v < 0.0f ? -det_powf(-v, 2.4f) : det_powf(v, 2.4f);
        case 1:  return det_powf(v, 2.4f);
        case 4:  return det_powf(v, 2.2f);
        case 5:  return det_powf(v, 2.8f);
        case 7:  return v < 0.0913f ? v / 4.0f : det_powf((v + 0.1115f) / 1.1115f, 1.0f / 0.45f);
        case 13: return v <= 0.04045f ? v / 12.92f : det_powf((v + 0.055f) / 1.055f, 2.4f);
        case 17: return v <= 0.0f ? 0.0f : (det_powf(v, 2.6f) * 52.37f) / 48.0f;
        case 16:                                                     /* ST 2084 EOTF, 1.0 = 10000 cd/m2 */
        {
            if (v <= 0.0f) return 0.0f;
            const float p = det_powf(v, 1.0f / 78.84375f);
            float num = p - 0.8359375f;
            if (num < 0.0f) num = 0.0f;
            float den = 18.8515625f - 18.6875f * p;
            if (den < 1e-6f) den = 1e-6f;
            return det_powf(num / den, 1.0f / 0.1593017578125f);
        }
        case 18:                                                     /* ARIB STD-B67 inverse OETF + 1.2 OOTF, 1.0 = 1000 cd/m2 */
        {
            const float x = v < 0.0f ? 0.0f : v;
            const float s = x <= 0.5f ? x * x / 3.0f : (det_expf((x - 0.55991073f) / 0.17883277f) + 0.28466892f) / 12.0f;
            return det_powf(s, 1.2f);
        }
    }
    return v;                                                        /* 8: linear */
}

static inline float to_gamma(int cls, float x)
{
    switch (cls)
    {
        case 9:  return x <= 0.01f ? 0.0f : 1.0f + (det_log2f(x) * 0.301029996f) / 2.0f;
        case 10: return x <= 0.00316227766f ? 0.0f : 1.0f + (det_log2f(x) * 0.301029996f) / 2.5f;
        case 11: return x < 0.0f ? -det_powf(-x, 1.0f / 2.4f) : det_powf(x, 1.0f / 2.4f);
        case 1:  return det_powf(x, 1.0f / 2.4f);
        case 4:  return det_powf(x, 1.0f / 2.2f);
        case 5:  return det_powf(x, 1.0f / 2.8f);
        case 7:  return x < 0.0228f ? 4.0f * x : 1.1115f * det_powf(x, 0.45f) - 0.1115f;
        case 13: return x <= 0.0031308f ? 12.92f * x : 1.055f * det_powf(x, 1.0f / 2.4f) - 0.055f;
        case 17: return x <= 0.0f ? 0.0f : det_powf((48.0f * x) / 52.37f, 1.0f / 2.6f);
        case 16:                                                     /* ST 2084 inverse EOTF */
        {
            if (x <= 0.0f) return 0.0f;
            const float xp = det_powf(x, 0.1593017578125f);
            const float num = (0.8359375f - 1.0f) + (18.8515625f - 18.6875f) * xp;
            const float den = 1.0f + 18.6875f * xp;
            return det_powf(1.0f + num / den, 78.84375f);
        }
        case 18:                                                     /* inverse 1.2 OOTF + ARIB STD-B67 OETF */
        {
            if (x <= 0.0f) return 0.0f;
            const float s = det_powf(x, 1.0f / 1.2f);
            return s <= 1.0f / 12.0f ? sqrtf(3.0f * s) : 0.17883277f * det_logf(12.0f * s - 0.28466892f) + 0.55991073f;
        }
    }
    return x;                                                        /* 8: linear */
}

/* ---- per-sample float pipeline (every operation in float, in this order) ---------------- */
static inline float hable(float in)
{
    const float a = 0.15f, b = 0.50f, c = 0.10f, d = 0.20f, e = 0.02f, f = 0.30f;
    return (in * (in * a + b * c) + d * e) / (in * (in * a + b) + d * f) - e / f;
}

/* vf_tonemap.c's operators on the brightest component; tm_a/b/c are per-filter constants */
static inline float tonemap_sig(const plan_t *p, float sig)
{
    switch (p->tonemap)
    {
        case 1: return sig * p->tm_param / p->tm_peak;                                       /* linear   */
        case 2:                                                                              /* gamma: tm_a = 1 / param, tm_b = pow(0.05 / peak, 1 / param) / 0.05 */
            return sig > 0.05f ? det_powf(sig / p->tm_peak, p->tm_a) : sig * p->tm_b;
        case 3: { const float v = sig * p->tm_param; return v < 0.f ? 0.f : v > 1.f ? 1.f : v; }   /* clip */
        case 4: return sig / (sig + p->tm_param) * (p->tm_peak + p->tm_param) / p->tm_peak;     /* reinhard */
        case 5: return hable(sig) / p->tm_a;                                                 /* hable: tm_a = hable(peak) */
        case 6:                                                                              /* mobius   */
            if (sig <= p->tm_param) return sig;
            return p->tm_c * (sig + p->tm_a) / (sig + p->tm_b);
    }
    return sig;                                                                              /* none     */
}

/* a row of a 3x3 matrix times a vector: one product, two fused multiply-adds, left to right */
static inline float row3(const float m[3], float a, float b, float c) { return fmaf(m[2], c, fmaf(m[1], b, m[0] * a)); }

static inline void convert_px(const plan_t *p, float y, float u, float v, float out[3])
{
    if (!p->need_linear)
    {
        for (int i = 0; i < 3; i++)
            out[i] = row3(p->m_direct[i], y, u, v);
        return;
    }
    float c[3], g[3];
    for (int i = 0; i < 3; i++)
    {
        const float e = row3(p->m_in[i], y, u, v);
        c[i] = to_linear(p->tc_in, e) * p->lin_scale;
    }
    if (p->tonemap >= 0)
    {
        float sig = c[0] > c[1] ? c[0] : c[1];
        sig = sig > c[2] ? sig : c[2];
        sig = sig > 1e-6f ? sig : 1e-6f;
        const float k = tonemap_sig(p, sig) / sig;
        for (int i = 0; i < 3; i++) c[i] *= k;
    }
    if (p->gamut)
        for (int i = 0; i < 3; i++)
            g[i] = row3(p->m_gamut[i], c[0], c[1], c[2]);
    else
        for (int i = 0; i < 3; i++) g[i] = c[i];
    for (int i = 0; i < 3; i++)
        g[i] = to_gamma(p->tc_out, g[i] * p->gam_scale);
    for (int i = 0; i < 3; i++)
        out[i] = row3(p->m_out[i], g[0], g[1], g[2]);
}

static inline int quant(float v, float mul, float off, int vmax)
{
    /* out-of-gamut input can reach +-inf / NaN on the way (e.g. the PQ EOTF beyond its pole): pinned to the clip
     * limits here rather than left to what the float -> integer conversion of the platform makes of them */
    float t = fmaf(v, mul, off);
    if (!(t > -1e9f)) t = -1e9f;
    if (t > 1e9f) t = 1e9f;
    const long q = lrintf(t);
    return q < 0 ? 0 : q > vmax ? vmax : (int)q;
}

/* ---- plan ---------------------------------------------------------------------------------- */
static int build_plan(plan_t *p, const orc_colorspace_params_t *cs, int depth)
{
    memset(p, 0, sizeof(*p));
    const int s = depth - 8;
    p->vmax = (1 << depth) - 1;
    double kr_i, kb_i, kr_o, kb_o;
    if (!matrix_coeffs(cs->in_matrix, &kr_i, &kb_i) || !matrix_coeffs(cs->out_matrix, &kr_o, &kb_o)) return -1;
    if (cs->in_range < 1 || cs->in_range > 2 || cs->out_range < 1 || cs->out_range > 2) return -1;
    const int lim_i = cs->in_range == 1, lim_o = cs->out_range == 1;
    p->yoff_in = lim_i ? (float)(16 << s) : 0.f;
    p->ymul_in = (float)(1.0 / (lim_i ? (double)(219 << s) : (double)p->vmax));
    p->coff_in = (float)(1 << (depth - 1));
    p->cmul_in = (float)(1.0 / (lim_i ? (double)(224 << s) : (double)p->vmax));
    p->yoff_out = lim_o ? (float)(16 << s) : 0.f;
    p->ymul_out = lim_o ? (float)(219 << s) : (float)p->vmax;
    p->coff_out = (float)(1 << (depth - 1));
    p->cmul_out = lim_o ? (float)(224 << s) : (float)p->vmax;

    const double kg_i = 1.0 - kr_i - kb_i, kg_o = 1.0 - kr_o - kb_o;
    double mi[3][3] = { { 1.0, 0.0, 2.0 * (1.0 - kr_i) },
                        { 1.0, -2.0 * kb_i * (1.0 - kb_i) / kg_i, -2.0 * kr_i * (1.0 - kr_i) / kg_i },
                        { 1.0, 2.0 * (1.0 - kb_i), 0.0 } };
    double mo[3][3] = { { kr_o, kg_o, kb_o },
                        { -kr_o / (2.0 * (1.0 - kb_o)), -kg_o / (2.0 * (1.0 - kb_o)), 0.5 },
                        { 0.5, -kg_o / (2.0 * (1.0 - kr_o)), -kb_o / (2.0 * (1.0 - kr_o)) } };
    /* YCgCo (AVCOL_SPC_YCGCO = 8; planes Y, Cg, Co): Y = (R + 2G + B) / 4, Cg = (-R + 2G - B) / 4, Co = (R - B) / 2 */
    static const double ycgco_i[3][3] = { { 1.0, -1.0, 1.0 }, { 1.0, 1.0, 0.0 }, { 1.0, -1.0, -1.0 } };
    static const double ycgco_o[3][3] = { { 0.25, 0.5, 0.25 }, { -0.25, 0.5, -0.25 }, { 0.5, 0.0, -0.5 } };
    if (cs->in_matrix == 8) memcpy(mi, ycgco_i, sizeof(mi));
    if (cs->out_matrix == 8) memcpy(mo, ycgco_o, sizeof(mo));
    const int tc_i = transfer_class(cs->in_transfer), tc_o = transfer_class(cs->out_transfer);
    const int pc_i = primaries_class(cs->in_prim), pc_o = primaries_class(cs->out_prim);
    p->need_linear = tc_i != tc_o || pc_i != pc_o;
    p->tonemap = -1;
    double md[3][3];
    mul3(md, mo, mi);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
        {
            p->m_in[i][j] = (float)mi[i][j];
            p->m_out[i][j] = (float)mo[i][j];
            p->m_direct[i][j] = (float)md[i][j];
        }
    if (!p->need_linear) return 0;

    p->gamut = pc_i != pc_o;
    if (p->gamut)
    {
        double xi[8], xo[8], g[3][3];
        if (!primaries_xy(pc_i, xi) || !primaries_xy(pc_o, xo)) return -1;
        gamut_matrix(g, xi, xo);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                p->m_gamut[i][j] = (float)g[i][j];
    }
    if (!transfer_known(tc_i, 0) || !transfer_known(tc_o, 1)) return -1;
    p->tc_in = tc_i;
    p->tc_out = tc_o;
    p->lin_scale = tc_i == 16 ? (float)(10000.0 / cs->npl) : tc_i == 18 ? (float)(1000.0 / cs->npl) : 1.0f;
    p->gam_scale = tc_o == 16 ? (float)(cs->npl / 10000.0) : tc_o == 18 ? (float)(cs->npl / 1000.0) : 1.0f;
    /* tone mapping only on the PQ / HLG -> other-transfer path (colorspace.c:126-127) */
    if ((cs->in_transfer == 16 || cs->in_transfer == 18) && tc_i != tc_o)
    {
        p->tonemap = cs->tonemap;
        float param = (float)cs->param;                 /* NAN = FFmpeg's default for the operator */
        const double peak = cs->peak;
        switch (cs->tonemap)
        {
            case 0: break;
            case 1: case 3: if (isnan(param)) param = 1.0f; break;
            case 2:                                                 /* gamma (vf_tonemap.c: default 1.8) */
                if (isnan(param)) param = 1.8f;
                p->tm_a = 1.0f / param;
                p->tm_b = det_powf(0.05f / (float)peak, p->tm_a) / 0.05f;
                break;
            case 4: param = isnan(param) ? 1.0f : (1.0f - param) / param; break;
            case 5: p->tm_a = hable((float)peak); break;
            case 6:
            {
                if (isnan(param)) param = 0.3f;
                const float j = param;
                const float a = -j * j * (peak - 1.0f) / (j * j - 2.0f * j + peak);
                const float b = (j * j - 2.0f * j * peak + peak) / (peak - 1.0f > 1e-6 ? peak - 1.0f : 1e-6);
                p->tm_a = a;
                p->tm_b = b;
                p->tm_c = (b * b + 2.0f * b * j + j * j) / (b - a);
                break;
            }
            default: return -1;
        }
        p->tm_param = param;
        p->tm_peak = (float)peak;
    }
    return 0;
}

/* ---- frame ---------------------------------------------------------------------------------- */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static inline float sample(const void *plane, int stride, int x, int y, int bps)
{
    const uint8_t *row = (const uint8_t *)plane + (size_t)y * stride;
    return bps == 1 ? (float)row[x] : (float)((const uint16_t *)row)[x];
}
static inline void store(void *plane, int stride, int x, int y, int bps, int v)
{
    uint8_t *row = (uint8_t *)plane + (size_t)y * stride;
    if (bps == 1) row[x] = (uint8_t)v; else ((uint16_t *)row)[x] = (uint16_t)v;
}

int orc_colorspace_frame(const orc_colorspace_params_t *cs, const void *const src[3], const int sstride[3],
                         void *const dst[3], const int dstride[3], int w, int h, int depth, int subw, int subh)
{
    plan_t p;
    if (depth < 8 || depth > 16 || build_plan(&p, cs, depth) != 0) return -1;
    const int bps = depth > 8 ? 2 : 1;
    const int cw = subw ? (w + 1) >> 1 : w, ch = subh ? (h + 1) >> 1 : h;
    float *oy = malloc(sizeof(float) * (size_t)w * h), *ou = malloc(sizeof(float) * (size_t)w * h),
          *ov = malloc(sizeof(float) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            /* chroma at this luma position: rows first, then columns */
            int r0 = y, r1 = y, c0 = x, c1 = x;
            float wy0 = 1.f, wy1 = 0.f, wx0 = 1.f, wx1 = 0.f;
            if (subh)
            {
                const int k = y >> 1;
                if (y & 1) { r0 = k; r1 = clampi(k + 1, 0, ch - 1); wy0 = 0.75f; wy1 = 0.25f; }
                else       { r0 = clampi(k - 1, 0, ch - 1); r1 = k; wy0 = 0.25f; wy1 = 0.75f; }
            }
            if (subw)
            {
                c0 = x >> 1; c1 = c0;
                if (x & 1) { c1 = clampi(c0 + 1, 0, cw - 1); wx0 = 0.5f; wx1 = 0.5f; }
            }
            float uv[2];
            for (int k = 0; k < 2; k++)
            {
                const void *pl = src[1 + k];
                const float s00 = (sample(pl, sstride[1 + k], c0, r0, bps) - p.coff_in) * p.cmul_in;
                const float s10 = (sample(pl, sstride[1 + k], c0, r1, bps) - p.coff_in) * p.cmul_in;
                const float s01 = (sample(pl, sstride[1 + k], c1, r0, bps) - p.coff_in) * p.cmul_in;
                const float s11 = (sample(pl, sstride[1 + k], c1, r1, bps) - p.coff_in) * p.cmul_in;
                /* rows first (a sample that is not interpolated is taken as it is), then columns */
                const float a = subh ? fmaf(wy1, s10, wy0 * s00) : s00;
                const float b = subh ? fmaf(wy1, s11, wy0 * s01) : s01;
                uv[k] = (x & 1) && subw ? fmaf(wx1, b, wx0 * a) : a;
            }
            const float yf = (sample(src[0], sstride[0], x, y, bps) - p.yoff_in) * p.ymul_in;
            float o[3];
            convert_px(&p, yf, uv[0], uv[1], o);
            oy[(size_t)y * w + x] = o[0];
            ou[(size_t)y * w + x] = o[1];
            ov[(size_t)y * w + x] = o[2];
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            store(dst[0], dstride[0], x, y, bps, quant(oy[(size_t)y * w + x], p.ymul_out, p.yoff_out, p.vmax));
    for (int k = 0; k < 2; k++)
    {
        const float *o = k ? ov : ou;
        for (int cy = 0; cy < ch; cy++)
            for (int cx = 0; cx < cw; cx++)
            {
                float col[3];
                const int xc = subw ? 2 * cx : cx;
                for (int i = 0; i < 3; i++)
                {
                    const int xx = clampi(xc - 1 + i, 0, w - 1);
                    if (subh)
                    {
                        const int y0 = clampi(2 * cy - 1, 0, h - 1), y1 = clampi(2 * cy, 0, h - 1),
                                  y2 = clampi(2 * cy + 1, 0, h - 1), y3 = clampi(2 * cy + 2, 0, h - 1);
                        col[i] = fmaf(0.125f, o[(size_t)y3 * w + xx], fmaf(0.375f, o[(size_t)y2 * w + xx],
                                 fmaf(0.375f, o[(size_t)y1 * w + xx], 0.125f * o[(size_t)y0 * w + xx])));
                    }
                    else
                        col[i] = o[(size_t)cy * w + xx];
                }
                const float v = subw ? fmaf(0.25f, col[2], fmaf(0.5f, col[1], 0.25f * col[0])) : col[1];
                store(dst[1 + k], dstride[1 + k], cx, cy, bps, quant(v, p.cmul_out, p.coff_out, p.vmax));
            }
    }
    free(oy); free(ou); free(ov);
    return 0;
}
