// colorspace.hip — the reference's colorspace filter (libhb/colorspace.c:20-207) for gfx950.
//
// In the reference colorspace_init only builds an FFmpeg graph
//     zscale=transfer=linear:npl -> format=gbrpf32le -> tonemap      (PQ / HLG sources only, :126-162)
//     -> zscale=primaries:transfer:matrix:range -> format=<pix_fmt>  (:170-193)
// which hb_avfilter_combine merges into HB_FILTER_AVFILTER; the arithmetic is zimg's and
// vf_tonemap.c's (not in the reference tree: "parity unpinned", pinned to oracle/colorspace_oracle.c,
// whose header lists what is restated and what is simplified).
//
// MI355X shape: the CPU graph makes five passes over float planes (to float, chroma to 4:4:4,
// colour conversion, chroma back to 4:2:0, to integer).  Here one launch does all of it for up to 16 frames: a
// workgroup owns a 60 x 30 luma tile (4:2:0; 64 / 32 where a direction is not subsampled), converts the tile plus the
// apron the chroma decimation needs (four columns to the left, one row above and below: a 64 x 32 region, two passes of
// 256 threads with four adjacent pixels each - dword loads and stores, the chroma samples of the four shared), keeps
// the converted Cb'/Cr' of the region in LDS (16 KB), writes luma straight out and then decimates chroma from LDS.
// HBM traffic is the input frame once and the output frame once.
#include "hbhip_internal.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr int CS_RW = 64, CS_RH = 32, CS_FRAMES = 16;

struct CsPlan
{
    int   need_linear, gamut, tonemap, vmax;
    float yoff_in, ymul_in, coff_in, cmul_in;
    float ymul_out, yoff_out, cmul_out, coff_out;
    float m_in[3][3], m_out[3][3], m_gamut[3][3], m_direct[3][3];
    float tm_param, tm_peak, tm_a, tm_b, tm_c;
    int   tc_in, tc_out;                 // transfer classes
    float lin_scale, gam_scale;          // PQ / HLG: display-light scale after the EOTF / before its inverse
};

struct CsBatch
{
    const uint8_t *src[CS_FRAMES][3];
    uint8_t       *dst[CS_FRAMES][3];
    int spitch[3], dpitch[3];
    int w, h, cw, ch;
};

// ---- per-sample pipeline: every operation in float, correctly rounded, in the oracle's order ----
// Deterministic float math: range reduction + fixed polynomials in IEEE single +, -, *, / and explicit fused
// multiply-adds (nothing contracted by the compiler, correctly rounded division) - the operation sequence of
// oracle/colorspace_oracle.c's det_* routines, so that transfer
// functions can be evaluated per sample (no tables, nothing clipped before the integer conversion) and still
// come out bit for bit as the checker's.  HOST_DEV: the same code builds the per-filter tone-map constants on the host.
#define HBHIP_HD __host__ __device__ __forceinline__
HBHIP_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
HBHIP_HD float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
HBHIP_HD float fdiv(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __fdiv_rn(a, b);
#else
    return a / b;
#endif
}

HBHIP_HD float ffma(float a, float b, float c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __fmaf_rn(a, b, c);
#else
    return fmaf(a, b, c);
#endif
}

HBHIP_HD float det_log2f(float x)
{
    const uint32_t bits = f2u(x);
    int e = (int)(bits >> 23) - 127;
    float m = u2f((bits & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f;
    const float z = f * f;
    float p = 7.0376836292e-2f;
    p = ffma(p, f, -1.1514610310e-1f);
    p = ffma(p, f, 1.1676998740e-1f);
    p = ffma(p, f, -1.2420140846e-1f);
    p = ffma(p, f, 1.4249322787e-1f);
    p = ffma(p, f, -1.6668057665e-1f);
    p = ffma(p, f, 2.0000714765e-1f);
    p = ffma(p, f, -2.4999993993e-1f);
    p = ffma(p, f, 3.3333331174e-1f);
    float y = (f * z) * p;
    y = ffma(-0.5f, z, y);
    return ffma(f + y, 1.44269502f, (float)e);
}

HBHIP_HD float det_exp2f(float y)
{
    if (!(y >= -126.0f)) return 0.0f;                  /* also what a NaN becomes */
    if (y > 127.0f) y = 127.0f;
    const int i = (int)(y + (y < 0.0f ? -0.5f : 0.5f));
    const float z = (y - (float)i) * 0.693147182f;
    float p = 0.000198412701f;
    p = ffma(p, z, 0.00138888892f);
    p = ffma(p, z, 0.00833333377f);
    p = ffma(p, z, 0.0416666679f);
    p = ffma(p, z, 0.166666672f);
    p = ffma(p, z, 0.5f);
    p = ffma(p, z, 1.0f);
    p = ffma(p, z, 1.0f);
    return p * u2f((uint32_t)(i + 127) << 23);
}

HBHIP_HD float det_powf(float x, float y) { return x <= 0.0f ? 0.0f : det_exp2f(y * det_log2f(x)); }
HBHIP_HD float det_expf(float x) { return det_exp2f(x * 1.44269502f); }
HBHIP_HD float det_logf(float x) { return det_log2f(x) * 0.693147182f; }

// transfer functions, display referred as zimg's set (gamma.cpp); classes as transfer_class() numbers them
// (9 / 10: log100 / log316, 11: xvYCC - oracle/colorspace_oracle.c has the definitions)
__device__ __forceinline__ float to_linear_dev(int cls, float v)
{
    switch (cls)
    {
        case 9:  return v <= 0.0f ? 0.01f : det_exp2f((2.0f * (v - 1.0f)) * 3.32192802f);
        case 10: return v <= 0.0f ? 0.00316227766f : det_exp2f((2.5f * (v - 1.0f)) * 3.32192802f);
        case 11: return v < 0.0f ? -det_powf(-v, 2.4f) : det_powf(v, 2.4f);
        case 1:  return det_powf(v, 2.4f);
        case 4:  return det_powf(v, 2.2f);
        case 5:  return det_powf(v, 2.8f);
        case 7:  return v < 0.0913f ? fdiv(v, 4.0f) : det_powf(fdiv(v + 0.1115f, 1.1115f), 1.0f / 0.45f);
        case 13: return v <= 0.04045f ? fdiv(v, 12.92f) : det_powf(fdiv(v + 0.055f, 1.055f), 2.4f);
        case 17: return v <= 0.0f ? 0.0f : fdiv(det_powf(v, 2.6f) * 52.37f, 48.0f);          // SMPTE ST 428-1
        case 16:
        {
            if (v <= 0.0f) return 0.0f;
            const float p = det_powf(v, 1.0f / 78.84375f);
            float num = p - 0.8359375f;
            if (num < 0.0f) num = 0.0f;
            float den = 18.8515625f - 18.6875f * p;
            if (den < 1e-6f) den = 1e-6f;
            return det_powf(fdiv(num, den), 1.0f / 0.1593017578125f);
        }
        case 18:
        {
            const float x = v < 0.0f ? 0.0f : v;
            const float s = x <= 0.5f ? fdiv(x * x, 3.0f) : fdiv(det_expf(fdiv(x - 0.55991073f, 0.17883277f)) + 0.28466892f, 12.0f);
            return det_powf(s, 1.2f);
        }
    }
    return v;
}

__device__ __forceinline__ float to_gamma_dev(int cls, float x)
{
    switch (cls)
    {
        case 9:  return x <= 0.01f ? 0.0f : 1.0f + fdiv(det_log2f(x) * 0.301029996f, 2.0f);
        case 10: return x <= 0.00316227766f ? 0.0f : 1.0f + fdiv(det_log2f(x) * 0.301029996f, 2.5f);
        case 11: return x < 0.0f ? -det_powf(-x, 1.0f / 2.4f) : det_powf(x, 1.0f / 2.4f);
        case 1:  return det_powf(x, 1.0f / 2.4f);
        case 4:  return det_powf(x, 1.0f / 2.2f);
        case 5:  return det_powf(x, 1.0f / 2.8f);
        case 7:  return x < 0.0228f ? 4.0f * x : 1.1115f * det_powf(x, 0.45f) - 0.1115f;
        case 13: return x <= 0.0031308f ? 12.92f * x : 1.055f * det_powf(x, 1.0f / 2.4f) - 0.055f;
        case 17: return x <= 0.0f ? 0.0f : det_powf(fdiv(48.0f * x, 52.37f), 1.0f / 2.6f);
        case 16:
        {
            if (x <= 0.0f) return 0.0f;
            const float xp = det_powf(x, 0.1593017578125f);
            const float num = (0.8359375f - 1.0f) + (18.8515625f - 18.6875f) * xp;
            const float den = 1.0f + 18.6875f * xp;
            return det_powf(1.0f + fdiv(num, den), 78.84375f);
        }
        case 18:
        {
            if (x <= 0.0f) return 0.0f;
            const float s = det_powf(x, 1.0f / 1.2f);
            return s <= 1.0f / 12.0f ? __builtin_sqrtf(3.0f * s) : 0.17883277f * det_logf(12.0f * s - 0.28466892f) + 0.55991073f;
        }
    }
    return x;
}

__device__ __forceinline__ float hable_dev(float in)
{
    const float a = 0.15f, b = 0.50f, c = 0.10f, d = 0.20f, e = 0.02f, f = 0.30f;
    return __fdiv_rn(in * (in * a + b * c) + d * e, in * (in * a + b) + d * f) - __fdiv_rn(e, f);
}

// vf_tonemap.c's operators on the brightest component
__device__ __forceinline__ float tonemap_sig(const CsPlan &p, float sig)
{
    switch (p.tonemap)
    {
        case 1: return __fdiv_rn(sig * p.tm_param, p.tm_peak);
        case 2: return sig > 0.05f ? det_powf(__fdiv_rn(sig, p.tm_peak), p.tm_a) : sig * p.tm_b;      // gamma: tm_a = 1 / param
        case 3: { const float v = sig * p.tm_param; return v < 0.f ? 0.f : v > 1.f ? 1.f : v; }
        case 4: return __fdiv_rn(__fdiv_rn(sig, sig + p.tm_param) * (p.tm_peak + p.tm_param), p.tm_peak);
        case 5: return __fdiv_rn(hable_dev(sig), p.tm_a);
        case 6:
            if (sig <= p.tm_param) return sig;
            return __fdiv_rn(p.tm_c * (sig + p.tm_a), sig + p.tm_b);
    }
    return sig;
}

__device__ __forceinline__ float row3(const float (&m)[3], float a, float b, float c)
{
    return __fmaf_rn(m[2], c, __fmaf_rn(m[1], b, m[0] * a));
}

// LINEAR = false: the instantiation for conversions without a trip through linear light (one matrix)
template <bool LINEAR>
__device__ __forceinline__ void convert_px(const CsPlan &p, float y, float u, float v, float out[3])
{
    if (!LINEAR || !p.need_linear)
    {
#pragma unroll
        for (int i = 0; i < 3; i++)
            out[i] = row3(p.m_direct[i], y, u, v);
        return;
    }
    float c[3], g[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
    {
        const float e = row3(p.m_in[i], y, u, v);
        c[i] = to_linear_dev(p.tc_in, e) * p.lin_scale;
    }
    if (p.tonemap >= 0)
    {
        float sig = c[0] > c[1] ? c[0] : c[1];
        sig = sig > c[2] ? sig : c[2];
        sig = sig > 1e-6f ? sig : 1e-6f;
        const float k = __fdiv_rn(tonemap_sig(p, sig), sig);
#pragma unroll
        for (int i = 0; i < 3; i++) c[i] *= k;
    }
    if (p.gamut)
    {
#pragma unroll
        for (int i = 0; i < 3; i++)
            g[i] = row3(p.m_gamut[i], c[0], c[1], c[2]);
    }
    else
    {
#pragma unroll
        for (int i = 0; i < 3; i++) g[i] = c[i];
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
        g[i] = to_gamma_dev(p.tc_out, g[i] * p.gam_scale);
#pragma unroll
    for (int i = 0; i < 3; i++)
        out[i] = row3(p.m_out[i], g[0], g[1], g[2]);
}

__device__ __forceinline__ int quant(float v, float mul, float off, int vmax)
{
    float t = __fmaf_rn(v, mul, off);        // +-inf / NaN (out-of-gamut input beyond a transfer function's pole): to the clip limits
    if (!(t > -1e9f)) t = -1e9f;
    if (t > 1e9f) t = 1e9f;
    const int q = __float2int_rn(t);
    return q < 0 ? 0 : q > vmax ? vmax : q;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

template <typename PIX>
__device__ __forceinline__ float samp(const uint8_t *row, int x) { return (float)reinterpret_cast<const PIX *>(row)[x]; }

// SUBW / SUBH: log2 of the chroma subsampling.  Region = CS_RW x CS_RH luma positions starting HX columns left of and HY
// rows above the tile; thread (gx, gr) converts the four positions 4 gx .. 4 gx + 3 of region rows gr and gr + 16.  The
// samples of both rows are fetched before the first is converted (one memory latency per workgroup, not two).
template <typename PIX, int SUBW, int SUBH, bool LINEAR>
__global__ __launch_bounds__(256) void colorspace_kernel(CsBatch a, CsPlan p)
{
    constexpr int HX = SUBW ? 4 : 0, HY = SUBH ? 1 : 0, TW = CS_RW - HX, TH = CS_RH - 2 * HY;
    constexpr bool LDS = SUBW || SUBH;
    constexpr int NC = SUBW ? 3 : 4;                               // chroma columns under four positions
    __shared__ __attribute__((aligned(16))) float s_u[LDS ? CS_RH : 1][CS_RW], s_v[LDS ? CS_RH : 1][CS_RW];
    const int f = blockIdx.z;
    const uint8_t *sy = a.src[f][0], *su = a.src[f][1], *sv = a.src[f][2];
    uint8_t *dy = a.dst[f][0];
    const int tid = threadIdx.x, gx = tid & 15, gr = tid >> 4;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int ux = x0 - HX + 4 * gx;                              // unclamped column of the thread's first position
    const bool wide = sizeof(PIX) == 1 ? ((a.dpitch[0] | a.spitch[0]) & 3) == 0 && (((uintptr_t)dy | (uintptr_t)sy) & 3) == 0
                                       : ((a.dpitch[0] | a.spitch[0]) & 7) == 0 && (((uintptr_t)dy | (uintptr_t)sy) & 7) == 0;
    // four positions none of which is clamped, rows that can be moved as dwords: luma as one load, the chroma columns
    // they share once
    const bool inner = wide && ux >= 0 && ux + 3 <= a.w - 1;
    int cc[NC];
    if (SUBW) { cc[0] = ux >> 1; cc[1] = cc[0] + 1; cc[NC - 1] = min(cc[0] + 2, a.cw - 1); }
    else
    {
#pragma unroll
        for (int j = 0; j < NC; j++) cc[j] = ux + j;
    }
    struct Rows { int y, r0, r1; float wy0, wy1; };
    auto rows_of = [&](int uy) __attribute__((always_inline)) {
        Rows r;
        r.y = clampi(uy, 0, a.h - 1);
        r.r0 = r.r1 = r.y; r.wy0 = 1.f; r.wy1 = 0.f;
        if (SUBH)
        {
            const int k = r.y >> 1;
            if (r.y & 1) { r.r0 = k; r.r1 = min(k + 1, a.ch - 1); r.wy0 = 0.75f; r.wy1 = 0.25f; }
            else         { r.r0 = max(k - 1, 0); r.r1 = k; r.wy0 = 0.25f; r.wy1 = 0.75f; }
        }
        return r;
    };
    struct Raw { uint32_t ly0, ly1; uint32_t u0[NC], u1[NC], v0[NC], v1[NC]; };
    auto fetch = [&](int uy) __attribute__((always_inline)) {
        Raw w = {};
        if (!inner || uy > a.h + 1) return w;
        const Rows r = rows_of(uy);
        const uint8_t *yrow = sy + (size_t)r.y * a.spitch[0];
        if (sizeof(PIX) == 1) w.ly0 = *reinterpret_cast<const uint32_t *>(yrow + ux);
        else { const uint2 v = *reinterpret_cast<const uint2 *>(yrow + 2 * ux); w.ly0 = v.x; w.ly1 = v.y; }
        const PIX *u0 = reinterpret_cast<const PIX *>(su + (size_t)r.r0 * a.spitch[1]), *u1 = reinterpret_cast<const PIX *>(su + (size_t)r.r1 * a.spitch[1]);
        const PIX *v0 = reinterpret_cast<const PIX *>(sv + (size_t)r.r0 * a.spitch[2]), *v1 = reinterpret_cast<const PIX *>(sv + (size_t)r.r1 * a.spitch[2]);
#pragma unroll
        for (int j = 0; j < NC; j++)
        {
            w.u0[j] = u0[cc[j]]; w.v0[j] = v0[cc[j]];
            if (SUBH) { w.u1[j] = u1[cc[j]]; w.v1[j] = v1[cc[j]]; }
        }
        return w;
    };
    // (the instantiation with the transfer functions has its registers full of them: it fetches a row when it gets there)
    Raw rawA = {}, rawB = {};
    if constexpr (!LINEAR) { rawA = fetch(y0 - HY + gr); rawB = fetch(y0 - HY + gr + 16); }

    // phase 1: convert the region at luma resolution (positions outside the picture are the nearest one inside: what
    // the decimation reads there).  (With the transfer functions inlined a conversion is a few hundred instructions: the
    // two passes are a loop there.)
    auto pass = [&](int it) __attribute__((always_inline)) {
        const int ry = gr + 16 * it;
        const int uy = y0 - HY + ry;
        if (ux > a.w || uy > a.h + 1) return;                      // nothing reads a position beyond column w, row h + 1
        const Rows r = rows_of(uy);
        const float wy0 = r.wy0, wy1 = r.wy1;
        float yf[4], uf[4], vf[4];
        if (inner)
        {
            Raw w;
            if constexpr (LINEAR) w = fetch(uy);
            else w = it ? rawB : rawA;
            float ys[4];
            if (sizeof(PIX) == 1)
            {
#pragma unroll
                for (int j = 0; j < 4; j++) ys[j] = (float)((w.ly0 >> (8 * j)) & 0xffu);
            }
            else
            {
                ys[0] = (float)(w.ly0 & 0xffffu); ys[1] = (float)(w.ly0 >> 16); ys[2] = (float)(w.ly1 & 0xffffu); ys[3] = (float)(w.ly1 >> 16);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) yf[j] = (ys[j] - p.yoff_in) * p.ymul_in;
            float cu[NC], cv[NC];
#pragma unroll
            for (int j = 0; j < NC; j++)
            {
                const float u0 = ((float)w.u0[j] - p.coff_in) * p.cmul_in, v0 = ((float)w.v0[j] - p.coff_in) * p.cmul_in;
                if (SUBH)
                {
                    const float u1 = ((float)w.u1[j] - p.coff_in) * p.cmul_in, v1 = ((float)w.v1[j] - p.coff_in) * p.cmul_in;
                    cu[j] = __fmaf_rn(wy1, u1, wy0 * u0);
                    cv[j] = __fmaf_rn(wy1, v1, wy0 * v0);
                }
                else { cu[j] = u0; cv[j] = v0; }
            }
            if (SUBW)
            {
                uf[0] = cu[0]; uf[1] = __fmaf_rn(0.5f, cu[1], 0.5f * cu[0]); uf[2] = cu[1]; uf[3] = __fmaf_rn(0.5f, cu[NC - 1], 0.5f * cu[1]);
                vf[0] = cv[0]; vf[1] = __fmaf_rn(0.5f, cv[1], 0.5f * cv[0]); vf[2] = cv[1]; vf[3] = __fmaf_rn(0.5f, cv[NC - 1], 0.5f * cv[1]);
            }
            else
            {
#pragma unroll
                for (int j = 0; j < 4; j++) { uf[j] = cu[j]; vf[j] = cv[j]; }
            }
        }
        else
        {
            const uint8_t *yrow = sy + (size_t)r.y * a.spitch[0];
            const uint8_t *urow0 = su + (size_t)r.r0 * a.spitch[1], *urow1 = su + (size_t)r.r1 * a.spitch[1];
            const uint8_t *vrow0 = sv + (size_t)r.r0 * a.spitch[2], *vrow1 = sv + (size_t)r.r1 * a.spitch[2];
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                const int x = clampi(ux + j, 0, a.w - 1);
                yf[j] = (samp<PIX>(yrow, x) - p.yoff_in) * p.ymul_in;
                int c0 = x, c1 = x;
                bool mix = false;
                if (SUBW) { c0 = x >> 1; c1 = c0; if (x & 1) { c1 = min(c0 + 1, a.cw - 1); mix = true; } }
                float au, bu, av, bv;
                {
                    const float u00 = (samp<PIX>(urow0, c0) - p.coff_in) * p.cmul_in, u01 = (samp<PIX>(urow0, c1) - p.coff_in) * p.cmul_in;
                    const float v00 = (samp<PIX>(vrow0, c0) - p.coff_in) * p.cmul_in, v01 = (samp<PIX>(vrow0, c1) - p.coff_in) * p.cmul_in;
                    if (SUBH)
                    {
                        const float u10 = (samp<PIX>(urow1, c0) - p.coff_in) * p.cmul_in, u11 = (samp<PIX>(urow1, c1) - p.coff_in) * p.cmul_in;
                        const float v10 = (samp<PIX>(vrow1, c0) - p.coff_in) * p.cmul_in, v11 = (samp<PIX>(vrow1, c1) - p.coff_in) * p.cmul_in;
                        au = __fmaf_rn(wy1, u10, wy0 * u00); bu = __fmaf_rn(wy1, u11, wy0 * u01);
                        av = __fmaf_rn(wy1, v10, wy0 * v00); bv = __fmaf_rn(wy1, v11, wy0 * v01);
                    }
                    else { au = u00; bu = u01; av = v00; bv = v01; }
                }
                uf[j] = mix ? __fmaf_rn(0.5f, bu, 0.5f * au) : au;
                vf[j] = mix ? __fmaf_rn(0.5f, bv, 0.5f * av) : av;
            }
        }
        float oy[4], ou[4], ov[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            float o[3];
            convert_px<LINEAR>(p, yf[j], uf[j], vf[j], o);
            oy[j] = o[0]; ou[j] = o[1]; ov[j] = o[2];
        }
        if (LDS)
        {
            *reinterpret_cast<float4 *>(&s_u[ry][4 * gx]) = make_float4(ou[0], ou[1], ou[2], ou[3]);
            *reinterpret_cast<float4 *>(&s_v[ry][4 * gx]) = make_float4(ov[0], ov[1], ov[2], ov[3]);
        }
        // the tile's own positions: luma out (and chroma, where it is not subsampled)
        if (ux >= x0 && ux < a.w && uy >= y0 && uy < y0 + TH && uy < a.h)
        {
            uint32_t q[4];
#pragma unroll
            for (int j = 0; j < 4; j++) q[j] = (uint32_t)quant(oy[j], p.ymul_out, p.yoff_out, p.vmax);
            PIX *d = reinterpret_cast<PIX *>(dy + (size_t)uy * a.dpitch[0]) + ux;
            if (inner)
            {
                if (sizeof(PIX) == 1) *reinterpret_cast<uint32_t *>(d) = q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24);
                else                  *reinterpret_cast<uint2 *>(d) = make_uint2(q[0] | (q[1] << 16), q[2] | (q[3] << 16));
            }
            else
            {
#pragma unroll
                for (int j = 0; j < 4; j++) if (ux + j < a.w) d[j] = (PIX)q[j];
            }
            if (!LDS)
            {
                PIX *du = reinterpret_cast<PIX *>(a.dst[f][1] + (size_t)uy * a.dpitch[1]) + ux;
                PIX *dv = reinterpret_cast<PIX *>(a.dst[f][2] + (size_t)uy * a.dpitch[2]) + ux;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (ux + j < a.w)
                    {
                        du[j] = (PIX)quant(ou[j], p.cmul_out, p.coff_out, p.vmax);
                        dv[j] = (PIX)quant(ov[j], p.cmul_out, p.coff_out, p.vmax);
                    }
            }
        }
    };
    if (LINEAR)
    {
#pragma unroll 1
        for (int it = 0; it < 2; it++) pass(it);
    }
    else { pass(0); pass(1); }
    if (!LDS) return;
    __syncthreads();

    // phase 2: chroma of the tile, decimated from LDS (rows first, then columns)
    constexpr int NCX = TW >> SUBW, NCY = TH >> SUBH;
    for (int i = tid; i < NCX * NCY; i += 256)
    {
        const int oy = i / NCX, ox = i - oy * NCX;
        const int cx = (x0 >> SUBW) + ox, cy = (y0 >> SUBH) + oy;
        if (cx >= a.cw || cy >= a.ch) continue;
        const int rx = (ox << SUBW) + HX, ry = (oy << SUBH) + HY;
#pragma unroll
        for (int k = 0; k < 2; k++)
        {
            float (*s)[CS_RW] = k ? s_v : s_u;
            float col[3];
#pragma unroll
            for (int j = 0; j < 3; j++)
            {
                if (!SUBW && j != 1) continue;
                const int xx = SUBW ? rx - 1 + j : rx;
                col[j] = SUBH ? __fmaf_rn(0.125f, s[ry + 2][xx], __fmaf_rn(0.375f, s[ry + 1][xx], __fmaf_rn(0.375f, s[ry][xx], 0.125f * s[ry - 1][xx])))
                              : s[ry][xx];
            }
            const float v = SUBW ? __fmaf_rn(0.25f, col[2], __fmaf_rn(0.5f, col[1], 0.25f * col[0])) : col[1];
            reinterpret_cast<PIX *>(a.dst[f][1 + k] + (size_t)cy * a.dpitch[1 + k])[cx] = (PIX)quant(v, p.cmul_out, p.coff_out, p.vmax);
        }
    }
}

// ---- plan construction on the host (double, host libm; the oracle builds the same tables) --------
void mul3(double r[3][3], const double a[3][3], const double b[3][3])
{
    double t[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            t[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j];
    memcpy(r, t, sizeof(t));
}

void inv3(double r[3][3], const double m[3][3])
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

// (Kr, Kb) of an AVCOL_SPC_* / HB_COLR_MAT_* id
bool luma_coefficients(int id, double &kr, double &kb)
{
    switch (id)
    {
        case 1: kr = 0.2126; kb = 0.0722; return true;
        case 4: kr = 0.30;   kb = 0.11;   return true;
        case 5: case 6: kr = 0.299; kb = 0.114; return true;
        case 7: kr = 0.212;  kb = 0.087;  return true;
        case 8: kr = 0.25; kb = 0.25; return true;          // YCgCo: placeholders, setup() installs its fixed matrix
        case 9: kr = 0.2627; kb = 0.0593; return true;
    }
    return false;
}

// chromaticities (xr yr xg yg xb yb xw yw) of an AVCOL_PRI_* / HB_COLR_PRI_* id
bool chromaticities(int id, double xy[8])
{
    struct Row { int id; double v[8]; };
    static const Row rows[] = {
        { 1,  { 0.640, 0.330, 0.300, 0.600, 0.150, 0.060, 0.3127, 0.3290 } },
        { 4,  { 0.670, 0.330, 0.210, 0.710, 0.140, 0.080, 0.310,  0.316  } },
        { 5,  { 0.640, 0.330, 0.290, 0.600, 0.150, 0.060, 0.3127, 0.3290 } },
        { 6,  { 0.630, 0.340, 0.310, 0.595, 0.155, 0.070, 0.3127, 0.3290 } },
        { 8,  { 0.681, 0.319, 0.243, 0.692, 0.145, 0.049, 0.310,  0.316  } },
        { 9,  { 0.708, 0.292, 0.170, 0.797, 0.131, 0.046, 0.3127, 0.3290 } },
        { 11, { 0.680, 0.320, 0.265, 0.690, 0.150, 0.060, 0.314,  0.351  } },
        { 12, { 0.680, 0.320, 0.265, 0.690, 0.150, 0.060, 0.3127, 0.3290 } },
        { 22, { 0.630, 0.340, 0.295, 0.605, 0.155, 0.077, 0.3127, 0.3290 } },
    };
    for (const Row &r : rows)
        if (r.id == id) { memcpy(xy, r.v, sizeof(r.v)); return true; }
    return false;
}

int primaries_class(int id) { return id == 7 ? 6 : id; }                            // smpte240m = smpte170m (SMPTE C)
int transfer_class(int id) { return (id == 6 || id == 14 || id == 15) ? 1 : id; }   // 601 / 2020 share 709's curve

void rgb_to_xyz(double m[3][3], const double xy[8])
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

void gamut_matrix(double g[3][3], const double in_xy[8], const double out_xy[8])
{
    double a[3][3], b[3][3], bi[3][3];
    rgb_to_xyz(a, in_xy);
    rgb_to_xyz(b, out_xy);
    inv3(bi, b);
    if (in_xy[6] != out_xy[6] || in_xy[7] != out_xy[7])
    {
        // Bradford adaptation between the white points
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

// display-referred transfer functions (zimg's set); PQ / HLG only towards linear light
bool transfer_known(int cls)
{
    return cls == 1 || cls == 4 || cls == 5 || cls == 7 || (cls >= 8 && cls <= 11) || cls == 13 || cls == 16 || cls == 17 || cls == 18;
}

float hable_host(float in)
{
    const float a = 0.15f, b = 0.50f, c = 0.10f, d = 0.20f, e = 0.02f, f = 0.30f;
    return (in * (in * a + b * c) + d * e) / (in * (in * a + b) + d * f) - e / f;
}

class ColorspaceFilter : public SimpleFilter
{
public:
    ColorspaceFilter(hbhip_ctx *c, const hbhip_colorspace_params &p) : SimpleFilter(c), par(p) {}
    int setup(int depth)
    {
        memset(&plan, 0, sizeof(plan));
        const int s = depth - 8;
        plan.vmax = (1 << depth) - 1;
        double kr_i, kb_i, kr_o, kb_o;
        if (!luma_coefficients(par.in_matrix, kr_i, kb_i) || !luma_coefficients(par.out_matrix, kr_o, kb_o)) return HBHIP_ERR_UNSUPPORTED;
        if (par.in_range < 1 || par.in_range > 2 || par.out_range < 1 || par.out_range > 2) return HBHIP_ERR_UNSUPPORTED;
        const bool lim_i = par.in_range == 1, lim_o = par.out_range == 1;
        plan.yoff_in = lim_i ? (float)(16 << s) : 0.f;
        plan.ymul_in = (float)(1.0 / (lim_i ? (double)(219 << s) : (double)plan.vmax));
        plan.coff_in = (float)(1 << (depth - 1));
        plan.cmul_in = (float)(1.0 / (lim_i ? (double)(224 << s) : (double)plan.vmax));
        plan.yoff_out = lim_o ? (float)(16 << s) : 0.f;
        plan.ymul_out = lim_o ? (float)(219 << s) : (float)plan.vmax;
        plan.coff_out = (float)(1 << (depth - 1));
        plan.cmul_out = lim_o ? (float)(224 << s) : (float)plan.vmax;

        const double kg_i = 1.0 - kr_i - kb_i, kg_o = 1.0 - kr_o - kb_o;
        double mi[3][3] = { { 1.0, 0.0, 2.0 * (1.0 - kr_i) },
                            { 1.0, -2.0 * kb_i * (1.0 - kb_i) / kg_i, -2.0 * kr_i * (1.0 - kr_i) / kg_i },
                            { 1.0, 2.0 * (1.0 - kb_i), 0.0 } };
        double mo[3][3] = { { kr_o, kg_o, kb_o },
                            { -kr_o / (2.0 * (1.0 - kb_o)), -kg_o / (2.0 * (1.0 - kb_o)), 0.5 },
                            { 0.5, -kg_o / (2.0 * (1.0 - kr_o)), -kb_o / (2.0 * (1.0 - kr_o)) } };
        // YCgCo (AVCOL_SPC_YCGCO = 8; planes Y, Cg, Co): Y = (R + 2G + B) / 4, Cg = (-R + 2G - B) / 4, Co = (R - B) / 2
        static const double ycgco_i[3][3] = { { 1.0, -1.0, 1.0 }, { 1.0, 1.0, 0.0 }, { 1.0, -1.0, -1.0 } };
        static const double ycgco_o[3][3] = { { 0.25, 0.5, 0.25 }, { -0.25, 0.5, -0.25 }, { 0.5, 0.0, -0.5 } };
        if (par.in_matrix == 8) memcpy(mi, ycgco_i, sizeof(mi));
        if (par.out_matrix == 8) memcpy(mo, ycgco_o, sizeof(mo));
        const int tc_i = transfer_class(par.in_transfer), tc_o = transfer_class(par.out_transfer);
        const int pc_i = primaries_class(par.in_prim), pc_o = primaries_class(par.out_prim);
        plan.need_linear = tc_i != tc_o || pc_i != pc_o;
        plan.tonemap = -1;
        double md[3][3];
        mul3(md, mo, mi);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
            {
                plan.m_in[i][j] = (float)mi[i][j];
                plan.m_out[i][j] = (float)mo[i][j];
                plan.m_direct[i][j] = (float)md[i][j];
            }
        if (!plan.need_linear) return HBHIP_OK;

        plan.gamut = pc_i != pc_o;
        if (plan.gamut)
        {
            double xi[8], xo[8], g[3][3];
            if (!chromaticities(pc_i, xi) || !chromaticities(pc_o, xo)) return HBHIP_ERR_UNSUPPORTED;
            gamut_matrix(g, xi, xo);
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++)
                    plan.m_gamut[i][j] = (float)g[i][j];
        }
        if (!transfer_known(tc_i) || !transfer_known(tc_o)) return HBHIP_ERR_UNSUPPORTED;
        plan.tc_in = tc_i;
        plan.tc_out = tc_o;
        plan.lin_scale = tc_i == 16 ? (float)(10000.0 / par.npl) : tc_i == 18 ? (float)(1000.0 / par.npl) : 1.0f;
        plan.gam_scale = tc_o == 16 ? (float)(par.npl / 10000.0) : tc_o == 18 ? (float)(par.npl / 1000.0) : 1.0f;
        // tone mapping only on the PQ / HLG -> other-transfer path (colorspace.c:126-127)
        if ((par.in_transfer == 16 || par.in_transfer == 18) && tc_i != tc_o)
        {
            plan.tonemap = par.tonemap;
            float param = (float)par.param;              // NAN = vf_tonemap's default for the operator
            const double peak = par.peak;
            if (!(peak > 0)) return HBHIP_ERR_ARG;
            switch (par.tonemap)
            {
                case HBHIP_TONEMAP_NONE: break;
                case HBHIP_TONEMAP_LINEAR: case HBHIP_TONEMAP_CLIP: if (std::isnan(param)) param = 1.0f; break;
                case HBHIP_TONEMAP_GAMMA:                  // vf_tonemap.c: default 1.8
                    if (std::isnan(param)) param = 1.8f;
                    plan.tm_a = 1.0f / param;
                    plan.tm_b = det_powf(0.05f / (float)peak, plan.tm_a) / 0.05f;
                    break;
                case HBHIP_TONEMAP_REINHARD: param = std::isnan(param) ? 1.0f : (1.0f - param) / param; break;
                case HBHIP_TONEMAP_HABLE: plan.tm_a = hable_host((float)peak); break;
                case HBHIP_TONEMAP_MOBIUS:
                {
                    if (std::isnan(param)) param = 0.3f;
                    const float j = param;
                    const float a = -j * j * (peak - 1.0f) / (j * j - 2.0f * j + peak);
                    const float b = (j * j - 2.0f * j * peak + peak) / (peak - 1.0f > 1e-6 ? peak - 1.0f : 1e-6);
                    plan.tm_a = a;
                    plan.tm_b = b;
                    plan.tm_c = (b * b + 2.0f * b * j + j * j) / (b - a);
                    break;
                }
                default: return HBHIP_ERR_UNSUPPORTED;
            }
            plan.tm_param = param;
            plan.tm_peak = (float)peak;
        }
        return HBHIP_OK;
    }
    template <typename PIX>
    void launch(const CsBatch &B, int nf)
    {
        const int sw = in_geo.log2_cw, sh = in_geo.log2_ch;
        const int tw = CS_RW - (sw ? 4 : 0), th = CS_RH - (sh ? 2 : 0);
        const dim3 grid(hbhip_grid_x((B.w + tw - 1) / tw), (B.h + th - 1) / th, nf);
        if (plan.need_linear)
        {
            if (sw && sh)       HBHIP_LAUNCH(ctx, "colorspace", (colorspace_kernel<PIX, 1, 1, true>), grid, dim3(256), 0, B, plan);
            else if (sw)        HBHIP_LAUNCH(ctx, "colorspace", (colorspace_kernel<PIX, 1, 0, true>), grid, dim3(256), 0, B, plan);
            else if (sh)        HBHIP_LAUNCH(ctx, "colorspace", (colorspace_kernel<PIX, 0, 1, true>), grid, dim3(256), 0, B, plan);
            else                HBHIP_LAUNCH(ctx, "colorspace", (colorspace_kernel<PIX, 0, 0, true>), grid, dim3(256), 0, B, plan);
        }
        else
        {
            if (sw && sh)       HBHIP_LAUNCH(ctx, "colorspace", (colorspace_kernel<PIX, 1, 1, false>), grid, dim3(256), 0, B, plan);
            else if (sw)        HBHIP_LAUNCH(ctx, "colorspace", (colorspace_kernel<PIX, 1, 0, false>), grid, dim3(256), 0, B, plan);
            else if (sh)        HBHIP_LAUNCH(ctx, "colorspace", (colorspace_kernel<PIX, 0, 1, false>), grid, dim3(256), 0, B, plan);
            else                HBHIP_LAUNCH(ctx, "colorspace", (colorspace_kernel<PIX, 0, 0, false>), grid, dim3(256), 0, B, plan);
        }
    }
    // the frames of a batch (a chain batch, a device-resident batch) in one launch where their pitches agree
    int process_many(DevPicture *const *ins, DevPicture *const *outs, int n) override
    {
        int at = 0;
        while (at < n)
        {
            int nf = 1;
            auto same = [&](int i) {
                for (int c = 0; c < 3; c++)
                    if (ins[i]->pitch[c] != ins[at]->pitch[c] || outs[i]->pitch[c] != outs[at]->pitch[c]) return false;
                return true;
            };
            while (at + nf < n && nf < CS_FRAMES && same(at + nf)) nf++;
            CsBatch B;
            memset(&B, 0, sizeof(B));
            for (int f = 0; f < nf; f++)
                for (int c = 0; c < 3; c++) { B.src[f][c] = ins[at + f]->plane[c]; B.dst[f][c] = outs[at + f]->plane[c]; }
            for (int c = 0; c < 3; c++) { B.spitch[c] = ins[at]->pitch[c]; B.dpitch[c] = outs[at]->pitch[c]; }
            B.w = ins[at]->width[0]; B.h = ins[at]->height[0]; B.cw = ins[at]->width[1]; B.ch = ins[at]->height[1];
            if (in_geo.bps == 1) launch<uint8_t>(B, nf);
            else                 launch<uint16_t>(B, nf);
            HBHIP_CHECK(ctx, hipGetLastError());
            at += nf;
        }
        return HBHIP_OK;
    }
    int process(DevPicture *in, DevPicture *out) override
    {
        DevPicture *i1[1] = { in }, *o1[1] = { out };
        return process_many(i1, o1, 1);
    }
    hbhip_colorspace_params par;
    CsPlan plan;
};

} // namespace

extern "C" int hbhip_colorspace_create(hbhip_ctx *ctx, const hbhip_colorspace_params *p, int width, int height,
                                       int depth, int log2_chroma_w, int log2_chroma_h, hbhip_filter **out)
{
    if (!ctx || !p || !out) return HBHIP_ERR_ARG;
    *out = nullptr;
    if (depth != 8 && depth != 10 && depth != 12) return HBHIP_ERR_UNSUPPORTED;
    if (log2_chroma_w < 0 || log2_chroma_w > 1 || log2_chroma_h < 0 || log2_chroma_h > 1) return HBHIP_ERR_UNSUPPORTED;
    if (width < 2 || height < 2 || !(p->npl > 0)) return HBHIP_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    ColorspaceFilter *f = new (std::nothrow) ColorspaceFilter(ctx, *p);
    if (!f) return HBHIP_ERR_NOMEM;
    PicGeometry g;
    g.set(width, height, depth, log2_chroma_w, log2_chroma_h);
    f->configure(g, g);
    int rc = f->setup(depth);
    if (rc != HBHIP_OK) { delete f; return rc; }
    *out = f;
    return HBHIP_OK;
}
