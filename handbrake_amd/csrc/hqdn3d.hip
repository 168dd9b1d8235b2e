// hqdn3d.hip — hqdn3d spatial/temporal IIR denoiser for gfx950 (8-bit).
//
//   hqdn3d_h_kernel    horizontal recurrence of hqdn3d_denoise_spatial   libhb/denoise.c:126-165
//   hqdn3d_vt_kernel   its vertical recurrence + the temporal step, and the
//                      first-frame seeding of hqdn3d_denoise_depth       :167-201
//   hqdn3d_t_kernel    hqdn3d_denoise_temporal (spatial strength 0)      :102-124
//
// The reference fuses three first-order recurrences in one raster scan.  They are
// separable: x-recurrence per row (rows in parallel), then y-recurrence per column on
// the x-filtered values plus the pointwise temporal step (columns in parallel).  Each
// step is cur + LUT[(prev-cur)>>4] with the 8192-entry int16 LUT the host builds with
// libm (denoise.c:78-94) held in LDS.  Latency-bound by the dependent LUT chain, not
// by HBM; bit-exact in integers.
#include "hbhip_internal.h"

namespace {

constexpr int LUT_N = 8192, CENTRE = 4096;

__device__ __forceinline__ uint32_t load8(uint32_t px) { return (px << 8) + 127u; }      // denoise.c:32-33
// LOAD / STORE for depth 8 + SH... i.e. SH = 16 - depth (denoise.c:32-35): 8 for bytes, 6 / 4 for
// 10 / 12-bit samples in 16-bit containers; the fixed-point state and the LUTs do not depend on it
template <int SH> __device__ __forceinline__ uint32_t load_sh(uint32_t px) { return (px << SH) + (((1u << SH) - 1u) >> 1); }
__device__ __forceinline__ uint32_t lowpass(int prev, int cur, const int16_t *coef)       // :96-100
{
    return (uint32_t)(cur + coef[CENTRE + ((prev - cur) >> 4)]);
}

__device__ __forceinline__ void stage_lut(int16_t *dst, const int16_t *src, int nthreads)
{
    for (int i = threadIdx.x; i < LUT_N / 2; i += nthreads)
        reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(src)[i];
}

// one thread per row; 16 pixels per global load
__global__ __launch_bounds__(64) void hqdn3d_h_kernel(const uint8_t *__restrict__ src, int spitch,
                                                      uint16_t *__restrict__ hbuf, int w, int h,
                                                      const int16_t *__restrict__ spatial_g)
{
    __shared__ int16_t lut[LUT_N];
    stage_lut(lut, spatial_g, 64);
    __syncthreads();
    const int y = blockIdx.x * 64 + threadIdx.x;
    if (y >= h) return;
    const uint8_t *s = src + (size_t)y * spitch;
    uint16_t *o = hbuf + (size_t)y * w;
    uint32_t run = load8(s[0]);
    if (y == 0) run = lowpass((int)run, (int)load8(s[0]), lut);      // row 0 quirk (:140-146)
    o[0] = (uint16_t)run;
    int x = 1;
    // head up to a 16-byte boundary, then 16 pixels per load
    for (; x < w && (x & 15); x++)
    {
        run = lowpass((int)run, (int)load8(s[x]), lut);
        o[x] = (uint16_t)run;
    }
    for (; x + 16 <= w; x += 16)
    {
        const uint4 v = *reinterpret_cast<const uint4 *>(s + x);
        const uint32_t wds[4] = { v.x, v.y, v.z, v.w };
        uint16_t r[16];
#pragma unroll
        for (int k = 0; k < 16; k++)
        {
            run = lowpass((int)run, (int)load8((wds[k >> 2] >> (8 * (k & 3))) & 0xffu), lut);
            r[k] = (uint16_t)run;
        }
#pragma unroll
        for (int k = 0; k < 16; k++) o[x + k] = r[k];
    }
    for (; x < w; x++)
    {
        run = lowpass((int)run, (int)load8(s[x]), lut);
        o[x] = (uint16_t)run;
    }
}

// one thread per column
__global__ __launch_bounds__(64) void hqdn3d_vt_kernel(const uint8_t *__restrict__ src, int spitch,
                                                       const uint16_t *__restrict__ hbuf,
                                                       uint16_t *__restrict__ ant, uint8_t *__restrict__ dst,
                                                       int dpitch, int w, int h, int seeded,
                                                       const int16_t *__restrict__ spatial_g,
                                                       const int16_t *__restrict__ temporal_g)
{
    __shared__ int16_t lut_s[LUT_N];
    __shared__ int16_t lut_t[LUT_N];
    stage_lut(lut_s, spatial_g, 64);
    stage_lut(lut_t, temporal_g, 64);
    __syncthreads();
    const int x = blockIdx.x * 64 + threadIdx.x;
    if (x >= w) return;
    uint32_t line = 0;
    // the loads of a row do not depend on the recurrence: fetch 4 rows ahead of the dependent
    // LUT chain, which is what bounds this kernel
    constexpr int U = 4;
    for (int y0 = 0; y0 < h; y0 += U)
    {
        uint32_t hv[U], pv[U];
#pragma unroll
        for (int k = 0; k < U; k++)
        {
            const int y = min(y0 + k, h - 1);
            hv[k] = hbuf[(size_t)y * w + x];
            pv[k] = seeded ? ant[(size_t)y * w + x] : (uint16_t)load8(src[(size_t)y * spitch + x]);
        }
#pragma unroll
        for (int k = 0; k < U; k++)
        {
            const int y = y0 + k;
            if (y >= h) break;
            const uint32_t v = y == 0 ? hv[k] : lowpass((int)(uint16_t)line, (int)hv[k], lut_s);
            line = v;
            const uint32_t t = lowpass((int)pv[k], (int)v, lut_t);
            ant[(size_t)y * w + x] = (uint16_t)t;
            dst[(size_t)y * dpitch + x] = (uint8_t)(t >> 8);
        }
    }
}

// temporal only: fully parallel
__global__ __launch_bounds__(256) void hqdn3d_t_kernel(const uint8_t *__restrict__ src, int spitch,
                                                       uint16_t *__restrict__ ant, uint8_t *__restrict__ dst,
                                                       int dpitch, int w, int h, int seeded,
                                                       const int16_t *__restrict__ temporal_g)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w || y >= h) return;
    const uint32_t cur = load8(src[(size_t)y * spitch + x]);
    const uint32_t prev = seeded ? ant[(size_t)y * w + x] : (uint16_t)cur;
    const uint32_t t = (uint32_t)((int)cur + temporal_g[CENTRE + (((int)prev - (int)cur) >> 4)]);
    ant[(size_t)y * w + x] = (uint16_t)t;
    dst[(size_t)y * dpitch + x] = (uint8_t)(t >> 8);
}

// ---- depth 10 / 12 (16-bit containers): the same three kernels with 16-bit samples ----------
template <int SH>
__global__ __launch_bounds__(64) void hqdn3d_h16_kernel(const uint8_t *__restrict__ src, int spitch,
                                                        uint16_t *__restrict__ hbuf, int w, int h,
                                                        const int16_t *__restrict__ spatial_g)
{
    __shared__ int16_t lut[LUT_N];
    stage_lut(lut, spatial_g, 64);
    __syncthreads();
    const int y = blockIdx.x * 64 + threadIdx.x;
    if (y >= h) return;
    const uint16_t *s = reinterpret_cast<const uint16_t *>(src + (size_t)y * spitch);
    uint16_t *o = hbuf + (size_t)y * w;
    uint32_t run = load_sh<SH>(s[0]);
    if (y == 0) run = lowpass((int)run, (int)load_sh<SH>(s[0]), lut);      // row 0 quirk (:140-146)
    o[0] = (uint16_t)run;
    for (int x = 1; x < w; x++)
    {
        run = lowpass((int)run, (int)load_sh<SH>(s[x]), lut);
        o[x] = (uint16_t)run;
    }
}

template <int SH>
__global__ __launch_bounds__(64) void hqdn3d_vt16_kernel(const uint8_t *__restrict__ src, int spitch,
                                                         const uint16_t *__restrict__ hbuf,
                                                         uint16_t *__restrict__ ant, uint8_t *__restrict__ dst,
                                                         int dpitch, int w, int h, int seeded,
                                                         const int16_t *__restrict__ spatial_g,
                                                         const int16_t *__restrict__ temporal_g)
{
    __shared__ int16_t lut_s[LUT_N];
    __shared__ int16_t lut_t[LUT_N];
    stage_lut(lut_s, spatial_g, 64);
    stage_lut(lut_t, temporal_g, 64);
    __syncthreads();
    const int x = blockIdx.x * 64 + threadIdx.x;
    if (x >= w) return;
    uint32_t line = 0;
    constexpr int U = 4;                                   // rows fetched ahead of the LUT chain
    for (int y0 = 0; y0 < h; y0 += U)
    {
        uint32_t hv[U], pv[U];
#pragma unroll
        for (int k = 0; k < U; k++)
        {
            const int y = min(y0 + k, h - 1);
            hv[k] = hbuf[(size_t)y * w + x];
            pv[k] = seeded ? ant[(size_t)y * w + x]
                           : (uint16_t)load_sh<SH>(reinterpret_cast<const uint16_t *>(src + (size_t)y * spitch)[x]);
        }
#pragma unroll
        for (int k = 0; k < U; k++)
        {
            const int y = y0 + k;
            if (y >= h) break;
            const uint32_t v = y == 0 ? hv[k] : lowpass((int)(uint16_t)line, (int)hv[k], lut_s);
            line = v;
            const uint32_t t = lowpass((int)pv[k], (int)v, lut_t);
            ant[(size_t)y * w + x] = (uint16_t)t;
            reinterpret_cast<uint16_t *>(dst + (size_t)y * dpitch)[x] = (uint16_t)(t >> SH);
        }
    }
}

template <int SH>
__global__ __launch_bounds__(256) void hqdn3d_t16_kernel(const uint8_t *__restrict__ src, int spitch,
                                                         uint16_t *__restrict__ ant, uint8_t *__restrict__ dst,
                                                         int dpitch, int w, int h, int seeded,
                                                         const int16_t *__restrict__ temporal_g)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w || y >= h) return;
    const uint32_t cur = load_sh<SH>(reinterpret_cast<const uint16_t *>(src + (size_t)y * spitch)[x]);
    const uint32_t prev = seeded ? ant[(size_t)y * w + x] : (uint16_t)cur;
    const uint32_t t = (uint32_t)((int)cur + temporal_g[CENTRE + (((int)prev - (int)cur) >> 4)]);
    ant[(size_t)y * w + x] = (uint16_t)t;
    reinterpret_cast<uint16_t *>(dst + (size_t)y * dpitch)[x] = (uint16_t)(t >> SH);
}

class Hqdn3dFilter : public SimpleFilter
{
public:
    Hqdn3dFilter(hbhip_ctx *c, const hbhip_hqdn3d_params &p) : SimpleFilter(c), par(p) {}
    ~Hqdn3dFilter() override
    {
        if (d_coef) (void)hipFree(d_coef);
        for (int c = 0; c < 3; c++) if (ant[c]) (void)hipFree(ant[c]);
        if (hbuf) (void)hipFree(hbuf);
    }
    int setup()
    {
        HBHIP_CHECK(ctx, hipMalloc((void **)&d_coef, sizeof(int16_t) * 6 * LUT_N));
        HBHIP_CHECK(ctx, hipMemcpyAsync(d_coef, par.coef, sizeof(int16_t) * 6 * LUT_N, hipMemcpyHostToDevice, ctx->stream));
        for (int c = 0; c < 3; c++)
            HBHIP_CHECK(ctx, hipMalloc((void **)&ant[c], sizeof(uint16_t) * (size_t)in_geo.pw[c] * in_geo.ph[c]));
        HBHIP_CHECK(ctx, hipMalloc((void **)&hbuf, sizeof(uint16_t) * (size_t)in_geo.pw[0] * in_geo.ph[0]));
        HBHIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        return HBHIP_OK;
    }
    int process(DevPicture *in, DevPicture *out) override
    {
        for (int c = 0; c < 3; c++)
        {
            const int w = in->width[c], h = in->height[c];
            const int16_t *sp = d_coef + (size_t)(2 * c) * LUT_N, *tp = sp + LUT_N;
            const bool spatial = par.coef[2 * c][0] != 0;             // spatial strength != 0 (denoise.c:191)
            const uint8_t *src = in->plane[c];
#define HQ_SPATIAL(HK, VK) do { \
                HBHIP_LAUNCH(ctx, "hqdn3d_h", HK, dim3((h + 63) / 64), dim3(64), 0, src, in->pitch[c], hbuf, w, h, sp); \
                HBHIP_LAUNCH(ctx, "hqdn3d_vt", VK, dim3((w + 63) / 64), dim3(64), 0, src, in->pitch[c], \
                             (const uint16_t *)hbuf, ant[c], out->plane[c], out->pitch[c], w, h, seeded[c], sp, tp); } while (0)
#define HQ_TEMPORAL(TK) HBHIP_LAUNCH(ctx, "hqdn3d_t", TK, dim3((w + 255) / 256, h), dim3(256), 0, src, in->pitch[c], \
                                     ant[c], out->plane[c], out->pitch[c], w, h, seeded[c], tp)
            if (in_geo.depth == 8)
            {
                if (spatial) HQ_SPATIAL(hqdn3d_h_kernel, hqdn3d_vt_kernel); else HQ_TEMPORAL(hqdn3d_t_kernel);
            }
            else if (in_geo.depth == 10)
            {
                if (spatial) HQ_SPATIAL(hqdn3d_h16_kernel<6>, hqdn3d_vt16_kernel<6>); else HQ_TEMPORAL(hqdn3d_t16_kernel<6>);
            }
            else
            {
                if (spatial) HQ_SPATIAL(hqdn3d_h16_kernel<4>, hqdn3d_vt16_kernel<4>); else HQ_TEMPORAL(hqdn3d_t16_kernel<4>);
            }
#undef HQ_SPATIAL
#undef HQ_TEMPORAL
            seeded[c] = 1;
        }
        HBHIP_CHECK(ctx, hipGetLastError());
        return HBHIP_OK;
    }
    hbhip_hqdn3d_params par;
    int16_t *d_coef = nullptr;
    uint16_t *ant[3] = {nullptr, nullptr, nullptr};
    uint16_t *hbuf = nullptr;
    int seeded[3] = {0, 0, 0};
};

} // namespace

extern "C" int hbhip_hqdn3d_create(hbhip_ctx *ctx, const hbhip_hqdn3d_params *p, int width, int height,
                                   int depth, int log2_chroma_w, int log2_chroma_h, hbhip_filter **out)
{
    if (!ctx || !p || !out) return HBHIP_ERR_ARG;
    *out = nullptr;
    if (depth != 8 && depth != 10 && depth != 12) return HBHIP_ERR_UNSUPPORTED;
    if (width < 1 || height < 1) return HBHIP_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    Hqdn3dFilter *f = new (std::nothrow) Hqdn3dFilter(ctx, *p);
    if (!f) return HBHIP_ERR_NOMEM;
    PicGeometry g;
    g.set(width, height, depth, log2_chroma_w, log2_chroma_h);
    f->configure(g, g);
    int rc = f->setup();
    if (rc != HBHIP_OK)
    {
        delete f;
        return rc;
    }
    *out = f;
    return HBHIP_OK;
}
