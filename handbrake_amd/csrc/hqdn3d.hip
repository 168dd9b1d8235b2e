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

// The three planes of a frame are independent, and each of these kernels is latency bound with
// few threads (one per row / per column), so one launch covers all planes: blockIdx.y (h, vt) or
// blockIdx.z (t) selects the plane.
struct HqPlane
{
    const uint8_t *src;
    uint8_t       *dst;
    uint16_t      *hbuf, *ant;
    const int16_t *spatial, *temporal;
    int spitch, dpitch, w, h, seeded, spatial_on;
};
struct HqArgs { HqPlane pl[3]; };

// PIX = uint8_t (SH = 8) or uint16_t (SH = 16 - depth): LOAD / STORE of denoise.c:32-35
template <typename PIX, int SH> __device__ __forceinline__ uint32_t hq_load(const uint8_t *row, int x)
{
    return load_sh<SH>(reinterpret_cast<const PIX *>(row)[x]);
}

// horizontal recurrence: one thread per row
template <typename PIX, int SH>
__global__ __launch_bounds__(64) void hqdn3d_h_kernel(HqArgs a)
{
    const HqPlane &P = a.pl[blockIdx.y];
    if (!P.spatial_on) return;
    __shared__ int16_t lut[LUT_N];
    stage_lut(lut, P.spatial, 64);
    __syncthreads();
    const int y = blockIdx.x * 64 + threadIdx.x;
    if (y >= P.h) return;
    const int w = P.w;
    const uint8_t *s = P.src + (size_t)y * P.spitch;
    uint16_t *o = P.hbuf + (size_t)y * w;
    uint32_t run = hq_load<PIX, SH>(s, 0);
    if (y == 0) run = lowpass((int)run, (int)hq_load<PIX, SH>(s, 0), lut);      // row 0 quirk (:140-146)
    o[0] = (uint16_t)run;
    int x = 1;
    if (sizeof(PIX) == 1)
    {
        // bytes: head up to a 16-byte boundary, then 16 pixels per load
        for (; x < w && (x & 15); x++)
        {
            run = lowpass((int)run, (int)hq_load<PIX, SH>(s, x), lut);
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
                run = lowpass((int)run, (int)load_sh<SH>((wds[k >> 2] >> (8 * (k & 3))) & 0xffu), lut);
                r[k] = (uint16_t)run;
            }
#pragma unroll
            for (int k = 0; k < 16; k++) o[x + k] = r[k];
        }
    }
    for (; x < w; x++)
    {
        run = lowpass((int)run, (int)hq_load<PIX, SH>(s, x), lut);
        o[x] = (uint16_t)run;
    }
}

// vertical recurrence + temporal step: one thread per column
template <typename PIX, int SH>
__global__ __launch_bounds__(64) void hqdn3d_vt_kernel(HqArgs a)
{
    const HqPlane &P = a.pl[blockIdx.y];
    if (!P.spatial_on) return;
    __shared__ int16_t lut_s[LUT_N];
    __shared__ int16_t lut_t[LUT_N];
    stage_lut(lut_s, P.spatial, 64);
    stage_lut(lut_t, P.temporal, 64);
    __syncthreads();
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int w = P.w, h = P.h;
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
            hv[k] = P.hbuf[(size_t)y * w + x];
            pv[k] = P.seeded ? P.ant[(size_t)y * w + x] : (uint16_t)hq_load<PIX, SH>(P.src + (size_t)y * P.spitch, x);
        }
#pragma unroll
        for (int k = 0; k < U; k++)
        {
            const int y = y0 + k;
            if (y >= h) break;
            const uint32_t v = y == 0 ? hv[k] : lowpass((int)(uint16_t)line, (int)hv[k], lut_s);
            line = v;
            const uint32_t t = lowpass((int)pv[k], (int)v, lut_t);
            P.ant[(size_t)y * w + x] = (uint16_t)t;
            reinterpret_cast<PIX *>(P.dst + (size_t)y * P.dpitch)[x] = (PIX)(t >> SH);
        }
    }
}

// temporal only (spatial strength 0): fully parallel
template <typename PIX, int SH>
__global__ __launch_bounds__(256) void hqdn3d_t_kernel(HqArgs a)
{
    const HqPlane &P = a.pl[blockIdx.z];
    if (P.spatial_on) return;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= P.w || y >= P.h) return;
    const uint32_t cur = hq_load<PIX, SH>(P.src + (size_t)y * P.spitch, x);
    const uint32_t prev = P.seeded ? P.ant[(size_t)y * P.w + x] : (uint16_t)cur;
    const uint32_t t = (uint32_t)((int)cur + P.temporal[CENTRE + (((int)prev - (int)cur) >> 4)]);
    P.ant[(size_t)y * P.w + x] = (uint16_t)t;
    reinterpret_cast<PIX *>(P.dst + (size_t)y * P.dpitch)[x] = (PIX)(t >> SH);
}

class Hqdn3dFilter : public SimpleFilter
{
public:
    Hqdn3dFilter(hbhip_ctx *c, const hbhip_hqdn3d_params &p) : SimpleFilter(c), par(p) {}
    ~Hqdn3dFilter() override
    {
        if (d_coef) (void)hipFree(d_coef);
        for (int c = 0; c < 3; c++) if (ant[c]) (void)hipFree(ant[c]);
        for (int c = 0; c < 3; c++) if (hbuf[c]) (void)hipFree(hbuf[c]);
    }
    int setup()
    {
        HBHIP_CHECK(ctx, hipMalloc((void **)&d_coef, sizeof(int16_t) * 6 * LUT_N));
        HBHIP_CHECK(ctx, hipMemcpyAsync(d_coef, par.coef, sizeof(int16_t) * 6 * LUT_N, hipMemcpyHostToDevice, ctx->stream));
        for (int c = 0; c < 3; c++)
            HBHIP_CHECK(ctx, hipMalloc((void **)&ant[c], sizeof(uint16_t) * (size_t)in_geo.pw[c] * in_geo.ph[c]));
        for (int c = 0; c < 3; c++)
            HBHIP_CHECK(ctx, hipMalloc((void **)&hbuf[c], sizeof(uint16_t) * (size_t)in_geo.pw[c] * in_geo.ph[c]));
        HBHIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        return HBHIP_OK;
    }
    int process(DevPicture *in, DevPicture *out) override
    {
        HqArgs a;
        bool any_spatial = false, any_temporal = false;
        int max_w = 0, max_h = 0;
        for (int c = 0; c < 3; c++)
        {
            HqPlane &P = a.pl[c];
            P.src = in->plane[c]; P.dst = out->plane[c];
            P.spitch = in->pitch[c]; P.dpitch = out->pitch[c];
            P.w = in->width[c]; P.h = in->height[c];
            P.hbuf = hbuf[c]; P.ant = ant[c];
            P.spatial = d_coef + (size_t)(2 * c) * LUT_N; P.temporal = P.spatial + LUT_N;
            P.seeded = seeded[c];
            P.spatial_on = par.coef[2 * c][0] != 0;                  // spatial strength != 0 (denoise.c:191)
            (P.spatial_on ? any_spatial : any_temporal) = true;
            max_w = std::max(max_w, P.w); max_h = std::max(max_h, P.h);
            seeded[c] = 1;
        }
#define HQ_GO(PIX, SH) do { \
            if (any_spatial) \
            { \
                HBHIP_LAUNCH(ctx, "hqdn3d_h", (hqdn3d_h_kernel<PIX, SH>), dim3((max_h + 63) / 64, 3), dim3(64), 0, a); \
                HBHIP_LAUNCH(ctx, "hqdn3d_vt", (hqdn3d_vt_kernel<PIX, SH>), dim3((max_w + 63) / 64, 3), dim3(64), 0, a); \
            } \
            if (any_temporal) \
                HBHIP_LAUNCH(ctx, "hqdn3d_t", (hqdn3d_t_kernel<PIX, SH>), dim3((max_w + 255) / 256, max_h, 3), dim3(256), 0, a); \
        } while (0)
        if (in_geo.depth == 8)       HQ_GO(uint8_t, 8);
        else if (in_geo.depth == 10) HQ_GO(uint16_t, 6);
        else                         HQ_GO(uint16_t, 4);
#undef HQ_GO
        HBHIP_CHECK(ctx, hipGetLastError());
        return HBHIP_OK;
    }
    hbhip_hqdn3d_params par;
    int16_t *d_coef = nullptr;
    uint16_t *ant[3] = {nullptr, nullptr, nullptr};
    uint16_t *hbuf[3] = {nullptr, nullptr, nullptr};   // h-filtered rows, one buffer per plane
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
