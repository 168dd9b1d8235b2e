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
// by HBM, so each chain is additionally cut into segments that are computed speculatively
// in parallel and verified / repaired afterwards (see below); bit-exact in integers.
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
    uint16_t      *ant_out;      // new temporal state (spatial path: `ant` stays intact for repairs)
    uint16_t      *vstate;       // vertical spatial state of every sample (what a repair compares with)
    const int16_t *spatial, *temporal;
    int spitch, dpitch, w, h, seeded, spatial_on;
};
struct HqArgs { HqPlane pl[3]; };

// PIX = uint8_t (SH = 8) or uint16_t (SH = 16 - depth): LOAD / STORE of denoise.c:32-35
template <typename PIX, int SH> __device__ __forceinline__ uint32_t hq_load(const uint8_t *row, int x)
{
    return load_sh<SH>(reinterpret_cast<const PIX *>(row)[x]);
}

// ---- exact speculative segmentation of the two recurrences -----------------------------------
// A row (column) is one chain of w (h) dependent LUT steps, ~45 ns each on this machine: a 1080p
// frame would cost 1920 + 1080 serial steps = 0.3 ms however many CUs idle next to it.  The chain
// is nonlinear (no scan), but it forgets: two runs over the same samples that start from different
// states meet after a few samples (a jump larger than the LUT's support resets the state to the
// sample itself; small differences decay and then quantise away) and stay together from then on.
// So every chain is cut into SEG segments.  Segment s > 0 starts WARM samples early from the state
// "sample itself" (how a row starts anyway), and only its results from its own first sample on are
// kept, together with the state it ENTERED its first sample with (`in`) and the state it left its
// last sample with (`out`).  After a barrier one thread per chain walks the segments in order: if
// in[s] equals the true out[s-1], segment s was computed from the right state and is exact; if not,
// it is recomputed serially from out[s-1] until its state meets the stored one (from there on the
// stored results are exact too) or the segment ends (then out[s] is replaced and the next
// comparison fails in turn).  The result is the serial result bit for bit for ANY input — a chain
// that never forgets merely degrades to the serial walk — while typical content needs WARM + w/SEG
// steps instead of w (1080p: 22 + 44 us instead of 99 + 200).
constexpr int MAX_SEG = 16;      // vertical chains: 64 columns x 16 segments = one workgroup
constexpr int H_SEG = 32, H_ROWS = 256 / H_SEG;   // horizontal chains: 8 rows x 32 segments

struct HqSeg { int seg, len, warm; };      // segments per chain, samples per segment, warm-up samples

// horizontal recurrence: a workgroup of 256 threads = H_ROWS rows x up to H_SEG segments
template <typename PIX, int SH>
__global__ __launch_bounds__(256) void hqdn3d_h_kernel(HqArgs a, HqSeg g)
{
    const HqPlane &P = a.pl[blockIdx.y];
    if (!P.spatial_on) return;
    __shared__ int16_t lut[LUT_N];
    __shared__ uint32_t s_in[H_ROWS][H_SEG], s_out[H_ROWS][H_SEG];
    stage_lut(lut, P.spatial, 256);
    __syncthreads();
    const int seg = threadIdx.x % H_SEG, rl = threadIdx.x / H_SEG;
    const int y = blockIdx.x * H_ROWS + rl;
    const int w = P.w;
    const bool live = y < P.h && seg < g.seg && seg * g.len < w;
    const uint8_t *s = P.src + (size_t)(live ? y : 0) * P.spitch;
    uint16_t *o = P.hbuf + (size_t)(live ? y : 0) * w;
    if (live)
    {
        const int x0 = seg * g.len, x1 = min(w, x0 + g.len);
        // warm-up samples [xb, x0) run through the same loops as the kept ones, with the stores off
        const int xb = seg == 0 ? 0 : max(0, x0 - g.warm);
        uint32_t run = hq_load<PIX, SH>(s, xb);
        if (seg == 0)
        {
            if (y == 0) run = lowpass((int)run, (int)hq_load<PIX, SH>(s, 0), lut);      // row 0 quirk (:140-146)
            o[0] = (uint16_t)run;
            s_in[rl][0] = 0;
        }
        else if (xb == x0) s_in[rl][seg] = run;                                         // no warm-up at all
        // with no warm-up at all the start sample is both the assumed state and the first sample
        int x = (seg != 0 && xb == x0) ? x0 : xb + 1;
        constexpr int NS = 16 / (int)sizeof(PIX);              // samples per 16-byte load
        for (; x < x1 && (x & (NS - 1)); x++)
        {
            if (x == x0) s_in[rl][seg] = run;
            run = lowpass((int)run, (int)hq_load<PIX, SH>(s, x), lut);
            if (x >= x0) o[x] = (uint16_t)run;
        }
        // one 16-byte load per NS samples, the next load in flight while the chain works through this one
        const PIX *sp = reinterpret_cast<const PIX *>(s);
        uint4 nxt = x + NS <= x1 ? *reinterpret_cast<const uint4 *>(sp + x) : make_uint4(0, 0, 0, 0);
        for (; x + NS <= x1; x += NS)
        {
            const uint4 v = nxt;
            if (x + 2 * NS <= x1) nxt = *reinterpret_cast<const uint4 *>(sp + x + NS);
            const uint32_t wds[4] = { v.x, v.y, v.z, v.w };
            uint16_t r[NS];
            if (x == x0) s_in[rl][seg] = run;                  // x0 is a multiple of 16: a group is kept or dropped whole
#pragma unroll
            for (int k = 0; k < NS; k++)
            {
                const uint32_t px = sizeof(PIX) == 1 ? (wds[k >> 2] >> (8 * (k & 3))) & 0xffu : (wds[k >> 1] >> (16 * (k & 1))) & 0xffffu;
                run = lowpass((int)run, (int)load_sh<SH>(px), lut);
                r[k] = (uint16_t)run;
            }
            if (x < x0) continue;
            if (((uintptr_t)(o + x) & 15) == 0)
            {
#pragma unroll
                for (int q = 0; q < NS / 8; q++)
                {
                    uint4 pk;
                    pk.x = r[8 * q + 0] | (r[8 * q + 1] << 16); pk.y = r[8 * q + 2] | (r[8 * q + 3] << 16);
                    pk.z = r[8 * q + 4] | (r[8 * q + 5] << 16); pk.w = r[8 * q + 6] | (r[8 * q + 7] << 16);
                    reinterpret_cast<uint4 *>(o + x)[q] = pk;
                }
            }
            else
            {
#pragma unroll
                for (int k = 0; k < NS; k++) o[x + k] = r[k];
            }
        }
        for (; x < x1; x++)
        {
            if (x == x0) s_in[rl][seg] = run;
            run = lowpass((int)run, (int)hq_load<PIX, SH>(s, x), lut);
            if (x >= x0) o[x] = (uint16_t)run;
        }
        s_out[rl][seg] = run;
    }
    __threadfence_block();
    __syncthreads();
    // verification / repair: one thread per row, segments in order
    if (seg != 0 || y >= P.h) return;
    uint32_t state = s_out[rl][0];
    for (int k = 1; k < g.seg && k * g.len < w; k++)
    {
        const int x0 = k * g.len, x1 = min(w, x0 + g.len);
        if (s_in[rl][k] == state) { state = s_out[rl][k]; continue; }
        uint32_t run = state;
        bool met = false;
        for (int x = x0; x < x1; x++)
        {
            run = lowpass((int)run, (int)hq_load<PIX, SH>(s, x), lut);
            if (run == (uint32_t)o[x] && x + 1 < x1) { met = true; break; }   // states are < 65536 (between sample and state)
            o[x] = (uint16_t)run;
        }
        state = met ? s_out[rl][k] : run;
    }
}

// vertical recurrence + temporal step: a workgroup = 64 columns x up to 16 segments of rows.
// `ant` (the previous frame's temporal state) is read-only here and the new state goes to `ant_out`:
// a repaired sample needs the ORIGINAL previous state.
template <typename PIX, int SH>
__global__ __launch_bounds__(1024) void hqdn3d_vt_kernel(HqArgs a, HqSeg g)
{
    const HqPlane &P = a.pl[blockIdx.y];
    if (!P.spatial_on) return;
    __shared__ int16_t lut_s[LUT_N];
    __shared__ int16_t lut_t[LUT_N];
    __shared__ uint32_t s_in[MAX_SEG][64], s_out[MAX_SEG][64];
    stage_lut(lut_s, P.spatial, blockDim.x);
    stage_lut(lut_t, P.temporal, blockDim.x);
    __syncthreads();
    const int xl = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + xl;
    const int w = P.w, h = P.h;
    const bool live = x < w && seg < g.seg && seg * g.len < h;
    auto prev_state = [&](int y) -> uint32_t {
        return P.seeded ? (uint32_t)P.ant[(size_t)y * w + x] : (uint32_t)(uint16_t)hq_load<PIX, SH>(P.src + (size_t)y * P.spitch, x);
    };
    auto emit = [&](int y, uint32_t v, uint32_t pv) {
        const uint32_t t = lowpass((int)pv, (int)v, lut_t);
        P.vstate[(size_t)y * w + x] = (uint16_t)v;
        P.ant_out[(size_t)y * w + x] = (uint16_t)t;
        reinterpret_cast<PIX *>(P.dst + (size_t)y * P.dpitch)[x] = (PIX)(t >> SH);
    };
    if (live)
    {
        const int y0 = seg * g.len, y1 = min(h, y0 + g.len);
        // warm-up rows [ys, y0) go through the same pipelined loop as the kept ones, with the outputs off
        const int ys = seg == 0 ? 0 : max(0, y0 - g.warm);
        uint32_t line = P.hbuf[(size_t)ys * w + x];                        // a chain starts from the h-filtered sample (:167-176)
        if (seg == 0)
        {
            emit(0, line, prev_state(0));
            s_in[0][xl] = 0;
        }
        else if (ys == y0) s_in[seg][xl] = (uint16_t)line;
        // the loads of a row do not depend on the recurrence: keep U rows in flight ahead of the chain
        constexpr int U = 8;
        const int yb = (seg != 0 && ys == y0) ? y0 : ys + 1;       // no warm-up: the start row is state and first sample
        uint32_t hv[U], pv[U];
#pragma unroll
        for (int k = 0; k < U; k++)
        {
            const int y = min(yb + k, h - 1);
            hv[k] = P.hbuf[(size_t)y * w + x];
            pv[k] = y >= y0 ? prev_state(y) : 0u;
        }
        for (int yq = yb; yq < y1; yq += U)
        {
            uint32_t hn[U], pn[U];
#pragma unroll
            for (int k = 0; k < U; k++)
            {
                const int y = min(yq + U + k, h - 1);
                hn[k] = P.hbuf[(size_t)y * w + x];
                pn[k] = y >= y0 ? prev_state(y) : 0u;
            }
#pragma unroll
            for (int k = 0; k < U; k++)
            {
                const int y = yq + k;
                if (y < y1)
                {
                    if (y == y0) s_in[seg][xl] = (uint16_t)line;
                    line = lowpass((int)(uint16_t)line, (int)hv[k], lut_s);
                    if (y >= y0) emit(y, line, pv[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < U; k++) { hv[k] = hn[k]; pv[k] = pn[k]; }
        }
        s_out[seg][xl] = (uint16_t)line;
    }
    __threadfence_block();
    __syncthreads();
    if (seg != 0 || x >= w) return;
    uint32_t state = s_out[0][xl];
    for (int k = 1; k < g.seg && k * g.len < h; k++)
    {
        const int y0 = k * g.len, y1 = min(h, y0 + g.len);
        if (s_in[k][xl] == state) { state = s_out[k][xl]; continue; }
        uint32_t line = state;
        bool met = false;
        for (int y = y0; y < y1; y++)
        {
            line = lowpass((int)(uint16_t)line, (int)P.hbuf[(size_t)y * w + x], lut_s);
            // the spatial state itself is not stored; two runs agree from the first row where their
            // states agree, and the stored temporal output is a function of (state, previous state)
            if ((uint16_t)line == P.vstate[(size_t)y * w + x] && y + 1 < y1) { met = true; break; }
            emit(y, line, prev_state(y));
        }
        state = met ? s_out[k][xl] : (uint32_t)(uint16_t)line;
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
        for (int c = 0; c < 3; c++) if (ant2[c]) (void)hipFree(ant2[c]);
        for (int c = 0; c < 3; c++) if (vstate[c]) (void)hipFree(vstate[c]);
    }
    int setup()
    {
        HBHIP_CHECK(ctx, hipMalloc((void **)&d_coef, sizeof(int16_t) * 6 * LUT_N));
        HBHIP_CHECK(ctx, hipMemcpyAsync(d_coef, par.coef, sizeof(int16_t) * 6 * LUT_N, hipMemcpyHostToDevice, ctx->stream));
        for (int c = 0; c < 3; c++)
            HBHIP_CHECK(ctx, hipMalloc((void **)&ant[c], sizeof(uint16_t) * (size_t)in_geo.pw[c] * in_geo.ph[c]));
        for (int c = 0; c < 3; c++)
        {
            const size_t n = sizeof(uint16_t) * (size_t)in_geo.pw[c] * in_geo.ph[c];
            HBHIP_CHECK(ctx, hipMalloc((void **)&hbuf[c], n));
            HBHIP_CHECK(ctx, hipMalloc((void **)&ant2[c], n));
            HBHIP_CHECK(ctx, hipMalloc((void **)&vstate[c], n));
        }
        if (const char *e = getenv("HBHIP_HQDN3D_WARMUP")) warm = std::max(0, atoi(e));     // test hook: 0 forces repairs
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
            P.hbuf = hbuf[c]; P.ant = ant[c]; P.ant_out = ant2[c]; P.vstate = vstate[c];
            P.spatial = d_coef + (size_t)(2 * c) * LUT_N; P.temporal = P.spatial + LUT_N;
            P.seeded = seeded[c];
            P.spatial_on = par.coef[2 * c][0] != 0;                  // spatial strength != 0 (denoise.c:191)
            (P.spatial_on ? any_spatial : any_temporal) = true;
            max_w = std::max(max_w, P.w); max_h = std::max(max_h, P.h);
            seeded[c] = 1;
            if (P.spatial_on) std::swap(ant[c], ant2[c]);            // the state this frame writes is the next frame's `ant`
        }
        // segments of the two recurrences (see the kernels): about 64 samples each
        auto cut = [&](int n, int target, int align, int most) {
            HqSeg g;
            g.seg = std::min(most, std::max(1, (n + target - 1) / target));
            g.len = ((n + g.seg - 1) / g.seg + align - 1) / align * align;
            g.warm = warm;
            return g;
        };
        const HqSeg gh = cut(max_w, 64, 16, H_SEG), gv = cut(max_h, 64, 1, MAX_SEG);
#define HQ_GO(PIX, SH) do { \
            if (any_spatial) \
            { \
                HBHIP_LAUNCH(ctx, "hqdn3d_h", (hqdn3d_h_kernel<PIX, SH>), dim3((max_h + H_ROWS - 1) / H_ROWS, 3), dim3(256), 0, a, gh); \
                HBHIP_LAUNCH(ctx, "hqdn3d_vt", (hqdn3d_vt_kernel<PIX, SH>), dim3((max_w + 63) / 64, 3), dim3(64 * gv.seg), 0, a, gv); \
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
    uint16_t *ant2[3] = {nullptr, nullptr, nullptr};   // the other half of the temporal-state double buffer
    uint16_t *vstate[3] = {nullptr, nullptr, nullptr}; // vertical spatial state per sample
    int warm = 64;                                     // warm-up samples of a speculative segment
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
