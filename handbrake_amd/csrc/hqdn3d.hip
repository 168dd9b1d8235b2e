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

// the 16 KB table into LDS: 16 bytes per load, and a thread's loads all in flight before its first LDS store (as a plain
// copy loop a workgroup of 256 spent sixteen dependent round trips to L2 here - most of a launch's time)
__device__ __forceinline__ void stage_lut(int16_t *dst, const int16_t *src, int nthreads)
{
    constexpr int NQ = LUT_N * 2 / 16;                 // 1024 uint4
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
    for (int base = threadIdx.x; base < NQ; base += 4 * nthreads)      // one round for workgroups of 256 threads and more
    {
        // (four named registers, not an array: conditionally written arrays go to scratch memory)
        const int i0 = base, i1 = base + nthreads, i2 = base + 2 * nthreads, i3 = base + 3 * nthreads;
        uint4 a = make_uint4(0, 0, 0, 0), b = a, c = a, d = a;
        if (i0 < NQ) a = s4[i0];
        if (i1 < NQ) b = s4[i1];
        if (i2 < NQ) c = s4[i2];
        if (i3 < NQ) d = s4[i3];
        if (i0 < NQ) d4[i0] = a;
        if (i1 < NQ) d4[i1] = b;
        if (i2 < NQ) d4[i2] = c;
        if (i3 < NQ) d4[i3] = d;
    }
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
// `groups`: row groups (of H_ROWS rows) a workgroup works off one after the other on the table it staged once
// HSEG: the segments a row is cut into at most (a power of two) = the lanes a row takes; 256 / HSEG rows per group
template <typename PIX, int SH, int HSEG>
__device__ __forceinline__ void hq_h_rows(const HqPlane &P, const HqSeg &g, int groups)
{
    constexpr int H_SEG = HSEG, H_ROWS = 256 / HSEG;
    if (!P.spatial_on || (int)blockIdx.x * groups * H_ROWS >= P.h) return;   // (the grid is sized for the luma plane)
    __shared__ __attribute__((aligned(16))) int16_t lut[LUT_N];
    __shared__ uint32_t s_in[H_ROWS][H_SEG], s_out[H_ROWS][H_SEG];
    stage_lut(lut, P.spatial, 256);
    __syncthreads();
    const int seg = threadIdx.x % H_SEG, rl = threadIdx.x / H_SEG;
    const int w = P.w;
    for (int grp = 0; grp < groups; grp++)
    {
    const int y = ((int)blockIdx.x * groups + grp) * H_ROWS + rl;
    if (y - rl >= P.h) break;                                              // (uniform)
    if (grp) __syncthreads();                                              // the group before is done with s_in / s_out
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
        // NS samples of one 16-byte load: the chain through them and, from the segment's own first sample on, their results
        const PIX *sp = reinterpret_cast<const PIX *>(s);
        // (results as packed pairs in dwords: arrays of halfwords end up in scratch memory)
        auto chain16 = [&](const uint4 &v, int xx, uint32_t (&rp)[NS / 2]) {
            const uint32_t wds[4] = { v.x, v.y, v.z, v.w };
            if (xx == x0) s_in[rl][seg] = run;                 // x0 is a multiple of 16: a block is kept or dropped whole
#pragma unroll
            for (int k = 0; k < NS; k += 2)
            {
                const uint32_t p0 = sizeof(PIX) == 1 ? (wds[k >> 2] >> (8 * (k & 3))) & 0xffu : wds[k >> 1] & 0xffffu;
                const uint32_t p1 = sizeof(PIX) == 1 ? (wds[k >> 2] >> (8 * ((k + 1) & 3))) & 0xffu : wds[k >> 1] >> 16;
                run = lowpass((int)run, (int)load_sh<SH>(p0), lut);
                const uint32_t lo = run & 0xffffu;
                run = lowpass((int)run, (int)load_sh<SH>(p1), lut);
                rp[k >> 1] = lo | (run << 16);
            }
        };
        auto store16 = [&](int xx, const uint32_t (&rp)[NS / 2]) {
            if (xx < x0) return;
            if (((uintptr_t)(o + xx) & 15) == 0)
            {
#pragma unroll
                for (int q = 0; q < NS / 8; q++)
                    reinterpret_cast<uint4 *>(o + xx)[q] = make_uint4(rp[4 * q], rp[4 * q + 1], rp[4 * q + 2], rp[4 * q + 3]);
            }
            else
            {
#pragma unroll
                for (int k = 0; k < NS / 2; k++) { o[xx + 2 * k] = (uint16_t)rp[k]; o[xx + 2 * k + 1] = (uint16_t)(rp[k] >> 16); }
            }
        };
        auto block = [&](const uint4 &v, int xx) {
            uint32_t rp[NS / 2];
            chain16(v, xx, rp);
            store16(xx, rp);
        };
        // Four loads = 64 bytes of the row at a time, and the 128 bytes of results stored together.  A lane reads its own
        // stretch of the row, 64 lanes 64 different cache lines per load instruction: taken 16 bytes at a time with sixteen
        // chain steps in between, a line is gone from the 16 KB L1 of a CU with 32 such waves before its next 16 bytes are
        // asked for; asked for together they are one line fetch (103 -> 95 us per 16 frames; the pass stays the slowest of
        // the three at 77 us - every memory instruction of a wave still touches 64 lines).
        constexpr int GL = 4;
        for (; x + GL * NS <= x1; x += GL * NS)
        {
            uint4 v[GL];
#pragma unroll
            for (int g4 = 0; g4 < GL; g4++) v[g4] = *reinterpret_cast<const uint4 *>(sp + x + g4 * NS);
            // ... and the results leave together too: 128 bytes of the lane's own line in one burst
            uint32_t rp[GL][NS / 2];
#pragma unroll
            for (int g4 = 0; g4 < GL; g4++) chain16(v[g4], x + g4 * NS, rp[g4]);
#pragma unroll
            for (int g4 = 0; g4 < GL; g4++) store16(x + g4 * NS, rp[g4]);
        }
        for (; x + NS <= x1; x += NS) block(*reinterpret_cast<const uint4 *>(sp + x), x);
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
    if (seg != 0 || y >= P.h) continue;
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
}

template <typename PIX, int SH>
__global__ __launch_bounds__(256) void hqdn3d_h_kernel(HqArgs a, HqSeg g)
{
    hq_h_rows<PIX, SH, H_SEG>(a.pl[blockIdx.y], g, 1);
}

// Several frames per launch.  The two spatial recurrences of a frame do not look at any other frame, so the frames of
// a batch go through them side by side (blockIdx.z = frame): sixteen times the workgroups of a launch whose time is
// the length of its longest chain, not its sample count.  Only the temporal step ties frame t to frame t - 1, and it
// is pointwise: hqdn3d_tn_kernel walks the frames of the batch per sample.  hbuf / vsp: the h-filtered samples and
// the vertical spatial results of every frame of the batch (frame f of plane c at + f * stride[c]).
constexpr int HQ_BATCH = 16;
// a workgroup of the batched horizontal pass takes three row groups: a 1080p batch is then 1 450 workgroups, all
// resident at once (eight to a CU), instead of 4 300 in two and a bit rounds of which the last runs nearly empty
// A batch has rows in plenty, so its chains are cut into FEWER and LONGER segments: a segment of 240 samples with its
// 64 samples of warm-up does 1.27 x the serial work where one of 64 does 2 x - and with eight waves to a SIMD the
// recurrences are bound by instruction issue, i.e. by the work (SQ_WAIT_ANY 68 % of the wave cycles with the short
// segments, but 22 M vector instructions where the serial walk has 11 M).
constexpr int HB_SEG = 8, HB_ROWS = 256 / HB_SEG, VB_SEG = 4;
constexpr int H_GROUPS = 1;
struct HqBatch
{
    const uint8_t *src[HQ_BATCH][3];
    uint8_t       *dst[HQ_BATCH][3];
    uint16_t      *hbuf[3], *vsp[3], *ant[3];
    size_t         stride[3];
    const int16_t *spatial[3], *temporal[3];
    int spitch[3], dpitch[3], w[3], h[3], seeded[3], spatial_on[3];
    int n;
};

__device__ __forceinline__ HqPlane hq_batch_plane(const HqBatch &B, int f, int c)
{
    HqPlane P;
    P.src = B.src[f][c]; P.dst = B.dst[f][c];
    P.hbuf = B.hbuf[c] + (size_t)f * B.stride[c];
    P.vstate = B.vsp[c] + (size_t)f * B.stride[c];
    P.ant = P.ant_out = nullptr;
    P.spatial = B.spatial[c]; P.temporal = B.temporal[c];
    P.spitch = B.spitch[c]; P.dpitch = B.dpitch[c]; P.w = B.w[c]; P.h = B.h[c];
    P.seeded = 0; P.spatial_on = B.spatial_on[c];
    return P;
}

template <typename PIX, int SH>
__global__ __launch_bounds__(256) void hqdn3d_h_batch_kernel(HqBatch B, HqSeg g)
{
    hq_h_rows<PIX, SH, HB_SEG>(hq_batch_plane(B, blockIdx.z, blockIdx.y), g, H_GROUPS);
}

// vertical recurrence + temporal step: a workgroup = 64 columns x up to 16 segments of rows.
// `ant` (the previous frame's temporal state) is read-only here and the new state goes to `ant_out`:
// a repaired sample needs the ORIGINAL previous state.
// TEMPORAL false (the batched form): the vertical results go to `vstate` and that is all - no previous state is read,
// nothing else written; the temporal step follows in hqdn3d_tn_kernel.
template <typename PIX, int SH, bool TEMPORAL>
__device__ __forceinline__ void hq_v_cols(const HqPlane &P, const HqSeg &g)
{
    if (!P.spatial_on || (int)blockIdx.x * 64 >= P.w) return;              // (the grid is sized for the luma plane)
    __shared__ __attribute__((aligned(16))) int16_t lut_s[LUT_N];
    __shared__ __attribute__((aligned(16))) int16_t lut_t[LUT_N];
    __shared__ uint32_t s_in[MAX_SEG][64], s_out[MAX_SEG][64];
    stage_lut(lut_s, P.spatial, blockDim.x);
    if (TEMPORAL) stage_lut(lut_t, P.temporal, blockDim.x);
    __syncthreads();
    const int xl = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + xl;
    const int w = P.w, h = P.h;
    const bool live = x < w && seg < g.seg && seg * g.len < h;
    auto prev_state = [&](int y) -> uint32_t {
        if (!TEMPORAL) return 0u;
        return P.seeded ? (uint32_t)P.ant[(size_t)y * w + x] : (uint32_t)(uint16_t)hq_load<PIX, SH>(P.src + (size_t)y * P.spitch, x);
    };
    auto emit = [&](int y, uint32_t v, uint32_t pv) {
        P.vstate[(size_t)y * w + x] = (uint16_t)v;
        if (!TEMPORAL) return;
        const uint32_t t = lowpass((int)pv, (int)v, lut_t);
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

template <typename PIX, int SH>
__global__ __launch_bounds__(1024) void hqdn3d_vt_kernel(HqArgs a, HqSeg g)
{
    hq_v_cols<PIX, SH, true>(a.pl[blockIdx.y], g);
}

template <typename PIX, int SH>
__global__ __launch_bounds__(1024) void hqdn3d_v_batch_kernel(HqBatch B, HqSeg g)
{
    hq_v_cols<PIX, SH, false>(hq_batch_plane(B, blockIdx.z, blockIdx.y), g);
}

// The temporal step of a batch (denoise.c:102-124 and the tail of :167-201): per sample, frame after frame.  The
// current value is the frame's vertical spatial result (vsp) or, for a plane without a spatial filter, its input
// sample; the previous one is the state left by the frame before - `ant` for the first frame of the batch, or that
// frame's own input sample when the filter has seen no frame yet (the seeding of hqdn3d_denoise_depth).
// grid: (samples of a row / 1024, rows, planes); a thread takes the samples x0 + tid + 256 k, k = 0 .. 3 (a wave's 64
// lanes read 64 consecutive samples), and has the values of ALL frames of the batch in flight before its chains start:
// a chain step is then an LDS look-up and nothing else.
template <typename PIX, int SH>
__global__ __launch_bounds__(256) void hqdn3d_tn_kernel(HqBatch B)
{
    __shared__ __attribute__((aligned(16))) int16_t lut_t[LUT_N];
    const int c = blockIdx.z;
    const int w = B.w[c], y = blockIdx.y;
    const int x0 = 1024 * blockIdx.x + threadIdx.x;
    if (y >= B.h[c] || 1024 * (int)blockIdx.x >= w) return;                // (the grid is sized for the luma plane)
    const bool spatial = B.spatial_on[c] != 0;
    const size_t at = (size_t)y * w + x0;
    uint32_t cur[HQ_BATCH][4];
#pragma unroll
    for (int f = 0; f < HQ_BATCH; f++)
    {
        if (f >= B.n) continue;                                            // (uniform; no break: the loop must unroll, cur[][] live in registers)
        if (spatial)
        {
            const uint16_t *v = B.vsp[c] + (size_t)f * B.stride[c] + at;
#pragma unroll
            for (int k = 0; k < 4; k++) cur[f][k] = x0 + 256 * k < w ? v[256 * k] : 0u;
        }
        else
        {
            const uint8_t *row = B.src[f][c] + (size_t)y * B.spitch[c];
#pragma unroll
            for (int k = 0; k < 4; k++) cur[f][k] = x0 + 256 * k < w ? hq_load<PIX, SH>(row, x0 + 256 * k) : 0u;
        }
    }
    uint32_t prev[4];
    if (B.seeded[c])
    {
#pragma unroll
        for (int k = 0; k < 4; k++) prev[k] = x0 + 256 * k < w ? B.ant[c][at + 256 * k] : 0u;
    }
    else
    {
        const uint8_t *row = B.src[0][c] + (size_t)y * B.spitch[c];
#pragma unroll
        for (int k = 0; k < 4; k++) prev[k] = x0 + 256 * k < w ? (uint32_t)(uint16_t)hq_load<PIX, SH>(row, x0 + 256 * k) : 0u;
    }
    stage_lut(lut_t, B.temporal[c], 256);
    __syncthreads();
#pragma unroll
    for (int f = 0; f < HQ_BATCH; f++)
    {
        if (f >= B.n) continue;
        PIX *d = reinterpret_cast<PIX *>(B.dst[f][c] + (size_t)y * B.dpitch[c]) + x0;
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            prev[k] = (uint32_t)(uint16_t)lowpass((int)prev[k], (int)cur[f][k], lut_t);
            if (x0 + 256 * k < w) d[256 * k] = (PIX)(prev[k] >> SH);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (x0 + 256 * k < w) B.ant[c][at + 256 * k] = (uint16_t)prev[k];
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
        for (int c = 0; c < 3; c++) if (hbuf_n[c]) (void)hipFree(hbuf_n[c]);
        for (int c = 0; c < 3; c++) if (vsp_n[c]) (void)hipFree(vsp_n[c]);
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
    // Several frames: the spatial passes of all of them in one launch each, then the temporal step frame after frame
    // per sample (see HqBatch).  The same numbers as frame-by-frame calls: the spatial passes never looked at another
    // frame, and the temporal chain is walked in the same order.
    int process_many(DevPicture *const *ins, DevPicture *const *outs, int n) override
    {
        if (n < 2) return SimpleFilter::process_many(ins, outs, n);
        for (int c = 0; c < 3 && !hbuf_n[c]; c++)
        {
            const size_t bytes = sizeof(uint16_t) * (size_t)in_geo.pw[c] * in_geo.ph[c] * HQ_BATCH;
            HBHIP_CHECK(ctx, hipMalloc((void **)&hbuf_n[c], bytes));
            HBHIP_CHECK(ctx, hipMalloc((void **)&vsp_n[c], bytes));
        }
        for (int at = 0; at < n; at += HQ_BATCH)
        {
            const int k = std::min(HQ_BATCH, n - at);
            HqBatch B;
            memset(&B, 0, sizeof(B));
            B.n = k;
            bool any_spatial = false, same = true;
            int max_w = 0, max_h = 0;
            for (int c = 0; c < 3; c++)
            {
                B.hbuf[c] = hbuf_n[c]; B.vsp[c] = vsp_n[c]; B.ant[c] = ant[c];
                B.stride[c] = (size_t)in_geo.pw[c] * in_geo.ph[c];
                B.spatial[c] = d_coef + (size_t)(2 * c) * LUT_N; B.temporal[c] = B.spatial[c] + LUT_N;
                B.spitch[c] = ins[at]->pitch[c]; B.dpitch[c] = outs[at]->pitch[c];
                B.w[c] = ins[at]->width[c]; B.h[c] = ins[at]->height[c];
                B.seeded[c] = seeded[c];
                B.spatial_on[c] = par.coef[2 * c][0] != 0;                  // spatial strength != 0 (denoise.c:191)
                any_spatial |= B.spatial_on[c] != 0;
                max_w = std::max(max_w, B.w[c]); max_h = std::max(max_h, B.h[c]);
                for (int f = 0; f < k; f++)
                {
                    B.src[f][c] = ins[at + f]->plane[c]; B.dst[f][c] = outs[at + f]->plane[c];
                    same &= ins[at + f]->pitch[c] == B.spitch[c] && outs[at + f]->pitch[c] == B.dpitch[c];
                }
            }
            if (!same)
            {
                int rc = SimpleFilter::process_many(ins + at, outs + at, k);
                if (rc != HBHIP_OK) return rc;
                continue;
            }
            auto cut = [&](int len, int target, int align, int most) {
                HqSeg g;
                g.seg = std::min(most, std::max(1, (len + target - 1) / target));
                g.len = ((len + g.seg - 1) / g.seg + align - 1) / align * align;
                g.warm = warm;
                return g;
            };
            const HqSeg gh = cut(max_w, (max_w + HB_SEG - 1) / HB_SEG, 16, HB_SEG), gv = cut(max_h, (max_h + VB_SEG - 1) / VB_SEG, 1, VB_SEG);
#define HQ_GO_N(PIX, SH) do { \
                if (any_spatial) \
                { \
                    HBHIP_LAUNCH(ctx, "hqdn3d_h", (hqdn3d_h_batch_kernel<PIX, SH>), dim3((max_h + HB_ROWS * H_GROUPS - 1) / (HB_ROWS * H_GROUPS), 3, k), dim3(256), 0, B, gh); \
                    HBHIP_LAUNCH(ctx, "hqdn3d_v", (hqdn3d_v_batch_kernel<PIX, SH>), dim3((max_w + 63) / 64, 3, k), dim3(64 * gv.seg), 0, B, gv); \
                } \
                HBHIP_LAUNCH(ctx, "hqdn3d_t", (hqdn3d_tn_kernel<PIX, SH>), dim3(hbhip_grid_x((max_w + 1023) / 1024), max_h, 3), dim3(256), 0, B); \
            } while (0)
            if (in_geo.depth == 8)       HQ_GO_N(uint8_t, 8);
            else if (in_geo.depth == 10) HQ_GO_N(uint16_t, 6);
            else                         HQ_GO_N(uint16_t, 4);
#undef HQ_GO_N
            HBHIP_CHECK(ctx, hipGetLastError());
            for (int c = 0; c < 3; c++) seeded[c] = 1;
        }
        return HBHIP_OK;
    }
    hbhip_hqdn3d_params par;
    uint16_t *hbuf_n[3] = {nullptr, nullptr, nullptr};   // batches: the h-filtered rows of HQ_BATCH frames per plane
    uint16_t *vsp_n[3] = {nullptr, nullptr, nullptr};    //          their vertical spatial results
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
