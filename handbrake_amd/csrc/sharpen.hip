// sharpen.hip — lapsharp, unsharp and chroma-smooth for gfx950 (8-bit).
//
//   lapsharp_kernel<S>   replaces lapsharp_8        libhb/lapsharp.c:125-182
//   blur_mix_kernel      replaces unsharp_8         libhb/unsharp.c:89-173
//                        and      chroma_smooth_8   libhb/chroma_smooth.c:87-172
//
// All three are one read + one write per pixel (HBM-bound, 2 B/pixel
// algorithmic); the neighbourhood reuse is served by LDS (blur) or L1/L2
// (3x3 / 5x5 Laplacian rows read as aligned dwords).  Results are bit-exact:
// lapsharp replays the reference's int16 accumulate and double mix; the
// binomial blur is exact in uint32 modular arithmetic (the reference's running
// pair sums are the same sums, unsharp.c:128-156).
#include <map>
#include <mutex>
#include <utility>
#include "hbhip_internal.h"

namespace {

// ------------------------------------------------------------------ lapsharp
struct LapKernel { int tap[25]; int size; double coef; };

constexpr LapKernel LAP_TABLE[4] = {
    { { 0, -1, 0, -1, 5, -1, 0, -1, 0 }, 3, 1.0 },                                      // lap     lapsharp.c:37-42
    { { -1, -4, -1, -4, 25, -4, -1, -4, -1 }, 3, 1.0 / 5 },                             // isolap  :47-52
    { { 0, 0, -1, 0, 0, 0, -1, -2, -1, 0, -1, -2, 21, -2, -1, 0, -1, -2, -1, 0, 0, 0, -1, 0, 0 }, 5, 1.0 / 5 },   // log :58-65
    { { 0, -1, -1, -1, 0, -1, -3, -4, -3, -1, -1, -4, 55, -4, -1, -1, -3, -4, -3, -1, 0, -1, -1, -1, 0 }, 5, 1.0 / 15 } }; // isolog :71-78

struct LapArgs
{
    const uint8_t *src;
    uint8_t       *dst;
    int            width, height, src_pitch, dst_pitch;
    int            stride_border;     // (caller stride - width) / 2, lapsharp.c:145
    int            valid_w;           // bytes at x >= valid_w read as 0 (device-resident input has no row padding)
    int            tap[25];
    double         coef, strength;
};

// the planes of a frame that share a kernel size go in one launch (blockIdx.z)
struct LapArgs3 { LapArgs p[3]; };

__device__ __forceinline__ uint32_t byte_at(const uint32_t (&v)[3], int k)   // k in [-4, 7]
{
    const int kk = k + 4;
    return (v[kk >> 2] >> (8 * (kk & 3))) & 0xffu;
}

// one thread = 4 horizontally adjacent pixels (one aligned dword of output)
template <int S>
__global__ __launch_bounds__(256) void lapsharp_kernel(LapArgs3 all)
{
    const LapArgs &a = all.p[blockIdx.z];
    constexpr int LO = -((S - 1) / 2), HI = (S + 1) / 2;
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y;
    if (x0 >= a.width || y >= a.height) return;

    const int pitch_dw = a.src_pitch >> 2;
    const int xd = x0 >> 2;
    uint32_t rows[S][3];
#pragma unroll
    for (int j = 0; j < S; j++)
    {
        int yy = y + LO + j;
        yy = min(max(yy, 0), a.height - 1);                  // only read for pixels that end up copied
        const uint32_t *r = reinterpret_cast<const uint32_t *>(a.src + (size_t)yy * a.src_pitch);
        rows[j][0] = r[max(xd - 1, 0)];
        rows[j][1] = r[xd];
        rows[j][2] = r[min(xd + 1, pitch_dw - 1)];
        if (x0 + 8 > a.valid_w)
        {
            // keep bytes [x0-4, x0+8) that lie below valid_w
#pragma unroll
            for (int k = 0; k < 3; k++)
            {
                const int base = x0 - 4 + 4 * k;
                const int keep = a.valid_w - base;            // number of valid low bytes in this dword
                if (keep <= 0) rows[j][k] = 0;
                else if (keep < 4) rows[j][k] &= (1u << (8 * keep)) - 1u;
            }
        }
    }

    uint32_t packed = 0;
#pragma unroll
    for (int p = 0; p < 4; p++)
    {
        const int x = x0 + p;
        const int centre = (int)byte_at(rows[-LO], p);
        int out = centre;
        const bool copy = (y < HI) || (y > a.height - HI) || (x < a.stride_border + HI) ||
                          (x > a.width + a.stride_border - HI);
        if (!copy)
        {
            int acc = 0;                                     // fits int16 for every table (lapsharp.c:148)
#pragma unroll
            for (int k = 0; k < S; k++)
#pragma unroll
                for (int j = 0; j < S; j++)
                    acc += a.tap[j * S + k] * (int)byte_at(rows[j], p + LO + k);
            const double mixed = (((double)acc * a.coef) - (double)centre) * a.strength;   // :174-175
            out = (int)(short)(int)mixed + centre;
            out = out < 0 ? 0 : out;
            out = out > 255 ? 255 : out;
        }
        packed |= (uint32_t)out << (8 * p);
    }
    uint8_t *d = a.dst + (size_t)y * a.dst_pitch + x0;
    if (x0 + 3 < a.width)
        *reinterpret_cast<uint32_t *>(d) = packed;
    else
        for (int p = 0; x0 + p < a.width; p++) d[p] = (uint8_t)(packed >> (8 * p));
}

// lapsharp_16 (DEF_LAPSHARP_FUNC(lapsharp, 16, 32), lapsharp.c:184): 16-bit samples, int32
// accumulator, clamp to (1 << depth) - 1.  One thread = 2 adjacent pixels (one dword);
// width / stride_border are in samples, pitches in bytes.
template <int S>
__global__ __launch_bounds__(256) void lapsharp16_kernel(LapArgs3 all, int max_value)
{
    const LapArgs &a = all.p[blockIdx.z];
    constexpr int LO = -((S - 1) / 2), HI = (S + 1) / 2;
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    const int y = blockIdx.y;
    if (x0 >= a.width || y >= a.height) return;
    const int pitch_dw = a.src_pitch >> 2;
    const int xd = x0 >> 1;
    uint32_t rows[S][3];                                     // samples x0-2 .. x0+3 of each row
#pragma unroll
    for (int j = 0; j < S; j++)
    {
        int yy = y + LO + j;
        yy = min(max(yy, 0), a.height - 1);
        const uint32_t *r = reinterpret_cast<const uint32_t *>(a.src + (size_t)yy * a.src_pitch);
        rows[j][0] = r[max(xd - 1, 0)];
        rows[j][1] = r[xd];
        rows[j][2] = r[min(xd + 1, pitch_dw - 1)];
        if (x0 + 4 > a.width)
        {
            // Right of the plane the reference reads its stride padding, which lapsharp has just
            // filled with hb_frame_buffer_mirror_stride (lapsharp.c:333 -> fifo.c:906-932; the 16-bit
            // variant is the one that really mirrors): sample w + i = sample w - 1 - i.
            const uint16_t *r16 = reinterpret_cast<const uint16_t *>(r);
#pragma unroll
            for (int k = 0; k < 6; k++)
            {
                const int xx = x0 - 2 + k;
                if (xx >= a.width)
                {
                    const uint32_t m = r16[max(2 * a.width - 1 - xx, 0)];
                    uint32_t &d = rows[j][k >> 1];
                    d = (k & 1) ? ((d & 0x0000ffffu) | (m << 16)) : ((d & 0xffff0000u) | m);
                }
            }
        }
    }
    auto at = [&](int j, int k) -> int {                    // sample x0 + k of row j, k in [-2, 3]
        const int kk = k + 2;
        return (int)((rows[j][kk >> 1] >> (16 * (kk & 1))) & 0xffffu);
    };
    uint32_t res[2];
#pragma unroll
    for (int p = 0; p < 2; p++)
    {
        const int x = x0 + p;
        const int centre = at(-LO, p);
        int out = centre;
        const bool copy = (y < HI) || (y > a.height - HI) || (x < a.stride_border + HI) ||
                          (x > a.width + a.stride_border - HI);
        if (!copy)
        {
            int acc = 0;
#pragma unroll
            for (int k = 0; k < S; k++)
#pragma unroll
                for (int j = 0; j < S; j++)
                    acc += a.tap[j * S + k] * at(j, p + LO + k);
            const double mixed = (((double)acc * a.coef) - (double)centre) * a.strength;
            out = (int)mixed + centre;
            out = out < 0 ? 0 : out;
            out = out > max_value ? max_value : out;
        }
        res[p] = (uint32_t)out;
    }
    uint16_t *d = reinterpret_cast<uint16_t *>(a.dst + (size_t)y * a.dst_pitch) + x0;
    if (x0 + 1 < a.width)
        *reinterpret_cast<uint32_t *>(d) = res[0] | (res[1] << 16);
    else
        d[0] = (uint16_t)res[0];
}

// ------------------------------------------------------------------ binomial blur + mix
constexpr int BT_W = 64, BT_H = 16, MAX_STEPS = 7;

struct BlurArgs
{
    const uint8_t *src;
    uint8_t       *dst;
    int            width, height, src_pitch, dst_pitch;
    int            steps, scalebits, halfscale, amount;
    int            sign, vmin, vmax;
    uint32_t       coef[2 * MAX_STEPS + 1];
};

// PIX = uint8_t, or uint16_t for the _16 instantiations (unsharp.c:171, chroma_smooth.c:170);
// pitches stay in bytes
struct BlurArgs3 { BlurArgs p[3]; };

// The blur size is a compile-time constant of the body (S = steps, size = 2 S + 1; the kernel switches on the plane's
// steps, uniform per workgroup): the tap loops unroll, the binomial coefficients sit in SGPRs, a multiply-add of a tap is
// one v_mad_u32_u24 - the sums are the reference's uint32 sums modulo 2^32, and with 8-bit samples every factor stays
// below 2^24 (coefficients <= 3432, horizontal sums <= 255 * 2^14), so the 24-bit multiply (low 32 bits of the full
// product) is exact; 16-bit samples keep the full multiply in the vertical pass (horizontal sums reach 2^30).  A
// thread's four output rows share their 4 + 2 S horizontal sums in registers.  (Sizes 11 - 15 and 10 / 12-bit samples
// run here; 8-bit planes with sizes up to 9 take blur_rows8_kernel below.)
template <typename PIX, int S>
__device__ __forceinline__ void blur_mix_body(const BlurArgs &a, PIX *s_src, uint32_t *s_h)
{
    constexpr int TW = BT_W + 2 * S, TH = BT_H + 2 * S, NT = 2 * S + 1;
    const int x0 = blockIdx.x * BT_W, y0 = blockIdx.y * BT_H;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;

    // edge-clamped tile (unsharp.c:126 clamps x, :117-120 / :166-170 clamp y)
    // (all of a wave's loads are issued before the first one is waited for: the staging is a chain of memory latencies
    // otherwise, one per tile row)
    {
        constexpr int NR = (TH + 3) / 4;
        PIX v0[NR], v1[NR];
        const int xa = min(max(x0 - S + lane, 0), a.width - 1), xb = min(max(x0 - S + lane + 64, 0), a.width - 1);
#pragma unroll
        for (int k = 0; k < NR; k++)
        {
            const int y = min(max(y0 - S + wv + 4 * k, 0), a.height - 1);      // rows past the tile are loaded, not stored
            const PIX *row = reinterpret_cast<const PIX *>(a.src + (uint32_t)__mul24(y, a.src_pitch));
            v0[k] = row[xa];
            v1[k] = row[xb];
        }
#pragma unroll
        for (int k = 0; k < NR; k++)
        {
            const int r = wv + 4 * k;
            if (r < TH)
            {
                s_src[r * TW + lane] = v0[k];
                if (lane + 64 < TW) s_src[r * TW + lane + 64] = v1[k];
            }
        }
    }
    __syncthreads();

    // horizontal binomial sums for every tile row
    for (int r = wv; r < TH; r += 4)
    {
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < NT; k++) sum = __umul24(a.coef[k] & 0xffffu, (uint32_t)s_src[r * TW + lane + k]) + sum;
        s_h[r * BT_W + lane] = sum;
    }
    __syncthreads();

    // vertical sums + mix; thread -> 4 rows of one column (coalesced rows)
    const int c = lane, x = x0 + c;
    if (x >= a.width) return;
    constexpr int Q = BT_H / 4;
    const int r0 = wv * Q;
    uint32_t h[Q + 2 * S];
#pragma unroll
    for (int k = 0; k < Q + 2 * S; k++) h[k] = s_h[(r0 + k) * BT_W + c];
#pragma unroll
    for (int q = 0; q < Q; q++)
    {
        const int r = r0 + q, y = y0 + r;
        if (y >= a.height) break;
        uint32_t t = 0;
#pragma unroll
        for (int k = 0; k < NT; k++)
            t = sizeof(PIX) == 1 ? __umul24(a.coef[k] & 0xffffu, h[q + k] & 0xffffffu) + t : a.coef[k] * h[q + k] + t;
        const int p = (int)s_src[(r + S) * TW + c + S];
        const int blur = (int)((t + (uint32_t)a.halfscale) >> a.scalebits);
        const int d = ((p - blur) * a.amount) >> 16;         // arithmetic shift, as gcc does
        int res = a.sign > 0 ? p + d : p - d;
        res = res > a.vmax ? a.vmax : res < a.vmin ? a.vmin : res;
        reinterpret_cast<PIX *>(a.dst + (size_t)y * a.dst_pitch)[x] = (PIX)res;
    }
}

template <typename PIX>
__global__ __launch_bounds__(256) void blur_mix_kernel(BlurArgs3 all)
{
    const BlurArgs &a = all.p[blockIdx.z];
    __shared__ PIX      s_src[(BT_H + 2 * MAX_STEPS) * (BT_W + 2 * MAX_STEPS)];
    __shared__ uint32_t s_h[(BT_H + 2 * MAX_STEPS) * BT_W];
    if ((int)(blockIdx.x * BT_W) >= a.width || (int)(blockIdx.y * BT_H) >= a.height) return;   // the grid is sized for the largest plane of the launch
    switch (a.steps)                                         // uniform per workgroup
    {
        case 1: blur_mix_body<PIX, 1>(a, s_src, s_h); break;
        case 2: blur_mix_body<PIX, 2>(a, s_src, s_h); break;
        case 3: blur_mix_body<PIX, 3>(a, s_src, s_h); break;
        case 4: blur_mix_body<PIX, 4>(a, s_src, s_h); break;
        case 5: blur_mix_body<PIX, 5>(a, s_src, s_h); break;
        case 6: blur_mix_body<PIX, 6>(a, s_src, s_h); break;
        case 7: blur_mix_body<PIX, 7>(a, s_src, s_h); break;
        default: break;
    }
}

// ---- unsharp / chroma smooth, 8-bit, sizes 3..9: the form that runs ------------------------------------------------
// The lapsharp rows kernel's shape.  A thread owns four adjacent columns (one dword of every row) and walks BR_ROWS rows
// down.  A row comes in as three dwords (columns x0-4 .. x0+7, enough for 2 S + 1 <= 9 taps), its four horizontal binomial
// sums stay in registers in a window of 2 S + 1 rows, and an output row is the vertical sum over the window, the
// reference's rounding shift, and its mix with the centre sample (unsharp.c:128-170): no LDS, every sample loaded three
// times from L1 / L2 instead of gathered bytewise, one dword stored per row.  Columns and rows outside the plane are
// clamped (unsharp.c:117-126, 166-170): threads whose window crosses a border gather their twelve bytes one by one.
// The sums are the reference's uint32 sums, taken exactly in float (see the kernel).  One launch covers the three planes
// of up to BR_FRAMES frames (blockIdx.z); a plane with amount 0 is copied (unsharp.c:111-115).  Strips away from the top
// and bottom of the plane take a branch-free form of the same arithmetic (byte dot products), see there.
#ifndef BR_ROWS_N
#define BR_ROWS_N 8
#endif
constexpr int BR_ROWS = BR_ROWS_N, BR_FRAMES = 16, BR_MAX_STEPS = 4;
struct BlurPlane8 { int width, height, src_pitch, dst_pitch, steps, scalebits, halfscale, amount, active; uint32_t coef[2 * BR_MAX_STEPS + 1]; };
struct BlurBatch8
{
    BlurPlane8     pl[3];
    int            sign, vmin, vmax;
    const uint8_t *src[BR_FRAMES][3];
    uint8_t       *dst[BR_FRAMES][3];
};

template <int S>
__global__ __launch_bounds__(256) void blur_rows8_kernel(BlurBatch8 B)
{
    constexpr int NT = 2 * S + 1;
    const int job = blockIdx.z, f = job / 3, c = job - 3 * f;
    const BlurPlane8 &P = B.pl[c];
    if (P.active == 0) return;
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int ys = (blockIdx.y * 4 + threadIdx.y) * BR_ROWS;
    if (x0 >= P.width || ys >= P.height) return;
    const uint8_t *src = B.src[f][c];
    uint8_t *dst = B.dst[f][c];
    const bool full_dword = x0 + 3 < P.width;
    auto store_row = [&](int y, uint32_t v) {
        uint8_t *d = dst + ((uint32_t)__mul24(y, P.dst_pitch) + (uint32_t)x0);
        if (full_dword) *reinterpret_cast<uint32_t *>(d) = v;
        else for (int k = 0; x0 + k < P.width; k++) d[k] = (uint8_t)(v >> (8 * k));
    };
    if (P.active == 2 || P.steps != S)                                     // a copied plane (the launch's S is another plane's)
    {
        if (P.active != 2) return;
        for (int r = 0; r < BR_ROWS && ys + r < P.height; r++)
        {
            const uint8_t *g = src + ((uint32_t)__mul24(ys + r, P.src_pitch) + (uint32_t)x0);
            uint32_t v = 0;
            if (full_dword) v = *reinterpret_cast<const uint32_t *>(g);
            else for (int k = 0; x0 + k < P.width; k++) v |= (uint32_t)g[k] << (8 * k);
            store_row(ys + r, v);
        }
        return;
    }
    // ---- the strips that need no row clamp (all but the plane's first and last ones), planes whose rows are whole dwords:
    // no per-row branch, no exit test; the horizontal sums as byte dot products (v_dot4_u32_u8: four taps an instruction,
    // the windows cut out of the row's three dwords with v_alignbyte), exact like everything else here.  Only the lanes at
    // the plane's left / right edge differ: the dword they have no neighbour for is their own edge sample four times.
    if ((P.width & 3) == 0 && ((P.src_pitch | P.dst_pitch) & 3) == 0 && ys >= S && ys + BR_ROWS + S <= P.height)      // wave-uniform
    {
        const bool lane_l = x0 == 0, lane_r = x0 + 4 >= P.width;
        const bool edges = __any(lane_l || lane_r);
        uint32_t cq[(NT + 3) / 4];                                            // the coefficients, four to a dword
#pragma unroll
        for (int q = 0; q < (NT + 3) / 4; q++)
        {
            cq[q] = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) if (4 * q + t < NT) cq[q] |= (P.coef[4 * q + t] & 0xffu) << (8 * t);
        }
        float cf[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) cf[t] = (float)P.coef[t];
        const uint8_t *g = src + ((uint32_t)__mul24(ys - S, P.src_pitch) + (uint32_t)x0);
        const int off_l = lane_l ? 0 : -4, off_r = lane_r ? 0 : 4;
        // (kept in vector registers and opaque to the optimiser: p + sgn * dd is then one v_mad_i32_i24, and the clamp can
        // be one v_med3_i32, which takes a single scalar operand)
        int sgn = B.sign > 0 ? 1 : -1, vlo = B.vmin, vhi = B.vmax;
        asm volatile("" : "+v"(sgn), "+v"(vlo), "+v"(vhi));
        uint8_t *d = dst + ((uint32_t)__mul24(ys, P.dst_pitch) + (uint32_t)x0);
        typedef float f2 __attribute__((ext_vector_type(2)));               // the vertical sums two columns at a time (v_pk_fma_f32)
        auto row_sums = [&](const uint8_t *row, f2 (&h)[2], uint32_t &centre) __attribute__((always_inline)) {
            uint32_t w[4];
            w[0] = *reinterpret_cast<const uint32_t *>(row + off_l);
            w[1] = *reinterpret_cast<const uint32_t *>(row);
            w[2] = *reinterpret_cast<const uint32_t *>(row + off_r);
            w[3] = 0;
            if (edges)
            {
                if (lane_l) w[0] = (w[1] & 0xffu) * 0x01010101u;
                if (lane_r) w[2] = (w[1] >> 24) * 0x01010101u;
            }
            centre = w[1];
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                uint32_t sum = 0;
#pragma unroll
                for (int q = 0; q < (NT + 3) / 4; q++)
                {
                    const int o = 4 - S + k + 4 * q;                          // window byte of this chunk's first tap
                    const uint32_t lo = w[o >> 2], hi = (o >> 2) + 1 < 4 ? w[(o >> 2) + 1] : 0u;
                    const uint32_t bytes = (o & 3) ? __builtin_amdgcn_alignbyte(hi, lo, o & 3) : lo;
                    sum = __builtin_amdgcn_udot4(bytes, cq[q], sum, false);
                }
                h[k >> 1][k & 1] = (float)sum;
            }
        };
        f2 H[NT][2];
        uint32_t C[NT];
#pragma unroll
        for (int r = 0; r < NT - 1; r++) row_sums(g + (uint32_t)__mul24(r, P.src_pitch), H[r], C[r]);
#pragma unroll
        for (int r = 0; r < BR_ROWS; r++)
        {
            row_sums(g + (uint32_t)__mul24(r + NT - 1, P.src_pitch), H[NT - 1], C[NT - 1]);
            uint32_t packed = 0;
            f2 tv[2] = { { 0.f, 0.f }, { 0.f, 0.f } };
#pragma unroll
            for (int j = 0; j < NT; j++)
            {
                const f2 cj = { cf[j], cf[j] };
                tv[0] = __builtin_elementwise_fma(cj, H[j][0], tv[0]);
                tv[1] = __builtin_elementwise_fma(cj, H[j][1], tv[1]);
            }
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const uint32_t t = (uint32_t)tv[k >> 1][k & 1];
                const int p = (int)((C[S] >> (8 * k)) & 0xffu);
                const int blur = (int)((t + (uint32_t)P.halfscale) >> P.scalebits);
                const int dd = ((p - blur) * P.amount) >> 16;                  // arithmetic shift, as gcc does
                int res = __mul24(sgn, dd) + p;                                // p + dd or p - dd (|dd| < 2^23): one v_mad_i32_i24
                asm("v_med3_i32 %0, %1, %2, %3" : "=v"(res) : "v"(res), "v"(vlo), "v"(vhi));   // the clamp (vlo <= vhi)
                packed |= (uint32_t)res << (8 * k);
            }
            *reinterpret_cast<uint32_t *>(d + (uint32_t)__mul24(r, P.dst_pitch)) = packed;
#pragma unroll
            for (int j = 0; j < NT - 1; j++)
            {
                C[j] = C[j + 1];
                H[j][0] = H[j + 1][0]; H[j][1] = H[j + 1][1];
            }
        }
        return;
    }
    const bool interior = x0 >= 4 && x0 + 7 <= P.width - 1;
    // The binomial sums are taken in float: every value is an integer below 2^24 (coefficients <= 70, horizontal sums <=
    // 255 * 2^8, vertical sums <= 255 * 2^16 = 2^24 - 65536) and every partial sum is no larger than the final one, so each
    // fused multiply-add is exact and the result is the reference's uint32 sum.  A byte becomes a float in one
    // instruction (v_cvt_f32_ubyteN) and v_fma_f32 issues in 2 cycles per wave (the compiler pairs them into
    // v_pk_fma_f32), where every integer multiply-add takes 4 (profiles/r02_valu_rate.json).
    float cf[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) cf[t] = (float)P.coef[t];
    // a row's three dwords (columns x0-4 .. x0+7, clamped into the plane)
    struct Raw { uint32_t w0, w1, w2; };
    auto load_raw = [&](int yy) -> Raw {
        yy = min(max(yy, 0), P.height - 1);
        const uint32_t ro = (uint32_t)__mul24(yy, P.src_pitch);
        Raw w;
        if (interior)
        {
            const uint32_t *g = reinterpret_cast<const uint32_t *>(src + (ro + (uint32_t)x0));
            w.w0 = g[-1]; w.w1 = g[0]; w.w2 = g[1];
        }
        else
        {
            w.w0 = w.w1 = w.w2 = 0;
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                w.w0 |= (uint32_t)src[ro + (uint32_t)min(max(x0 - 4 + i, 0), P.width - 1)] << (8 * i);
                w.w1 |= (uint32_t)src[ro + (uint32_t)min(max(x0 + i, 0), P.width - 1)] << (8 * i);
                w.w2 |= (uint32_t)src[ro + (uint32_t)min(max(x0 + 4 + i, 0), P.width - 1)] << (8 * i);
            }
        }
        return w;
    };
    // its four horizontal sums and its own dword (the centre samples)
    auto sums = [&](const Raw &w, float (&h)[4], uint32_t &centre) {
        centre = w.w1;
        float b[12];
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            b[i] = (float)((w.w0 >> (8 * i)) & 0xffu); b[4 + i] = (float)((w.w1 >> (8 * i)) & 0xffu); b[8 + i] = (float)((w.w2 >> (8 * i)) & 0xffu);
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < NT; t++) sum = __fmaf_rn(cf[t], b[4 - S + k + t], sum);
            h[k] = sum;
        }
    };
    float H[NT][4];                                                        // rows y-S .. y+S of the row being made
    uint32_t C[NT];
    {
        // the rows before the first output row: all their loads first, then the arithmetic
        Raw w[NT - 1];
#pragma unroll
        for (int r = 0; r < NT - 1; r++) w[r] = load_raw(ys - S + r);
#pragma unroll
        for (int r = 0; r < NT - 1; r++) sums(w[r], H[r], C[r]);
    }
    // (issuing a row's loads one row ahead of their use changes nothing: 111 against 103 us per 16 frames)
#pragma unroll
    for (int r = 0; r < BR_ROWS; r++)
    {
        const int y = ys + r;
        if (y >= P.height) break;
        sums(load_raw(y + S), H[NT - 1], C[NT - 1]);
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            float tf = 0.f;
#pragma unroll
            for (int j = 0; j < NT; j++) tf = __fmaf_rn(cf[j], H[j][k], tf);
            const uint32_t t = (uint32_t)tf;
            const int p = (int)((C[S] >> (8 * k)) & 0xffu);
            const int blur = (int)((t + (uint32_t)P.halfscale) >> P.scalebits);
            const int d = ((p - blur) * P.amount) >> 16;                   // arithmetic shift, as gcc does
            int res = B.sign > 0 ? p + d : p - d;
            res = res > B.vmax ? B.vmax : res < B.vmin ? B.vmin : res;
            packed |= (uint32_t)res << (8 * k);
        }
        store_row(y, packed);
#pragma unroll
        for (int j = 0; j < NT - 1; j++)
        {
            C[j] = C[j + 1];
#pragma unroll
            for (int k = 0; k < 4; k++) H[j][k] = H[j + 1][k];
        }
    }
}

__global__ void plane_copy_kernel(uint8_t *dst, int dst_pitch, const uint8_t *src, int src_pitch,
                                  int row_bytes, int rows)
{
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y;
    if (y >= rows || x >= row_bytes) return;
    const uint8_t *s = src + (size_t)y * src_pitch + x;
    uint8_t *d = dst + (size_t)y * dst_pitch + x;
    if (x + 3 < row_bytes)
        *reinterpret_cast<uint32_t *>(d) = *reinterpret_cast<const uint32_t *>(s);
    else
        for (int i = 0; x + i < row_bytes; i++) d[i] = s[i];
}

int launch_copy(hbhip_ctx *ctx, const char *name, DevPicture *in, DevPicture *out, int c)
{
    const int row = in->width[c] * in->bps;
    dim3 grid((row / 4 + 255) / 256 + 1, in->height[c]);
    HBHIP_LAUNCH(ctx, name, plane_copy_kernel, grid, dim3(256), 0, out->plane[c], out->pitch[c],
                 (const uint8_t *)in->plane[c], in->pitch[c], row, in->height[c]);
    HBHIP_CHECK(ctx, hipGetLastError());
    return HBHIP_OK;
}

// ---- 3x3 kernels (lap, isolap), 8-bit: the form that runs --------------------------------------------------------
// Both 3x3 tables are symmetric (corner a, edge b, centre c), so a row contributes to the rows above / below as
// u = a (l + r) + b m and to its own row as v = b (l + r) + c m, and the convolution of output row y is
// u[y-1] + v[y] + u[y+1].  A thread owns four adjacent columns and walks LS_ROWS rows down, carrying u / v / m of the
// rows it has passed in registers: every source dword is loaded once per thread instead of three times, each byte is
// unpacked once, and the 9-tap sum costs ~7 integer operations per pixel instead of 18.  The mix is the reference's
// double arithmetic, operation for operation (lapsharp.c:174-175).  One launch covers the three planes of up to
// LS_FRAMES frames (blockIdx.z): single 1080p planes are too small to fill the GPU or hide a launch.
#ifndef LS_ROWS_N
#define LS_ROWS_N 8
#endif
#ifndef LS_BX_N
#define LS_BX_N 64
#endif
// a workgroup is LS_BX x LS_BY threads = 4 LS_BX columns x LS_BY LS_ROWS rows of a plane
#ifndef LS_ROWS16_N
#define LS_ROWS16_N 4
#endif
// (16-bit samples: four rows per thread - 2160p x 16 frames 197 us per launch against 211 at eight; at 8 bits eight rows are
// 7 % ahead of four, profiles/r6g_lapsharp_block_shapes.log)
constexpr int LS_ROWS = LS_ROWS_N, LS_ROWS16 = LS_ROWS16_N, LS_FRAMES = 16, LS_BX = LS_BX_N, LS_BY = 256 / LS_BX_N;
struct LapPlane3 { int width, height, src_pitch, dst_pitch, stride_border, valid_w, a, b, c, active; double coef, strength; int fast, kinv; float mixf; };
struct LapBatch3
{
    LapPlane3      pl[3];
    const uint8_t *src[LS_FRAMES][3];
    uint8_t       *dst[LS_FRAMES][3];
};

__global__ __launch_bounds__(256) void lapsharp3_rows_kernel(LapBatch3 B)
{
    const int job = blockIdx.z, f = job / 3, c = job - 3 * f;
    const LapPlane3 &P = B.pl[c];
    if (!P.active) return;
    const int x0 = (blockIdx.x * LS_BX + threadIdx.x) * 4;
    const int ys = (blockIdx.y * LS_BY + threadIdx.y) * LS_ROWS;
    if (x0 >= P.width || ys >= P.height) return;
    const uint8_t *src = B.src[f][c];
    uint8_t *dst = B.dst[f][c];
    const int pitch_dw = P.src_pitch >> 2, xd = x0 >> 2;
    const bool tail = x0 + 8 > P.valid_w;

    int u_prev[4], u_cur[4], v_cur[4], m_cur[4];
    // byte offsets of the three dwords of a row this thread reads (clamped into the row: only read for pixels that end up
    // copied), as 32-bit unsigned offsets from the plane's base: a row costs one multiply and three adds of address work
    const uint32_t o0 = 4u * (uint32_t)max(xd - 1, 0), o1 = 4u * (uint32_t)xd, o2 = 4u * (uint32_t)min(xd + 1, pitch_dw - 1);
    // All of the thread's rows are fetched before the first of them is used: the stores of a row may alias the loads of the
    // next as far as the compiler knows, so a load inside the row loop waited for the previous row's store to be issued -
    // one row of loads in flight per thread, ten round trips to memory in a row.
    uint32_t raw[LS_ROWS + 2][3];
#pragma unroll
    for (int i = 0; i < LS_ROWS + 2; i++)
    {
        const int yy = min(max(ys - 1 + i, 0), P.height - 1);                  // (outside the plane: only read for pixels that end up copied)
        const uint32_t ro = (uint32_t)__mul24(yy, P.src_pitch);      // rows and pitches stay below 2^23
        raw[i][0] = *reinterpret_cast<const uint32_t *>(src + (ro + o0));
        raw[i][1] = *reinterpret_cast<const uint32_t *>(src + (ro + o1));
        raw[i][2] = *reinterpret_cast<const uint32_t *>(src + (ro + o2));
    }
    auto load_row = [&](int i, int (&u)[4], int (&v)[4], int (&m)[4]) {
        uint32_t w[3] = { raw[i][0], raw[i][1], raw[i][2] };
        if (tail)
        {
#pragma unroll
            for (int k = 0; k < 3; k++)
            {
                const int keep = P.valid_w - (x0 - 4 + 4 * k);   // valid low bytes of this dword
                if (keep <= 0) w[k] = 0;
                else if (keep < 4) w[k] &= (1u << (8 * keep)) - 1u;
            }
        }
        const int b_1 = (int)(w[0] >> 24), b0 = (int)(w[1] & 0xff), b1 = (int)((w[1] >> 8) & 0xff),
                  b2 = (int)((w[1] >> 16) & 0xff), b3 = (int)(w[1] >> 24), b4 = (int)(w[2] & 0xff);
        const int bb[6] = { b_1, b0, b1, b2, b3, b4 };
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const int h = bb[k] + bb[k + 2];
            m[k] = bb[k + 1];
            // (|taps| <= 25, samples <= 510: 24-bit multiply-adds, no 64-bit v_mad_u64_u32 forms)
            u[k] = __mul24(P.a, h) + __mul24(P.b, m[k]);
            v[k] = __mul24(P.b, h) + __mul24(P.c, m[k]);
        }
    };
    {
        int v_tmp[4], m_tmp[4];
        load_row(0, u_prev, v_tmp, m_tmp);
        load_row(1, u_cur, v_cur, m_cur);
    }
    const int y_end = min(ys + LS_ROWS, P.height);
    // which of the thread's four columns are copied whatever the row (x < stride_border + HI || x > width + stride_border
    // - HI, lapsharp.c:167-171); interior threads (nearly all) then run the row body without any per-pixel border test
    uint32_t copy_cols = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (x0 + k < P.stride_border + 2 || x0 + k > P.width + P.stride_border - 2) copy_cols |= 1u << k;
    const bool full_dword = x0 + 3 < P.width;
    // fully unrolled: the carried rows (u_prev / u_cur / v_cur / m_cur) rotate by renaming instead of sixteen moves per row
#pragma unroll
    for (int r = 0; r < LS_ROWS; r++)
    {
        const int y = ys + r;
        if (y >= y_end) break;
        int u_next[4], v_next[4], m_next[4];
        load_row(r + 2, u_next, v_next, m_next);
        const bool row_copy = (y < 2) || (y > P.height - 2);                 // y < HI || y > height - HI, HI = 2
        uint32_t packed = 0;
        if (row_copy)                                                        // wave-uniform (a wave works on one row)
        {
#pragma unroll
            for (int k = 0; k < 4; k++) packed |= (uint32_t)m_cur[k] << (8 * k);
        }
        else
        {
            if (P.fast)                                                      // one float multiply stands for the mix (lap_float_mix)
            {
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const int centre = m_cur[k];
                    const int acc = u_prev[k] + v_cur[k] + u_next[k];
                    int out = (int)((float)(acc - __mul24(P.kinv, centre)) * P.mixf) + centre;
                    out = min(max(out, 0), 255);
                    packed |= (uint32_t)out << (8 * k);
                }
            }
            else
            {
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const int centre = m_cur[k];
                    const int acc = u_prev[k] + v_cur[k] + u_next[k];
                    const double mixed = (((double)acc * P.coef) - (double)centre) * P.strength;   // lapsharp.c:174-175
                    int out = (int)(short)(int)mixed + centre;
                    out = min(max(out, 0), 255);
                    packed |= (uint32_t)out << (8 * k);
                }
            }
            if (copy_cols)                                                   // border columns keep the source sample
            {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if ((copy_cols >> k) & 1u) packed = (packed & ~(0xffu << (8 * k))) | ((uint32_t)m_cur[k] << (8 * k));
            }
        }
        uint8_t *d = dst + ((uint32_t)__mul24(y, P.dst_pitch) + (uint32_t)x0);
        if (full_dword) *reinterpret_cast<uint32_t *>(d) = packed;
        else for (int k = 0; x0 + k < P.width; k++) d[k] = (uint8_t)(packed >> (8 * k));
#pragma unroll
        for (int k = 0; k < 4; k++) { u_prev[k] = u_cur[k]; u_cur[k] = u_next[k]; v_cur[k] = v_next[k]; m_cur[k] = m_next[k]; }
    }
}

// The same walk for 10 / 12-bit samples (lapsharp_16, lapsharp.c:184): a thread owns four adjacent columns = two dwords of
// every row and reads four (samples x0 - 2 .. x0 + 5), int32 sums, clamp to the depth.  Right of the plane the reference
// reads its stride padding, which the 16-bit hb_frame_buffer_mirror_stride really fills (fifo.c:906-932): sample
// w + i = sample w - 1 - i - applied while a row's samples are unpacked, as lapsharp16_kernel does.
__global__ __launch_bounds__(256) void lapsharp3_rows16_kernel(LapBatch3 B, int max_value)
{
    const int job = blockIdx.z, f = job / 3, c = job - 3 * f;
    const LapPlane3 &P = B.pl[c];
    if (!P.active) return;
    const int x0 = (blockIdx.x * LS_BX + threadIdx.x) * 4;
    const int ys = (blockIdx.y * LS_BY + threadIdx.y) * LS_ROWS16;
    if (x0 >= P.width || ys >= P.height) return;
    const uint8_t *src = B.src[f][c];
    uint8_t *dst = B.dst[f][c];
    const int pitch_dw = P.src_pitch >> 2, xd = x0 >> 1;
    const bool tail = x0 + 5 > P.width;
    const uint32_t o0 = 4u * (uint32_t)max(xd - 1, 0), o1 = 4u * (uint32_t)xd, o2 = 4u * (uint32_t)min(xd + 1, pitch_dw - 1),
                   o3 = 4u * (uint32_t)min(xd + 2, pitch_dw - 1);
    int u_prev[4], u_cur[4], v_cur[4], m_cur[4];
    // all rows of the thread fetched ahead of its first store (lapsharp3_rows_kernel)
    uint32_t raw[LS_ROWS16 + 2][4];
#pragma unroll
    for (int i = 0; i < LS_ROWS16 + 2; i++)
    {
        const uint32_t ro = (uint32_t)__mul24(min(max(ys - 1 + i, 0), P.height - 1), P.src_pitch);
        raw[i][0] = *reinterpret_cast<const uint32_t *>(src + (ro + o0)); raw[i][1] = *reinterpret_cast<const uint32_t *>(src + (ro + o1));
        raw[i][2] = *reinterpret_cast<const uint32_t *>(src + (ro + o2)); raw[i][3] = *reinterpret_cast<const uint32_t *>(src + (ro + o3));
    }
    auto load_row = [&](int i, int (&u)[4], int (&v)[4], int (&m)[4]) {
        const uint32_t ro = (uint32_t)__mul24(min(max(ys - 1 + i, 0), P.height - 1), P.src_pitch);   // only read for samples that end up copied
        const uint32_t w0 = raw[i][0], w1 = raw[i][1], w2 = raw[i][2], w3 = raw[i][3];
        int bb[6] = { (int)(w0 >> 16), (int)(w1 & 0xffffu), (int)(w1 >> 16), (int)(w2 & 0xffffu), (int)(w2 >> 16), (int)(w3 & 0xffffu) };
        if (tail)
        {
            const uint16_t *r16 = reinterpret_cast<const uint16_t *>(src + ro);
#pragma unroll
            for (int k = 0; k < 6; k++)
            {
                const int xx = x0 - 1 + k;
                if (xx >= P.width) bb[k] = (int)r16[max(2 * P.width - 1 - xx, 0)];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const int h = bb[k] + bb[k + 2];
            m[k] = bb[k + 1];
            u[k] = __mul24(P.a, h) + __mul24(P.b, m[k]);       // |taps| <= 25, samples < 2^13: 24-bit multiplies
            v[k] = __mul24(P.b, h) + __mul24(P.c, m[k]);
        }
    };
    {
        int v_tmp[4], m_tmp[4];
        load_row(0, u_prev, v_tmp, m_tmp);
        load_row(1, u_cur, v_cur, m_cur);
    }
    const int y_end = min(ys + LS_ROWS16, P.height);
    uint32_t copy_cols = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (x0 + k < P.stride_border + 2 || x0 + k > P.width + P.stride_border - 2) copy_cols |= 1u << k;
    const bool full = x0 + 3 < P.width;
#pragma unroll
    for (int r = 0; r < LS_ROWS16; r++)
    {
        const int y = ys + r;
        if (y >= y_end) break;
        int u_next[4], v_next[4], m_next[4];
        load_row(r + 2, u_next, v_next, m_next);
        const bool row_copy = (y < 2) || (y > P.height - 2);
        int out[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const int centre = m_cur[k];
            out[k] = centre;
            if (!row_copy && !((copy_cols >> k) & 1u))
            {
                const int acc = u_prev[k] + v_cur[k] + u_next[k];
                const double mixed = (((double)acc * P.coef) - (double)centre) * P.strength;   // lapsharp.c:174-175
                out[k] = min(max((int)mixed + centre, 0), max_value);
            }
        }
        uint16_t *d = reinterpret_cast<uint16_t *>(dst + (uint32_t)__mul24(y, P.dst_pitch)) + x0;
        if (full && (((uintptr_t)d) & 7) == 0)
            *reinterpret_cast<uint2 *>(d) = make_uint2((uint32_t)out[0] | ((uint32_t)out[1] << 16), (uint32_t)out[2] | ((uint32_t)out[3] << 16));
        else for (int k = 0; k < 4 && x0 + k < P.width; k++) d[k] = (uint16_t)out[k];
#pragma unroll
        for (int k = 0; k < 4; k++) { u_prev[k] = u_cur[k]; u_cur[k] = u_next[k]; v_cur[k] = v_next[k]; m_cur[k] = m_next[k]; }
    }
}

// The mix of lapsharp.c:174-175 - (((double)sum * coef) - centre) * strength, truncated - as ONE float multiply for 8-bit
// planes: 1 / coef is an integer k for every table, so the difference is (sum - k centre) coef up to the doubles' rounding,
// and the product with coef * strength is either an integer (where a float constant a hair above or below the real one
// lands on the side the double form lands on) or far from one.  Whether some float constant reproduces the double form
// is not argued but tried: every (sum, centre) the taps can produce, 2.9 M cases for isolap, once per filter; strengths for
// which none does (0.35, 0.7) keep the double form (tests/test_eedi2_identities_cpu.py::test_lapsharp_mix_as_one_float_multiply
// runs the same search in numpy; tests/test_sharpen_gpu.py::test_lapsharp_mix_forms both forms against the oracle).
static bool lap_float_mix(const LapKernel &k, double strength, int &kinv, float &mixf)
{
    if (!(strength >= 0.0) || !(k.coef > 0.0) || k.size != 3) return false;
    const int ki = (int)std::lround(1.0 / k.coef);
    if (ki < 1 || ki > 255 || std::fabs(k.coef * ki - 1.0) > 1e-12) return false;
    int lo = 0, hi = 0;
    for (int i = 0; i < 9; i++) (k.tap[i] < 0 ? lo : hi) += 255 * k.tap[i];
    const float s = (float)(k.coef * strength);
    const float cands[3] = { s, std::nextafterf(s, INFINITY), std::nextafterf(s, -INFINITY) };
    for (float c : cands)
    {
        bool ok = true;
        for (int sum = lo; sum <= hi && ok; sum++)
            for (int centre = 0; centre < 256; centre++)
            {
                const double mixed = (((double)sum * k.coef) - (double)centre) * strength;
                const volatile float prod = (float)(sum - ki * centre) * c;      // one rounding, as v_mul_f32
                if ((int)(short)(int)mixed != (int)prod) { ok = false; break; }
            }
        if (ok) { kinv = ki; mixf = c; return true; }
    }
    return false;
}

// ------------------------------------------------------------------ filter classes
// the result per (kernel of LAP_TABLE, strength), kept for the process: every filter instance of a job - and of the jobs
// after it - with the same settings asks once
static bool lap_float_mix_cached(int kernel, double strength, int &kinv, float &mixf)
{
    struct Entry { bool ok; int kinv; float mixf; };
    static std::mutex lock;
    static std::map<std::pair<int, double>, Entry> cache;
    std::lock_guard<std::mutex> lk(lock);
    const auto key = std::make_pair(kernel, strength);
    auto it = cache.find(key);
    if (it == cache.end())
    {
        Entry e = { false, 0, 0.f };
        e.ok = lap_float_mix(LAP_TABLE[kernel], strength, e.kinv, e.mixf);
        it = cache.emplace(key, e).first;
    }
    kinv = it->second.kinv;
    mixf = it->second.mixf;
    return it->second.ok;
}

class LapsharpFilter : public SimpleFilter
{
public:
    LapsharpFilter(hbhip_ctx *c, const hbhip_lapsharp_params &p) : SimpleFilter(c), par(p)
    {
        // the search behind lap_float_mix (tens of milliseconds) runs here, at create time - init() of the plugin -, once
        // per (kernel, strength) of the process, not inside the first frame's work()
        for (int pl = 0; pl < 3; pl++)
            mix_fast[pl] = LAP_TABLE[par.kernel[pl]].size == 3 && lap_float_mix_cached(par.kernel[pl], par.strength[pl], mix_k[pl], mix_f[pl]);
    }
    // up to LS_FRAMES frames per launch when every plane uses a 3x3 kernel
    int process_many(DevPicture *const *ins, DevPicture *const *outs, int n) override
    {
        bool rows3 = true;
        for (int c = 0; c < 3; c++) rows3 &= LAP_TABLE[par.kernel[c]].size == 3;
        if (!rows3) return SimpleFilter::process_many(ins, outs, n);
        for (int at = 0; at < n; at += LS_FRAMES)
        {
            const int nf = std::min(LS_FRAMES, n - at);
            LapBatch3 B;
            int max_w = 0, max_h = 0;
            for (int c = 0; c < 3; c++)
            {
                const LapKernel &k = LAP_TABLE[par.kernel[c]];
                const DevPicture *in = ins[at];
                LapPlane3 &P = B.pl[c];
                P.width = in->width[c]; P.height = in->height[c];
                P.src_pitch = in->pitch[c]; P.dst_pitch = outs[at]->pitch[c];
                const int hb_stride = in_is_dev ? hbhip_align_up(in->width[c] * in->bps, 64) / in->bps : in_stride[c] / in->bps;
                P.stride_border = (hb_stride - in->width[c]) / 2;
                P.valid_w = in_is_dev ? in->width[c] : (1 << 30);
                P.a = k.tap[0]; P.b = k.tap[1]; P.c = k.tap[4];
                P.coef = k.coef; P.strength = par.strength[c];
                P.active = 1;
                P.fast = in_geo.bps == 1 && mix_fast[c]; P.kinv = mix_k[c]; P.mixf = mix_f[c];
                max_w = std::max(max_w, P.width); max_h = std::max(max_h, P.height);
                for (int f = 0; f < nf; f++)
                {
                    if (ins[at + f]->pitch[c] != P.src_pitch || outs[at + f]->pitch[c] != P.dst_pitch) return SimpleFilter::process_many(ins, outs, n);
                    B.src[f][c] = ins[at + f]->plane[c];
                    B.dst[f][c] = outs[at + f]->plane[c];
                }
            }
            // (hbhip_grid_x: never a multiple of 8 workgroups per row of strips - see hbhip_internal.h)
            const int rows = LS_BY * (in_geo.bps == 1 ? LS_ROWS : LS_ROWS16);
            const dim3 grid(hbhip_grid_x(((max_w + 3) / 4 + LS_BX - 1) / LS_BX), (max_h + rows - 1) / rows, 3 * nf);
            if (in_geo.bps == 1) HBHIP_LAUNCH(ctx, "lapsharp_3x3", lapsharp3_rows_kernel, grid, dim3(LS_BX, LS_BY), 0, B);
            else                 HBHIP_LAUNCH(ctx, "lapsharp_3x3", lapsharp3_rows16_kernel, grid, dim3(LS_BX, LS_BY), 0, B, (1 << in_geo.depth) - 1);
            HBHIP_CHECK(ctx, hipGetLastError());
        }
        return HBHIP_OK;
    }
    int process(DevPicture *in, DevPicture *out) override
    {
        if (LAP_TABLE[par.kernel[0]].size == 3 && LAP_TABLE[par.kernel[1]].size == 3 && LAP_TABLE[par.kernel[2]].size == 3)
        {
            DevPicture *i1[1] = { in }, *o1[1] = { out };
            return process_many(i1, o1, 1);
        }
        // one launch per kernel size present (3x3: lap / isolap, 5x5: log / isolog), covering the planes that use it
        for (int size : {3, 5})
        {
            LapArgs3 all;
            int n = 0, max_w = 0, max_h = 0;
            for (int c = 0; c < 3; c++)
            {
                const LapKernel &k = LAP_TABLE[par.kernel[c]];
                if (k.size != size) continue;
                LapArgs &a = all.p[n++];
                a.src = in->plane[c]; a.dst = out->plane[c];
                a.width = in->width[c]; a.height = in->height[c];
                a.src_pitch = in->pitch[c]; a.dst_pitch = out->pitch[c];
                // host buffers: the caller's stride decides (lapsharp.c:145); device-resident frames have no
                // hb_buffer stride, so use what hb_image_stride would be and read the padding as zeros
                // (in samples: the reference divides the strides by bps first, lapsharp.c:137-138)
                const int hb_stride = in_is_dev ? hbhip_align_up(in->width[c] * in->bps, 64) / in->bps : in_stride[c] / in->bps;
                a.stride_border = (hb_stride - in->width[c]) / 2;
                a.valid_w = in_is_dev ? in->width[c] : (1 << 30);
                for (int i = 0; i < 25; i++) a.tap[i] = k.tap[i];
                a.coef = k.coef; a.strength = par.strength[c];
                max_w = std::max(max_w, a.width); max_h = std::max(max_h, a.height);
            }
            if (n == 0) continue;
            const char *name = size == 3 ? "lapsharp_3x3" : "lapsharp_5x5";
            if (in->bps == 2)
            {
                const dim3 grid(((max_w + 1) / 2 + 255) / 256, max_h, n);
                const int max_value = (1 << in_geo.depth) - 1;
                if (size == 3) HBHIP_LAUNCH(ctx, name, lapsharp16_kernel<3>, grid, dim3(256), 0, all, max_value);
                else           HBHIP_LAUNCH(ctx, name, lapsharp16_kernel<5>, grid, dim3(256), 0, all, max_value);
            }
            else
            {
                const dim3 grid(((max_w + 3) / 4 + 255) / 256, max_h, n);
                if (size == 3) HBHIP_LAUNCH(ctx, name, lapsharp_kernel<3>, grid, dim3(256), 0, all);
                else           HBHIP_LAUNCH(ctx, name, lapsharp_kernel<5>, grid, dim3(256), 0, all);
            }
            HBHIP_CHECK(ctx, hipGetLastError());
        }
        return HBHIP_OK;
    }
    hbhip_lapsharp_params par;
    bool  mix_fast[3] = {};                        // lap_float_mix per plane (the constructor)
    int   mix_k[3] = {};
    float mix_f[3] = {};
};

class BlurMixFilter : public SimpleFilter
{
public:
    BlurMixFilter(hbhip_ctx *c, const hbhip_blur_params &p, int sign_, int vmin_, int vmax_, const char *nm)
        : SimpleFilter(c), par(p), sign(sign_), vmin(vmin_), vmax(vmax_), name(nm) {}
    // the rows kernel takes 8-bit planes with sizes up to 9 whose rows are dword aligned
    bool rows_ok(const DevPicture *in, const DevPicture *out) const
    {
        if (in_geo.bps != 1) return false;
        for (int c = 0; c < 3; c++)
        {
            if (par.amount[c] && par.size[c] / 2 > BR_MAX_STEPS) return false;
            if ((in->pitch[c] & 3) || (out->pitch[c] & 3) || ((uintptr_t)in->plane[c] & 3) || ((uintptr_t)out->plane[c] & 3)) return false;
        }
        return true;
    }
    int run_rows(DevPicture *const *ins, DevPicture *const *outs, int nf)
    {
        BlurBatch8 B;
        memset(&B, 0, sizeof(B));
        B.sign = sign; B.vmin = vmin; B.vmax = vmax;
        int max_w = 0, max_h = 0;
        bool want[BR_MAX_STEPS + 1] = {false};
        for (int c = 0; c < 3; c++)
        {
            BlurPlane8 &P = B.pl[c];
            P.width = ins[0]->width[c]; P.height = ins[0]->height[c];
            P.src_pitch = ins[0]->pitch[c]; P.dst_pitch = outs[0]->pitch[c];
            P.amount = par.amount[c];
            P.active = P.amount ? 1 : 2;
            P.steps = P.amount ? par.size[c] / 2 : 0;
            P.scalebits = P.steps * 4;
            P.halfscale = P.steps ? 1 << (P.scalebits - 1) : 0;
            uint32_t row[2 * BR_MAX_STEPS + 1] = {1};                       // binomial row of order 2*steps
            for (int k = 1; k <= 2 * P.steps; k++)
            {
                row[k] = 1;
                for (int i = k - 1; i >= 1; i--) row[i] += row[i - 1];
            }
            for (int i = 0; i <= 2 * BR_MAX_STEPS; i++) P.coef[i] = i <= 2 * P.steps ? row[i] : 0;
            if (P.amount) want[P.steps] = true;
            max_w = std::max(max_w, P.width); max_h = std::max(max_h, P.height);
            for (int f = 0; f < nf; f++) { B.src[f][c] = ins[f]->plane[c]; B.dst[f][c] = outs[f]->plane[c]; }
        }
        const dim3 grid(hbhip_grid_x(((max_w + 3) / 4 + 63) / 64), (max_h + 4 * BR_ROWS - 1) / (4 * BR_ROWS), 3 * nf), block(64, 4);
        bool copies_done = false;
        for (int st = 1; st <= BR_MAX_STEPS; st++)
        {
            if (!want[st]) continue;
            if (copies_done)
                for (int c = 0; c < 3; c++) if (B.pl[c].active == 2) B.pl[c].active = 0;     // copied by the first launch
            switch (st)
            {
                case 1: HBHIP_LAUNCH(ctx, name, blur_rows8_kernel<1>, grid, block, 0, B); break;
                case 2: HBHIP_LAUNCH(ctx, name, blur_rows8_kernel<2>, grid, block, 0, B); break;
                case 3: HBHIP_LAUNCH(ctx, name, blur_rows8_kernel<3>, grid, block, 0, B); break;
                default: HBHIP_LAUNCH(ctx, name, blur_rows8_kernel<4>, grid, block, 0, B); break;
            }
            copies_done = true;
        }
        if (!copies_done)                                                    // nothing filtered: every plane is a copy
            HBHIP_LAUNCH(ctx, name, blur_rows8_kernel<1>, grid, block, 0, B);
        HBHIP_CHECK(ctx, hipGetLastError());
        return HBHIP_OK;
    }
    // up to BR_FRAMES frames per launch (hbhip_filter_process_dev / a chain batch)
    int process_many(DevPicture *const *ins, DevPicture *const *outs, int n) override
    {
        bool ok = n > 0;
        for (int f = 0; f < n && ok; f++)
        {
            ok = rows_ok(ins[f], outs[f]);
            for (int c = 0; c < 3 && ok; c++) ok = ins[f]->pitch[c] == ins[0]->pitch[c] && outs[f]->pitch[c] == outs[0]->pitch[c];
        }
        if (!ok) return SimpleFilter::process_many(ins, outs, n);
        for (int at = 0; at < n; at += BR_FRAMES)
        {
            const int rc = run_rows(ins + at, outs + at, std::min(BR_FRAMES, n - at));
            if (rc != HBHIP_OK) return rc;
        }
        return HBHIP_OK;
    }
    int process(DevPicture *in, DevPicture *out) override
    {
        if (rows_ok(in, out)) return run_rows(&in, &out, 1);
        BlurArgs3 all;
        int n = 0, max_w = 0, max_h = 0;
        for (int c = 0; c < 3; c++)
        {
            const int amount = par.amount[c];
            if (!amount)
            {
                int rc = launch_copy(ctx, "plane_copy", in, out, c);    // unsharp.c:111-115
                if (rc != HBHIP_OK) return rc;
                continue;
            }
            BlurArgs &a = all.p[n++];
            a.src = in->plane[c]; a.dst = out->plane[c];
            a.width = in->width[c]; a.height = in->height[c];
            a.src_pitch = in->pitch[c]; a.dst_pitch = out->pitch[c];
            a.steps = par.size[c] / 2;
            a.scalebits = a.steps * 4;
            a.halfscale = 1 << (a.scalebits - 1);
            a.amount = amount;
            a.sign = sign; a.vmin = vmin; a.vmax = vmax;
            // binomial row of order 2*steps
            uint32_t row[2 * MAX_STEPS + 1] = {1};
            for (int k = 1; k <= 2 * a.steps; k++)
            {
                row[k] = 1;
                for (int i = k - 1; i >= 1; i--) row[i] += row[i - 1];
            }
            for (int i = 0; i <= 2 * MAX_STEPS; i++) a.coef[i] = i <= 2 * a.steps ? row[i] : 0;
            max_w = std::max(max_w, a.width); max_h = std::max(max_h, a.height);
        }
        if (n > 0)
        {
            // the filtered planes of the frame in one launch (blockIdx.z)
            dim3 grid((max_w + BT_W - 1) / BT_W, (max_h + BT_H - 1) / BT_H, n), block(256);
            if (in->bps == 2) HBHIP_LAUNCH(ctx, name, blur_mix_kernel<uint16_t>, grid, block, 0, all);
            else              HBHIP_LAUNCH(ctx, name, blur_mix_kernel<uint8_t>, grid, block, 0, all);
            HBHIP_CHECK(ctx, hipGetLastError());
        }
        return HBHIP_OK;
    }
    hbhip_blur_params par;
    int sign, vmin, vmax;
    const char *name;
};

template <class F, class... A>
int create_simple(hbhip_ctx *ctx, int width, int height, int depth, int lcw, int lch, hbhip_filter **out, A &&...args)
{
    if (!ctx || !out) return HBHIP_ERR_ARG;
    *out = nullptr;
    if (depth != 8 && depth != 10 && depth != 12) return HBHIP_ERR_UNSUPPORTED;
    if (width < 1 || height < 1) return HBHIP_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    F *f = new (std::nothrow) F(ctx, std::forward<A>(args)...);
    if (!f) return HBHIP_ERR_NOMEM;
    PicGeometry g;
    g.set(width, height, depth, lcw, lch);
    f->configure(g, g);
    *out = f;
    return HBHIP_OK;
}

} // namespace

extern "C" int hbhip_lapsharp_create(hbhip_ctx *ctx, const hbhip_lapsharp_params *p, int width, int height,
                                     int depth, int log2_chroma_w, int log2_chroma_h, hbhip_filter **out)
{
    if (!p) return HBHIP_ERR_ARG;
    for (int c = 0; c < 3; c++)
        if (p->kernel[c] < 0 || p->kernel[c] > 3) return HBHIP_ERR_ARG;
    return create_simple<LapsharpFilter>(ctx, width, height, depth, log2_chroma_w, log2_chroma_h, out, *p);
}

static int check_blur(const hbhip_blur_params *p)
{
    if (!p) return HBHIP_ERR_ARG;
    for (int c = 0; c < 3; c++)
        if (p->amount[c] != 0 && (p->size[c] < 3 || p->size[c] > 15 || !(p->size[c] & 1))) return HBHIP_ERR_ARG;
    return HBHIP_OK;
}

extern "C" int hbhip_unsharp_create(hbhip_ctx *ctx, const hbhip_blur_params *p, int width, int height,
                                    int depth, int log2_chroma_w, int log2_chroma_h, hbhip_filter **out)
{
    int rc = check_blur(p);
    if (rc != HBHIP_OK) return rc;
    return create_simple<BlurMixFilter>(ctx, width, height, depth, log2_chroma_w, log2_chroma_h, out,
                                        *p, +1, 0, (1 << depth) - 1, "unsharp_blur_mix");
}

extern "C" int hbhip_chroma_smooth_create(hbhip_ctx *ctx, const hbhip_blur_params *p, int width, int height,
                                          int depth, int log2_chroma_w, int log2_chroma_h, hbhip_filter **out)
{
    int rc = check_blur(p);
    if (rc != HBHIP_OK) return rc;
    // clamp range max/16 .. max - max/16 (chroma_smooth.c:233-235)
    const int max = 1 << depth;
    return create_simple<BlurMixFilter>(ctx, width, height, depth, log2_chroma_w, log2_chroma_h, out,
                                        *p, -1, max / 16, max - max / 16, "chroma_smooth_blur_mix");
}
