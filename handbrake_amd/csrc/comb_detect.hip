// comb_detect.hip — comb detection for gfx950 (8-bit luma).
//
//   comb_detect_kernel     replaces detect_gamma_combed_segment_8 / detect_combed_segment_8
//                          (libhb/templates/comb_detect_template.c:288-402, 789-933)
//   comb_mask_pass_kernel  replaces mask_filter_work / mask_erode_work / mask_dilate_work
//                          (libhb/comb_detect.c:901-966, 726-792, 556-622)
//   comb_score_kernel      replaces check_filtered_combing_mask / check_combing_mask +
//                          check_combing_results (comb_detect.c:221-276, 384-454, 1029-1049)
//
// The masks are kept in HBM exactly as the reference keeps them: one byte per
// pixel at hb_image_stride(GRAY8, width) (= width rounded up to 64), rows
// contiguous, because the 3x3 mask passes index from column 1 and therefore read
// column `width` (comb_detect.c:939-947) — the stride padding, or the first byte of
// the next row when stride == width.  All passes are HBM-bound byte streams; the
// classification is reduced on the device to one int (atomicMax) and read back.
#include "hbhip_internal.h"

namespace {

struct CombConst
{
    int   mode, spatial_metric, motion_threshold, spatial_threshold, filter_mode;
    int   block_threshold, block_width, block_height;
    float g_mthresh, g_athresh, g_athresh6;
    int   athresh_sq, athresh6;
    int   c32_min, c32_max;          // 10 / 15 scaled to the sample depth (comb_detect.c:1161-1162)
    int   lut_len;                   // 1 << depth
};

// PIX = uint8_t or uint16_t (detect_*_combed_segment_16): `pitch` is in samples; the gamma
// table (1 << depth floats, built by the host) is staged in dynamic LDS.
template <typename PIX>
__global__ __launch_bounds__(256) void comb_detect_kernel(const PIX *__restrict__ prev,
                                                          const PIX *__restrict__ cur,
                                                          const PIX *__restrict__ next, int pitch,
                                                          uint8_t *__restrict__ mask, int mask_stride,
                                                          int width, int height, CombConst k,
                                                          const float *__restrict__ lut_g, int force)
{
    extern __shared__ float L[];
    if (k.mode & 1)
    {
        for (int i = threadIdx.y * blockDim.x + threadIdx.x; i < k.lut_len; i += 256) L[i] = lut_g[i];
        __syncthreads();
    }
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = 2 + blockIdx.y * blockDim.y + threadIdx.y;        // rows 2 .. height-3 (:312-319)
    if (y >= height - 2 || x >= mask_stride) return;
    uint8_t out = 0;                                                  // memset(mask, 0, mask_stride) (:342)
    if (x < width)
    {
        const PIX *c = cur + (size_t)y * pitch + x;
        const PIX *p = prev + (size_t)y * pitch + x;
        const PIX *n = next + (size_t)y * pitch + x;
        const int v = c[0], u1 = c[-pitch], d1 = c[pitch], u2 = c[-2 * pitch], d2 = c[2 * pitch];
        if (k.mode & 1)
        {
            const float up = L[v] - L[u1], dn = L[v] - L[d1];
            if ((up > k.g_athresh && dn > k.g_athresh) || (up < -k.g_athresh && dn < -k.g_athresh))
            {
                int motion = 0;
                if (k.g_mthresh > 0)
                {
                    if (fabsf(L[p[0]] - L[v]) > k.g_mthresh && fabsf(L[u1] - L[n[-pitch]]) > k.g_mthresh &&
                        fabsf(L[d1] - L[n[pitch]]) > k.g_mthresh)
                        motion++;
                    if (fabsf(L[n[0]] - L[v]) > k.g_mthresh && fabsf(L[p[-pitch]] - L[u1]) > k.g_mthresh &&
                        fabsf(L[p[pitch]] - L[d1]) > k.g_mthresh)
                        motion++;
                }
                else
                    motion = 1;
                if (motion || force)
                {
                    // (:382-386) same association as the reference, no contraction
                    const float combing = fabsf(L[u2] + (4 * L[v]) + L[d2] - (3 * (L[u1] + L[d1])));
                    if (combing > k.g_athresh6) out = 1;
                }
            }
        }
        else
        {
            const int at = k.spatial_threshold, mt = k.motion_threshold;
            const int up = v - u1, dn = v - d1;
            if ((up > at && dn > at) || (up < -at && dn < -at))
            {
                int motion = 0;
                if (mt > 0)
                {
                    if (abs((int)p[0] - v) > mt && abs(u1 - (int)n[-pitch]) > mt && abs(d1 - (int)n[pitch]) > mt) motion++;
                    if (abs((int)n[0] - v) > mt && abs((int)p[-pitch] - u1) > mt && abs((int)p[pitch] - d1) > mt) motion++;
                }
                else
                    motion = 1;
                if (motion || force)
                {
                    if (k.spatial_metric == 0)      { if (abs(v - d2) < k.c32_min && abs(v - d1) > k.c32_max) out = 1; }
                    else if (k.spatial_metric == 1) { if ((u1 - v) * (d1 - v) > k.athresh_sq) out = 1; }
                    else if (k.spatial_metric == 2) { if (abs(u2 + 4 * v + d2 - 3 * (u1 + d1)) > k.athresh6) out = 1; }
                }
            }
        }
    }
    mask[(size_t)y * mask_stride + x] = out;
}

// ---- 8-bit, four pixels per thread, several frames per launch -------------------------------------------------------
// A 1080p luma is 2 MB: one frame per launch is dispatch latency, one byte per thread is load instructions.  Here
// grid.z runs over up to CB_FRAMES frames (frame f is classified from lumas f, f+1, f+2 of the batch - consecutive
// frames share two of their three planes, which then come from L2), a thread owns an aligned dword of its row, and
// only threads with a pixel that passes the first (vertical) test load the other two frames' rows.  (16 pixels per
// thread measured slower, 86 vs 72 us for 16 frames: with the gamma table the kernel is bound by its LDS look-ups and
// float compares - about 60 operations per pixel - not by the loads.)
constexpr int CB_FRAMES = 16;

struct CombBatch
{
    const uint8_t *luma[CB_FRAMES + 2];
    uint32_t force;                  // bit f: frame f is checked exhaustively (no motion test)
    int n;
};

__device__ __forceinline__ int cb_b(uint32_t d, int k) { return (int)((d >> (8 * k)) & 0xffu); }

// TAB (gamma metric only): the threshold tests on differences of table values as integer compares.  The gamma table is
// monotone, so for a sample value v the values a with L[v] - L[a] > t are a prefix 0 .. hi[v] - 1 of the value range and
// those with L[v] - L[a] < -t a suffix lo[v] .. 255: the host finds hi / lo by evaluating the float expression itself for
// all 256 x 256 pairs (and keeps the float form should a table ever not be monotone).  A test then costs one look-up on
// one of its two operands and two integer compares instead of two look-ups, a subtraction and the float compares; the six
// motion tests share three look-ups.  tab: 256 dwords hi | lo << 16 for the spatial threshold, 256 for the motion one.
template <bool GAMMA, bool TAB>
__global__ __launch_bounds__(256) void comb_detect4_kernel(CombBatch B, int pitch, uint8_t *__restrict__ mask_base, size_t mask_fs,
                                                           int mask_stride, int width, int height, CombConst k,
                                                           const float *__restrict__ lut_g, const uint32_t *__restrict__ tab)
{
    __shared__ float L[256];
    __shared__ uint32_t TA[TAB ? 256 : 1], TM[TAB ? 256 : 1];
    if (GAMMA)
    {
        L[threadIdx.y * 64 + threadIdx.x] = lut_g[threadIdx.y * 64 + threadIdx.x];
        if (TAB)
        {
            TA[threadIdx.y * 64 + threadIdx.x] = tab[threadIdx.y * 64 + threadIdx.x];
            TM[threadIdx.y * 64 + threadIdx.x] = tab[256 + threadIdx.y * 64 + threadIdx.x];
        }
        __syncthreads();
    }
    const int f = blockIdx.z;
    const int x = 4 * (blockIdx.x * 64 + threadIdx.x);
    const int y = 2 + blockIdx.y * 4 + threadIdx.y;                   // rows 2 .. height-3 (:312-319)
    if (y >= height - 2 || x >= mask_stride) return;
    const int force = (B.force >> f) & 1;
    uint32_t out = 0;                                                  // memset(mask, 0, mask_stride) (:342)
    if (x < width)
    {
        const size_t at = (size_t)y * pitch + x;
        const uint8_t *c = B.luma[f + 1] + at;
        auto ld = [](const uint8_t *q) { return *reinterpret_cast<const uint32_t *>(q); };
        const uint32_t cv = ld(c), cu1 = ld(c - pitch), cd1 = ld(c + pitch);
        uint32_t cand = 0;
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            const int v = cb_b(cv, j), u1 = cb_b(cu1, j), d1 = cb_b(cd1, j);
            bool pass;
            if (GAMMA && TAB)
            {
                const uint32_t t = TA[v];
                pass = max(u1, d1) < (int)(t & 0xffffu) || min(u1, d1) >= (int)(t >> 16);
            }
            else if (GAMMA)
            {
                const float up = L[v] - L[u1], dn = L[v] - L[d1];
                pass = (up > k.g_athresh && dn > k.g_athresh) || (up < -k.g_athresh && dn < -k.g_athresh);
            }
            else
            {
                const int at_ = k.spatial_threshold, up = v - u1, dn = v - d1;
                pass = (up > at_ && dn > at_) || (up < -at_ && dn < -at_);
            }
            if (pass && x + j < width) cand |= 1u << j;
        }
        if (cand)
        {
            const uint8_t *p = B.luma[f] + at, *n = B.luma[f + 2] + at;
            const uint32_t cu2 = ld(c - 2 * pitch), cd2 = ld(c + 2 * pitch);
            const uint32_t pv = ld(p), pu1 = ld(p - pitch), pd1 = ld(p + pitch);
            const uint32_t nv = ld(n), nu1 = ld(n - pitch), nd1 = ld(n + pitch);
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                if (!((cand >> j) & 1u)) continue;
                const int v = cb_b(cv, j), u1 = cb_b(cu1, j), d1 = cb_b(cd1, j), u2 = cb_b(cu2, j), d2 = cb_b(cd2, j);
                int motion = 0;
                bool comb = false;
                if (GAMMA)
                {
                    if (k.g_mthresh > 0 && TAB)
                    {
                        const uint32_t tv = TM[v], tu = TM[u1], td = TM[d1];
                        auto far = [](uint32_t t, int b) { return b < (int)(t & 0xffffu) || b >= (int)(t >> 16); };   // |L[a] - L[b]| > mthresh
                        if (far(tv, cb_b(pv, j)) && far(tu, cb_b(nu1, j)) && far(td, cb_b(nd1, j))) motion++;
                        if (far(tv, cb_b(nv, j)) && far(tu, cb_b(pu1, j)) && far(td, cb_b(pd1, j))) motion++;
                    }
                    else if (k.g_mthresh > 0)
                    {
                        if (fabsf(L[cb_b(pv, j)] - L[v]) > k.g_mthresh && fabsf(L[u1] - L[cb_b(nu1, j)]) > k.g_mthresh &&
                            fabsf(L[d1] - L[cb_b(nd1, j)]) > k.g_mthresh)
                            motion++;
                        if (fabsf(L[cb_b(nv, j)] - L[v]) > k.g_mthresh && fabsf(L[cb_b(pu1, j)] - L[u1]) > k.g_mthresh &&
                            fabsf(L[cb_b(pd1, j)] - L[d1]) > k.g_mthresh)
                            motion++;
                    }
                    else
                        motion = 1;
                    if (motion || force)
                    {
                        // (:382-386) same association as the reference, no contraction
                        const float combing = fabsf(L[u2] + (4 * L[v]) + L[d2] - (3 * (L[u1] + L[d1])));
                        comb = combing > k.g_athresh6;
                    }
                }
                else
                {
                    const int mt = k.motion_threshold;
                    if (mt > 0)
                    {
                        if (abs(cb_b(pv, j) - v) > mt && abs(u1 - cb_b(nu1, j)) > mt && abs(d1 - cb_b(nd1, j)) > mt) motion++;
                        if (abs(cb_b(nv, j) - v) > mt && abs(cb_b(pu1, j) - u1) > mt && abs(cb_b(pd1, j) - d1) > mt) motion++;
                    }
                    else
                        motion = 1;
                    if (motion || force)
                    {
                        if (k.spatial_metric == 0)      comb = abs(v - d2) < k.c32_min && abs(v - d1) > k.c32_max;
                        else if (k.spatial_metric == 1) comb = (u1 - v) * (d1 - v) > k.athresh_sq;
                        else if (k.spatial_metric == 2) comb = abs(u2 + 4 * v + d2 - 3 * (u1 + d1)) > k.athresh6;
                    }
                }
                if (comb) out |= 1u << (8 * j);
            }
        }
    }
    *reinterpret_cast<uint32_t *>(mask_base + (size_t)f * mask_fs + (size_t)y * mask_stride + x) = out;
}

// op 0: filter (classic -> h, else h&v), 1: erode (thr 2), 2: dilate (thr 4).
// Row pointers start at column 1 and columns 1..width-2 are indexed from there.
__global__ __launch_bounds__(256) void comb_mask_pass_kernel(const uint8_t *__restrict__ src,
                                                             uint8_t *__restrict__ dst, int stride,
                                                             int width, int height, int op, int classic)
{
    const int xx = 1 + blockIdx.x * blockDim.x + threadIdx.x;        // 1 .. width-2
    const int y = 1 + blockIdx.y * blockDim.y + threadIdx.y;         // 1 .. height-2
    if (xx >= width - 1 || y >= height - 1) return;
    const uint8_t *q = src + (size_t)y * stride + 1 + xx;
    const uint8_t *p = q - stride, *n = q + stride;
    int r;
    if (op == 0)
    {
        const int hc = q[-1] & q[0] & q[1];
        const int vc = p[0] & q[0] & n[0];
        r = classic ? hc : (hc & vc);
    }
    else
    {
        const int count = p[-1] + p[0] + p[1] + q[-1] + q[1] + n[-1] + n[0] + n[1];
        r = op == 1 ? (q[0] == 0 ? 0 : count >= 2) : (q[0] ? 1 : count >= 4);
    }
    dst[(size_t)y * stride + 1 + xx] = (uint8_t)r;
}

// The four mask passes of filter mode 2 (mask_filter_work h&v -> mask_erode_work -> mask_dilate_work ->
// mask_erode_work, comb_detect.c:556-966) in one launch: each pass looks one cell around itself, so a
// workgroup carries a 64 x 16 tile of the final mask through all four in LDS with a shrinking apron
// instead of four round trips through HBM.  Every pass writes only rows 1..height-2, columns
// 2..width-1 (the reference's row pointers start at column 1); everything else in the intermediate
// buffers is the zero they were allocated with, which is what `live` restores here.  The first pass
// reads column `width` of the detector's mask through the same flat addressing as the reference
// (the next row's column 0 when the stride equals the width).
constexpr int CF_W = 64, CF_H = 16, CF_A = 4, CF_LW = CF_W + 2 * CF_A, CF_LH = CF_H + 2 * CF_A;

__global__ __launch_bounds__(256) void comb_mask_fused_kernel(const uint8_t *__restrict__ mask, uint8_t *__restrict__ dst,
                                                              int stride, int width, int height)
{
    __shared__ uint8_t s_a[CF_LH][CF_LW], s_b[CF_LH][CF_LW];
    const int c0 = 2 + blockIdx.x * CF_W - CF_A, r0 = 1 + blockIdx.y * CF_H - CF_A;     // global position of LDS cell (0, 0)
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    auto live = [&](int r, int c) { return r >= 1 && r <= height - 2 && c >= 2 && c <= width - 1; };

    for (int i = tid; i < CF_LH * CF_LW; i += 256)
    {
        const int lr = i / CF_LW, lc = i - lr * CF_LW;
        const int r = r0 + lr, c = c0 + lc;
        uint8_t v = 0;
        if (r >= 0 && r < height && c >= 0 && c <= width && !(c == width && r == height - 1 && stride == width))
            v = mask[(size_t)r * stride + c];
        s_a[lr][lc] = v;
    }
    __syncthreads();
    // pass 1: horizontal & vertical triple test, apron 3
    for (int i = tid; i < (CF_LH - 2) * (CF_LW - 2); i += 256)
    {
        const int lr = 1 + i / (CF_LW - 2), lc = 1 + i % (CF_LW - 2);
        int v = 0;
        if (live(r0 + lr, c0 + lc))
            v = s_a[lr][lc - 1] & s_a[lr][lc] & s_a[lr][lc + 1] & s_a[lr - 1][lc] & s_a[lr + 1][lc];
        s_b[lr][lc] = (uint8_t)v;
    }
    __syncthreads();
    // passes 2-4: erode (>= 2 neighbours), dilate (>= 4), erode; ping-pong s_b -> s_a -> s_b -> out
    for (int pass = 0; pass < 3; pass++)
    {
        uint8_t (*in)[CF_LW] = (pass & 1) ? s_a : s_b;
        uint8_t (*out)[CF_LW] = (pass & 1) ? s_b : s_a;
        const int ap = 2 + pass;                                   // cells dropped on each side
        const int nw = CF_LW - 2 * ap, nh = CF_LH - 2 * ap;
        for (int i = tid; i < nw * nh; i += 256)
        {
            const int lr = ap + i / nw, lc = ap + i % nw;
            const int r = r0 + lr, c = c0 + lc;
            int v = 0;
            if (live(r, c))
            {
                const int count = in[lr - 1][lc - 1] + in[lr - 1][lc] + in[lr - 1][lc + 1] + in[lr][lc - 1] + in[lr][lc + 1] +
                                  in[lr + 1][lc - 1] + in[lr + 1][lc] + in[lr + 1][lc + 1];
                v = pass == 1 ? (in[lr][lc] ? 1 : count >= 4) : (in[lr][lc] == 0 ? 0 : count >= 2);
                if (pass == 2) dst[(size_t)r * stride + c] = (uint8_t)v;
            }
            if (pass < 2) out[lr][lc] = (uint8_t)v;
        }
        __syncthreads();
    }
}

// comb_mask_fused_kernel on dwords, several frames per launch (grid.z).  The cells are 0 / 1 bytes, so four of them go
// through each pass in one 32-bit operation: the triple tests of pass 1 are ANDs of byte-shifted dwords, the
// 8-neighbour counts of the erode / dilate passes sums of them (at most 8 per byte) and the thresholds one add each
// (bit 7 of count + 0x80 - thr).  A thread owns a dword column of the LDS frame and two rows per pass.  The tile is
// 64 x 16 cells at a dword-aligned column (the `live` test keeps columns 0, 1 and the border rows at zero).
constexpr int CQ_DW = (64 + 8) / 4, CQ_DP = CQ_DW + 2, CQ_ROWS = 16 + 8;      // 18 dwords (+1 pad either side) x 24 rows

__device__ __forceinline__ uint32_t cq_bytes_in(int X, int lo, int hi)            // 0x01 in byte k when lo <= X + k <= hi
{
    uint32_t m = 0x01010101u;
    const int a = lo - X, b = hi + 1 - X;
    if (a > 0) m = a >= 4 ? 0u : (m << (8 * a));
    if (b < 4) m = b <= 0 ? 0u : (m & (0x01010101u >> (8 * (4 - b))));
    return m;
}

__global__ __launch_bounds__(256) void comb_mask_fused4_kernel(const uint8_t *__restrict__ mask_base, uint8_t *__restrict__ dst_base,
                                                               size_t fs, int stride, int width, int height)
{
    __shared__ uint32_t s_a[CQ_ROWS][CQ_DP], s_b[CQ_ROWS][CQ_DP];
    const uint8_t *mask = mask_base + (size_t)blockIdx.z * fs;
    uint8_t *dst = dst_base + (size_t)blockIdx.z * fs;
    const int fc = blockIdx.x * 64 - 4, fr = blockIdx.y * 16 - 4;             // plane position of frame cell (0, 0)
    const int t = threadIdx.y * 64 + threadIdx.x;
    for (int i = t; i < CQ_ROWS * CQ_DW; i += 256)
    {
        const int lr = i / CQ_DW, q = i - lr * CQ_DW;
        const int r = fr + lr, c = fc + 4 * q;
        uint32_t v = 0;
        if (r >= 0 && r < height && c >= 0 && c <= width)
        {
            v = *reinterpret_cast<const uint32_t *>(mask + (size_t)r * stride + c);
            // column `width` of the last row lies past the plane when the rows are packed (elsewhere the reference reads
            // the next row's column 0 there, and so does the flat dword)
            if (r == height - 1 && stride == width && c + 3 >= width) v = c >= width ? 0u : (v & (0xffffffffu >> (8 * (c + 4 - width))));
        }
        s_a[lr][q + 1] = v;
    }
    __syncthreads();
    const int q = t % CQ_DW, strip = t / CQ_DW;                                   // 14 strips of 2 rows
    const uint32_t lcol = cq_bytes_in(fc + 4 * q, 2, width - 1);                  // live columns of this dword
    auto live_row = [&](int lr) { const int r = fr + lr; return r >= 1 && r <= height - 2; };
    // pass 1: horizontal & vertical triple test, frame rows 1 .. 22
    {
        const int r0 = 1 + 2 * strip;
#pragma unroll
        for (int i = 0; i < 2; i++)
        {
            const int lr = r0 + i;
            if (lr > CQ_ROWS - 2 || strip >= 14) break;
            const uint32_t l = s_a[lr][q], c = s_a[lr][q + 1], r = s_a[lr][q + 2];
            const uint32_t v = __builtin_amdgcn_alignbyte(c, l, 3) & c & __builtin_amdgcn_alignbyte(r, c, 1) & s_a[lr - 1][q + 1] & s_a[lr + 1][q + 1];
            s_b[lr][q + 1] = live_row(lr) ? (v & lcol) : 0u;
        }
    }
    __syncthreads();
    // passes 2-4: erode (>= 2 neighbours), dilate (>= 4), erode; s_b -> s_a -> s_b -> out
#pragma unroll
    for (int pass = 0; pass < 3; pass++)
    {
        uint32_t (*in)[CQ_DP] = (pass & 1) ? s_a : s_b;
        uint32_t (*out)[CQ_DP] = (pass & 1) ? s_b : s_a;
        const int ap = 2 + pass;                                               // frame rows ap .. 23 - ap
        const int r0 = ap + 2 * strip;
        if (strip < 14 && r0 <= CQ_ROWS - 1 - ap)
        {
            const uint32_t K = (uint32_t)(0x80 - (pass == 1 ? 4 : 2)) * 0x01010101u;
            uint32_t S2[4], S3[4], C[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                const int lr = min(r0 - 1 + i, CQ_ROWS - 1);
                const uint32_t l = in[lr][q], c = in[lr][q + 1], r = in[lr][q + 2];
                C[i] = c;
                S2[i] = __builtin_amdgcn_alignbyte(c, l, 3) + __builtin_amdgcn_alignbyte(r, c, 1);
                S3[i] = S2[i] + c;
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
            {
                const int lr = r0 + i;
                if (lr > CQ_ROWS - 1 - ap) break;
                const uint32_t count = S3[i] + S2[i + 1] + S3[i + 2];
                const uint32_t ge = ((count + K) >> 7) & 0x01010101u;
                const uint32_t c = C[i + 1];
                uint32_t v = pass == 1 ? (c | ge) : (c & ge);
                v = live_row(lr) ? (v & lcol) : 0u;
                if (pass < 2) out[lr][q + 1] = v;
                else
                {
                    // the tile: frame rows 4 .. 19, dword columns 1 .. 16
                    const int r = fr + lr, cc = fc + 4 * q;
                    if (q >= 1 && q <= 16 && r >= 0 && r < height && cc < stride)
                        *reinterpret_cast<uint32_t *>(dst + (size_t)r * stride + cc) = v;
                }
            }
        }
        __syncthreads();
    }
}

// one wave per block_width x block_height block; result = max category seen
// One workgroup (4 waves) per row of blocks: each wave sums one block at a time, the categories are
// maximised inside the workgroup and one atomic per workgroup reaches the result word (a global
// atomic - or even an uncached read - per block would queue 8100 requests on one L2 channel).
__global__ __launch_bounds__(256) void comb_score_kernel(const uint8_t *__restrict__ mask, int stride,
                                                         int width, int height, int bw, int bh, int thr,
                                                         int filtered, int blocks_x, int *result, int overlay, size_t fs)
{
    __shared__ int s_cat;
    mask += (size_t)blockIdx.y * fs;                           // grid.y = frames of a batch, 4 result words each
    result += 4 * blockIdx.y;
    if (threadIdx.x == 0) s_cat = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int by = blockIdx.x, y0 = by * bh;
    int best = 0;
    for (int bx = wave; bx < blocks_x; bx += 4)
    {
        const int x0 = bx * bw;
        int score = 0;
        if (filtered && bw == 16 && bh == 16 && !(stride & 3) && !overlay)     // overlay modes: cells may hold 128
        {
            // the default 16 x 16 block of a filtered mask: one aligned dword of 0/1 bytes per lane
            const int iy = lane >> 2, ix = (lane & 3) * 4;
            const uint32_t v = *reinterpret_cast<const uint32_t *>(mask + (size_t)(y0 + iy) * stride + x0 + ix);
            score = (int)((v * 0x01010101u) >> 24);
        }
        else
        {
            for (int i = lane; i < bw * bh; i += 64)
            {
                const int iy = i / bw, ix = i - iy * bw;
                const uint8_t *r = mask + (size_t)(y0 + iy) * stride + x0 + ix;
                const int xa = x0 + ix;
                if (filtered)             score += r[0];
                else if (xa == 0)         score += r[0] & r[1];
                else if (xa == width - 1) score += r[-1] & r[0];
                else                      score += r[-1] & r[0] & r[1];
            }
        }
        for (int off = 32; off > 0; off >>= 1) score += __shfl_down(score, off, 64);
        score = __shfl(score, 0, 64);
        const int cat = score > thr ? 2 : (score >= thr / 2 ? 1 : 0);
        if (overlay && cat && lane == 0)
        {
            // mask_box_x / _y as ONE check thread leaves them (comb_detect.c:205-214): the first block in raster
            // order above the threshold, else the last one at or above half of it
            const int idx = by * blocks_x + bx;
            if (cat == 2) atomicMax(result + 1, 0x7fffffff - idx);
            atomicMax(result + 2, idx + 1);
        }
        best = max(best, cat);
        if (best == 2) break;                                   // HEAVY: nothing can raise it (comb_detect.c:211-214)
    }
    if (lane == 0 && best) atomicMax(&s_cat, best);
    __syncthreads();
    if (threadIdx.x == 0 && s_cat) atomicMax(result, s_cat);
}

// draw_mask_box (comb_detect_template.c:21-53): the outline of the recorded block, value 128, into the mask itself
// (it stays there: later frames' passes and block sums see whatever cell no pass rewrites).  The bottom line of a
// box in the last block row lands on row `height` when the height is a multiple of the block height - one row
// past the plane, where the reference scribbles into its allocation padding; skipped here.
__global__ void comb_draw_box_kernel(uint8_t *mask, int stride, int height, int x, int y, int bw, int bh)
{
    for (int i = threadIdx.x; i < bw; i += blockDim.x)
    {
        mask[(size_t)y * stride + x + i] = 128;
        if (y + bh < height) mask[(size_t)(y + bh) * stride + x + i] = 128;
    }
    for (int i = threadIdx.x; i < bh; i += blockDim.x)
    {
        mask[(size_t)(y + i) * stride + x] = 128;
        mask[(size_t)(y + i) * stride + x + bw] = 128;
    }
}

struct OverlayArgs
{
    uint8_t *plane[3];
    int pitch[3], w[3], h[3];
    const uint8_t *mask;
    int mstride, composite, maxv, half;
};

// apply_mask (comb_detect_template.c:72-136): mask-only mode blanks the picture (luma 0, chroma half); luma then takes
// the maximum where the mask is 1 and half where it is 128 (the box).
template <typename PIX>
__global__ __launch_bounds__(256) void comb_overlay_kernel(OverlayArgs a)
{
    const int c = blockIdx.z;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= a.w[c] || y >= a.h[c]) return;
    PIX *p = reinterpret_cast<PIX *>(a.plane[c] + (size_t)y * a.pitch[c]) + x;
    if (c != 0)
    {
        if (!a.composite) *p = (PIX)a.half;
        return;
    }
    const int m = a.mask[(size_t)y * a.mstride + x];
    if (m == 1)        *p = (PIX)a.maxv;
    else if (m == 128) *p = (PIX)a.half;
    else if (!a.composite) *p = 0;
}

class CombDetectFilter : public hbhip_filter
{
public:
    CombDetectFilter(hbhip_ctx *c, const hbhip_comb_detect_params &p) : hbhip_filter(c), par(p) {}
    ~CombDetectFilter() override
    {
        for (int i = 0; i < 3; i++) if (luma_alloc[i]) (void)hipFree(luma_alloc[i]);
        if (masks) (void)hipFree(masks);
        if (d_lut) (void)hipFree(d_lut);
        if (d_tab) (void)hipFree(d_tab);
        if (d_result) (void)hipFree(d_result);
        if (h_result) (void)hipHostFree(h_result);
        if (stage) (void)hipFree(stage);
        if (bmasks) (void)hipFree(bmasks);
        if (d_bresult) (void)hipFree(d_bresult);
        if (h_bresult) (void)hipHostFree(h_bresult);
    }

    int setup(int w, int h, int depth)
    {
        width = w; height = h;
        in_geo.set(w, h, depth, 1, 1);
        bps = in_geo.bps;
        const int up = depth - 8;                                   // thresholds scale with the depth (:1151-1153)
        const int max_value = (1 << depth) - 1;
        par.motion_threshold <<= up;
        par.spatial_threshold <<= up;
        out_geo = in_geo;
        if (par.block_width > w) par.block_width = w;               // comb_detect.c:1139-1146
        if (par.block_height > h) par.block_height = h;
        if (par.block_width < 1 || par.block_height < 1) return HBHIP_ERR_ARG;
        if (par.mode & ~15) return HBHIP_ERR_UNSUPPORTED;
        overlay = (par.mode & 12) != 0;                             // MODE_MASK / MODE_COMPOSITE (comb_detect.c:25-26)
        if (h < 5 || w < 4) return HBHIP_ERR_UNSUPPORTED;
        pitch = hbhip_align_up(w * bps, 256);                      // bytes
        mstride = hbhip_align_up(w, 64);                            // hb_image_stride(GRAY8, w)
        for (int i = 0; i < 3; i++)
        {
            HBHIP_CHECK(ctx, hipMalloc((void **)&luma_alloc[i], (size_t)pitch * h));
            slot[i] = -1;
        }
        const size_t msz = (size_t)mstride * h + 256;               // tail guard for the column-`width` reads
        HBHIP_CHECK(ctx, hipMalloc((void **)&masks, 3 * msz));
        HBHIP_CHECK(ctx, hipMemsetAsync(masks, 0, 3 * msz, ctx->stream));
        mask = masks; mask_filtered = masks + msz; mask_temp = masks + 2 * msz;
        HBHIP_CHECK(ctx, hipMalloc((void **)&d_lut, sizeof(float) * (max_value + 1)));
        if (depth == 8)
        {
            HBHIP_CHECK(ctx, hipMemcpyAsync(d_lut, par.gamma_lut, sizeof(float) * 256, hipMemcpyHostToDevice, ctx->stream));
            lut_ready = true;
        }
        HBHIP_CHECK(ctx, hipMalloc((void **)&d_result, sizeof(int) * 4));
        HBHIP_CHECK(ctx, hipHostMalloc((void **)&h_result, sizeof(int) * 4, hipHostMallocDefault));
        HBHIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        k.mode = par.mode; k.spatial_metric = par.spatial_metric;
        k.motion_threshold = par.motion_threshold; k.spatial_threshold = par.spatial_threshold;
        k.filter_mode = par.filter_mode; k.block_threshold = par.block_threshold;
        k.block_width = par.block_width; k.block_height = par.block_height;
        k.g_mthresh = (float)par.motion_threshold / (float)max_value;     // comb_detect.c:1153-1158
        k.g_athresh = (float)par.spatial_threshold / (float)max_value;
        k.g_athresh6 = 6 * k.g_athresh;
        k.athresh_sq = par.spatial_threshold * par.spatial_threshold;
        k.athresh6 = 6 * par.spatial_threshold;
        k.c32_min = 10 << up; k.c32_max = 15 << up;
        k.lut_len = max_value + 1;
        if (depth == 8 && (par.mode & 1)) return build_threshold_tables();
        return HBHIP_OK;
    }

    // comb_detect4_kernel's integer form of the gamma metric's threshold tests (see there)
    int build_threshold_tables()
    {
        const float *L = par.gamma_lut;
        std::vector<uint32_t> tab(512);
        bool ok = true;
        for (int which = 0; which < 2 && ok; which++)
        {
            const float t = which ? k.g_mthresh : k.g_athresh;
            for (int v = 0; v < 256 && ok; v++)
            {
                int hi = 0, lo = 256;
                while (hi < 256 && L[v] - L[hi] > t) hi++;
                for (int a = 255; a >= 0 && L[v] - L[a] < -t; a--) lo = a;
                for (int a = 0; a < 256 && ok; a++)                        // a prefix and a suffix, nothing else
                    ok = ((L[v] - L[a] > t) == (a < hi)) && ((L[v] - L[a] < -t) == (a >= lo)) &&
                         ((fabsf(L[a] - L[v]) > t) == (a < hi || a >= lo));
                tab[256 * which + v] = (uint32_t)hi | ((uint32_t)lo << 16);
            }
        }
        if (!ok) return HBHIP_OK;                                          // not monotone: the float form stays
        HBHIP_CHECK(ctx, hipMalloc((void **)&d_tab, sizeof(uint32_t) * 512));
        HBHIP_CHECK(ctx, hipMemcpy(d_tab, tab.data(), sizeof(uint32_t) * 512, hipMemcpyHostToDevice));
        return HBHIP_OK;
    }

    // depth > 8: the (1 << depth)-entry gamma table arrives separately (the params struct holds 256)
    int set_gamma_lut(const float *lut, int entries)
    {
        if (!lut || entries != k.lut_len) return HBHIP_ERR_ARG;
        HBHIP_CHECK(ctx, hipMemcpyAsync(d_lut, lut, sizeof(float) * entries, hipMemcpyHostToDevice, ctx->stream));
        HBHIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        lut_ready = true;
        if (k.lut_len == 256 && (par.mode & 1))
        {
            // the 8-bit kernel classifies through integer thresholds derived from the table (build_threshold_tables):
            // a new table means new thresholds, or the float form when the new one is not monotone
            memcpy(par.gamma_lut, lut, sizeof(float) * 256);
            if (d_tab) { (void)hipFree(d_tab); d_tab = nullptr; }
            return build_threshold_tables();
        }
        return HBHIP_OK;
    }

    // ring of device luma planes: ref[i] -> which allocation
    int store(const void *luma, int stride, bool device)
    {
        // rotate: the allocation that held ref[0] is reused unless ref[1]/ref[2] still point at it
        int freed = ref[0];
        ref[0] = ref[1]; ref[1] = ref[2];
        if (luma == nullptr)
        {
            if (ref[2] < 0) return HBHIP_ERR_STATE;
            return HBHIP_OK;                                          // ref[2] repeated
        }
        int use = -1;
        for (int i = 0; i < 3 && use < 0; i++)
            if (i != ref[0] && i != ref[1]) use = i;
        (void)freed;
        if (stride < width * bps) return HBHIP_ERR_ARG;
        if (device)
            HBHIP_CHECK(ctx, hipMemcpy2DAsync(luma_alloc[use], pitch, luma, stride, (size_t)width * bps, height,
                                              hipMemcpyDeviceToDevice, ctx->stream));
        else
        {
            // on the context's upload stream: wait for this copy only, not for what the job's other filters have
            // queued on the compute stream.  The slot being overwritten was last read by a classify() two frames ago,
            // and classify() waits for its kernels, so nothing can still be reading it.
            hipEvent_t done = ctx->sync_ev_get();
            if (!done) return ctx->fail(hipErrorOutOfMemory, "hipEventCreate(store)");
            HBHIP_CHECK(ctx, hipMemcpy2DAsync(luma_alloc[use], pitch, luma, stride, (size_t)width * bps, height,
                                              hipMemcpyHostToDevice, ctx->up()));
            HBHIP_CHECK(ctx, hipEventRecord(done, ctx->up()));
            const hipError_t e = hipEventSynchronize(done);
            ctx->sync_ev_put(done);
            if (e != hipSuccess) return ctx->fail(e, "hipEventSynchronize(store)");
        }
        ref[2] = use;
        return HBHIP_OK;
    }

    int classify(int force, int *combed)
    {
        if (ref[0] < 0 || ref[1] < 0 || ref[2] < 0) return HBHIP_ERR_STATE;
        if ((par.mode & 1) && !lut_ready) return HBHIP_ERR_STATE;
        dim3 b(64, 4);
        dim3 g((mstride + 63) / 64, (height - 4 + 3) / 4);
        const size_t lds = (par.mode & 1) ? sizeof(float) * k.lut_len : 0;
        const bool quad = bps == 1 && !overlay;
        if (quad)
        {
            CombBatch B;
            for (int i = 0; i < 3; i++) B.luma[i] = luma_alloc[ref[i]];
            B.force = force ? 1u : 0u; B.n = 1;
            const dim3 g4(hbhip_grid_x((mstride / 4 + 63) / 64), (height - 4 + 3) / 4, 1);
            if ((par.mode & 1) && d_tab) HBHIP_LAUNCH(ctx, "comb_detect", (comb_detect4_kernel<true, true>), g4, b, 0, B, pitch, mask, (size_t)0, mstride, width, height, k, (const float *)d_lut, (const uint32_t *)d_tab);
            else if (par.mode & 1)       HBHIP_LAUNCH(ctx, "comb_detect", (comb_detect4_kernel<true, false>), g4, b, 0, B, pitch, mask, (size_t)0, mstride, width, height, k, (const float *)d_lut, (const uint32_t *)nullptr);
            else                         HBHIP_LAUNCH(ctx, "comb_detect", (comb_detect4_kernel<false, false>), g4, b, 0, B, pitch, mask, (size_t)0, mstride, width, height, k, (const float *)d_lut, (const uint32_t *)nullptr);
        }
        else if (bps == 2)
            HBHIP_LAUNCH(ctx, "comb_detect", comb_detect_kernel<uint16_t>, g, b, lds, (const uint16_t *)luma_alloc[ref[0]],
                         (const uint16_t *)luma_alloc[ref[1]], (const uint16_t *)luma_alloc[ref[2]], pitch / 2, mask, mstride,
                         width, height, k, (const float *)d_lut, force);
        else
            HBHIP_LAUNCH(ctx, "comb_detect", comb_detect_kernel<uint8_t>, g, b, lds, (const uint8_t *)luma_alloc[ref[0]],
                         (const uint8_t *)luma_alloc[ref[1]], (const uint8_t *)luma_alloc[ref[2]], pitch, mask, mstride,
                         width, height, k, (const float *)d_lut, force);
        const bool filt = (par.mode & 2) != 0;
        dim3 gm((width - 2 + 63) / 64, (height - 2 + 3) / 4);
        if (filt)
        {
            // overlay modes: the box outline written into the filtered mask persists in the cells no pass rewrites
            // and the passes read it, so they run one by one on the real buffers (the fused kernel carries its
            // intermediate masks in LDS and assumes those cells are zero)
            const bool fused = par.filter_mode == 2 && !overlay;
            if (fused && quad)
                HBHIP_LAUNCH(ctx, "comb_mask_passes", comb_mask_fused4_kernel, dim3((width + 63) / 64, (height + 15) / 16, 1), b, 0,
                             (const uint8_t *)mask, mask_filtered, (size_t)0, mstride, width, height);
            else if (fused)
                HBHIP_LAUNCH(ctx, "comb_mask_passes", comb_mask_fused_kernel,
                             dim3((width - 2 + CF_W - 1) / CF_W, (height - 2 + CF_H - 1) / CF_H), b, 0,
                             (const uint8_t *)mask, mask_filtered, mstride, width, height);
            else if (par.filter_mode == 1)
                HBHIP_LAUNCH(ctx, "comb_mask_filter", comb_mask_pass_kernel, gm, b, 0, (const uint8_t *)mask, mask_filtered, mstride, width, height, 0, 1);
            else
                HBHIP_LAUNCH(ctx, "comb_mask_filter", comb_mask_pass_kernel, gm, b, 0, (const uint8_t *)mask, mask_temp, mstride, width, height, 0, 0);
            if (par.filter_mode == 2 && !fused)
            {
                HBHIP_LAUNCH(ctx, "comb_mask_erode", comb_mask_pass_kernel, gm, b, 0, (const uint8_t *)mask_temp, mask_filtered, mstride, width, height, 1, 0);
                HBHIP_LAUNCH(ctx, "comb_mask_dilate", comb_mask_pass_kernel, gm, b, 0, (const uint8_t *)mask_filtered, mask_temp, mstride, width, height, 2, 0);
                HBHIP_LAUNCH(ctx, "comb_mask_erode", comb_mask_pass_kernel, gm, b, 0, (const uint8_t *)mask_temp, mask_filtered, mstride, width, height, 1, 0);
            }
        }
        HBHIP_CHECK(ctx, hipMemsetAsync(d_result, 0, sizeof(int) * 4, ctx->stream));
        const int bw = par.block_width, bh = par.block_height;
        const int blocks_x = (width - bw + bw - 1) / bw;             // x = 0, bw, ... while x < width - bw
        const int blocks_y = height / bh;                            // y + bh <= height
        if (blocks_x > 0 && blocks_y > 0)
        {
            HBHIP_LAUNCH(ctx, "comb_block_score", comb_score_kernel, dim3(blocks_y), dim3(256), 0,
                         (const uint8_t *)(filt ? mask_filtered : mask), mstride, width, height, bw, bh,
                         par.block_threshold, filt ? 1 : 0, blocks_x, d_result, overlay ? 1 : 0, (size_t)0);
        }
        HBHIP_CHECK(ctx, hipGetLastError());
        HBHIP_CHECK(ctx, hipMemcpyAsync(h_result, d_result, sizeof(int) * 4, hipMemcpyDeviceToHost, ctx->stream));
        {
            // wait for the verdict - an event behind the read-back rather than the whole stream, which other filter
            // threads keep filling
            hipEvent_t done = ctx->sync_ev_get();
            if (!done) return ctx->fail(hipErrorOutOfMemory, "hipEventCreate(classify)");
            HBHIP_CHECK(ctx, hipEventRecord(done, ctx->stream));
            const hipError_t e = hipEventSynchronize(done);
            ctx->sync_ev_put(done);
            if (e != hipSuccess) return ctx->fail(e, "hipEventSynchronize(classify)");
        }
        *combed = h_result[0];
        if (overlay && h_result[2] > 0 && blocks_x > 0)
        {
            const int idx = h_result[1] > 0 ? 0x7fffffff - h_result[1] : h_result[2] - 1;
            box_x = (idx % blocks_x) * bw;
            box_y = (idx / blocks_x) * bh;
        }
        return HBHIP_OK;
    }

    // n frames at once: frame i is classified from lumas[i], lumas[i + 1], lumas[i + 2] (device pointers, `stride` bytes a
    // row), in three launches and one read-back for all of them.  Independent of the ring store() / classify() work on.
    int classify_many(const void *const *lumas, int stride, int n, uint32_t force_bits, int *combed)
    {
        if (n < 1 || n > CB_FRAMES || bps != 1 || overlay) return HBHIP_ERR_UNSUPPORTED;
        const bool filt = (par.mode & 2) != 0;
        if (filt && par.filter_mode != 2) return HBHIP_ERR_UNSUPPORTED;
        if ((par.mode & 1) && !lut_ready) return HBHIP_ERR_STATE;
        if (stride < width || (stride & 3)) return HBHIP_ERR_ARG;
        CombBatch B;
        for (int i = 0; i < n + 2; i++)
        {
            if (!lumas[i] || ((uintptr_t)lumas[i] & 3)) return HBHIP_ERR_ARG;
            B.luma[i] = (const uint8_t *)lumas[i];
        }
        B.force = force_bits; B.n = n;
        const size_t msz = (size_t)mstride * height + 256, fs = 2 * msz;
        if (!bmasks)
        {
            HBHIP_CHECK(ctx, hipMalloc((void **)&bmasks, fs * CB_FRAMES));
            HBHIP_CHECK(ctx, hipMemsetAsync(bmasks, 0, fs * CB_FRAMES, ctx->stream));
            HBHIP_CHECK(ctx, hipMalloc((void **)&d_bresult, sizeof(int) * 4 * CB_FRAMES));
            HBHIP_CHECK(ctx, hipHostMalloc((void **)&h_bresult, sizeof(int) * 4 * CB_FRAMES, hipHostMallocDefault));
        }
        const dim3 b(64, 4), g4(hbhip_grid_x((mstride / 4 + 63) / 64), (height - 4 + 3) / 4, n);
        if ((par.mode & 1) && d_tab) HBHIP_LAUNCH(ctx, "comb_detect", (comb_detect4_kernel<true, true>), g4, b, 0, B, stride, bmasks, fs, mstride, width, height, k, (const float *)d_lut, (const uint32_t *)d_tab);
        else if (par.mode & 1)       HBHIP_LAUNCH(ctx, "comb_detect", (comb_detect4_kernel<true, false>), g4, b, 0, B, stride, bmasks, fs, mstride, width, height, k, (const float *)d_lut, (const uint32_t *)nullptr);
        else                         HBHIP_LAUNCH(ctx, "comb_detect", (comb_detect4_kernel<false, false>), g4, b, 0, B, stride, bmasks, fs, mstride, width, height, k, (const float *)d_lut, (const uint32_t *)nullptr);
        if (filt)
            HBHIP_LAUNCH(ctx, "comb_mask_passes", comb_mask_fused4_kernel, dim3((width + 63) / 64, (height + 15) / 16, n), b, 0,
                         (const uint8_t *)bmasks, bmasks + msz, fs, mstride, width, height);
        HBHIP_CHECK(ctx, hipMemsetAsync(d_bresult, 0, sizeof(int) * 4 * n, ctx->stream));
        const int bw = par.block_width, bh = par.block_height;
        const int blocks_x = (width - bw + bw - 1) / bw, blocks_y = height / bh;
        if (blocks_x > 0 && blocks_y > 0)
            HBHIP_LAUNCH(ctx, "comb_block_score", comb_score_kernel, dim3(blocks_y, n), dim3(256), 0,
                         (const uint8_t *)(filt ? bmasks + msz : bmasks), mstride, width, height, bw, bh,
                         par.block_threshold, filt ? 1 : 0, blocks_x, d_bresult, 0, fs);
        HBHIP_CHECK(ctx, hipGetLastError());
        HBHIP_CHECK(ctx, hipMemcpyAsync(h_bresult, d_bresult, sizeof(int) * 4 * n, hipMemcpyDeviceToHost, ctx->stream));
        hipEvent_t done = ctx->sync_ev_get();
        if (!done) return ctx->fail(hipErrorOutOfMemory, "hipEventCreate(classify_many)");
        HBHIP_CHECK(ctx, hipEventRecord(done, ctx->stream));
        const hipError_t e = hipEventSynchronize(done);
        ctx->sync_ev_put(done);
        if (e != hipSuccess) return ctx->fail(e, "hipEventSynchronize(classify_many)");
        for (int i = 0; i < n; i++) combed[i] = h_bresult[4 * i];
        return HBHIP_OK;
    }

    // draw_mask_box + apply_mask (comb_detect_template.c:21-136) on a picture holding a copy of the frame that was
    // just classified: planes in HBM, pitches in bytes.
    int overlay_dev(const hbhip_dev_frame *fr, const int plane_w[3], const int plane_h[3])
    {
        if (!overlay) return HBHIP_ERR_STATE;
        uint8_t *m = (par.mode & 2) ? mask_filtered : mask;
        HBHIP_LAUNCH(ctx, "comb_draw_box", comb_draw_box_kernel, dim3(1), dim3(256), 0, m, mstride, height, box_x, box_y,
                     par.block_width, par.block_height);
        const int maxv = (1 << in_geo.depth) - 1, half = 1 << (in_geo.depth - 1);
        OverlayArgs a;
        for (int c = 0; c < 3; c++)
        {
            a.plane[c] = (uint8_t *)fr->plane[c]; a.pitch[c] = fr->stride[c];
            a.w[c] = plane_w[c]; a.h[c] = plane_h[c];
        }
        a.mask = m; a.mstride = mstride; a.composite = (par.mode & 8) != 0; a.maxv = maxv; a.half = half;
        const dim3 grid((plane_w[0] + 63) / 64, (plane_h[0] + 3) / 4, 3);
        if (bps == 2) HBHIP_LAUNCH(ctx, "comb_overlay", comb_overlay_kernel<uint16_t>, grid, dim3(64, 4), 0, a);
        else          HBHIP_LAUNCH(ctx, "comb_overlay", comb_overlay_kernel<uint8_t>, grid, dim3(64, 4), 0, a);
        HBHIP_CHECK(ctx, hipGetLastError());
        return HBHIP_OK;
    }

    // the same for a frame in host memory: staged through a device picture
    int overlay_host(const hbhip_host_frame *fr, const int plane_w[3], const int plane_h[3])
    {
        if (!overlay) return HBHIP_ERR_STATE;
        size_t off[3], total = 0;
        int pitchv[3];
        for (int c = 0; c < 3; c++)
        {
            pitchv[c] = hbhip_align_up(plane_w[c] * bps, 256);
            off[c] = total;
            total += (size_t)pitchv[c] * plane_h[c];
        }
        if (total > stage_bytes)
        {
            if (stage) (void)hipFree(stage);
            stage = nullptr; stage_bytes = 0;
            HBHIP_CHECK(ctx, hipMalloc((void **)&stage, total));
            stage_bytes = total;
        }
        hbhip_dev_frame d;
        for (int c = 0; c < 3; c++)
        {
            d.plane[c] = stage + off[c]; d.stride[c] = pitchv[c];
            if (fr->plane[c] == nullptr || fr->stride[c] < plane_w[c] * bps) return HBHIP_ERR_ARG;
            HBHIP_CHECK(ctx, hipMemcpy2DAsync(d.plane[c], pitchv[c], fr->plane[c], fr->stride[c], (size_t)plane_w[c] * bps,
                                              plane_h[c], hipMemcpyHostToDevice, ctx->stream));
        }
        int rc = overlay_dev(&d, plane_w, plane_h);
        if (rc != HBHIP_OK) return rc;
        for (int c = 0; c < 3; c++)
            HBHIP_CHECK(ctx, hipMemcpy2DAsync(fr->plane[c], fr->stride[c], d.plane[c], pitchv[c], (size_t)plane_w[c] * bps,
                                              plane_h[c], hipMemcpyDeviceToHost, ctx->stream));
        HBHIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        return HBHIP_OK;
    }

    // generic surface: comb detect does not transform pictures
    DevPicture *acquire_input() override { return nullptr; }
    int submit(DevPicture *) override { return HBHIP_ERR_STATE; }
    int flush() override { return HBHIP_OK; }
    int pending() override { return 0; }
    DevPicture *pop_output() override { return nullptr; }
    void recycle_output(DevPicture *) override {}

private:
    hbhip_comb_detect_params par;
    CombConst k;
    int width = 0, height = 0, pitch = 0, mstride = 0, bps = 1;
    bool lut_ready = false;
    uint8_t *luma_alloc[3] = {nullptr, nullptr, nullptr};
    int slot[3];
    int ref[3] = {-1, -1, -1};
    uint8_t *masks = nullptr, *mask = nullptr, *mask_filtered = nullptr, *mask_temp = nullptr;
    float *d_lut = nullptr;
    uint32_t *d_tab = nullptr;          // comb_detect4_kernel<true, true>'s threshold tables (8-bit gamma metric)
    int *d_result = nullptr, *h_result = nullptr;
    bool overlay = false;
    int box_x = 0, box_y = 0;                        // pv->mask_box_x / _y
    uint8_t *stage = nullptr;
    size_t stage_bytes = 0;
    uint8_t *bmasks = nullptr;                       // classify_many: CB_FRAMES x (mask, filtered mask)
    int *d_bresult = nullptr, *h_bresult = nullptr;
};

} // namespace

extern "C" int hbhip_comb_detect_create(hbhip_ctx *ctx, const hbhip_comb_detect_params *p, int width, int height,
                                        int depth, hbhip_filter **out)
{
    if (!ctx || !p || !out) return HBHIP_ERR_ARG;
    *out = nullptr;
    if (depth != 8 && depth != 10 && depth != 12) return HBHIP_ERR_UNSUPPORTED;
    (void)hipSetDevice(ctx->device);
    CombDetectFilter *f = new (std::nothrow) CombDetectFilter(ctx, *p);
    if (!f) return HBHIP_ERR_NOMEM;
    int rc = f->setup(width, height, depth);
    if (rc != HBHIP_OK)
    {
        delete f;
        return rc;
    }
    *out = f;
    return HBHIP_OK;
}

extern "C" int hbhip_comb_detect_store(hbhip_filter *f, const uint8_t *luma, int stride)
{
    CombDetectFilter *c = dynamic_cast<CombDetectFilter *>(f);
    if (!c) return HBHIP_ERR_ARG;
    (void)hipSetDevice(f->ctx->device);
    return c->store(luma, stride, false);
}

extern "C" int hbhip_comb_detect_store_dev(hbhip_filter *f, const void *luma, int stride)
{
    CombDetectFilter *c = dynamic_cast<CombDetectFilter *>(f);
    if (!c) return HBHIP_ERR_ARG;
    (void)hipSetDevice(f->ctx->device);
    return c->store(luma, stride, true);
}

extern "C" int hbhip_comb_detect_set_gamma_lut(hbhip_filter *f, const float *lut, int entries)
{
    CombDetectFilter *c = dynamic_cast<CombDetectFilter *>(f);
    if (!c) return HBHIP_ERR_ARG;
    (void)hipSetDevice(f->ctx->device);
    return c->set_gamma_lut(lut, entries);
}

extern "C" int hbhip_comb_detect_overlay(hbhip_filter *f, const hbhip_host_frame *frame, const int plane_w[3], const int plane_h[3])
{
    CombDetectFilter *c = dynamic_cast<CombDetectFilter *>(f);
    if (!c || !frame || !plane_w || !plane_h) return HBHIP_ERR_ARG;
    (void)hipSetDevice(f->ctx->device);
    return c->overlay_host(frame, plane_w, plane_h);
}

extern "C" int hbhip_comb_detect_overlay_dev(hbhip_filter *f, const hbhip_dev_frame *frame, const int plane_w[3], const int plane_h[3])
{
    CombDetectFilter *c = dynamic_cast<CombDetectFilter *>(f);
    if (!c || !frame || !plane_w || !plane_h) return HBHIP_ERR_ARG;
    (void)hipSetDevice(f->ctx->device);
    return c->overlay_dev(frame, plane_w, plane_h);
}

extern "C" int hbhip_comb_detect_classify_many_dev(hbhip_filter *f, const void *const *lumas, int stride, int n_frames,
                                                   unsigned force_bits, int *combed)
{
    CombDetectFilter *c = dynamic_cast<CombDetectFilter *>(f);
    if (!c || !lumas || !combed) return HBHIP_ERR_ARG;
    (void)hipSetDevice(f->ctx->device);
    return c->classify_many(lumas, stride, n_frames, force_bits, combed);
}

extern "C" int hbhip_comb_detect_classify(hbhip_filter *f, int force_exhaustive, int *combed)
{
    CombDetectFilter *c = dynamic_cast<CombDetectFilter *>(f);
    if (!c || !combed) return HBHIP_ERR_ARG;
    (void)hipSetDevice(f->ctx->device);
    return c->classify(force_exhaustive, combed);
}
