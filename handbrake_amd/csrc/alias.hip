// alias.hip — the libavfilter/zimg-backed "alias" filters as real GPU filters (8-bit):
//
//   rotate_kernel       transpose / hflip / vflip as rotate_init composes them
//                       (libhb/rotate.c:169-256 -> FFmpeg transpose,hflip,vflip)
//   monochrome_kernel   FFmpeg monochrome as grayscale_init configures it
//                       (libhb/grayscale.c:32-68; same math as the reference's Metal
//                       shader platform/macosx/shaders/grayscale_vt.metal:68-136)
//   cropscale_kernel    crop + zscale(filter=lanczos) as crop_scale_init configures them
//                       (libhb/cropscale.c:52-185)
//
// PARITY UNPINNED: the arithmetic of these filters is in FFmpeg / zimg, which are not in
// the reference tree.  These kernels are bit-exact against OUR restatement
// (oracle/alias_oracle.c): pure permutations for rotate; host-built exp() table + plain
// IEEE float ops for monochrome; host-built Lanczos tap tables (double, libm sin) and
// double accumulation in the oracle's order for the scaler.
// All three are HBM-bound (1 read + 1 write per pixel; the scaler reads each source
// pixel ~taps^2/scale^2 times from L2).
#include "hbhip_internal.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace {

// ------------------------------------------------------------------ rotate
enum { T_NONE = 0, T_CCLOCK_FLIP, T_CLOCK, T_CCLOCK, T_CLOCK_FLIP };

struct RotArgs
{
    const uint8_t *src;
    uint8_t       *dst;
    int sw, sh, spitch, dw, dh, dpitch;
    int trans, hflip, vflip;
};

__device__ __forceinline__ void rot_map(const RotArgs &a, int x, int y, int &sx, int &sy)
{
    switch (a.trans)
    {
        case T_CCLOCK_FLIP: sx = y;            sy = x;            break;
        case T_CLOCK:       sx = y;            sy = a.sh - 1 - x; break;
        case T_CCLOCK:      sx = a.sw - 1 - y; sy = x;            break;
        case T_CLOCK_FLIP:  sx = a.sw - 1 - y; sy = a.sh - 1 - x; break;
        default:            sx = a.hflip ? a.sw - 1 - x : x; sy = a.vflip ? a.sh - 1 - y : y; break;
    }
}

// 64x64 output tile per block of 64x4 threads.  Transposing modes go through LDS so
// that both the gather from the source rows and the store to the output rows coalesce.
template <typename PIX>
__global__ __launch_bounds__(256) void rotate_kernel(RotArgs a)
{
    __shared__ PIX tile[64][65];
    const int ox = blockIdx.x * 64, oy = blockIdx.y * 64;
    if (a.trans == T_NONE)
    {
        for (int r = threadIdx.y; r < 64; r += 4)
        {
            const int x = ox + threadIdx.x, y = oy + r;
            if (x < a.dw && y < a.dh)
            {
                int sx, sy;
                rot_map(a, x, y, sx, sy);
                reinterpret_cast<PIX *>(a.dst + (size_t)y * a.dpitch)[x] =
                    reinterpret_cast<const PIX *>(a.src + (size_t)sy * a.spitch)[sx];
            }
        }
        return;
    }
    // gather: consecutive lanes walk consecutive output ROWS (= consecutive source columns)
    for (int r = threadIdx.y; r < 64; r += 4)
    {
        const int x = ox + r, y = oy + threadIdx.x;
        if (x < a.dw && y < a.dh)
        {
            int sx, sy;
            rot_map(a, x, y, sx, sy);
            tile[r][threadIdx.x] = reinterpret_cast<const PIX *>(a.src + (size_t)sy * a.spitch)[sx];
        }
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 64; r += 4)
    {
        const int x = ox + threadIdx.x, y = oy + r;
        if (x < a.dw && y < a.dh)
            reinterpret_cast<PIX *>(a.dst + (size_t)y * a.dpitch)[x] = tile[threadIdx.x][r];
    }
}

// rotate_kernel for 8-bit planes, all three planes of up to RB_FRAMES frames per launch, dwords to and from HBM.
// A workgroup makes one 64 x 64 output tile: the source rectangle it maps to goes into LDS as it lies (16 dwords per
// row), every thread then assembles output dwords from four LDS bytes (rot_map per byte) - the same for the
// transposing and the mirroring modes.  A 1080p plane is ~2 MB, a launch per plane mostly dispatch latency.
// (128 x 128 tiles - full 128-byte lines on both sides - measured slower: 48.9 vs 39.3 us for 16 frames.)
constexpr int RB_FRAMES = 16;

struct RotBatch
{
    int sw[3], sh[3], spitch[3], dw[3], dh[3], dpitch[3];
    int trans, hflip, vflip, n;
    const uint8_t *src[RB_FRAMES][3];
    uint8_t       *dst[RB_FRAMES][3];
};

__global__ __launch_bounds__(256) void rotate8_batch_kernel(RotBatch B)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[64][68];
    const int job = blockIdx.z, f = job / 3, c = job - 3 * f;
    RotArgs a;
    a.src = B.src[f][c]; a.dst = B.dst[f][c];
    a.sw = B.sw[c]; a.sh = B.sh[c]; a.spitch = B.spitch[c]; a.dw = B.dw[c]; a.dh = B.dh[c]; a.dpitch = B.dpitch[c];
    a.trans = B.trans; a.hflip = B.hflip; a.vflip = B.vflip;
    const int ox = blockIdx.x * 64, oy = blockIdx.y * 64;
    if (ox >= a.dw || oy >= a.dh) return;
    const int tw = min(64, a.dw - ox), th = min(64, a.dh - oy);
    // the source rectangle: the maps are monotonic per axis, so two opposite corners bound it
    int ax, ay, bx, by;
    rot_map(a, ox, oy, ax, ay);
    rot_map(a, ox + tw - 1, oy + th - 1, bx, by);
    const int sx0 = min(ax, bx), sy0 = min(ay, by), sx1 = max(ax, bx), sy1 = max(ay, by);
    const int t = threadIdx.y * 64 + threadIdx.x;
    const int cols = sx1 - sx0 + 1, rows = sy1 - sy0 + 1;
    // (offsets inside a plane fit 32 bits and rows / pitches 24: one v_mul_u32_u24 and a 32-bit offset instead of 64-bit
    // address arithmetic per access)
    if ((sx0 & 3) == 0 && cols == 64)
    {
        for (int i = t; i < rows * 16; i += 256)
        {
            const int r = i >> 4, q = i & 15;
            *reinterpret_cast<uint32_t *>(&tile[r][4 * q]) =
                *reinterpret_cast<const uint32_t *>(a.src + ((uint32_t)__mul24(sy0 + r, a.spitch) + (uint32_t)(sx0 + 4 * q)));
        }
    }
    else
    {
        for (int i = t; i < rows * 64; i += 256)
        {
            const int r = i >> 6, q = i & 63;
            if (q < cols) tile[r][q] = a.src[(uint32_t)__mul24(sy0 + r, a.spitch) + (uint32_t)(sx0 + q)];
        }
    }
    __syncthreads();
    // rot_map is affine in (x, y), so is the position of an output pixel's source byte inside the tile: its value at the
    // tile's origin and its steps in x and y (uniform) replace a map evaluation and a row multiply per byte
    int i00, ix, iy;
    {
        int sx, sy;
        rot_map(a, ox, oy, sx, sy);     i00 = (sy - sy0) * 68 + (sx - sx0);
        rot_map(a, ox + 1, oy, sx, sy); ix = (sy - sy0) * 68 + (sx - sx0) - i00;
        rot_map(a, ox, oy + 1, sx, sy); iy = (sy - sy0) * 68 + (sx - sx0) - i00;
    }
    const uint8_t *flat = &tile[0][0];
    for (int i = t; i < th * 16; i += 256)
    {
        const int r = i >> 4, q = i & 15;
        const int x = ox + 4 * q, y = oy + r;
        if (x >= a.dw) continue;
        const int base = i00 + __mul24(iy, r);
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            v |= (uint32_t)flat[base + __mul24(ix, min(4 * q + k, tw - 1))] << (8 * k);      // past the width: any byte of the tile
        uint8_t *d = a.dst + ((uint32_t)__mul24(y, a.dpitch) + (uint32_t)x);
        if (x + 3 < a.dw) *reinterpret_cast<uint32_t *>(d) = v;
        else for (int k = 0; k < 4 && x + k < a.dw; k++) d[k] = (uint8_t)(v >> (8 * k));
    }
}

class RotateFilter : public SimpleFilter
{
public:
    RotateFilter(hbhip_ctx *c, int angle, int flip) : SimpleFilter(c)
    {
        switch (angle)                                     // rotate.c:190-215
        {
            case 0:   hflip = flip; break;
            case 90:  trans = flip ? T_CLOCK_FLIP : T_CLOCK; break;
            case 180: vflip = 1; hflip = !flip; break;
            case 270: trans = flip ? T_CCLOCK_FLIP : T_CCLOCK; break;
        }
    }
    bool transposes() const { return trans != T_NONE; }
    // the batch kernel moves dwords: 8-bit planes whose rows are dword aligned
    bool batch_ok(const DevPicture *in, const DevPicture *out) const
    {
        if (in_geo.bps != 1 || getenv("HBHIP_ROTATE_OLD")) return false;
        for (int c = 0; c < 3; c++)
            if ((in->pitch[c] & 3) || (out->pitch[c] & 3) || ((uintptr_t)in->plane[c] & 3) || ((uintptr_t)out->plane[c] & 3)) return false;
        return true;
    }
    int process_many(DevPicture *const *ins, DevPicture *const *outs, int n) override
    {
        bool ok = n > 0;
        for (int f = 0; ok && f < n; f++)
        {
            ok = batch_ok(ins[f], outs[f]);
            for (int c = 0; ok && c < 3; c++) ok = ins[f]->pitch[c] == ins[0]->pitch[c] && outs[f]->pitch[c] == outs[0]->pitch[c];
        }
        if (!ok) return SimpleFilter::process_many(ins, outs, n);          // frame by frame (process() does not come back here then)
        for (int at = 0; at < n; at += RB_FRAMES)
        {
            const int nf = std::min(RB_FRAMES, n - at);
            RotBatch B;
            int max_w = 0, max_h = 0;
            for (int c = 0; c < 3; c++)
            {
                B.sw[c] = ins[at]->width[c]; B.sh[c] = ins[at]->height[c]; B.spitch[c] = ins[at]->pitch[c];
                B.dw[c] = outs[at]->width[c]; B.dh[c] = outs[at]->height[c]; B.dpitch[c] = outs[at]->pitch[c];
                max_w = std::max(max_w, B.dw[c]); max_h = std::max(max_h, B.dh[c]);
                for (int f = 0; f < nf; f++)
                {
                    B.src[f][c] = ins[at + f]->plane[c];
                    B.dst[f][c] = outs[at + f]->plane[c];
                }
            }
            B.trans = trans; B.hflip = hflip; B.vflip = vflip; B.n = nf;
            HBHIP_LAUNCH(ctx, "rotate", rotate8_batch_kernel, dim3((max_w + 63) / 64, (max_h + 63) / 64, 3 * nf), dim3(64, 4), 0, B);
            HBHIP_CHECK(ctx, hipGetLastError());
        }
        return HBHIP_OK;
    }
    int process(DevPicture *in, DevPicture *out) override
    {
        if (batch_ok(in, out))
        {
            DevPicture *i1[1] = { in }, *o1[1] = { out };
            return process_many(i1, o1, 1);
        }
        for (int c = 0; c < 3; c++)
        {
            RotArgs a;
            a.src = in->plane[c]; a.dst = out->plane[c];
            a.sw = in->width[c]; a.sh = in->height[c]; a.spitch = in->pitch[c];
            a.dw = out->width[c]; a.dh = out->height[c]; a.dpitch = out->pitch[c];
            a.trans = trans; a.hflip = hflip; a.vflip = vflip;
            if (in_geo.bps == 1) HBHIP_LAUNCH(ctx, "rotate", rotate_kernel<uint8_t>, dim3((a.dw + 63) / 64, (a.dh + 63) / 64), dim3(64, 4), 0, a);
            else                 HBHIP_LAUNCH(ctx, "rotate", rotate_kernel<uint16_t>, dim3((a.dw + 63) / 64, (a.dh + 63) / 64), dim3(64, 4), 0, a);
        }
        HBHIP_CHECK(ctx, hipGetLastError());
        return HBHIP_OK;
    }
    int trans = T_NONE, hflip = 0, vflip = 0;
};

// ------------------------------------------------------------------ monochrome
__device__ __forceinline__ float envelope(const float x)
{
    const float beta = 0.6f;
    if (x < beta)
    {
        const float tmp = fabsf(x / beta - 1.f);
        return 1.f - tmp * tmp;
    }
    const float tmp = (1.f - x) / (1.f - beta);
    return tmp * tmp * (3.f - 2.f * tmp);
}

template <typename PIX>
__global__ __launch_bounds__(256) void monochrome_kernel(const uint8_t *__restrict__ yp, int ypitch,
                                                         const uint8_t *__restrict__ up,
                                                         const uint8_t *__restrict__ vp, int cpitch,
                                                         uint8_t *__restrict__ dst, int dpitch, int w, int h,
                                                         int subw, int subh, const float *__restrict__ wlut,
                                                         float ihigh, int max)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    const float imax = 1.f / max;
    const int cx = x >> subw, cy = y >> subh;
    const float fy = reinterpret_cast<const PIX *>(yp + (size_t)y * ypitch)[x] * imax;
    const int u = reinterpret_cast<const PIX *>(up + (size_t)cy * cpitch)[cx];
    const int v = reinterpret_cast<const PIX *>(vp + (size_t)cy * cpitch)[cx];
    float ny = wlut[(size_t)u * (max + 1) + v];            // exp(-clip(dist/size)) built on the host
    const float tt = envelope(fy);
    const float t = tt + (1.f - tt) * ihigh;
    ny = (1.f - t) * fy + t * ny * fy;
    int q = __float2int_rn(ny * max);
    q = q < 0 ? 0 : q > max ? max : q;
    reinterpret_cast<PIX *>(dst + (size_t)y * dpitch)[x] = (PIX)q;
}

template <typename PIX>
__global__ void fill_plane_kernel(uint8_t *dst, int pitch, int w, int h, int value)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x < w && y < h) reinterpret_cast<PIX *>(dst + (size_t)y * pitch)[x] = (PIX)value;
}

// monochrome_kernel for 8-bit 4:2:0, up to MB_FRAMES frames per launch.  grid.z = 2 jobs per frame: the luma job
// gives a thread 16 pixels of two rows (16 bytes each) and the eight (Cb, Cr) pairs above them - so the weight table is
// looked up once per chroma sample, not once per pixel, and a wave has 2.5 KB in flight instead of a few hundred
// bytes (these kernels wait for memory, not for the ALUs) - the other job fills both chroma planes, 16 bytes per store.
constexpr int MB_FRAMES = 16;

struct MonoBatch
{
    const uint8_t *y[MB_FRAMES], *u[MB_FRAMES], *v[MB_FRAMES];
    uint8_t       *dy[MB_FRAMES], *du[MB_FRAMES], *dv[MB_FRAMES];
    int ypitch, cpitch, dpitch, dcpitch, w, h, cw, ch;
    const float *wlut;
    float ihigh;
};

__global__ __launch_bounds__(256) void monochrome8_batch_kernel(MonoBatch B)
{
    const int f = blockIdx.z >> 1;
    if (blockIdx.z & 1)
    {
        // chroma fill: rows of cw bytes, both planes
        const int x = 16 * (blockIdx.x * 64 + threadIdx.x);
        const int row = blockIdx.y * 4 + threadIdx.y;
        if (x >= B.cw || row >= 2 * B.ch) return;
        uint8_t *d = (row < B.ch ? B.du[f] + (size_t)row * B.dcpitch : B.dv[f] + (size_t)(row - B.ch) * B.dcpitch) + x;
        if (x + 15 < B.cw && (((uintptr_t)d) & 15) == 0) *reinterpret_cast<uint4 *>(d) = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
        else for (int k = 0; k < 16 && x + k < B.cw; k++) d[k] = 128;
        return;
    }
    // everything that depends on the luma value alone, once per workgroup: fy, t and (1 - t) * fy for the 256 values
    // (the per-pixel expression below then has the scalar kernel's operations and their order, minus two divisions)
    __shared__ float4 s_tab[256];
    {
        const int v = threadIdx.y * 64 + threadIdx.x;
        const float imax = 1.f / 255;
        const float fy = (float)v * imax;
        const float tt = envelope(fy);
        const float t = tt + (1.f - tt) * B.ihigh;
        s_tab[v] = make_float4((1.f - t) * fy, t, fy, 0.f);
    }
    __syncthreads();
    const int x = 16 * (blockIdx.x * 64 + threadIdx.x);
    const int y = 2 * (blockIdx.y * 4 + threadIdx.y);
    if (x >= B.w || y >= B.h) return;
    // 16 pixels of two rows and the 8 (Cb, Cr) pairs above them; every load is issued before the first use
    const uint8_t *crow_u = B.u[f] + (size_t)(y >> 1) * B.cpitch + (x >> 1), *crow_v = B.v[f] + (size_t)(y >> 1) * B.cpitch + (x >> 1);
    const uint2 u8 = *reinterpret_cast<const uint2 *>(crow_u), v8 = *reinterpret_cast<const uint2 *>(crow_v);
    const bool two = y + 1 < B.h;
    const uint8_t *yr0 = B.y[f] + (size_t)y * B.ypitch + x;
    const uint4 in0 = *reinterpret_cast<const uint4 *>(yr0);
    const uint4 in1 = two ? *reinterpret_cast<const uint4 *>(yr0 + B.ypitch) : make_uint4(0, 0, 0, 0);
    float wgt[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
    {
        // samples right of the plane's last chroma column are never used (their pixels are not stored)
        const uint32_t cu = ((k < 4 ? u8.x : u8.y) >> (8 * (k & 3))) & 0xffu, cv = ((k < 4 ? v8.x : v8.y) >> (8 * (k & 3))) & 0xffu;
        wgt[k] = B.wlut[(size_t)cu * 256 + cv];                           // exp(-clip(dist/size)) built on the host
    }
#pragma unroll
    for (int r = 0; r < 2; r++)
    {
        if (r == 1 && !two) break;
        const uint4 in = r ? in1 : in0;
        const uint32_t iw[4] = { in.x, in.y, in.z, in.w };
        uint32_t ow[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            uint32_t out = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const float4 e = s_tab[(iw[j] >> (8 * k)) & 0xffu];
                const float ny = e.x + e.y * wgt[2 * j + (k >> 1)] * e.z;      // (1 - t) * fy + t * ny * fy
                int q = __float2int_rn(ny * 255);
                q = q < 0 ? 0 : q > 255 ? 255 : q;
                out |= (uint32_t)q << (8 * k);
            }
            ow[j] = out;
        }
        uint8_t *d = B.dy[f] + (size_t)(y + r) * B.dpitch + x;
        if (x + 15 < B.w) *reinterpret_cast<uint4 *>(d) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        else for (int k = 0; k < 16 && x + k < B.w; k++) d[k] = (uint8_t)(ow[k >> 2] >> (8 * (k & 3)));
    }
}

class MonochromeFilter : public SimpleFilter
{
public:
    MonochromeFilter(hbhip_ctx *c, double cb_, double cr_, double size_, double high_)
        : SimpleFilter(c), cb(cb_), cr(cr_), size(size_), high(high_) {}
    ~MonochromeFilter() override { if (d_lut) (void)hipFree(d_lut); }
    int setup()
    {
        // one entry per (Cb, Cr) pair: 256 KB at 8 bits, 4 MB at 10, 64 MB at 12
        const int n = 1 << in_geo.depth;
        std::vector<float> lut((size_t)n * n);
        const float imax = 1.f / (n - 1);
        const float isize = 1.f / (float)size;
        const float b = (float)cb * .5f, r = (float)cr * .5f;
        for (int u = 0; u < n; u++)
            for (int v = 0; v < n; v++)
            {
                const float fu = u * imax - .5f, fv = v * imax - .5f;
                float d = ((b - fu) * (b - fu) + (r - fv) * (r - fv)) * isize;
                d = d < 0.f ? 0.f : d > 1.f ? 1.f : d;
                lut[(size_t)u * n + v] = expf(-d);
            }
        HBHIP_CHECK(ctx, hipMalloc((void **)&d_lut, sizeof(float) * lut.size()));
        HBHIP_CHECK(ctx, hipMemcpyAsync(d_lut, lut.data(), sizeof(float) * lut.size(), hipMemcpyHostToDevice, ctx->stream));
        HBHIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        return HBHIP_OK;
    }
    // the batch kernel moves 16 luma / 8 chroma bytes at a time: 8-bit 4:2:0 planes aligned accordingly
    bool batch_ok(const DevPicture *in, const DevPicture *out) const
    {
        return in_geo.bps == 1 && in_geo.log2_cw == 1 && in_geo.log2_ch == 1 && !getenv("HBHIP_MONOCHROME_OLD") &&
               (in->pitch[0] & 15) == 0 && (out->pitch[0] & 15) == 0 && ((uintptr_t)in->plane[0] & 15) == 0 && ((uintptr_t)out->plane[0] & 15) == 0 &&
               (in->pitch[1] & 7) == 0 && ((uintptr_t)in->plane[1] & 7) == 0 && ((uintptr_t)in->plane[2] & 7) == 0;
    }
    int process_many(DevPicture *const *ins, DevPicture *const *outs, int n) override
    {
        bool ok = n > 0;
        for (int i = 0; ok && i < n; i++)
        {
            ok = batch_ok(ins[i], outs[i]);
            for (int c = 0; ok && c < 3; c++) ok = ins[i]->pitch[c] == ins[0]->pitch[c] && outs[i]->pitch[c] == outs[0]->pitch[c];
        }
        if (!ok) return SimpleFilter::process_many(ins, outs, n);          // frame by frame (process() does not come back here then)
        for (int at = 0; at < n; at += MB_FRAMES)
        {
            const int nf = std::min(MB_FRAMES, n - at);
            MonoBatch B;
            for (int f = 0; f < nf; f++)
            {
                B.y[f] = ins[at + f]->plane[0]; B.u[f] = ins[at + f]->plane[1]; B.v[f] = ins[at + f]->plane[2];
                B.dy[f] = outs[at + f]->plane[0]; B.du[f] = outs[at + f]->plane[1]; B.dv[f] = outs[at + f]->plane[2];
            }
            B.ypitch = ins[at]->pitch[0]; B.cpitch = ins[at]->pitch[1]; B.dpitch = outs[at]->pitch[0]; B.dcpitch = outs[at]->pitch[1];
            B.w = ins[at]->width[0]; B.h = ins[at]->height[0]; B.cw = outs[at]->width[1]; B.ch = outs[at]->height[1];
            B.wlut = d_lut; B.ihigh = 1.f - (float)high;
            // luma job: 16 x 2 pixels per thread; fill job: 16 bytes per thread over 2 * ch rows - one grid covers both
            const int gx = std::max(((B.w + 15) / 16 + 63) / 64, ((B.cw + 15) / 16 + 63) / 64);
            const int gy = std::max(((B.h + 1) / 2 + 3) / 4, (2 * B.ch + 3) / 4);
            HBHIP_LAUNCH(ctx, "monochrome", monochrome8_batch_kernel, dim3(gx, gy, 2 * nf), dim3(64, 4), 0, B);
            HBHIP_CHECK(ctx, hipGetLastError());
        }
        return HBHIP_OK;
    }
    int process(DevPicture *in, DevPicture *out) override
    {
        if (batch_ok(in, out))
        {
            DevPicture *i1[1] = { in }, *o1[1] = { out };
            return process_many(i1, o1, 1);
        }
        const int w = in->width[0], h = in->height[0];
        const int max = (1 << in_geo.depth) - 1, mid = 1 << (in_geo.depth - 1);
        if (in_geo.bps == 1)
            HBHIP_LAUNCH(ctx, "monochrome", monochrome_kernel<uint8_t>, dim3((w + 63) / 64, (h + 3) / 4), dim3(64, 4), 0,
                         (const uint8_t *)in->plane[0], in->pitch[0], (const uint8_t *)in->plane[1],
                         (const uint8_t *)in->plane[2], in->pitch[1], out->plane[0], out->pitch[0], w, h,
                         in_geo.log2_cw, in_geo.log2_ch, (const float *)d_lut, 1.f - (float)high, max);
        else
            HBHIP_LAUNCH(ctx, "monochrome", monochrome_kernel<uint16_t>, dim3((w + 63) / 64, (h + 3) / 4), dim3(64, 4), 0,
                         (const uint8_t *)in->plane[0], in->pitch[0], (const uint8_t *)in->plane[1],
                         (const uint8_t *)in->plane[2], in->pitch[1], out->plane[0], out->pitch[0], w, h,
                         in_geo.log2_cw, in_geo.log2_ch, (const float *)d_lut, 1.f - (float)high, max);
        for (int c = 1; c < 3; c++)
        {
            const dim3 grid((out->width[c] + 255) / 256, out->height[c]);
            if (in_geo.bps == 1) HBHIP_LAUNCH(ctx, "monochrome_fill", fill_plane_kernel<uint8_t>, grid, dim3(256), 0,
                                              out->plane[c], out->pitch[c], out->width[c], out->height[c], mid);
            else                 HBHIP_LAUNCH(ctx, "monochrome_fill", fill_plane_kernel<uint16_t>, grid, dim3(256), 0,
                                              out->plane[c], out->pitch[c], out->width[c], out->height[c], mid);
        }
        HBHIP_CHECK(ctx, hipGetLastError());
        return HBHIP_OK;
    }
    double cb, cr, size, high;
    float *d_lut = nullptr;
};

// ------------------------------------------------------------------ crop + lanczos scale
struct ScaleArgs
{
    const uint8_t *src;      // already offset to the crop window
    uint8_t       *dst;
    int spitch, dpitch, dw, dh, tx, ty;
    const int    *ix, *iy;
    const double *cx, *cy;
    double vmax;             // (1 << depth) - 1
};

// Pass 1: H[r][x] = sum_i cx[x][i] * src[r][ix[x][i]] for every source row r of the crop window.
// Pass 2: out[y][x] = round(clamp(sum_j cy[y][j] * H[iy[y][j]][x])).
// Same products, same order of additions as the one-loop form in oracle/alias_oracle.c, so the
// split changes nothing numerically; it turns taps^2 gathers per pixel into 2*taps.
template <typename PIX>
__global__ __launch_bounds__(256) void cropscale_h_kernel(ScaleArgs a, double *__restrict__ hbuf, int src_rows)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= a.dw || r >= src_rows) return;
    const int *ix = a.ix + (size_t)x * a.tx;
    const double *cx = a.cx + (size_t)x * a.tx;
    const PIX *row = reinterpret_cast<const PIX *>(a.src + (size_t)r * a.spitch);
    double h = 0.0;
    for (int i = 0; i < a.tx; i++)
        h += cx[i] * (double)row[ix[i]];
    hbuf[(size_t)r * a.dw + x] = h;
}

template <typename PIX>
__global__ __launch_bounds__(256) void cropscale_v_kernel(ScaleArgs a, const double *__restrict__ hbuf)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= a.dw || y >= a.dh) return;
    double acc = 0.0;
    for (int j = 0; j < a.ty; j++)
        acc += a.cy[(size_t)y * a.ty + j] * hbuf[(size_t)a.iy[(size_t)y * a.ty + j] * a.dw + x];
    acc = acc < 0.0 ? 0.0 : acc > a.vmax ? a.vmax : acc;
    reinterpret_cast<PIX *>(a.dst + (size_t)y * a.dpitch)[x] = (PIX)(int)(acc + 0.5);
}

// Both passes in one kernel: a workgroup owns a FS_TW x FS_TH tile of the output.  It first forms
// the horizontally filtered values H[r][x] of the few source rows its output rows tap (those sit in
// LDS as doubles instead of making a round trip through HBM), then the vertical sums.  Each H value
// and each output is the same sequence of double operations as in the two-pass kernels above.
constexpr int FS_TW = 64, FS_TH = 32, FS_MAXR = 40;
struct ScaleArgs3 { ScaleArgs p[3]; int active[3]; };

// Any tap counts (read from the arguments); the 6 x 6 case has its own kernel below.
template <int TX, int TY, typename PIX>
__global__ __launch_bounds__(256) void cropscale_fused_kernel(ScaleArgs3 all)
{
    __shared__ double s_h[FS_MAXR][FS_TW];
    __shared__ int s_rmin, s_rmax;
    const int pl = blockIdx.z;
    if (!all.active[pl]) return;
    const ScaleArgs &a = all.p[pl];
    const int x0 = blockIdx.x * FS_TW, y0 = blockIdx.y * FS_TH;
    if (x0 >= a.dw || y0 >= a.dh) return;
    const int t = threadIdx.x;
    const int tx = TX > 0 ? TX : a.tx, ty = TY > 0 ? TY : a.ty;
    const int rows = min(FS_TH, a.dh - y0);
    if (t == 0) { s_rmin = 0x7fffffff; s_rmax = -1; }
    __syncthreads();
    {
        int lo = 0x7fffffff, hi = -1;
        for (int i = t; i < rows * ty; i += 256)
        {
            const int r = a.iy[(size_t)y0 * ty + i];
            lo = min(lo, r); hi = max(hi, r);
        }
        if (hi >= 0) { atomicMin(&s_rmin, lo); atomicMax(&s_rmax, hi); }
    }
    __syncthreads();
    const int rmin = s_rmin, nr = s_rmax - rmin + 1;            // nr <= FS_MAXR (checked by the host)
    const int xl = t & (FS_TW - 1), x = x0 + xl;
    const int wave_row = __builtin_amdgcn_readfirstlane(t / FS_TW);   // a wave = one tile row: uniform
    if (x < a.dw)
    {
        const int *ix = a.ix + (size_t)x * tx;
        const double *cx = a.cx + (size_t)x * tx;
        for (int rr = wave_row; rr < nr; rr += 256 / FS_TW)
        {
            const PIX *row = reinterpret_cast<const PIX *>(a.src + (size_t)(rmin + rr) * a.spitch);
            double h = 0.0;
            for (int i = 0; i < tx; i++) h += cx[i] * (double)row[ix[i]];
            s_h[rr][xl] = h;
        }
    }
    __syncthreads();
    if (x >= a.dw) return;
    for (int yy = wave_row; yy < rows; yy += 256 / FS_TW)
    {
        const int y = y0 + yy;                                   // uniform: the taps come through scalar loads
        const double *cy = a.cy + (size_t)y * ty;
        const int *iy = a.iy + (size_t)y * ty;
        double acc = 0.0;
        for (int j = 0; j < ty; j++) acc += cy[j] * s_h[iy[j] - rmin][xl];
        acc = acc < 0.0 ? 0.0 : acc > a.vmax ? a.vmax : acc;
        reinterpret_cast<PIX *>(a.dst + (size_t)y * a.dpitch)[x] = (PIX)(int)(acc + 0.5);
    }
}

// The 6 x 6-tap case (every upscale, e.g. 1080p -> 2160p), the form that runs for it.  What bounded the generic kernel
// above was not arithmetic but a chain of dependent latencies per workgroup at low occupancy: tap rows from memory, a
// barrier, per-thread tap tables from memory, byte gathers from memory for every horizontal tap, a barrier, then scalar
// loads of the vertical taps for every output row - with 20 KB of LDS allowing five workgroups per CU.  Here
//   * the host precomputes, per tile row and tile column, which source rows / columns the tile taps (tile_y, tile_x);
//   * ONE round of loads fetches everything a workgroup needs: the tapped source samples (dwords -> LDS), the vertical
//     taps of its 32 output rows (-> LDS) and each thread's own six horizontal taps (-> registers);
//   * the horizontal pass gathers from LDS, the vertical pass reads its taps from LDS (broadcast);
//   * LDS is sized to what the tile really taps (dynamic): 22 rows for a 2x upscale instead of 40.
// Each H value and each output is the same sequence of double operations as in the kernels above.
struct Tile6 { const int *tile_y[3]; const int *tile_x[3]; int nr_max, span_max; };

template <typename PIX>
__global__ __launch_bounds__(256) void cropscale_fused6_kernel(ScaleArgs3 all, Tile6 T)
{
    extern __shared__ __attribute__((aligned(16))) double smem6[];
    const int pl = blockIdx.z;
    if (!all.active[pl]) return;
    const ScaleArgs &a = all.p[pl];
    const int x0 = blockIdx.x * FS_TW, y0 = blockIdx.y * FS_TH;
    if (x0 >= a.dw || y0 >= a.dh) return;
    const int t = threadIdx.x;
    const int rows = min(FS_TH, a.dh - y0);
    const int rmin = T.tile_y[pl][2 * blockIdx.y], nr = T.tile_y[pl][2 * blockIdx.y + 1];
    const int c0 = T.tile_x[pl][2 * blockIdx.x], span = T.tile_x[pl][2 * blockIdx.x + 1];
    double *s_h = smem6;                                             // [nr_max][FS_TW]
    double *s_cy = s_h + (size_t)T.nr_max * FS_TW;                   // [FS_TH][6]
    int *s_iy = reinterpret_cast<int *>(s_cy + FS_TH * 6);           // [FS_TH][6]
    const int sp = (T.span_max + 4 + 3) & ~3;                        // samples per staged row
    PIX *s_src = reinterpret_cast<PIX *>(s_iy + FS_TH * 6);          // [nr_max][sp]
    const int xl = t & (FS_TW - 1), x = x0 + xl;
    const int wave_row = __builtin_amdgcn_readfirstlane(t / FS_TW);

    // one round of loads
    int ixr[6];
    double cxr[6];
    if (x < a.dw)
    {
#pragma unroll
        for (int i = 0; i < 6; i++) { ixr[i] = a.ix[(size_t)x * 6 + i] - c0; cxr[i] = a.cx[(size_t)x * 6 + i]; }
    }
    if (t < rows * 6)
    {
        s_cy[t] = a.cy[(size_t)y0 * 6 + t];
        s_iy[t] = a.iy[(size_t)y0 * 6 + t] - rmin;
    }
    {
        const int ndw = (span * (int)sizeof(PIX) + 3) / 4;
        const int avail = a.spitch - c0 * (int)sizeof(PIX);          // bytes from c0 to the end of the row's pitch
        const bool aligned = ((a.spitch | (int)((uintptr_t)a.src & 3)) & 3) == 0;
        for (int i = t; i < nr * ndw; i += 256)
        {
            const int rr = i / ndw, d = i - rr * ndw;
            const uint8_t *g = a.src + (size_t)(rmin + rr) * a.spitch + (size_t)c0 * sizeof(PIX) + 4 * d;
            uint32_t v;
            if (aligned && 4 * d + 4 <= avail) v = *reinterpret_cast<const uint32_t *>(g);
            else
            {
                v = 0;
                for (int k = 0; k < 4 && 4 * d + k < avail; k++) v |= (uint32_t)g[k] << (8 * k);
            }
            reinterpret_cast<uint32_t *>(s_src + (size_t)rr * sp)[d] = v;
        }
    }
    __syncthreads();
    if (x < a.dw)
        for (int rr = wave_row; rr < nr; rr += 256 / FS_TW)
        {
            const PIX *row = s_src + (size_t)rr * sp;
            double h = 0.0;
#pragma unroll
            for (int i = 0; i < 6; i++) h += cxr[i] * (double)row[ixr[i]];
            s_h[rr * FS_TW + xl] = h;
        }
    __syncthreads();
    if (x >= a.dw) return;
    for (int yy = wave_row; yy < rows; yy += 256 / FS_TW)
    {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 6; j++) acc += s_cy[yy * 6 + j] * s_h[s_iy[yy * 6 + j] * FS_TW + xl];
        acc = acc < 0.0 ? 0.0 : acc > a.vmax ? a.vmax : acc;
        reinterpret_cast<PIX *>(a.dst + (size_t)(y0 + yy) * a.dpitch)[x] = (PIX)(int)(acc + 0.5);
    }
}

__global__ void crop_copy_kernel(const uint8_t *src, int spitch, uint8_t *dst, int dpitch, int w, int h)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x < w && y < h) dst[(size_t)y * dpitch + x] = src[(size_t)y * spitch + x];
}

double lanczos3(double x)
{
    const double pi = 3.14159265358979323846;
    x = std::fabs(x);
    if (x >= 3.0) return 0.0;
    if (x == 0.0) return 1.0;
    const double a = x * pi;
    return (std::sin(a) / a) * (std::sin(a / 3.0) / (a / 3.0));
}

// zimg-style tap table of one dimension (see oracle/alias_oracle.c for the conventions)
int lanczos_table(int src_dim, int dst_dim, double shift, std::vector<int> &idx, std::vector<double> &coef)
{
    const double scale = (double)dst_dim / (double)src_dim;
    const double step = scale < 1.0 ? scale : 1.0;
    const double support = 3.0 / step;
    int taps = (int)std::ceil(support) * 2;
    if (taps < 1) taps = 1;
    if (taps > 64) taps = 64;
    idx.assign((size_t)dst_dim * taps, 0);
    coef.assign((size_t)dst_dim * taps, 0.0);
    for (int i = 0; i < dst_dim; i++)
    {
        const double pos = (i + 0.5) / scale + shift;
        const double begin = std::floor(pos - taps / 2.0 + 0.5);
        double w[64], total = 0.0;
        for (int k = 0; k < taps; k++)
        {
            w[k] = lanczos3((begin + k + 0.5 - pos) * step);
            total += w[k];
        }
        for (int k = 0; k < taps; k++)
        {
            long j = (long)begin + k;
            if (j < 0) j = -j - 1;
            if (j >= src_dim) j = 2L * src_dim - 1 - j;
            if (j < 0) j = 0;
            if (j >= src_dim) j = src_dim - 1;
            idx[(size_t)i * taps + k] = (int)j;
            coef[(size_t)i * taps + k] = w[k] / total;
        }
    }
    return taps;
}

class CropScaleFilter : public SimpleFilter
{
public:
    CropScaleFilter(hbhip_ctx *c, const hbhip_cropscale_params &p) : SimpleFilter(c), par(p) {}
    ~CropScaleFilter() override
    {
        for (int c = 0; c < 3; c++)
        {
            if (d_ix[c]) (void)hipFree(d_ix[c]);
            if (d_iy[c]) (void)hipFree(d_iy[c]);
            if (d_cx[c]) (void)hipFree(d_cx[c]);
            if (d_cy[c]) (void)hipFree(d_cy[c]);
            if (d_tile_y[c]) (void)hipFree(d_tile_y[c]);
            if (d_tile_x[c]) (void)hipFree(d_tile_x[c]);
        }
        if (hbuf) (void)hipFree(hbuf);
    }
    int setup()
    {
        const int cw = in_geo.width - par.crop_left - par.crop_right;
        const int ch = in_geo.height - par.crop_top - par.crop_bottom;
        if (cw < 1 || ch < 1) return HBHIP_ERR_ARG;
        for (int c = 0; c < 3; c++)
        {
            const int lw = c ? in_geo.log2_cw : 0, lh = c ? in_geo.log2_ch : 0;
            crop_x[c] = par.crop_left >> lw;
            crop_y[c] = par.crop_top >> lh;
            crop_w[c] = c ? -((-cw) >> lw) : cw;
            crop_h[c] = c ? -((-ch) >> lh) : ch;
            const int dw = out_geo.pw[c], dh = out_geo.ph[c];
            // left-sited chroma: 0.25 * (1 - src/dst) of a chroma sample, horizontally only
            const double sx = (c && lw) ? 0.25 * (1.0 - (double)cw / (double)out_geo.width) : 0.0;
            identity[c] = (dw == crop_w[c] && dh == crop_h[c] && sx == 0.0);
            if (identity[c]) continue;
            std::vector<int> ix, iy;
            std::vector<double> cx, cy;
            tx[c] = lanczos_table(crop_w[c], dw, sx, ix, cx);
            ty[c] = lanczos_table(crop_h[c], dh, 0.0, iy, cy);
            // the fused kernel keeps the tapped source rows of a 32-row output tile in LDS
            for (int y0 = 0; y0 < dh && fused; y0 += FS_TH)
            {
                int lo = 0x7fffffff, hi = -1;
                for (int i = 0; i < std::min(FS_TH, dh - y0) * ty[c]; i++)
                {
                    lo = std::min(lo, iy[(size_t)y0 * ty[c] + i]);
                    hi = std::max(hi, iy[(size_t)y0 * ty[c] + i]);
                }
                if (hi - lo + 1 > FS_MAXR) fused = false;
            }
            if (tx[c] == 6 && ty[c] == 6)
            {
                // which source rows / columns each output tile taps (cropscale_fused6_kernel); the column origin is
                // rounded down to a dword of samples
                std::vector<int> tyv, txv;
                const int al = 4 / in_geo.bps;
                for (int y0 = 0; y0 < dh; y0 += FS_TH)
                {
                    int lo = 0x7fffffff, hi = -1;
                    for (int i = 0; i < std::min(FS_TH, dh - y0) * 6; i++) { lo = std::min(lo, iy[(size_t)y0 * 6 + i]); hi = std::max(hi, iy[(size_t)y0 * 6 + i]); }
                    tyv.push_back(lo); tyv.push_back(hi - lo + 1);
                    nr_max = std::max(nr_max, hi - lo + 1);
                }
                for (int x0 = 0; x0 < dw; x0 += FS_TW)
                {
                    int lo = 0x7fffffff, hi = -1;
                    for (int i = 0; i < std::min(FS_TW, dw - x0) * 6; i++) { lo = std::min(lo, ix[(size_t)x0 * 6 + i]); hi = std::max(hi, ix[(size_t)x0 * 6 + i]); }
                    lo = lo / al * al;
                    txv.push_back(lo); txv.push_back(hi - lo + 1);
                    span_max = std::max(span_max, hi - lo + 1);
                }
                HBHIP_CHECK(ctx, hipMalloc((void **)&d_tile_y[c], sizeof(int) * tyv.size()));
                HBHIP_CHECK(ctx, hipMalloc((void **)&d_tile_x[c], sizeof(int) * txv.size()));
                HBHIP_CHECK(ctx, hipMemcpy(d_tile_y[c], tyv.data(), sizeof(int) * tyv.size(), hipMemcpyHostToDevice));
                HBHIP_CHECK(ctx, hipMemcpy(d_tile_x[c], txv.data(), sizeof(int) * txv.size(), hipMemcpyHostToDevice));
            }
            HBHIP_CHECK(ctx, hipMalloc((void **)&d_ix[c], sizeof(int) * ix.size()));
            HBHIP_CHECK(ctx, hipMalloc((void **)&d_iy[c], sizeof(int) * iy.size()));
            HBHIP_CHECK(ctx, hipMalloc((void **)&d_cx[c], sizeof(double) * cx.size()));
            HBHIP_CHECK(ctx, hipMalloc((void **)&d_cy[c], sizeof(double) * cy.size()));
            HBHIP_CHECK(ctx, hipMemcpy(d_ix[c], ix.data(), sizeof(int) * ix.size(), hipMemcpyHostToDevice));
            HBHIP_CHECK(ctx, hipMemcpy(d_iy[c], iy.data(), sizeof(int) * iy.size(), hipMemcpyHostToDevice));
            HBHIP_CHECK(ctx, hipMemcpy(d_cx[c], cx.data(), sizeof(double) * cx.size(), hipMemcpyHostToDevice));
            HBHIP_CHECK(ctx, hipMemcpy(d_cy[c], cy.data(), sizeof(double) * cy.size(), hipMemcpyHostToDevice));
        }
        if (getenv("HBHIP_SCALE_TWO_PASS")) fused = false;
        size_t need = 0;
        for (int c = 0; c < 3; c++)
            if (!identity[c]) need = std::max(need, (size_t)out_geo.pw[c] * crop_h[c]);
        if (need && !fused) HBHIP_CHECK(ctx, hipMalloc((void **)&hbuf, sizeof(double) * need));
        return HBHIP_OK;
    }
    int process(DevPicture *in, DevPicture *out) override
    {
        ScaleArgs3 all;
        memset(&all, 0, sizeof(all));
        for (int c = 0; c < 3; c++)
        {
            const uint8_t *win = in->plane[c] + (size_t)crop_y[c] * in->pitch[c] + (size_t)crop_x[c] * in_geo.bps;
            const int dw = out->width[c], dh = out->height[c];
            const double vmax = (double)((1 << in_geo.depth) - 1);
            if (identity[c])
            {
                HBHIP_LAUNCH(ctx, "crop_copy", crop_copy_kernel, dim3((dw * in_geo.bps + 255) / 256, dh), dim3(256), 0,
                             win, in->pitch[c], out->plane[c], out->pitch[c], dw * in_geo.bps, dh);
                continue;
            }
            if (fused)
            {
                ScaleArgs &f = all.p[c];
                f.src = win; f.dst = out->plane[c];
                f.spitch = in->pitch[c]; f.dpitch = out->pitch[c];
                f.dw = dw; f.dh = dh; f.tx = tx[c]; f.ty = ty[c];
                f.ix = d_ix[c]; f.iy = d_iy[c]; f.cx = d_cx[c]; f.cy = d_cy[c]; f.vmax = vmax;
                all.active[c] = 1;
                continue;
            }
            ScaleArgs a;
            a.src = win; a.dst = out->plane[c];
            a.spitch = in->pitch[c]; a.dpitch = out->pitch[c];
            a.dw = dw; a.dh = dh; a.tx = tx[c]; a.ty = ty[c];
            a.ix = d_ix[c]; a.iy = d_iy[c]; a.cx = d_cx[c]; a.cy = d_cy[c]; a.vmax = vmax;
            const dim3 gh((dw + 63) / 64, (crop_h[c] + 3) / 4), gv((dw + 63) / 64, (dh + 3) / 4);
            if (in_geo.bps == 1)
            {
                HBHIP_LAUNCH(ctx, "cropscale_lanczos_h", cropscale_h_kernel<uint8_t>, gh, dim3(64, 4), 0, a, hbuf, crop_h[c]);
                HBHIP_LAUNCH(ctx, "cropscale_lanczos_v", cropscale_v_kernel<uint8_t>, gv, dim3(64, 4), 0, a, (const double *)hbuf);
            }
            else
            {
                HBHIP_LAUNCH(ctx, "cropscale_lanczos_h", cropscale_h_kernel<uint16_t>, gh, dim3(64, 4), 0, a, hbuf, crop_h[c]);
                HBHIP_LAUNCH(ctx, "cropscale_lanczos_v", cropscale_v_kernel<uint16_t>, gv, dim3(64, 4), 0, a, (const double *)hbuf);
            }
        }
        if (all.active[0] || all.active[1] || all.active[2])
        {
            const dim3 grid((out->width[0] + FS_TW - 1) / FS_TW, (out->height[0] + FS_TH - 1) / FS_TH, 3);
            bool six = true;
            for (int c = 0; c < 3; c++) six &= !all.active[c] || (tx[c] == 6 && ty[c] == 6);
            const int sp = (span_max + 4 + 3) & ~3;
            const size_t lds6 = sizeof(double) * ((size_t)nr_max * FS_TW + FS_TH * 6) + sizeof(int) * FS_TH * 6 +
                                (size_t)nr_max * sp * in_geo.bps + 16;
            if (six && lds6 <= 64 * 1024 && getenv("HBHIP_SCALE_GENERIC") == nullptr)
            {
                Tile6 T;
                for (int c = 0; c < 3; c++) { T.tile_y[c] = d_tile_y[c]; T.tile_x[c] = d_tile_x[c]; }
                T.nr_max = nr_max; T.span_max = span_max;
                if (in_geo.bps == 1) HBHIP_LAUNCH(ctx, "cropscale_lanczos_fused", cropscale_fused6_kernel<uint8_t>, grid, dim3(256), lds6, all, T);
                else                 HBHIP_LAUNCH(ctx, "cropscale_lanczos_fused", cropscale_fused6_kernel<uint16_t>, grid, dim3(256), lds6, all, T);
            }
            else if (in_geo.bps == 1)
                HBHIP_LAUNCH(ctx, "cropscale_lanczos_fused", (cropscale_fused_kernel<0, 0, uint8_t>), grid, dim3(256), 0, all);
            else
                HBHIP_LAUNCH(ctx, "cropscale_lanczos_fused", (cropscale_fused_kernel<0, 0, uint16_t>), grid, dim3(256), 0, all);
        }
        HBHIP_CHECK(ctx, hipGetLastError());
        return HBHIP_OK;
    }
    bool fused = true;          // all scaled planes fit the fused kernel's LDS budget
    hbhip_cropscale_params par;
    int crop_x[3], crop_y[3], crop_w[3], crop_h[3], tx[3] = {0, 0, 0}, ty[3] = {0, 0, 0};
    bool identity[3] = {false, false, false};
    int *d_ix[3] = {nullptr, nullptr, nullptr}, *d_iy[3] = {nullptr, nullptr, nullptr};
    double *d_cx[3] = {nullptr, nullptr, nullptr}, *d_cy[3] = {nullptr, nullptr, nullptr};
    double *hbuf = nullptr;     // horizontally filtered rows of one plane (dst_w x crop_h doubles)
    int *d_tile_y[3] = {nullptr, nullptr, nullptr}, *d_tile_x[3] = {nullptr, nullptr, nullptr};
    int nr_max = 0, span_max = 0;
};

// ------------------------------------------------------------------ pad
// FFmpeg vf_pad as pad_init configures it (libhb/pad.c:40-148): the picture at (x, y) of a larger
// one, the rest one colour.  One launch for the three planes; HBM-bound (read in, write out).
struct PadArgs
{
    const uint8_t *src[3];
    uint8_t       *dst[3];
    int spitch[3], dpitch[3], sw[3], sh[3], dw[3], dh[3], x[3], y[3], fill[3];
};

template <typename PIX>
__global__ __launch_bounds__(256) void pad_kernel(PadArgs a)
{
    const int c = blockIdx.z;
    const int xx = blockIdx.x * blockDim.x + threadIdx.x, yy = blockIdx.y * blockDim.y + threadIdx.y;
    if (xx >= a.dw[c] || yy >= a.dh[c]) return;
    const int sx = xx - a.x[c], sy = yy - a.y[c];
    const bool inside = sx >= 0 && sx < a.sw[c] && sy >= 0 && sy < a.sh[c];
    const PIX v = inside ? reinterpret_cast<const PIX *>(a.src[c] + (size_t)sy * a.spitch[c])[sx] : (PIX)a.fill[c];
    reinterpret_cast<PIX *>(a.dst[c] + (size_t)yy * a.dpitch[c])[xx] = v;
}

class PadFilter : public SimpleFilter
{
public:
    PadFilter(hbhip_ctx *c, const hbhip_pad_params &p) : SimpleFilter(c), par(p) {}
    int process(DevPicture *in, DevPicture *out) override
    {
        PadArgs a;
        for (int c = 0; c < 3; c++)
        {
            a.src[c] = in->plane[c]; a.dst[c] = out->plane[c];
            a.spitch[c] = in->pitch[c]; a.dpitch[c] = out->pitch[c];
            a.sw[c] = in->width[c]; a.sh[c] = in->height[c]; a.dw[c] = out->width[c]; a.dh[c] = out->height[c];
            a.x[c] = c ? par.x >> in_geo.log2_cw : par.x;
            a.y[c] = c ? par.y >> in_geo.log2_ch : par.y;
            a.fill[c] = par.fill[c];
        }
        const dim3 grid((a.dw[0] + 63) / 64, (a.dh[0] + 3) / 4, 3);
        if (in_geo.bps == 1) HBHIP_LAUNCH(ctx, "pad", pad_kernel<uint8_t>, grid, dim3(64, 4), 0, a);
        else                 HBHIP_LAUNCH(ctx, "pad", pad_kernel<uint16_t>, grid, dim3(64, 4), 0, a);
        HBHIP_CHECK(ctx, hipGetLastError());
        return HBHIP_OK;
    }
    hbhip_pad_params par;
};

// ------------------------------------------------------------------ format (depth conversion)
// `format=pix_fmts=...` (libhb/format.c:13-111): libavfilter inserts a same-size `scale`, i.e. libswscale's
// unscaled planar copy (swscale_unscaled.c:planarCopyWrapper; parity unpinned, restated in
// oracle/alias_oracle.c:orc_format_plane).  Up: shift (limited range, chroma) or top-bit replication
// (full-range luma); down: ordered dither then the overflow clamp tmp - (tmp >> depth).  One launch for the
// three planes, four samples per thread, HBM-bound (read in + write out).
struct FormatArgs
{
    const uint8_t *src[3];
    uint8_t       *dst[3];
    int spitch[3], dpitch[3], w[3], h[3];
    int sdepth, ddepth, full_range;
};

template <typename SRC, typename DST>
__global__ __launch_bounds__(256) void format_kernel(FormatArgs a)
{
    const int c = blockIdx.z;
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x0 >= a.w[c] || y >= a.h[c]) return;
    const SRC *s = reinterpret_cast<const SRC *>(a.src[c] + (size_t)y * a.spitch[c]);
    DST *d = reinterpret_cast<DST *>(a.dst[c] + (size_t)y * a.dpitch[c]);
    const bool shiftonly = c != 0 || !a.full_range;
    const int up = a.ddepth - a.sdepth;
    for (int i = 0; i < 4 && x0 + i < a.w[c]; i++)
    {
        const int x = x0 + i;
        const unsigned v = s[x];
        unsigned o;
        if (up == 0) o = v;
        else if (up > 0) o = shiftonly ? v << up : (v << up) | (v >> (2 * a.sdepth - a.ddepth));
        else
        {
            const int shift = -up;
            // dithers[shift - 1][y & 7][x & 7] of swscale_unscaled.c for shift 2 / 4 (2x2 / 4x4 ordered matrices)
            // one nibble per cell: {1,2;3,0} and {4,8,7,11; 12,0,15,3; 6,10,5,9; 14,2,13,1}
            constexpr uint64_t D4 = 0xB784ull | (0x3F0Cull << 16) | (0x95A6ull << 32) | (0x1D2Eull << 48);
            const unsigned dm = shift == 2 ? ((0x0321u >> (4 * ((y & 1) * 2 + (x & 1)))) & 15u)
                                           : (unsigned)((D4 >> (16 * (y & 3) + 4 * (x & 3))) & 15u);
            const unsigned tmp = (v + dm) >> shift;
            o = tmp - (tmp >> a.ddepth);
        }
        d[x] = (DST)o;
    }
}

class FormatFilter : public SimpleFilter
{
public:
    FormatFilter(hbhip_ctx *c, int full) : SimpleFilter(c), full_range(full) {}
    int process(DevPicture *in, DevPicture *out) override
    {
        FormatArgs a;
        for (int c = 0; c < 3; c++)
        {
            a.src[c] = in->plane[c]; a.dst[c] = out->plane[c];
            a.spitch[c] = in->pitch[c]; a.dpitch[c] = out->pitch[c];
            a.w[c] = in_geo.pw[c]; a.h[c] = in_geo.ph[c];
        }
        a.sdepth = in_geo.depth; a.ddepth = out_geo.depth; a.full_range = full_range;
        const dim3 grid((a.w[0] + 255) / 256, (a.h[0] + 3) / 4, 3);
        if (in_geo.bps == 1 && out_geo.bps == 1)      HBHIP_LAUNCH(ctx, "format", (format_kernel<uint8_t, uint8_t>), grid, dim3(64, 4), 0, a);
        else if (in_geo.bps == 1)                     HBHIP_LAUNCH(ctx, "format", (format_kernel<uint8_t, uint16_t>), grid, dim3(64, 4), 0, a);
        else if (out_geo.bps == 1)                    HBHIP_LAUNCH(ctx, "format", (format_kernel<uint16_t, uint8_t>), grid, dim3(64, 4), 0, a);
        else                                          HBHIP_LAUNCH(ctx, "format", (format_kernel<uint16_t, uint16_t>), grid, dim3(64, 4), 0, a);
        HBHIP_CHECK(ctx, hipGetLastError());
        return HBHIP_OK;
    }
    int full_range;
};

} // namespace

extern "C" int hbhip_format_create(hbhip_ctx *ctx, int width, int height, int src_depth, int dst_depth,
                                   int log2_chroma_w, int log2_chroma_h, int full_range, hbhip_filter **out)
{
    if (!ctx || !out) return HBHIP_ERR_ARG;
    *out = nullptr;
    for (int d : {src_depth, dst_depth})
        if (d != 8 && d != 10 && d != 12) return HBHIP_ERR_UNSUPPORTED;
    // the range-stretching dither of full-range luma on the way down is not restated
    if (dst_depth < src_depth && full_range) return HBHIP_ERR_UNSUPPORTED;
    if (width < 1 || height < 1) return HBHIP_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    FormatFilter *f = new (std::nothrow) FormatFilter(ctx, full_range);
    if (!f) return HBHIP_ERR_NOMEM;
    PicGeometry gi, go;
    gi.set(width, height, src_depth, log2_chroma_w, log2_chroma_h);
    go.set(width, height, dst_depth, log2_chroma_w, log2_chroma_h);
    f->configure(gi, go);
    *out = f;
    return HBHIP_OK;
}

extern "C" int hbhip_pad_create(hbhip_ctx *ctx, const hbhip_pad_params *p, int width, int height, int depth,
                                int log2_chroma_w, int log2_chroma_h, hbhip_filter **out)
{
    if (!ctx || !p || !out) return HBHIP_ERR_ARG;
    *out = nullptr;
    if (depth != 8 && depth != 10 && depth != 12) return HBHIP_ERR_UNSUPPORTED;
    if (p->width < width || p->height < height || p->x < 0 || p->y < 0 ||
        p->x + width > p->width || p->y + height > p->height) return HBHIP_ERR_ARG;
    // vf_pad rounds the offsets down to the chroma subsampling; the caller has done so
    if ((p->x & ((1 << log2_chroma_w) - 1)) || (p->y & ((1 << log2_chroma_h) - 1))) return HBHIP_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    PadFilter *f = new (std::nothrow) PadFilter(ctx, *p);
    if (!f) return HBHIP_ERR_NOMEM;
    PicGeometry gi, go;
    gi.set(width, height, depth, log2_chroma_w, log2_chroma_h);
    go.set(p->width, p->height, depth, log2_chroma_w, log2_chroma_h);
    f->configure(gi, go);
    *out = f;
    return HBHIP_OK;
}

extern "C" int hbhip_rotate_create(hbhip_ctx *ctx, int angle, int hflip, int width, int height, int depth,
                                   int log2_chroma_w, int log2_chroma_h, hbhip_filter **out)
{
    if (!ctx || !out) return HBHIP_ERR_ARG;
    *out = nullptr;
    if (depth != 8 && depth != 10 && depth != 12) return HBHIP_ERR_UNSUPPORTED;
    if (angle != 0 && angle != 90 && angle != 180 && angle != 270) return HBHIP_ERR_ARG;
    if ((angle == 90 || angle == 270) && log2_chroma_w != log2_chroma_h) return HBHIP_ERR_UNSUPPORTED;
    (void)hipSetDevice(ctx->device);
    RotateFilter *f = new (std::nothrow) RotateFilter(ctx, angle, hflip);
    if (!f) return HBHIP_ERR_NOMEM;
    PicGeometry gi, go;
    gi.set(width, height, depth, log2_chroma_w, log2_chroma_h);
    if (f->transposes()) go.set(height, width, depth, log2_chroma_w, log2_chroma_h);
    else                 go = gi;
    f->configure(gi, go);
    *out = f;
    return HBHIP_OK;
}

extern "C" int hbhip_grayscale_create(hbhip_ctx *ctx, double cb, double cr, double size, double high,
                                      int width, int height, int depth, int log2_chroma_w, int log2_chroma_h,
                                      hbhip_filter **out)
{
    if (!ctx || !out) return HBHIP_ERR_ARG;
    *out = nullptr;
    if (depth != 8 && depth != 10 && depth != 12) return HBHIP_ERR_UNSUPPORTED;
    if (!(size > 0)) return HBHIP_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    MonochromeFilter *f = new (std::nothrow) MonochromeFilter(ctx, cb, cr, size, high);
    if (!f) return HBHIP_ERR_NOMEM;
    PicGeometry g;
    g.set(width, height, depth, log2_chroma_w, log2_chroma_h);
    f->configure(g, g);
    int rc = f->setup();
    if (rc != HBHIP_OK) { delete f; return rc; }
    *out = f;
    return HBHIP_OK;
}

extern "C" int hbhip_cropscale_create(hbhip_ctx *ctx, const hbhip_cropscale_params *p, int width, int height,
                                      int depth, int log2_chroma_w, int log2_chroma_h, hbhip_filter **out)
{
    if (!ctx || !p || !out) return HBHIP_ERR_ARG;
    *out = nullptr;
    if (depth != 8 && depth != 10 && depth != 12) return HBHIP_ERR_UNSUPPORTED;
    if (p->width < 1 || p->height < 1) return HBHIP_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    CropScaleFilter *f = new (std::nothrow) CropScaleFilter(ctx, *p);
    if (!f) return HBHIP_ERR_NOMEM;
    PicGeometry gi, go;
    gi.set(width, height, depth, log2_chroma_w, log2_chroma_h);
    go.set(p->width, p->height, depth, log2_chroma_w, log2_chroma_h);
    f->configure(gi, go);
    int rc = f->setup();
    if (rc != HBHIP_OK) { delete f; return rc; }
    *out = f;
    return HBHIP_OK;
}
