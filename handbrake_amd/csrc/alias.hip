// alias.hip — the libavfilter/zimg-backed "alias" filters as real GPU filters (8-bit):
//
//   rotate_kernel       transpose / hflip / vflip as rotate_init composes them
//                       (libhb/rotate.c:169-256 -> FFmpeg transpose,hflip,vflip)
//   monochrome_kernel   FFmpeg monochrome as grayscale_init configures it
//                       (libhb/grayscale.c:32-68; same math as the reference's Metal
//                       shader platform/macosx/shaders/grayscale_vt.metal:68-136)
//   scale8_*_kernel     crop + zscale(filter=lanczos) as crop_scale_init configures them
//   cropscale_*_kernel  (libhb/cropscale.c:52-185): 8-bit planes in zimg's 16-bit fixed point, 10 / 12-bit ones in double
//
// PARITY UNPINNED: the arithmetic of these filters is in FFmpeg / zimg, which are not in
// the reference tree.  These kernels are bit-exact against OUR restatement
// (oracle/alias_oracle.c): pure permutations for rotate; host-built exp() table + plain
// IEEE float ops for monochrome; host-built Lanczos tap tables (double, libm sin), quantised
// to 14 bits for 8-bit planes (integer passes, orc_cropscale_plane_fx) and used as they are
// with double accumulation in the oracle's order for 10 / 12-bit planes.
// All three are HBM-bound by their bytes (1 read + 1 write per pixel).
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
        if (in_geo.bps != 1) return false;
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
            HBHIP_LAUNCH(ctx, "rotate", rotate8_batch_kernel, dim3(hbhip_grid_x((max_w + 63) / 64), (max_h + 63) / 64, 3 * nf), dim3(64, 4), 0, B);
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
        return in_geo.bps == 1 && in_geo.log2_cw == 1 && in_geo.log2_ch == 1 &&
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
// ---- 8-bit planes: zimg's own arithmetic ------------------------------------------------------------------------
// zimg resizes an 8-bit plane as a 16-bit one (v << 8) in 16-bit fixed point: per pass
//     dst = clamp((sum_k c[k] * src[k] + (1 << 13)) >> 14, 0, 65535),    c = the filter row with 14 fractional bits,
// horizontal pass first, then back to 8 bits with round half up of v / 256 (oracle/alias_oracle.c:
// orc_cropscale_plane_fx has the derivation and the caveats - parity unpinned).  Integer throughout, so the result
// does not depend on evaluation order and the kernels below are free to use packed dot products: two taps per
// v_dot2_i32_i16, the 16-bit value between the passes kept biased by -32768 as zimg keeps it (a filter row sums to
// exactly 1 << 14, so the bias comes out as a constant).
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), acc, false);
}
__device__ __forceinline__ int reflect_idx(int j, int n)          // taps outside the plane fold back in (edge sample repeated)
{
    if (j < 0) j = -j - 1;
    if (j >= n) j = 2 * n - 1 - j;
    return min(max(j, 0), n - 1);
}
__device__ __forceinline__ int fx_round14(int acc) { return min(max(((acc + 8192) >> 14) + 32768, 0), 65535); }
__device__ __forceinline__ int fx_to8(int v16) { return min((v16 + 128) >> 8, 255); }

// ---- the swscale branch (cropscale.c:159-165: `scale=flags=lanczos+accurate_rnd` for sizes zscale is not used for -
// an odd width or height): libswscale's arithmetic for planar YUV at 8 / 10 / 12 bits, restated in oracle/alias_oracle.c
// (orc_cropscale_plane_sws / _sws16; PARITY UNPINNED like the zimg form).  px / py: first tapped source column / row of an output
// column / row (taps outside the plane already folded onto the edge sample by the table), qx: 14-bit, qy: 12-bit
// coefficients.  Two plain launches per plane: the sizes that come here are the odd ones, a fallback, not a hot path.
struct ScaleSwsArgs
{
    const uint8_t *src; uint8_t *dst;
    int spitch, dpitch, dw, dh, tx, ty, src_rows;
    const int *px, *py;
    const short *qx, *qy;
};

template <typename PIX>
__global__ __launch_bounds__(256) void scale_sws_h_kernel(ScaleSwsArgs a, int16_t *__restrict__ hbuf, int sh)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= a.dw || r >= a.src_rows) return;
    const PIX *row = reinterpret_cast<const PIX *>(a.src + (size_t)r * a.spitch) + a.px[x];
    const short *q = a.qx + (size_t)x * a.tx;
    int val = 0;
    for (int j = 0; j < a.tx; j++) val += (int)row[j] * (int)q[j];
    hbuf[(size_t)r * a.dw + x] = (int16_t)min(val >> sh, (1 << 15) - 1);            // hScale8To15_c (sh 7) / hScale16To15_c (depth - 1)
}

// yuv2planeX_8_c: the flat dither of 64 (round = 64 << 12, shift 19); yuv2planeX_10 / _12: half of the shift 27 - depth
template <typename PIX>
__global__ __launch_bounds__(256) void scale_sws_v_kernel(ScaleSwsArgs a, const int16_t *__restrict__ hbuf, int round, int shift, int vmax)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= a.dw || y >= a.dh) return;
    const int16_t *col = hbuf + (size_t)a.py[y] * a.dw + x;
    const short *q = a.qy + (size_t)y * a.ty;
    int val = round;
    for (int j = 0; j < a.ty; j++) val += (int)col[(size_t)j * a.dw] * (int)q[j];
    reinterpret_cast<PIX *>(a.dst + (size_t)y * a.dpitch)[x] = (PIX)min(max(val >> shift, 0), vmax);
}

// 6 x 6 taps (every upscale and same-size resampling, e.g. 1080p -> 2160p), both passes in one kernel, the three
// planes of up to SU_FRAMES frames per launch.  A workgroup owns a 256 x 16 tile of the output:
//   1. the source rows / columns the tile taps go into LDS as they lie (dwords), the reflection at the plane's
//      edges applied while staging - so every tap window is six CONSECUTIVE bytes of a staged row;
//   2. horizontal pass: a thread owns one output column; its six coefficients (three packed pairs) and the three
//      v_perm selectors that cut its window out of three staged dwords are constants of the thread, a staged row
//      costs it three LDS reads, three v_perm, three v_dot2.  The 16-bit results are stored as PAIRS of rows
//      (row 2p in the low half, 2p + 1 in the high half of a dword) ...
//   3. ... so that the vertical pass takes two taps per v_dot2 as well: a thread owns four adjacent columns of an
//      output row, a 16-byte LDS read brings the row pairs of all four; a window that starts on an odd row uses
//      the coefficient pairs (0, c0) (c1, c2) (c3, c4) (c5, 0) over four row pairs instead of realigning.
// bx / by: first tapped source column / row of an output column / row before reflection (may be < 0).
constexpr int SU_TW = 256, SU_TH = 16, SU_SRC_DW = 68, SU_MAXR = 24, SU_PAIRS = SU_MAXR / 2 + 2, SU_FRAMES = 16;
struct ScalePlane8
{
    const int *bx, *by;
    const uint32_t *qx, *qy;         // [dw][3], [dh][3]: coefficient pairs (c0, c1) (c2, c3) (c4, c5)
    int dw, dh, sw, sh, active;
};
struct ScaleBatch8
{
    ScalePlane8 p[3];
    int spitch[3], dpitch[3];
    const uint8_t *src[SU_FRAMES][3];    // crop window origins
    uint8_t       *dst[SU_FRAMES][3];
};

// TH: rows of a tile - 32 where the source rows that many output rows tap fit the LDS frame (every upscale by 4/3 or more:
// 22 source rows under 32 output rows at 2x, against 14 under 16 - a fifth less staging and horizontal work), else 16
template <int TH>
__global__ __launch_bounds__(256) void scale8_up_kernel(ScaleBatch8 B)
{
    __shared__ uint32_t s_src[SU_MAXR][SU_SRC_DW];
    __shared__ __attribute__((aligned(16))) uint32_t s_h[SU_PAIRS][SU_TW];
    __shared__ uint4 s_v[TH];
    const int f = (int)blockIdx.z / 3, pl = (int)blockIdx.z - 3 * f;
    const ScalePlane8 &P = B.p[pl];
    if (!P.active) return;
    const int x0 = blockIdx.x * SU_TW, y0 = blockIdx.y * TH;
    if (x0 >= P.dw || y0 >= P.dh) return;
    const int t = threadIdx.x;
    const int xe = min(x0 + SU_TW, P.dw) - 1, ye = min(y0 + TH, P.dh) - 1;
    const int cmin = P.bx[x0] & ~3, cmax = P.bx[xe] + 5;           // bx, by are non-decreasing
    const int rmin = P.by[y0], nr = P.by[ye] + 5 - rmin + 1;
    const int ndw = (cmax - cmin) / 4 + 1;                          // <= SU_SRC_DW - 2, nr <= SU_MAXR (checked by the host)
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    // the thread's column of the horizontal pass: its window and coefficients are on their way while the tile is staged
    const int xh = min(x0 + t, P.dw - 1);
    const int bxh = P.bx[xh];
    const uint32_t hc01 = P.qx[3 * (size_t)xh], hc23 = P.qx[3 * (size_t)xh + 1], hc45 = P.qx[3 * (size_t)xh + 2];
    // the vertical pass's rows: first tapped row and the three coefficient pairs of the tile's output rows
    if (t < TH && y0 + t <= ye)
    {
        const int y = y0 + t;
        s_v[t] = make_uint4((uint32_t)(P.by[y] - rmin), P.qy[3 * (size_t)y], P.qy[3 * (size_t)y + 1], P.qy[3 * (size_t)y + 2]);
    }
    {
        const uint8_t *src = B.src[f][pl];
        const int spitch = B.spitch[pl];
        // a wave stages the rows wave, wave + 4, ..: a lane per dword of the row (a 2x tile: 35 of them, two more lanes'
        // worth at most), the row loop without a division, all loads of a thread in flight before its first LDS store
        constexpr int RPW = SU_MAXR / 4;
        uint32_t v[RPW][2];
#pragma unroll
        for (int j = 0; j < RPW; j++)
        {
            const int rr = wave + 4 * j;
            v[j][0] = v[j][1] = 0;
            if (rr < nr)
            {
                const uint8_t *row = src + (size_t)reflect_idx(rmin + rr, P.sh) * spitch;
#pragma unroll
                for (int h = 0; h < 2; h++)
                {
                    const int d = lane + 64 * h, col = cmin + 4 * d;
                    if (d < ndw)
                    {
                        if (col >= 0 && col + 3 < P.sw && (((uintptr_t)(row + col)) & 3) == 0) v[j][h] = *reinterpret_cast<const uint32_t *>(row + col);
                        else
                        {
#pragma unroll
                            for (int k = 0; k < 4; k++) v[j][h] |= (uint32_t)row[reflect_idx(col + k, P.sw)] << (8 * k);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < RPW; j++)
        {
            const int rr = wave + 4 * j;
            if (rr < nr)
            {
                if (lane < ndw) s_src[rr][lane] = v[j][0];
                if (lane + 64 < ndw) s_src[rr][lane + 64] = v[j][1];
            }
        }
    }
    __syncthreads();
    if (x0 + t < P.dw)
    {
        const int o = bxh - cmin, dq = o >> 2, ob = o & 3;
        // selectors of v_perm_b32(hi, lo, sel): bytes ob, ob + 1 (ob + 2, ob + 3) of {hi:lo} zero-extended to halves
        const uint32_t sel01 = (uint32_t)ob | 0x0c000c00u | ((uint32_t)(ob + 1) << 16);
        const uint32_t sel23 = (uint32_t)(ob + 2) | 0x0c000c00u | ((uint32_t)(ob + 3) << 16);
        const uint32_t c01 = hc01, c23 = hc23, c45 = hc45;
        uint16_t *hp = reinterpret_cast<uint16_t *>(&s_h[0][0]) + 2 * t;
        for (int rr = 0; rr < nr; rr++)
        {
            const uint32_t d0 = s_src[rr][dq], d1 = s_src[rr][dq + 1], d2 = s_src[rr][dq + 2];
            int s = dot2(__builtin_amdgcn_perm(d1, d0, sel01), c01, 32);             // + 32: the rounding of (s + 32) >> 6
            s = dot2(__builtin_amdgcn_perm(d1, d0, sel23), c23, s);
            s = dot2(__builtin_amdgcn_perm(d2, d1, sel01), c45, s);
            const int h = min(max(s >> 6, 0), 65535);                // the 16-bit plane between the passes
            hp[(size_t)(rr >> 1) * (2 * SU_TW) + (rr & 1)] = (uint16_t)(h ^ 0x8000);    // biased, as zimg holds it
        }
    }
    __syncthreads();
    const int xq = x0 + 4 * lane;
    if (xq >= P.dw) return;
    for (int y = y0 + wave; y <= ye; y += 4)
    {
        const uint4 vt = s_v[y - y0];
        const int ob = __builtin_amdgcn_readfirstlane((int)vt.x);
        const uint32_t c01 = vt.y, c23 = vt.z, c45 = vt.w;
        // fx_to8(fx_round14(acc)) = clamp((((acc + 8192) >> 14) + 32768 + 128) >> 8, 0, 255) with the 16-bit clamp folded in
        // (a value outside 0 .. 65535 lands outside 0 .. 255 either way) = clamp((acc + K) >> 22, 0, 255): the sums start at K
        constexpr int K = 8192 + (32768 << 14) + (128 << 14);       // |acc| < 2^30, so acc + K stays inside int
        int acc[4] = {K, K, K, K};
        const uint4 *hq = reinterpret_cast<const uint4 *>(&s_h[ob >> 1][4 * lane]);
        auto tap = [&](int pair_row, uint32_t cpair) {
            const uint4 q = hq[(size_t)pair_row * (SU_TW / 4)];
            acc[0] = dot2(q.x, cpair, acc[0]); acc[1] = dot2(q.y, cpair, acc[1]);
            acc[2] = dot2(q.z, cpair, acc[2]); acc[3] = dot2(q.w, cpair, acc[3]);
        };
        if (!(ob & 1)) { tap(0, c01); tap(1, c23); tap(2, c45); }
        else
        {
            // rows ob .. ob + 5 against the pairs (ob - 1, ob) (ob + 1, ob + 2) (ob + 3, ob + 4) (ob + 5, ob + 6)
            tap(0, c01 << 16); tap(1, (c01 >> 16) | (c23 << 16)); tap(2, (c23 >> 16) | (c45 << 16)); tap(3, c45 >> 16);
        }
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) out |= (uint32_t)min(max(acc[k] >> 22, 0), 255) << (8 * k);
        uint8_t *d = B.dst[f][pl] + (size_t)y * B.dpitch[pl] + xq;
        if (xq + 3 < P.dw && (((uintptr_t)d) & 3) == 0) *reinterpret_cast<uint32_t *>(d) = out;
        else for (int k = 0; k < 4 && xq + k < P.dw; k++) d[k] = (uint8_t)(out >> (8 * k));
    }
}

// ---- 10 / 12-bit planes: the same arithmetic at the samples' own depth --------------------------------------------
// The two-pass form: any tap counts, the 16-bit plane between the passes in HBM - what a downscale, or any resize the fused
// kernels' LDS frame does not hold, runs through; an upscale takes the fused kernels.  ix / iy: tap positions (already
// reflected), qx / qy: 14-bit coefficients.  The planes of up to SD_FRAMES frames per launch (grid.z = 3 * frame + plane): a
// launch per plane and pass of every frame was 6 launches of 8 - 14 us per 1080p frame (28 % of the kernel time of a
// 1080i -> 540p list through the plugin surface).  8-bit planes: zimg's 16-bit fixed point between the passes (biased by
// -32768, fx_round14 / fx_to8).  10 / 12 bits: zimg resizes a uint16 plane in place of its depth, per pass
// dst = clamp((sum c[k] * src[k] + (1 << 13)) >> 14, 0, vmax) (oracle/alias_oracle.c: orc_cropscale_plane_fx16).
constexpr int SD_FRAMES = 8;
struct ScaleBatchHV
{
    const uint8_t *src[SD_FRAMES][3];
    uint8_t       *dst[SD_FRAMES][3];
    int spitch[3], dpitch[3], dw[3], dh[3], tx[3], ty[3], src_rows[3], src_bytes[3];   // src_bytes: a source row's samples, in bytes
    const uint32_t *tqx[3];             // horizontal taps, tap-major: [i * dw + x] = position << 16 | coefficient & 0xffff (a lane per column: coalesced)
    const int *iy[3];
    const short *qy[3];
    size_t hoff[3], hframe;             // the plane between the passes: plane c of frame f at hbuf + f * hframe + hoff[c] (samples)
    unsigned active;                    // bit c: plane c is resized (the others were copied)
};

// A thread makes SD_HR consecutive rows of its column: the taps (position and coefficient in one word, a lane per column) are
// fetched once for them, and all CH taps of a chunk - the whole filter where it has at most CH taps: the host picks the
// instantiation - are in flight before the first multiply.  As a plain loop over the taps every trip waited for its own two
// loads and the pass ran at the latency of 2 tx round trips (141 us per 8 frames 1080p -> 540p).
#ifndef SD_HR_N
#define SD_HR_N 4
#endif
constexpr int SD_HR = SD_HR_N;
template <typename PIX, int CH>
__global__ __launch_bounds__(256) void scale_h_batch_kernel(ScaleBatchHV a, uint16_t *__restrict__ hbuf, int vmax)
{
    const int c = blockIdx.z % 3, f = blockIdx.z / 3;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int r0 = (blockIdx.y * blockDim.y + threadIdx.y) * SD_HR;
    if (!((a.active >> c) & 1u) || x >= a.dw[c] || r0 >= a.src_rows[c]) return;
    const int tx = a.tx[c], dw = a.dw[c], nr = min(SD_HR, a.src_rows[c] - r0);
    const uint32_t *tq = a.tqx[c] + x;
    const uint8_t *rows = a.src[f][c] + (size_t)r0 * a.spitch[c];
    uint16_t *h = hbuf + (size_t)f * a.hframe + a.hoff[c] + (size_t)r0 * dw + x;
    int s[SD_HR];
#pragma unroll
    for (int k = 0; k < SD_HR; k++) s[k] = sizeof(PIX) == 1 ? 0 : 8192;
    for (int i0 = 0; i0 < tx; i0 += CH)
    {
        uint32_t t[CH];
#pragma unroll
        for (int i = 0; i < CH; i++) t[i] = i0 + i < tx ? tq[(size_t)(i0 + i) * dw] : 0u;          // (a tap beyond tx: coefficient 0 on sample 0)
        // Away from the plane's edges (where the table folds the taps back) a column's taps are consecutive samples: they
        // come as the aligned dwords that hold them, realigned by the column's byte offset - a third of the load
        // instructions of the sample-by-sample form (a quarter at 8 bits), which is what the pass is bound by (a wave's
        // byte load costs the texture path what its dword load does).
        const uint32_t base = t[0] >> 16;
        bool consecutive = true;
#pragma unroll
        for (int i = 1; i < CH; i++) consecutive &= i0 + i >= tx || (t[i] >> 16) == base + (uint32_t)i;
        if (consecutive)
        {
            constexpr int SZ = (int)sizeof(PIX), NW = CH * SZ / 4 + 1;
            const int mis = (int)(reinterpret_cast<uintptr_t>(rows) & 3u);           // (the pitch is a multiple of 4: the same for every row)
            const int bo = (int)base * SZ + mis, ao = (bo & ~3) - mis, last = ((a.src_bytes[c] - 1 + mis) & ~3) - mis;
            const uint32_t sh = (uint32_t)bo & 3u;
            uint32_t w[SD_HR][NW];
#pragma unroll
            for (int k = 0; k < SD_HR; k++)
            {
                const uint8_t *row = rows + (size_t)min(k, nr - 1) * a.spitch[c];
#pragma unroll
                for (int q = 0; q < NW; q++) w[k][q] = *reinterpret_cast<const uint32_t *>(row + min(ao + 4 * q, last));
            }
#pragma unroll
            for (int k = 0; k < SD_HR; k++)
#pragma unroll
                for (int i = 0; i < CH; i++)
                {
                    const int q = i * SZ / 4;
                    const uint32_t u = __builtin_amdgcn_alignbyte(w[k][q + 1], w[k][q], sh);
                    const int px = SZ == 1 ? (int)((u >> (8 * (i & 3))) & 0xffu) : (int)((u >> (16 * (i & 1))) & 0xffffu);
                    s[k] += (int)(int16_t)(t[i] & 0xffffu) * px;
                }
        }
        else
        {
#pragma unroll
            for (int k = 0; k < SD_HR; k++)
            {
                const PIX *row = reinterpret_cast<const PIX *>(rows + (size_t)min(k, nr - 1) * a.spitch[c]);
#pragma unroll
                for (int i = 0; i < CH; i++) s[k] += (int)(int16_t)(t[i] & 0xffffu) * (int)row[t[i] >> 16];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < SD_HR; k++)
        if (k < nr)
            h[(size_t)k * dw] = sizeof(PIX) == 1 ? (uint16_t)min(max((s[k] + 32) >> 6, 0), 65535)       // = (256 s + 8192) >> 14
                                                 : (uint16_t)min(max(s[k] >> 14, 0), vmax);
}

// The vertical pass: two adjacent columns per thread (a dword of the plane between the passes), the taps of a row - the same
// for the whole wave - through the scalar unit, a chunk's loads in flight together.
template <typename PIX, int CH>
__global__ __launch_bounds__(256) void scale_v_batch_kernel(ScaleBatchHV a, const uint16_t *__restrict__ hbuf, int vmax)
{
    const int c = blockIdx.z % 3, f = blockIdx.z / 3;
    const int x = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    const int y = __builtin_amdgcn_readfirstlane(blockIdx.y * blockDim.y + threadIdx.y);      // (a wave is one row of the block)
    if (!((a.active >> c) & 1u) || y >= a.dh[c]) return;
    const int ty = a.ty[c], dw = a.dw[c];
    const short *qy = a.qy[c] + (size_t)y * ty;
    const int *iy = a.iy[c] + (size_t)y * ty;
    if (x >= dw) return;
    const uint16_t *h = hbuf + (size_t)f * a.hframe + a.hoff[c] + x;
    const bool pair = x + 1 < dw, dword = pair && (dw & 1) == 0;     // (an even row length: every pair starts on a dword)
    int acc0 = sizeof(PIX) == 1 ? 0 : 8192, acc1 = acc0;
    for (int j0 = 0; j0 < ty; j0 += CH)
    {
        uint32_t v[CH];
        int q[CH];
#pragma unroll
        for (int j = 0; j < CH; j++)
        {
            const bool on = j0 + j < ty;
            q[j] = on ? (int)qy[j0 + j] : 0;
            const uint16_t *p = h + (size_t)(on ? iy[j0 + j] : iy[0]) * dw;
            v[j] = dword ? *reinterpret_cast<const uint32_t *>(p) : (uint32_t)p[0] | (pair ? (uint32_t)p[1] << 16 : 0u);
        }
#pragma unroll
        for (int j = 0; j < CH; j++)
        {
            const int bias = sizeof(PIX) == 1 ? 32768 : 0;
            acc0 += q[j] * ((int)(v[j] & 0xffffu) - bias);
            acc1 += q[j] * ((int)(v[j] >> 16) - bias);
        }
    }
    if (sizeof(PIX) == 1)
    {
        uint8_t *d = a.dst[f][c] + (size_t)y * a.dpitch[c] + x;
        d[0] = (uint8_t)fx_to8(fx_round14(acc0));
        if (pair) d[1] = (uint8_t)fx_to8(fx_round14(acc1));
    }
    else
    {
        uint16_t *d = reinterpret_cast<uint16_t *>(a.dst[f][c] + (size_t)y * a.dpitch[c]) + x;
        d[0] = (uint16_t)min(max(acc0 >> 14, 0), vmax);
        if (pair) d[1] = (uint16_t)min(max(acc1 >> 14, 0), vmax);
    }
}

// 6 x 6 taps, both passes in one kernel, the planes of up to SU_FRAMES frames per launch: scale8_up_kernel's shape on
// uint16 samples.  The staged source rows hold two samples per dword; a thread's six-sample window starts on an even
// or an odd sample, so it reads four dwords of a row and realigns them by 0 or 2 bytes (v_alignbyte_b32 with the
// thread's own shift) into the three sample pairs its three v_dot2 take.  The horizontal results go to LDS as pairs
// of rows, the vertical pass is the 8-bit kernel's with another rounding and a 16-bit store.
constexpr int SW_SRC_DW = 138;          // 2 * 136 samples of a row (+ the dwords the realignment reads past the window)
template <int TH>
__global__ __launch_bounds__(256) void scale16_up_kernel(ScaleBatch8 B, int vmax)
{
    __shared__ uint32_t s_src[SU_MAXR][SW_SRC_DW];
    __shared__ __attribute__((aligned(16))) uint32_t s_h[SU_PAIRS][SU_TW];
    __shared__ uint4 s_v[TH];
    const int f = (int)blockIdx.z / 3, pl = (int)blockIdx.z - 3 * f;
    const ScalePlane8 &P = B.p[pl];
    if (!P.active) return;
    const int x0 = blockIdx.x * SU_TW, y0 = blockIdx.y * TH;
    if (x0 >= P.dw || y0 >= P.dh) return;
    const int t = threadIdx.x;
    const int xe = min(x0 + SU_TW, P.dw) - 1, ye = min(y0 + TH, P.dh) - 1;
    const int cmin = P.bx[x0] & ~1, cmax = P.bx[xe] + 5;           // in samples; bx, by are non-decreasing
    const int rmin = P.by[y0], nr = P.by[ye] + 5 - rmin + 1;
    const int ndw = (cmax - cmin) / 2 + 2;                          // + 1: the dword an odd window's last pair reaches into
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    // the thread's column of the horizontal pass and the tile's rows of the vertical one, fetched ahead of the staging
    // (scale8_up_kernel)
    const int xh = min(x0 + t, P.dw - 1);
    const int bxh = P.bx[xh];
    const uint32_t hc01 = P.qx[3 * (size_t)xh], hc23 = P.qx[3 * (size_t)xh + 1], hc45 = P.qx[3 * (size_t)xh + 2];
    if (t < TH && y0 + t <= ye)
    {
        const int y = y0 + t;
        s_v[t] = make_uint4((uint32_t)(P.by[y] - rmin), P.qy[3 * (size_t)y], P.qy[3 * (size_t)y + 1], P.qy[3 * (size_t)y + 2]);
    }
    {
        const uint8_t *src = B.src[f][pl];
        const int spitch = B.spitch[pl];
        // a wave stages the rows wave, wave + 4, ..: a lane per dword (two samples) of the row, all loads of a thread in flight
        // before its first LDS store
        constexpr int RPW = SU_MAXR / 4, CH = (SW_SRC_DW + 63) / 64;
        uint32_t v[RPW][CH];
#pragma unroll
        for (int j = 0; j < RPW; j++)
        {
            const int rr = wave + 4 * j;
#pragma unroll
            for (int h = 0; h < CH; h++) v[j][h] = 0;
            if (rr < nr)
            {
                const uint16_t *row = reinterpret_cast<const uint16_t *>(src + (size_t)reflect_idx(rmin + rr, P.sh) * spitch);
#pragma unroll
                for (int h = 0; h < CH; h++)
                {
                    const int d = lane + 64 * h, col = cmin + 2 * d;
                    if (d < ndw)
                    {
                        if (col >= 0 && col + 1 < P.sw && (((uintptr_t)(row + col)) & 3) == 0) v[j][h] = *reinterpret_cast<const uint32_t *>(row + col);
                        else v[j][h] = (uint32_t)row[reflect_idx(col, P.sw)] | ((uint32_t)row[reflect_idx(col + 1, P.sw)] << 16);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < RPW; j++)
        {
            const int rr = wave + 4 * j;
            if (rr < nr)
            {
#pragma unroll
                for (int h = 0; h < CH; h++)
                    if (lane + 64 * h < ndw) s_src[rr][lane + 64 * h] = v[j][h];
            }
        }
    }
    __syncthreads();
    if (x0 + t < P.dw)
    {
        const int o = bxh - cmin, dq = o >> 1;
        const uint32_t shift = (uint32_t)(o & 1) * 2u;              // bytes
        const uint32_t c01 = hc01, c23 = hc23, c45 = hc45;
        uint16_t *hp = reinterpret_cast<uint16_t *>(&s_h[0][0]) + 2 * t;
        for (int rr = 0; rr < nr; rr++)
        {
            const uint32_t d0 = s_src[rr][dq], d1 = s_src[rr][dq + 1], d2 = s_src[rr][dq + 2], d3 = s_src[rr][dq + 3];
            int s = dot2(__builtin_amdgcn_alignbyte(d1, d0, shift), c01, 8192);
            s = dot2(__builtin_amdgcn_alignbyte(d2, d1, shift), c23, s);
            s = dot2(__builtin_amdgcn_alignbyte(d3, d2, shift), c45, s);
            hp[(size_t)(rr >> 1) * (2 * SU_TW) + (rr & 1)] = (uint16_t)min(max(s >> 14, 0), vmax);
        }
    }
    __syncthreads();
    const int xq = x0 + 4 * lane;
    if (xq >= P.dw) return;
    for (int y = y0 + wave; y <= ye; y += 4)
    {
        const uint4 vt = s_v[y - y0];
        const int ob = __builtin_amdgcn_readfirstlane((int)vt.x);
        const uint32_t c01 = vt.y, c23 = vt.z, c45 = vt.w;
        int acc[4] = {8192, 8192, 8192, 8192};
        const uint4 *hq = reinterpret_cast<const uint4 *>(&s_h[ob >> 1][4 * lane]);
        auto tap = [&](int pair_row, uint32_t cpair) {
            const uint4 q = hq[(size_t)pair_row * (SU_TW / 4)];
            acc[0] = dot2(q.x, cpair, acc[0]); acc[1] = dot2(q.y, cpair, acc[1]);
            acc[2] = dot2(q.z, cpair, acc[2]); acc[3] = dot2(q.w, cpair, acc[3]);
        };
        if (!(ob & 1)) { tap(0, c01); tap(1, c23); tap(2, c45); }
        else { tap(0, c01 << 16); tap(1, (c01 >> 16) | (c23 << 16)); tap(2, (c23 >> 16) | (c45 << 16)); tap(3, c45 >> 16); }
        uint32_t o4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) o4[k] = (uint32_t)min(max(acc[k] >> 14, 0), vmax);
        uint16_t *d = reinterpret_cast<uint16_t *>(B.dst[f][pl] + (size_t)y * B.dpitch[pl]) + xq;
        if (xq + 3 < P.dw && (((uintptr_t)d) & 7) == 0) *reinterpret_cast<uint2 *>(d) = make_uint2(o4[0] | (o4[1] << 16), o4[2] | (o4[3] << 16));
        else for (int k = 0; k < 4 && xq + k < P.dw; k++) d[k] = (uint16_t)o4[k];
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
int lanczos_table(int src_dim, int dst_dim, double shift, std::vector<int> &idx, std::vector<double> &coef,
                  std::vector<int> *first = nullptr)
{
    if (first) first->assign((size_t)dst_dim, 0);
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
        if (first) (*first)[i] = (int)begin;
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

// libswscale's filter of one dimension for SWS_LANCZOS (utils.c:initFilter), formula for formula as oracle/alias_oracle.c:
// orc_sws_filter states it (same libm, same expression order: the tables come out equal).  Returns the tap count.
int sws_filter(int src, int dst, int one, int src_pos, int dst_pos, std::vector<int> &pos, std::vector<short> &coef)
{
    auto ilog2 = [](unsigned v) { int n = 0; while (v >>= 1) n++; return n; };
    const int64_t fone = 1LL << (54 - std::min(ilog2((unsigned)(src / dst)), 8));
    const int x_inc = (int)((((int64_t)src << 16) + (dst >> 1)) / dst);
    pos.assign((size_t)dst, 0);
    int size;
    std::vector<int64_t> f;
    if (std::llabs((long long)x_inc - 0x10000) < 10 && src_pos == dst_pos)
    {
        size = 1;
        f.assign((size_t)dst, fone);
        for (int i = 0; i < dst; i++) pos[i] = i;
    }
    else
    {
        const int size_factor = 6;
        size = x_inc <= (1 << 16) ? 1 + size_factor : 1 + (size_factor * src + dst - 1) / dst;
        size = std::max(std::min(size, src - 2), 1);
        f.assign((size_t)dst * size, 0);
        int64_t x_dst_in_src = (((int64_t)dst_pos * x_inc) >> 7) - (((int64_t)src_pos * 0x10000LL) >> 7);
        for (int i = 0; i < dst; i++)
        {
            int xx = (int)((x_dst_in_src - (int64_t)(size - 2) * (1LL << 16)) / (1 << 17));
            pos[i] = xx;
            for (int j = 0; j < size; j++)
            {
                int64_t d = std::llabs(((int64_t)xx * (1 << 17)) - x_dst_in_src) << 13;
                if (x_inc > (1 << 16)) d = d * dst / src;
                const double fd = (double)d * (1.0 / (1 << 30));
                int64_t c = (int64_t)((d ? std::sin(fd * M_PI) * std::sin(fd * M_PI / 3.0) / (fd * fd * M_PI * M_PI / 3.0) : 1.0) * (double)fone);
                if (fd > 3.0) c = 0;
                f[(size_t)i * size + j] = c;
                xx++;
            }
            x_dst_in_src += 2 * (int64_t)x_inc;
        }
    }
    int min_size = 0;
    const double cut = 0.002 * (double)fone;
    for (int i = dst - 1; i >= 0; i--)
    {
        int64_t *row = &f[(size_t)i * size];
        int mn = size;
        int64_t cut_off = 0;
        for (int j = 0; j < size; j++)
        {
            cut_off += std::llabs(row[0]);
            if ((double)cut_off > cut) break;
            if (i < dst - 1 && pos[i] >= pos[i + 1]) break;
            for (int k = 1; k < size; k++) row[k - 1] = row[k];
            row[size - 1] = 0;
            pos[i]++;
        }
        cut_off = 0;
        for (int j = size - 1; j > 0; j--)
        {
            cut_off += std::llabs(row[j]);
            if ((double)cut_off > cut) break;
            mn--;
        }
        min_size = std::max(min_size, mn);
    }
    const int fsize = min_size;
    std::vector<int64_t> g((size_t)dst * fsize, 0);
    for (int i = 0; i < dst; i++)
        for (int j = 0; j < fsize && j < size; j++) g[(size_t)i * fsize + j] = f[(size_t)i * size + j];
    for (int i = 0; i < dst; i++)
    {
        int64_t *row = &g[(size_t)i * fsize];
        if (pos[i] < 0)
        {
            for (int j = 1; j < fsize; j++)
            {
                const int left = std::max(j + pos[i], 0);
                row[left] += row[j];
                row[j] = 0;
            }
            pos[i] = 0;
        }
        if (pos[i] + fsize > src)
        {
            const int shift = pos[i] + std::min(fsize - src, 0);
            int64_t acc = 0;
            for (int j = fsize - 1; j >= 0; j--)
                if (pos[i] + j >= src) { acc += row[j]; row[j] = 0; }
            for (int j = fsize - 1; j >= 0; j--) row[j] = j < shift ? 0 : row[j - shift];
            pos[i] -= shift;
            row[src - 1 - pos[i]] += acc;
        }
    }
    coef.assign((size_t)dst * fsize, 0);
    for (int i = 0; i < dst; i++)
    {
        const int64_t *row = &g[(size_t)i * fsize];
        int64_t error = 0, sum = 0;
        for (int j = 0; j < fsize; j++) sum += row[j];
        sum = (sum + one / 2) / one;
        if (!sum) sum = 1;
        for (int j = 0; j < fsize; j++)
        {
            const int64_t v = row[j] + error;
            const int64_t iv = v >= 0 ? (v + (sum >> 1)) / sum : -((-v + (sum >> 1)) / sum);
            coef[(size_t)i * fsize + j] = (short)iv;
            error = v - iv * sum;
        }
    }
    return fsize;
}

// a filter row with 14 fractional bits that still sums to 1 << 14: the rounding residue goes to the largest tap
// (oracle/alias_oracle.c: orc_quantize_taps)
void quantize_taps(const double *coef, int taps, short *q)
{
    int sum = 0, big = 0;
    for (int k = 0; k < taps; k++)
    {
        const long v = std::lrint(coef[k] * 16384.0);
        q[k] = (short)v;
        sum += (int)v;
        if (std::fabs(coef[k]) > std::fabs(coef[big])) big = k;
    }
    q[big] = (short)(q[big] + (16384 - sum));
}

class CropScaleFilter : public SimpleFilter
{
public:
    CropScaleFilter(hbhip_ctx *c, const hbhip_cropscale_params &p) : SimpleFilter(c), par(p) {}
    ~CropScaleFilter() override
    {
        for (int c = 0; c < 3; c++)
        {
            if (d_tqx[c]) (void)hipFree(d_tqx[c]);
            if (d_iy[c]) (void)hipFree(d_iy[c]);
            if (d_bx[c]) (void)hipFree(d_bx[c]);
            if (d_by[c]) (void)hipFree(d_by[c]);
            if (d_qx[c]) (void)hipFree(d_qx[c]);
            if (d_qy[c]) (void)hipFree(d_qy[c]);
        }
        if (hbuf16) (void)hipFree(hbuf16);
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
            identity[c] = (dw == crop_w[c] && dh == crop_h[c] && (sws || sx == 0.0));
            if (identity[c]) continue;
            if (sws)
            {
                // libswscale's tables: positions + 14-bit (horizontal) / 12-bit (vertical) coefficients; left-sited chroma
                // at horizontal position 0 (srcPos = dstPos = 64), everything else centred (128)
                std::vector<int> px, py;
                std::vector<short> qx, qy;
                const int hpos = (c && lw) ? 64 : 128;
                tx[c] = sws_filter(crop_w[c], dw, 1 << 14, hpos, hpos, px, qx);
                ty[c] = sws_filter(crop_h[c], dh, 1 << 12, 128, 128, py, qy);
                auto up = [&](auto *&dptr, const auto &v) -> int {
                    HBHIP_CHECK(ctx, hipMalloc((void **)&dptr, sizeof(v[0]) * v.size()));
                    HBHIP_CHECK(ctx, hipMemcpy(dptr, v.data(), sizeof(v[0]) * v.size(), hipMemcpyHostToDevice));
                    return HBHIP_OK;
                };
                int rc = up(d_bx[c], px);
                if (rc == HBHIP_OK) rc = up(d_by[c], py);
                if (rc == HBHIP_OK) rc = up(d_qx[c], qx);
                if (rc == HBHIP_OK) rc = up(d_qy[c], qy);
                if (rc != HBHIP_OK) return rc;
                up6 = false;
                continue;
            }
            std::vector<int> ix, iy, bx, by;
            std::vector<double> cx, cy;
            tx[c] = lanczos_table(crop_w[c], dw, sx, ix, cx, &bx);
            ty[c] = lanczos_table(crop_h[c], dh, 0.0, iy, cy, &by);
            auto upload = [&](auto *&dptr, const auto &v) -> int {
                HBHIP_CHECK(ctx, hipMalloc((void **)&dptr, sizeof(v[0]) * v.size()));
                HBHIP_CHECK(ctx, hipMemcpy(dptr, v.data(), sizeof(v[0]) * v.size(), hipMemcpyHostToDevice));
                return HBHIP_OK;
            };
            int rc = upload(d_iy[c], iy);
            if (rc != HBHIP_OK) return rc;
            {
                // zimg's fixed-point arithmetic at every depth (scale8_* / scale16_* kernels)
                std::vector<short> qx(cx.size()), qy(cy.size());
                for (int x = 0; x < dw; x++) quantize_taps(&cx[(size_t)x * tx[c]], tx[c], &qx[(size_t)x * tx[c]]);
                for (int y = 0; y < dh; y++) quantize_taps(&cy[(size_t)y * ty[c]], ty[c], &qy[(size_t)y * ty[c]]);
                rc = upload(d_qx[c], qx);
                if (rc == HBHIP_OK) rc = upload(d_qy[c], qy);
                if (rc == HBHIP_OK)
                {
                    std::vector<uint32_t> tq((size_t)dw * tx[c]);
                    for (int x = 0; x < dw; x++)
                        for (int i = 0; i < tx[c]; i++)
                            tq[(size_t)i * dw + x] = ((uint32_t)ix[(size_t)x * tx[c] + i] << 16) | ((uint32_t)(uint16_t)qx[(size_t)x * tx[c] + i]);
                    rc = upload(d_tqx[c], tq);
                }
                if (rc == HBHIP_OK) rc = upload(d_bx[c], bx);
                if (rc == HBHIP_OK) rc = upload(d_by[c], by);
                if (rc != HBHIP_OK) return rc;
                // the fused kernels: six taps either way, and what a 256 x 16 tile taps must fit their LDS frame
                if (tx[c] != 6 || ty[c] != 6) up6 = false;
                for (int x0 = 0; x0 < dw && up6; x0 += SU_TW)
                {
                    const int last = bx[std::min(x0 + SU_TW, dw) - 1] + 5;
                    if (in_geo.bps == 1 ? (last - (bx[x0] & ~3)) / 4 + 1 > SU_SRC_DW - 2 : (last - (bx[x0] & ~1)) / 2 + 2 > SW_SRC_DW) up6 = false;
                }
                for (int y0 = 0; y0 < dh && up6; y0 += SU_TH)
                    if (by[std::min(y0 + SU_TH, dh) - 1] + 5 - by[y0] + 1 > SU_MAXR) up6 = false;
                for (int y0 = 0; y0 < dh && tall; y0 += 2 * SU_TH)
                    if (by[std::min(y0 + 2 * SU_TH, dh) - 1] + 5 - by[y0] + 1 > SU_MAXR) tall = false;
            }
        }
        size_t need = 0;
        hframe = 0;
        for (int c = 0; c < 3; c++)
            if (!identity[c])
            {
                need = std::max(need, (size_t)out_geo.pw[c] * crop_h[c]);
                hoff[c] = hframe;
                hframe += ((size_t)out_geo.pw[c] * crop_h[c] + 127) & ~(size_t)127;
            }
        if (!sws) need = hframe * SD_FRAMES;        // the batched two-pass form: every plane of SD_FRAMES frames between the passes
        if (need && !up6) HBHIP_CHECK(ctx, hipMalloc((void **)&hbuf16, sizeof(uint16_t) * need));
        return HBHIP_OK;
    }

    const uint8_t *window(const DevPicture *in, int c) const
    {
        return in->plane[c] + (size_t)crop_y[c] * in->pitch[c] + (size_t)crop_x[c] * in_geo.bps;
    }
    int copy_identity_planes(DevPicture *in, DevPicture *out)
    {
        for (int c = 0; c < 3; c++)
            if (identity[c])
                HBHIP_LAUNCH(ctx, "crop_copy", crop_copy_kernel, dim3((out->width[c] * in_geo.bps + 255) / 256, out->height[c]), dim3(256), 0,
                             window(in, c), in->pitch[c], out->plane[c], out->pitch[c], out->width[c] * in_geo.bps, out->height[c]);
        return HBHIP_OK;
    }

    int process_many(DevPicture *const *ins, DevPicture *const *outs, int n) override
    {
        const int vmax = (1 << in_geo.depth) - 1;
        for (int i = 0; i < n; i++) (void)copy_identity_planes(ins[i], outs[i]);
        if (identity[0] && identity[1] && identity[2]) { HBHIP_CHECK(ctx, hipGetLastError()); return HBHIP_OK; }
        if (up6)
            for (int i0 = 0; i0 < n; i0 += SU_FRAMES)
            {
                const int m = std::min(SU_FRAMES, n - i0);
                ScaleBatch8 B;
                memset(&B, 0, sizeof(B));
                for (int c = 0; c < 3; c++)
                {
                    ScalePlane8 &P = B.p[c];
                    P.bx = d_bx[c]; P.by = d_by[c];
                    P.qx = reinterpret_cast<const uint32_t *>(d_qx[c]); P.qy = reinterpret_cast<const uint32_t *>(d_qy[c]);
                    P.dw = out_geo.pw[c]; P.dh = out_geo.ph[c]; P.sw = crop_w[c]; P.sh = crop_h[c];
                    P.active = !identity[c];
                    B.spitch[c] = ins[i0]->pitch[c]; B.dpitch[c] = outs[i0]->pitch[c];
                    for (int k = 0; k < m; k++)
                    {
                        if (ins[i0 + k]->pitch[c] != B.spitch[c] || outs[i0 + k]->pitch[c] != B.dpitch[c]) return HBHIP_ERR_ARG;
                        B.src[k][c] = window(ins[i0 + k], c); B.dst[k][c] = outs[i0 + k]->plane[c];
                    }
                }
                const dim3 grid((out_geo.pw[0] + SU_TW - 1) / SU_TW, (out_geo.ph[0] + SU_TH - 1) / SU_TH, 3 * m);
                const dim3 grid_tall(grid.x, (out_geo.ph[0] + 2 * SU_TH - 1) / (2 * SU_TH), grid.z);
                if (in_geo.bps == 1 && tall) HBHIP_LAUNCH(ctx, "cropscale_lanczos_fused", scale8_up_kernel<2 * SU_TH>, grid_tall, dim3(256), 0, B);
                else if (in_geo.bps == 1) HBHIP_LAUNCH(ctx, "cropscale_lanczos_fused", scale8_up_kernel<SU_TH>, grid, dim3(256), 0, B);
                else if (tall)       HBHIP_LAUNCH(ctx, "cropscale_lanczos_fused", scale16_up_kernel<2 * SU_TH>, grid_tall, dim3(256), 0, B, vmax);
                else                 HBHIP_LAUNCH(ctx, "cropscale_lanczos_fused", scale16_up_kernel<SU_TH>, grid, dim3(256), 0, B, vmax);
            }
        else if (!sws)
            for (int i0 = 0; i0 < n; i0 += SD_FRAMES)
            {
                const int m = std::min(SD_FRAMES, n - i0);
                ScaleBatchHV B;
                memset(&B, 0, sizeof(B));
                int gw = 0, ghr = 0, gvr = 0;
                for (int c = 0; c < 3; c++)
                {
                    if (identity[c]) continue;
                    B.active |= 1u << c;
                    B.spitch[c] = ins[i0]->pitch[c]; B.dpitch[c] = outs[i0]->pitch[c];
                    B.dw[c] = out_geo.pw[c]; B.dh[c] = out_geo.ph[c]; B.tx[c] = tx[c]; B.ty[c] = ty[c]; B.src_rows[c] = crop_h[c];
                    B.src_bytes[c] = crop_w[c] * in_geo.bps;
                    B.tqx[c] = d_tqx[c]; B.iy[c] = d_iy[c]; B.qy[c] = d_qy[c];
                    B.hoff[c] = hoff[c];
                    for (int k = 0; k < m; k++)
                    {
                        if (ins[i0 + k]->pitch[c] != B.spitch[c] || outs[i0 + k]->pitch[c] != B.dpitch[c]) return HBHIP_ERR_ARG;
                        B.src[k][c] = window(ins[i0 + k], c); B.dst[k][c] = outs[i0 + k]->plane[c];
                    }
                    gw = std::max(gw, B.dw[c]); ghr = std::max(ghr, B.src_rows[c]); gvr = std::max(gvr, B.dh[c]);
                }
                B.hframe = hframe;
                const dim3 gh((gw + 63) / 64, (ghr + 4 * SD_HR - 1) / (4 * SD_HR), 3 * m), gv(((gw + 1) / 2 + 63) / 64, (gvr + 3) / 4, 3 * m);
                // the instantiation that holds the planes' longest filter in one chunk (or the widest one, in several)
                int mtx = 0, mty = 0;
                for (int c = 0; c < 3; c++)
                    if (!identity[c]) { mtx = std::max(mtx, tx[c]); mty = std::max(mty, ty[c]); }
#define SD_GO(K, PIX, CH, G, HB) HBHIP_LAUNCH(ctx, "cropscale_lanczos_" #K, (scale_##K##_batch_kernel<PIX, CH>), G, dim3(64, 4), 0, B, HB, vmax)
#define SD_PICK(K, PIX, T, G, HB) do { if ((T) <= 8) SD_GO(K, PIX, 8, G, HB); else if ((T) <= 12) SD_GO(K, PIX, 12, G, HB); \
                                       else if ((T) <= 16) SD_GO(K, PIX, 16, G, HB); else SD_GO(K, PIX, 24, G, HB); } while (0)
                if (in_geo.bps == 1)
                {
                    SD_PICK(h, uint8_t, mtx, gh, hbuf16);
                    SD_PICK(v, uint8_t, mty, gv, (const uint16_t *)hbuf16);
                }
                else
                {
                    SD_PICK(h, uint16_t, mtx, gh, hbuf16);
                    SD_PICK(v, uint16_t, mty, gv, (const uint16_t *)hbuf16);
                }
#undef SD_PICK
#undef SD_GO
            }
        else
            for (int i = 0; i < n; i++)
                for (int c = 0; c < 3; c++)
                {
                    if (identity[c]) continue;
                    {
                        ScaleSwsArgs w;                                     // (swscale's arithmetic: odd sizes, a plane and a pass per launch)
                        w.src = window(ins[i], c); w.dst = outs[i]->plane[c];
                        w.spitch = ins[i]->pitch[c]; w.dpitch = outs[i]->pitch[c];
                        w.dw = out_geo.pw[c]; w.dh = out_geo.ph[c]; w.tx = tx[c]; w.ty = ty[c]; w.src_rows = crop_h[c];
                        w.px = d_bx[c]; w.py = d_by[c]; w.qx = d_qx[c]; w.qy = d_qy[c];
                        const dim3 gh((w.dw + 63) / 64, (crop_h[c] + 3) / 4), gv((w.dw + 63) / 64, (w.dh + 3) / 4);
                        if (in_geo.bps == 1)
                        {
                            HBHIP_LAUNCH(ctx, "cropscale_sws_h", scale_sws_h_kernel<uint8_t>, gh, dim3(64, 4), 0, w, (int16_t *)hbuf16, 7);
                            HBHIP_LAUNCH(ctx, "cropscale_sws_v", scale_sws_v_kernel<uint8_t>, gv, dim3(64, 4), 0, w, (const int16_t *)hbuf16, 64 << 12, 19, 255);
                        }
                        else
                        {
                            const int d = in_geo.depth, shift = 11 + 16 - d;
                            HBHIP_LAUNCH(ctx, "cropscale_sws_h", scale_sws_h_kernel<uint16_t>, gh, dim3(64, 4), 0, w, (int16_t *)hbuf16, d - 1);
                            HBHIP_LAUNCH(ctx, "cropscale_sws_v", scale_sws_v_kernel<uint16_t>, gv, dim3(64, 4), 0, w, (const int16_t *)hbuf16, 1 << (shift - 1), shift, (1 << d) - 1);
                        }
                    }
                }
        HBHIP_CHECK(ctx, hipGetLastError());
        return HBHIP_OK;
    }

    int process(DevPicture *in, DevPicture *out) override { return process_many(&in, &out, 1); }
    bool sws = false;           // libswscale's arithmetic (the reference's path for odd sizes) instead of zimg's
    bool up6 = true;            // six taps either way and every tile fits the fused kernel's LDS frame
    bool tall = true;           // ... also at 32 rows per tile
    hbhip_cropscale_params par;
    int crop_x[3], crop_y[3], crop_w[3], crop_h[3], tx[3] = {0, 0, 0}, ty[3] = {0, 0, 0};
    bool identity[3] = {false, false, false};
    int *d_iy[3] = {nullptr, nullptr, nullptr};
    uint32_t *d_tqx[3] = {nullptr, nullptr, nullptr};   // the two-pass form's horizontal taps (ScaleBatchHV::tqx)
    uint16_t *hbuf16 = nullptr; // the 16-bit plane between the passes (two-launch form); zimg form: the planes of SD_FRAMES frames
    size_t hoff[3] = {0, 0, 0}, hframe = 0;
    int *d_bx[3] = {nullptr, nullptr, nullptr}, *d_by[3] = {nullptr, nullptr, nullptr};
    short *d_qx[3] = {nullptr, nullptr, nullptr}, *d_qy[3] = {nullptr, nullptr, nullptr};
};

// ------------------------------------------------------------------ pad
// FFmpeg vf_pad as pad_init configures it (libhb/pad.c:40-148): the picture at (x, y) of a larger
// one, the rest one colour.  One launch for the three planes; HBM-bound (read in, write out).
constexpr int PAD_FRAMES = 16;
struct PadArgs
{
    const uint8_t *src[PAD_FRAMES][3];
    uint8_t       *dst[PAD_FRAMES][3];
    int spitch[3], dpitch[3], sw[3], sh[3], dw[3], dh[3], x[3], y[3], fill[3];       // widths and x in BYTES
};

// A thread makes one dword of an output row: the fill colour, four bytes of the picture (two aligned loads and a
// v_alignbyte when the picture sits at an odd byte offset, as a 4:2:0 chroma plane at x = 2 (mod 4) does), or - the
// dword the picture's edge runs through - byte by byte.  grid.z = 3 * frame + plane: the frames of a batch in one launch.
__global__ __launch_bounds__(256) void pad_kernel(PadArgs a, int bps)
{
    const int c = blockIdx.z % 3, f = blockIdx.z / 3;
    const int ob = 4 * (blockIdx.x * blockDim.x + threadIdx.x), yy = blockIdx.y * blockDim.y + threadIdx.y;
    if (ob >= a.dw[c] || yy >= a.dh[c]) return;
    const uint32_t fillw = bps == 1 ? (uint32_t)a.fill[c] * 0x01010101u : (uint32_t)a.fill[c] * 0x00010001u;
    uint8_t *drow = a.dst[f][c] + (size_t)yy * a.dpitch[c];
    const int sy = yy - a.y[c], lo = ob - a.x[c];                    // source row, source byte of the dword's first byte
    uint32_t v = fillw;
    if (sy >= 0 && sy < a.sh[c] && lo + 3 >= 0 && lo < a.sw[c])
    {
        const uint8_t *srow = a.src[f][c] + (size_t)sy * a.spitch[c];
        if (lo >= 0 && lo + 3 < a.sw[c])
        {
            const int al = lo & ~3, sh = lo & 3;
            const uint32_t w0 = *reinterpret_cast<const uint32_t *>(srow + al);
            const uint32_t w1 = sh ? *reinterpret_cast<const uint32_t *>(srow + al + 4) : 0u;     // (al + 4 <= lo + 3 < sw: inside the row)
            v = __builtin_amdgcn_alignbyte(w1, w0, (uint32_t)sh);
        }
        else
        {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (lo + k >= 0 && lo + k < a.sw[c]) v = (v & ~(0xffu << (8 * k))) | ((uint32_t)srow[lo + k] << (8 * k));
        }
    }
    if (ob + 3 < a.dw[c]) *reinterpret_cast<uint32_t *>(drow + ob) = v;
    else for (int k = 0; ob + k < a.dw[c]; k++) drow[ob + k] = (uint8_t)(v >> (8 * k));
}

class PadFilter : public SimpleFilter
{
public:
    PadFilter(hbhip_ctx *c, const hbhip_pad_params &p) : SimpleFilter(c), par(p) {}
    int process_many(DevPicture *const *ins, DevPicture *const *outs, int n) override
    {
        const int bps = in_geo.bps;
        int at = 0;
        while (at < n)
        {
            int nf = 1;
            auto same = [&](int i) {
                for (int c = 0; c < 3; c++)
                    if (ins[i]->pitch[c] != ins[at]->pitch[c] || outs[i]->pitch[c] != outs[at]->pitch[c]) return false;
                return true;
            };
            while (at + nf < n && nf < PAD_FRAMES && same(at + nf)) nf++;
            PadArgs a;
            memset(&a, 0, sizeof(a));
            bool aligned = true;
            for (int c = 0; c < 3; c++)
            {
                DevPicture *in = ins[at], *out = outs[at];
                a.spitch[c] = in->pitch[c]; a.dpitch[c] = out->pitch[c];
                a.sw[c] = in->width[c] * bps; a.sh[c] = in->height[c]; a.dw[c] = out->width[c] * bps; a.dh[c] = out->height[c];
                a.x[c] = (c ? par.x >> in_geo.log2_cw : par.x) * bps;
                a.y[c] = c ? par.y >> in_geo.log2_ch : par.y;
                a.fill[c] = par.fill[c];
                aligned = aligned && ((a.spitch[c] | a.dpitch[c]) & 3) == 0;
                for (int f = 0; f < nf; f++)
                {
                    a.src[f][c] = ins[at + f]->plane[c]; a.dst[f][c] = outs[at + f]->plane[c];
                    aligned = aligned && (((uintptr_t)a.src[f][c] | (uintptr_t)a.dst[f][c]) & 3) == 0;
                }
            }
            if (!aligned) return HBHIP_ERR_ARG;                              // planes and pitches of this library are 64-byte aligned
            const dim3 grid(hbhip_grid_x(((a.dw[0] + 3) / 4 + 63) / 64), (a.dh[0] + 3) / 4, 3 * nf);
            HBHIP_LAUNCH(ctx, "pad", pad_kernel, grid, dim3(64, 4), 0, a, bps);
            HBHIP_CHECK(ctx, hipGetLastError());
            at += nf;
        }
        return HBHIP_OK;
    }
    int process(DevPicture *in, DevPicture *out) override { return process_many(&in, &out, 1); }
    hbhip_pad_params par;
};

// ------------------------------------------------------------------ format (depth conversion)
// `format=pix_fmts=...` (libhb/format.c:13-111): libavfilter inserts a same-size `scale`, i.e. libswscale's
// unscaled planar copy (swscale_unscaled.c:planarCopyWrapper; parity unpinned, restated in
// oracle/alias_oracle.c:orc_format_plane).  Up: shift (limited range, chroma) or top-bit replication
// (full-range luma); down: ordered dither then the overflow clamp tmp - (tmp >> depth).  One launch for the
// three planes, four samples per thread, HBM-bound (read in + write out).
constexpr int FMT_FRAMES = 16;
struct FormatArgs
{
    const uint8_t *src[FMT_FRAMES][3];
    uint8_t       *dst[FMT_FRAMES][3];
    int spitch[3], dpitch[3], w[3], h[3];
    int sdepth, ddepth, full_range;
};

// four samples per thread, moved as one dword (bytes) or two (16-bit samples) where the row has them; grid.z =
// 3 * frame + plane: the frames of a batch in one launch
template <typename SRC, typename DST>
__global__ __launch_bounds__(256) void format_kernel(FormatArgs a)
{
    const int c = blockIdx.z % 3, f = blockIdx.z / 3;
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x0 >= a.w[c] || y >= a.h[c]) return;
    const SRC *s = reinterpret_cast<const SRC *>(a.src[f][c] + (size_t)y * a.spitch[c]);
    DST *d = reinterpret_cast<DST *>(a.dst[f][c] + (size_t)y * a.dpitch[c]);
    const bool shiftonly = c != 0 || !a.full_range;
    const int up = a.ddepth - a.sdepth;
    const bool whole = x0 + 3 < a.w[c];
    unsigned v4[4] = { 0, 0, 0, 0 };
    if (whole)
    {
        if (sizeof(SRC) == 1)
        {
            const uint32_t w = *reinterpret_cast<const uint32_t *>(s + x0);
#pragma unroll
            for (int i = 0; i < 4; i++) v4[i] = (w >> (8 * i)) & 0xffu;
        }
        else
        {
            const uint2 w = *reinterpret_cast<const uint2 *>(s + x0);
            v4[0] = w.x & 0xffffu; v4[1] = w.x >> 16; v4[2] = w.y & 0xffffu; v4[3] = w.y >> 16;
        }
    }
    else
        for (int i = 0; x0 + i < a.w[c]; i++) v4[i] = s[x0 + i];
    unsigned o4[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const int x = x0 + i;
        const unsigned v = v4[i];
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
            if (shiftonly)
            {
                const unsigned tmp = (v + dm) >> shift;
                o = tmp - (tmp >> a.ddepth);
            }
            else
                o = (v - (v >> a.ddepth) + dm) >> shift;                // full-range luma: DITHER_COPY's other arm
        }
        o4[i] = o;
    }
    if (whole)
    {
        if (sizeof(DST) == 1) *reinterpret_cast<uint32_t *>(d + x0) = o4[0] | (o4[1] << 8) | (o4[2] << 16) | (o4[3] << 24);
        else                  *reinterpret_cast<uint2 *>(d + x0) = make_uint2(o4[0] | (o4[1] << 16), o4[2] | (o4[3] << 16));
    }
    else
        for (int i = 0; x0 + i < a.w[c]; i++) d[x0 + i] = (DST)o4[i];
}

class FormatFilter : public SimpleFilter
{
public:
    FormatFilter(hbhip_ctx *c, int full) : SimpleFilter(c), full_range(full) {}
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
            while (at + nf < n && nf < FMT_FRAMES && same(at + nf)) nf++;
            FormatArgs a;
            memset(&a, 0, sizeof(a));
            bool aligned = true;
            for (int c = 0; c < 3; c++)
            {
                a.spitch[c] = ins[at]->pitch[c]; a.dpitch[c] = outs[at]->pitch[c];
                a.w[c] = in_geo.pw[c]; a.h[c] = in_geo.ph[c];
                aligned = aligned && ((a.spitch[c] | a.dpitch[c]) & 7) == 0;
                for (int f = 0; f < nf; f++)
                {
                    a.src[f][c] = ins[at + f]->plane[c]; a.dst[f][c] = outs[at + f]->plane[c];
                    aligned = aligned && (((uintptr_t)a.src[f][c] | (uintptr_t)a.dst[f][c]) & 7) == 0;
                }
            }
            if (!aligned) return HBHIP_ERR_ARG;                              // planes and pitches of this library are 64-byte aligned
            a.sdepth = in_geo.depth; a.ddepth = out_geo.depth; a.full_range = full_range;
            const dim3 grid(hbhip_grid_x((a.w[0] + 255) / 256), (a.h[0] + 3) / 4, 3 * nf);
            if (in_geo.bps == 1 && out_geo.bps == 1)      HBHIP_LAUNCH(ctx, "format", (format_kernel<uint8_t, uint8_t>), grid, dim3(64, 4), 0, a);
            else if (in_geo.bps == 1)                     HBHIP_LAUNCH(ctx, "format", (format_kernel<uint8_t, uint16_t>), grid, dim3(64, 4), 0, a);
            else if (out_geo.bps == 1)                    HBHIP_LAUNCH(ctx, "format", (format_kernel<uint16_t, uint8_t>), grid, dim3(64, 4), 0, a);
            else                                          HBHIP_LAUNCH(ctx, "format", (format_kernel<uint16_t, uint16_t>), grid, dim3(64, 4), 0, a);
            HBHIP_CHECK(ctx, hipGetLastError());
            at += nf;
        }
        return HBHIP_OK;
    }
    int process(DevPicture *in, DevPicture *out) override { return process_many(&in, &out, 1); }
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

static int cropscale_create(hbhip_ctx *ctx, const hbhip_cropscale_params *p, int width, int height,
                            int depth, int log2_chroma_w, int log2_chroma_h, bool sws, hbhip_filter **out);

extern "C" int hbhip_cropscale_create(hbhip_ctx *ctx, const hbhip_cropscale_params *p, int width, int height,
                                      int depth, int log2_chroma_w, int log2_chroma_h, hbhip_filter **out)
{
    return cropscale_create(ctx, p, width, height, depth, log2_chroma_w, log2_chroma_h, false, out);
}

extern "C" int hbhip_cropscale_sws_create(hbhip_ctx *ctx, const hbhip_cropscale_params *p, int width, int height,
                                          int depth, int log2_chroma_w, int log2_chroma_h, hbhip_filter **out)
{
    return cropscale_create(ctx, p, width, height, depth, log2_chroma_w, log2_chroma_h, true, out);
}

static int cropscale_create(hbhip_ctx *ctx, const hbhip_cropscale_params *p, int width, int height,
                            int depth, int log2_chroma_w, int log2_chroma_h, bool sws, hbhip_filter **out)
{
    if (!ctx || !p || !out) return HBHIP_ERR_ARG;
    *out = nullptr;
    if (depth != 8 && depth != 10 && depth != 12) return HBHIP_ERR_UNSUPPORTED;
    if (p->width < 1 || p->height < 1) return HBHIP_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    CropScaleFilter *f = new (std::nothrow) CropScaleFilter(ctx, *p);
    if (!f) return HBHIP_ERR_NOMEM;
    f->sws = sws;
    PicGeometry gi, go;
    gi.set(width, height, depth, log2_chroma_w, log2_chroma_h);
    go.set(p->width, p->height, depth, log2_chroma_w, log2_chroma_h);
    f->configure(gi, go);
    int rc = f->setup();
    if (rc != HBHIP_OK) { delete f; return rc; }
    *out = f;
    return HBHIP_OK;
}
