// blend.hip — the reference's subtitle compositor (libhb/blend.c) for planar frames on gfx950.
//
//   blend_same_kernel        blend8on8 :425-509 / blend8on1x :511-604   (overlay in the frame's subsampling)
//   blend_subsample_kernel   blend_subsample_8on8 :236-328 / blend_subsample_8on1x :48-140
//                            (4:4:4 overlay on a chroma-subsampled frame, chroma-location aware)
//
// The 8-bit functions are the 16-bit ones with shift 0, so each kernel is one template over the
// sample type.  Integer arithmetic throughout: bit-exact with the reference.  A thread owns one
// chroma sample of the frame and the (1 << wshift) x (1 << hshift) luma samples that go with it, so a
// frame sample is read and written by exactly one thread; overlays are composited in list order
// (they may overlap, hb_blend_work :866-869): consecutive overlays that touch disjoint parts of the frame - the
// usual case, a few lines of text - go into one launch (grid.z = overlay; the groups are found when the list is
// set), one that overlaps an earlier one of its group starts the next launch.  Overlay bitmaps are uploaded once per
// change (rendersub's `changed`), not per frame.
//
// Not reproduced: the reference's stray chroma writes one sample before the row when a
// same-subsampling overlay hangs over the left / top edge by an odd amount (:485-505), and its
// running past the row / plane when an overlay sticks out to the right / bottom (:74-75): writes stop
// at the frame edge.  Biplanar (NV12 / P010) frames are refused (planar frames only).
#include "hbhip_internal.h"

#include <algorithm>
#include <vector>

namespace {

struct OverlayDev
{
    const uint8_t *plane[4];
    int stride[4];
    int x, y, width, height;
};

constexpr int BL_GROUP = 8;          // overlays per launch
struct OverlayGroup
{
    OverlayDev o[BL_GROUP];
    int bx0[BL_GROUP], by0[BL_GROUP];    // blend_subsample_kernel: the first frame chroma sample the overlay touches
};

struct BlendArgs
{
    uint8_t *dst[3];
    int pitch[3];
    int width, height, cw, ch;       // frame luma and chroma dimensions
    int wshift, hshift, shift;
    unsigned coeff[2][2];            // chroma-location weights of the samples under one chroma sample
};

template <typename PIX> __device__ __forceinline__ PIX *row_of(uint8_t *plane, int pitch, int y)
{
    return reinterpret_cast<PIX *>(plane + (size_t)y * pitch);
}

// grid: overlay chroma samples (xx, yy) in the overlay's own coordinates
template <typename PIX>
__global__ __launch_bounds__(256) void blend_same_kernel(BlendArgs a, OverlayGroup G)
{
    const OverlayDev &o = G.o[blockIdx.z];
    const int xx = blockIdx.x * blockDim.x + threadIdx.x, yy = blockIdx.y * blockDim.y + threadIdx.y;
    const int left = o.x, top = o.y;
    const int x0 = left < 0 ? -left : 0, y0 = top < 0 ? -top : 0;
    int ww = o.width, hh = o.height;
    if (o.width - x0 > a.width - left) ww = a.width - left + x0;
    if (o.height - y0 > a.height - top) hh = a.height - top + y0;
    const unsigned max = (256u << a.shift) - 1;

    // the luma samples of this block
    for (int j = 0; j < (1 << a.hshift); j++)
        for (int i = 0; i < (1 << a.wshift); i++)
        {
            const int lx = (xx << a.wshift) + i, ly = (yy << a.hshift) + j;
            if (lx < x0 || lx >= ww || ly < y0 || ly >= hh) continue;
            const int dx = left + lx, dy = top + ly;
            if (dx >= a.width || dy >= a.height) continue;
            const unsigned al = (unsigned)o.plane[3][(size_t)ly * o.stride[3] + lx] << a.shift;
            const unsigned s = (unsigned)o.plane[0][(size_t)ly * o.stride[0] + lx] << a.shift;
            PIX *d = row_of<PIX>(a.dst[0], a.pitch[0], dy) + dx;
            *d = (PIX)(((unsigned)*d * (max - al) + s * al) / max);
        }
    // its chroma sample
    if (xx < (x0 >> a.wshift) || xx >= (ww >> a.wshift) || yy < (y0 >> a.hshift) || yy >= (hh >> a.hshift)) return;
    const int dx = (left >> a.wshift) + xx, dy = yy + (top >> a.hshift);
    if (dx < 0 || dy < 0 || dx >= a.cw || dy >= a.ch) return;
    const unsigned al = (unsigned)o.plane[3][(size_t)(yy << a.hshift) * o.stride[3] + (xx << a.wshift)] << a.shift;
#pragma unroll
    for (int c = 1; c < 3; c++)
    {
        const unsigned s = (unsigned)o.plane[c][(size_t)yy * o.stride[c] + xx] << a.shift;
        PIX *d = row_of<PIX>(a.dst[c], a.pitch[c], dy) + dx;
        *d = (PIX)(((unsigned)*d * (max - al) + s * al) / max);
    }
}

// grid: frame chroma samples starting at (bx0, by0) = the first one the overlay touches
template <typename PIX>
__global__ __launch_bounds__(256) void blend_subsample_kernel(BlendArgs a, OverlayGroup G)
{
    const OverlayDev &o = G.o[blockIdx.z];
    const int bx0 = G.bx0[blockIdx.z], by0 = G.by0[blockIdx.z];
    const int cx = bx0 + blockIdx.x * blockDim.x + threadIdx.x, cy = by0 + blockIdx.y * blockDim.y + threadIdx.y;
    const int x0 = o.x, y0 = o.y;
    const int ow = o.width <= a.width ? o.width : a.width;          // :74-75 with left == x0
    const int oh = o.height <= a.height ? o.height : a.height;
    const int xx = cx << a.wshift, yy = cy << a.hshift;
    const int ox = xx - x0, oy = yy - y0;
    if (cx >= a.cw || cy >= a.ch || ox >= ow || oy >= oh) return;
    const unsigned max = (256u << a.shift) - 1;

    PIX *du = row_of<PIX>(a.dst[1], a.pitch[1], cy) + cx, *dv = row_of<PIX>(a.dst[2], a.pitch[2], cy) + cx;
    const unsigned cur_u = *du, cur_v = *dv;
    unsigned acc_u = 0, acc_v = 0, total = 0;
    for (int yz = 0; yz < (1 << a.hshift) && oy + yz < oh; yz++)
        for (int xz = 0; xz < (1 << a.wshift) && ox + xz < ow; xz++)
        {
            const unsigned coeff = a.coeff[0][xz] * a.coeff[1][yz];
            unsigned ru = cur_u, rv = cur_v;
            if (ox + xz >= 0 && oy + yz >= 0)
            {
                const size_t row = (size_t)(oy + yz);
                const int col = ox + xz;
                const unsigned al = (unsigned)o.plane[3][row * o.stride[3] + col] << a.shift;
                const unsigned su = (unsigned)o.plane[1][row * o.stride[1] + col] << a.shift;
                const unsigned sv = (unsigned)o.plane[2][row * o.stride[2] + col] << a.shift;
                ru = (ru * (max - al) + su * al + (max >> 1)) / max;
                rv = (rv * (max - al) + sv * al + (max >> 1)) / max;
                // the luma sample at the same place
                if (xx + xz < a.width && yy + yz < a.height)
                {
                    const unsigned sy = (unsigned)o.plane[0][row * o.stride[0] + col] << a.shift;
                    PIX *d = row_of<PIX>(a.dst[0], a.pitch[0], yy + yz) + xx + xz;
                    *d = (PIX)(((unsigned)*d * (max - al) + sy * al + (max >> 1)) / max);
                }
            }
            acc_u += coeff * ru;
            acc_v += coeff * rv;
            total += coeff;
        }
    if (total)
    {
        *du = (PIX)((acc_u + (total >> 1)) / total);
        *dv = (PIX)((acc_v + (total >> 1)) / total);
    }
}

} // namespace

struct hbhip_blend
{
    hbhip_ctx *ctx = nullptr;
    PicGeometry geo;
    int chroma_location = 1, ov_wshift = 0, ov_hshift = 0;
    bool subsample = false;
    unsigned coeff[2][2] = {{1, 1}, {1, 1}};
    uint8_t *d_store = nullptr;          // the uploaded overlay bitmaps, back to back
    size_t   store_bytes = 0;
    std::vector<OverlayDev> overlays;
    struct Launch { OverlayGroup g; int n; dim3 grid; };
    std::vector<Launch> launches;        // the overlays in list order, grouped (build_launches)
    hbhip_frame *staging = nullptr;      // device frame of the host-frame entry point

    ~hbhip_blend()
    {
        if (d_store) (void)hipFree(d_store);
        if (staging) hbhip_frame_release(staging);
    }
};

// The launches of an overlay list: what a launch of one overlay used to cover (its grid and, for the subsampling
// kernel, its origin), and consecutive overlays joined while the frame rectangles they touch - luma, widened to whole
// chroma samples and by one more sample for the odd-origin cases - stay disjoint.
static void build_launches(hbhip_blend *b)
{
    b->launches.clear();
    const int ws = b->geo.log2_cw, hs = b->geo.log2_ch, W = b->geo.width, H = b->geo.height;
    struct Rect { int x0, y0, x1, y1; };
    std::vector<Rect> rects;                                    // of the overlays in the group being filled
    hbhip_blend::Launch cur;
    cur.n = 0;
    cur.grid = dim3(0, 0, 0);
    auto flush = [&]() {
        if (cur.n) { cur.grid.z = cur.n; b->launches.push_back(cur); }
        cur.n = 0; cur.grid = dim3(0, 0, 0); rects.clear();
    };
    for (const OverlayDev &o : b->overlays)
    {
        int bx0 = 0, by0 = 0, nx, ny;
        if (b->subsample)
        {
            int x0c = o.x & ~((1 << ws) - 1), y0c = o.y & ~((1 << hs) - 1);
            if (x0c < 0) x0c = 0;
            if (y0c < 0) y0c = 0;
            const int ow = o.width <= W ? o.width : W, oh = o.height <= H ? o.height : H;
            int x1 = o.x + ow, y1 = o.y + oh;                     // one past the last frame sample touched
            if (x1 > W) x1 = W;
            if (y1 > H) y1 = H;
            if (x1 <= x0c || y1 <= y0c) continue;
            bx0 = x0c >> ws; by0 = y0c >> hs;
            nx = ((x1 - 1) >> ws) - bx0 + 1; ny = ((y1 - 1) >> hs) - by0 + 1;
        }
        else
        {
            nx = -((-o.width) >> ws); ny = -((-o.height) >> hs);
        }
        const int m = 2 << (ws > hs ? ws : hs);
        const Rect r = { o.x - m, o.y - m, o.x + o.width + m, o.y + o.height + m };
        bool clash = cur.n == BL_GROUP;
        for (const Rect &q : rects) clash = clash || (r.x0 < q.x1 && q.x0 < r.x1 && r.y0 < q.y1 && q.y0 < r.y1);
        if (clash) flush();
        cur.g.o[cur.n] = o; cur.g.bx0[cur.n] = bx0; cur.g.by0[cur.n] = by0;
        cur.grid.x = std::max<unsigned>(cur.grid.x, (nx + 63) / 64);
        cur.grid.y = std::max<unsigned>(cur.grid.y, (ny + 3) / 4);
        cur.n++;
        rects.push_back(r);
    }
    flush();
}

extern "C" int hbhip_blend_create(hbhip_ctx *ctx, int width, int height, int depth, int log2_chroma_w, int log2_chroma_h,
                                  int chroma_location, int overlay_log2_chroma_w, int overlay_log2_chroma_h,
                                  hbhip_blend **out)
{
    if (!ctx || !out) return HBHIP_ERR_ARG;
    *out = nullptr;
    if (depth != 8 && depth != 10 && depth != 12) return HBHIP_ERR_UNSUPPORTED;
    if (log2_chroma_w < 0 || log2_chroma_w > 1 || log2_chroma_h < 0 || log2_chroma_h > 1 || width < 1 || height < 1)
        return HBHIP_ERR_ARG;
    const bool subsample = log2_chroma_w != overlay_log2_chroma_w || log2_chroma_h != overlay_log2_chroma_h;
    // the reference's subsampling functions index the overlay's chroma at full resolution (blend.c:117-122)
    if (subsample && (overlay_log2_chroma_w || overlay_log2_chroma_h)) return HBHIP_ERR_UNSUPPORTED;
    hbhip_blend *b = new (std::nothrow) hbhip_blend;
    if (!b) return HBHIP_ERR_NOMEM;
    b->ctx = ctx;
    b->geo.set(width, height, depth, log2_chroma_w, log2_chroma_h);
    b->chroma_location = chroma_location;
    b->ov_wshift = overlay_log2_chroma_w;
    b->ov_hshift = overlay_log2_chroma_h;
    b->subsample = subsample;
    // hb_compute_chroma_smoothing_coefficient (common.c:7054-7091): window into 1 3 9 27 9 3 1
    static const unsigned base[] = { 1, 3, 9, 27, 9, 3, 1 };
    int wx = 4 - (1 << log2_chroma_w), wy = 4 - (1 << log2_chroma_h);
    const bool left = chroma_location == 1 || chroma_location == 3 || chroma_location == 5;
    const bool vert = chroma_location >= 3 && chroma_location <= 6;       // the switch falls through top / bottom alike
    if (left) wx += (1 << log2_chroma_w) - 1;
    if (vert) wy += (1 << log2_chroma_h) - 1;
    for (int i = 0; i < 2; i++)
    {
        b->coeff[0][i] = (base[i + wx] + base[i + wx + !(wx & 1)]) >> 1;
        b->coeff[1][i] = (base[i + wy] + base[i + wy + !(wy & 1)]) >> 1;
    }
    *out = b;
    return HBHIP_OK;
}

extern "C" void hbhip_blend_destroy(hbhip_blend *b)
{
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    delete b;
}

extern "C" int hbhip_blend_set_overlays(hbhip_blend *b, const hbhip_overlay *ov, int n)
{
    if (!b || n < 0 || (n > 0 && !ov)) return HBHIP_ERR_ARG;
    hbhip_ctx *ctx = b->ctx;
    (void)hipSetDevice(ctx->device);
    // launches of the previous set may still be reading the store
    HBHIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    b->overlays.clear();
    b->launches.clear();
    size_t total = 0;
    for (int i = 0; i < n; i++)
    {
        if (ov[i].width < 1 || ov[i].height < 1) return HBHIP_ERR_ARG;
        const int cw = -((-ov[i].width) >> b->ov_wshift), ch = -((-ov[i].height) >> b->ov_hshift);
        total += 2 * (size_t)hbhip_align_up(ov[i].width, 16) * ov[i].height + 2 * (size_t)hbhip_align_up(cw, 16) * ch;
    }
    if (total > b->store_bytes)
    {
        if (b->d_store) (void)hipFree(b->d_store);
        b->d_store = nullptr;
        b->store_bytes = 0;
        HBHIP_CHECK(ctx, hipMalloc((void **)&b->d_store, total));
        b->store_bytes = total;
    }
    uint8_t *at = b->d_store;
    for (int i = 0; i < n; i++)
    {
        OverlayDev d;
        d.x = ov[i].x; d.y = ov[i].y; d.width = ov[i].width; d.height = ov[i].height;
        const int cw = -((-ov[i].width) >> b->ov_wshift), ch = -((-ov[i].height) >> b->ov_hshift);
        for (int p = 0; p < 4; p++)
        {
            const bool chroma = p == 1 || p == 2;
            const int w = chroma ? cw : ov[i].width, h = chroma ? ch : ov[i].height;
            d.stride[p] = hbhip_align_up(w, 16);
            d.plane[p] = at;
            HBHIP_CHECK(ctx, hipMemcpy2DAsync(at, d.stride[p], ov[i].plane[p], ov[i].stride[p], w, h,
                                              hipMemcpyHostToDevice, ctx->stream));
            at += (size_t)d.stride[p] * h;
        }
        b->overlays.push_back(d);
    }
    build_launches(b);
    // the caller may free its bitmaps as soon as this returns
    HBHIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return HBHIP_OK;
}

extern "C" int hbhip_blend_apply_dev(hbhip_blend *b, const hbhip_dev_frame *frame)
{
    if (!b || !frame) return HBHIP_ERR_ARG;
    hbhip_ctx *ctx = b->ctx;
    (void)hipSetDevice(ctx->device);
    BlendArgs a;
    for (int c = 0; c < 3; c++) { a.dst[c] = (uint8_t *)frame->plane[c]; a.pitch[c] = frame->stride[c]; }
    a.width = b->geo.width; a.height = b->geo.height; a.cw = b->geo.pw[1]; a.ch = b->geo.ph[1];
    a.wshift = b->geo.log2_cw; a.hshift = b->geo.log2_ch; a.shift = b->geo.depth - 8;
    for (int i = 0; i < 2; i++) { a.coeff[0][i] = b->coeff[0][i]; a.coeff[1][i] = b->coeff[1][i]; }
    const dim3 blk(64, 4);
    for (const hbhip_blend::Launch &l : b->launches)
    {
        if (b->subsample)
        {
            if (b->geo.bps == 1) HBHIP_LAUNCH(ctx, "blend_subsample", blend_subsample_kernel<uint8_t>, l.grid, blk, 0, a, l.g);
            else                 HBHIP_LAUNCH(ctx, "blend_subsample", blend_subsample_kernel<uint16_t>, l.grid, blk, 0, a, l.g);
        }
        else
        {
            if (b->geo.bps == 1) HBHIP_LAUNCH(ctx, "blend", blend_same_kernel<uint8_t>, l.grid, blk, 0, a, l.g);
            else                 HBHIP_LAUNCH(ctx, "blend", blend_same_kernel<uint16_t>, l.grid, blk, 0, a, l.g);
        }
    }
    HBHIP_CHECK(ctx, hipGetLastError());
    return HBHIP_OK;
}

extern "C" int hbhip_blend_apply(hbhip_blend *b, const hbhip_host_frame *frame)
{
    if (!b || !frame) return HBHIP_ERR_ARG;
    if (b->overlays.empty()) return HBHIP_OK;
    if (!b->staging)
    {
        int rc = hbhip_frame_alloc(b->ctx, b->geo.width, b->geo.height, b->geo.depth, b->geo.log2_cw, b->geo.log2_ch, &b->staging);
        if (rc != HBHIP_OK) return rc;
    }
    int rc = hbhip_frame_upload(b->staging, frame);
    if (rc != HBHIP_OK) return rc;
    hbhip_dev_frame d;
    rc = hbhip_frame_describe(b->staging, &d, nullptr, nullptr);
    if (rc != HBHIP_OK) return rc;
    rc = hbhip_blend_apply_dev(b, &d);
    if (rc != HBHIP_OK) return rc;
    return hbhip_frame_download(b->staging, frame);
}
