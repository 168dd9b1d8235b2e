// motion_metric.hip — the frame-difference metric vfr.c uses to choose the frame to drop
// (libhb/motion_metric.c; the object hb_motion_metric, :306-312) on gfx950.
//
// Luma of two frames -> one float.  Samples go through the reference's scaled 2.2-gamma table
// (built by the host mirror exactly as build_gamma_lut does, :36-42, and handed over), squared
// differences are summed per 16 x 16 block in 32 bits (sse_block16 returns `unsigned`, :141-158: the
// wrap for a block of extreme differences is part of the result), blocks in 64 bits, and the host
// divides by width * height in float (:190).  Pictures >= 1920 wide or >= 1080 high are compared at
// quarter size (:245-259): each sample is a tree of rounded pair averages over 4 x 4 source samples
// (approximate_frame_data, :44-75) — formed on the fly here, the reduced pictures never exist in HBM.
// Kept from the reference: above 8 bits motion_metric_16 divides the (sample) stride of the reduced
// pictures by the sample size once more (:176-177 after :207-210), so it walks them with half their
// stride; sample (x, y) of the comparison is element y * (w / 2) + x of the reduced picture.
//
// One workgroup of 256 threads = one 16 x 16 block; wave-level then LDS reduction of the wrapped
// 32-bit block sum; one 64-bit atomic per block.  Integer adds commute, so the result is exact
// whatever the order.  HBM-bound: both lumas are read once (2 x 2.07 MB at 1080p).
#include "hbhip_internal.h"

namespace {

template <typename PIX> __device__ __forceinline__ unsigned avg2(unsigned a, unsigned b) { return (a + b + 1) >> 1; }

template <typename PIX>
__device__ __forceinline__ unsigned sample_at(const uint8_t *plane, int pitch, int x, int y, bool fast)
{
    if (!fast) return reinterpret_cast<const PIX *>(plane + (size_t)y * pitch)[x];
    unsigned q[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        const int sx = 4 * x + 2 * (k & 1), sy = 4 * y + 2 * (k >> 1);
        const PIX *r0 = reinterpret_cast<const PIX *>(plane + (size_t)sy * pitch) + sx;
        const PIX *r1 = reinterpret_cast<const PIX *>(plane + (size_t)(sy + 1) * pitch) + sx;
        q[k] = avg2<PIX>(avg2<PIX>(r0[0], r1[0]), avg2<PIX>(r0[1], r1[1]));
    }
    return avg2<PIX>(avg2<PIX>(q[0], q[1]), avg2<PIX>(q[2], q[3]));
}

template <typename PIX>
__global__ __launch_bounds__(256) void motion_metric_kernel(const uint8_t *__restrict__ a, int pitch_a,
                                                            const uint8_t *__restrict__ b, int pitch_b,
                                                            const unsigned *__restrict__ lut, int fast, int w,
                                                            unsigned long long *__restrict__ total)
{
    __shared__ unsigned s_part[4];
    int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (fast && sizeof(PIX) == 2)
    {
        // the reference walks its 16-bit reduced pictures with half their stride (see the file header)
        const int at = y * (w / 2) + x;
        x = at % w;
        y = at / w;
    }
    const unsigned va = sample_at<PIX>(a, pitch_a, x, y, fast != 0), vb = sample_at<PIX>(b, pitch_b, x, y, fast != 0);
    const int diff = (int)(lut[va] - lut[vb]);
    unsigned v = (unsigned)(diff * diff);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicAdd(total, (unsigned long long)(unsigned)(s_part[0] + s_part[1] + s_part[2] + s_part[3]));
}

} // namespace

struct hbhip_motion_metric
{
    hbhip_ctx *ctx = nullptr;
    int width = 0, height = 0, depth = 8, bps = 1;
    unsigned *d_lut = nullptr;
    unsigned long long *d_total = nullptr;
    uint8_t *d_a = nullptr, *d_b = nullptr;      // staging of the host entry point
    int pitch = 0;

    ~hbhip_motion_metric()
    {
        if (d_lut) (void)hipFree(d_lut);
        if (d_total) (void)hipFree(d_total);
        if (d_a) (void)hipFree(d_a);
        if (d_b) (void)hipFree(d_b);
    }
};

extern "C" int hbhip_motion_metric_create(hbhip_ctx *ctx, int width, int height, int depth,
                                          const unsigned *gamma_lut, int entries, hbhip_motion_metric **out)
{
    if (!ctx || !out || !gamma_lut) return HBHIP_ERR_ARG;
    *out = nullptr;
    if (depth < 8 || depth > 16 || entries != (1 << depth)) return HBHIP_ERR_ARG;
    if (width < 16 || height < 16) return HBHIP_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    hbhip_motion_metric *m = new (std::nothrow) hbhip_motion_metric;
    if (!m) return HBHIP_ERR_NOMEM;
    m->ctx = ctx; m->width = width; m->height = height; m->depth = depth; m->bps = depth > 8 ? 2 : 1;
    if (hipMalloc((void **)&m->d_lut, sizeof(unsigned) * entries) != hipSuccess ||
        hipMalloc((void **)&m->d_total, sizeof(unsigned long long)) != hipSuccess ||
        hipMemcpy(m->d_lut, gamma_lut, sizeof(unsigned) * entries, hipMemcpyHostToDevice) != hipSuccess)
    {
        delete m;
        return HBHIP_ERR_NOMEM;
    }
    *out = m;
    return HBHIP_OK;
}

extern "C" void hbhip_motion_metric_destroy(hbhip_motion_metric *m)
{
    if (!m) return;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    delete m;
}

extern "C" int hbhip_motion_metric_run_dev(hbhip_motion_metric *m, const void *luma_a, int stride_a,
                                           const void *luma_b, int stride_b, float *out)
{
    if (!m || !luma_a || !luma_b || !out) return HBHIP_ERR_ARG;
    hbhip_ctx *ctx = m->ctx;
    (void)hipSetDevice(ctx->device);
    const int fast = m->width >= 1920 || m->height >= 1080;                       // motion_metric.c:245-246
    const int w = fast ? m->width / 4 : m->width, h = fast ? m->height / 4 : m->height;
    const int bw = w / 16, bh = h / 16;
    unsigned long long sum = 0;
    if (bw > 0 && bh > 0)
    {
        HBHIP_CHECK(ctx, hipMemsetAsync(m->d_total, 0, sizeof(unsigned long long), ctx->stream));
        if (m->bps == 1)
            HBHIP_LAUNCH(ctx, "motion_metric", motion_metric_kernel<uint8_t>, dim3(bw, bh), dim3(256), 0,
                         (const uint8_t *)luma_a, stride_a, (const uint8_t *)luma_b, stride_b, (const unsigned *)m->d_lut, fast, w, m->d_total);
        else
            HBHIP_LAUNCH(ctx, "motion_metric", motion_metric_kernel<uint16_t>, dim3(bw, bh), dim3(256), 0,
                         (const uint8_t *)luma_a, stride_a, (const uint8_t *)luma_b, stride_b, (const unsigned *)m->d_lut, fast, w, m->d_total);
        HBHIP_CHECK(ctx, hipGetLastError());
        HBHIP_CHECK(ctx, hipMemcpyAsync(&sum, m->d_total, sizeof(sum), hipMemcpyDeviceToHost, ctx->stream));
        HBHIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    *out = (float)sum / (w * h);                                                  // :190
    return HBHIP_OK;
}

extern "C" int hbhip_motion_metric_run(hbhip_motion_metric *m, const uint8_t *luma_a, int stride_a,
                                       const uint8_t *luma_b, int stride_b, float *out)
{
    if (!m || !luma_a || !luma_b || !out) return HBHIP_ERR_ARG;
    hbhip_ctx *ctx = m->ctx;
    (void)hipSetDevice(ctx->device);
    if (!m->d_a)
    {
        m->pitch = hbhip_align_up(m->width * m->bps, 256);
        HBHIP_CHECK(ctx, hipMalloc((void **)&m->d_a, (size_t)m->pitch * m->height));
        HBHIP_CHECK(ctx, hipMalloc((void **)&m->d_b, (size_t)m->pitch * m->height));
    }
    HBHIP_CHECK(ctx, hipMemcpy2DAsync(m->d_a, m->pitch, luma_a, stride_a, (size_t)m->width * m->bps, m->height,
                                      hipMemcpyHostToDevice, ctx->stream));
    HBHIP_CHECK(ctx, hipMemcpy2DAsync(m->d_b, m->pitch, luma_b, stride_b, (size_t)m->width * m->bps, m->height,
                                      hipMemcpyHostToDevice, ctx->stream));
    return hbhip_motion_metric_run_dev(m, m->d_a, m->pitch, m->d_b, m->pitch, out);
}
