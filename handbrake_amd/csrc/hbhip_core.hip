// hbhip_core.hip — context, stream, event-based per-kernel timing, device
// picture pool and the generic push/pull half of the C ABI (include/hbhip.h).
#include "hbhip_internal.h"

#include <algorithm>
#include <new>

// ---------------------------------------------------------------- ctx helpers
int hbhip_ctx::fail(hipError_t e, const char *what)
{
    std::lock_guard<std::recursive_mutex> lk(state_lock);
    last_error = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    if (e == hipErrorOutOfMemory) return HBHIP_ERR_NOMEM;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return HBHIP_ERR_NODEVICE;
    return HBHIP_ERR_HIP;
}

hipEvent_t hbhip_ctx::ev_get()
{
    if (!ev_pool.empty())
    {
        hipEvent_t e = ev_pool.back();
        ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

hipEvent_t hbhip_ctx::sync_ev_get()
{
    {
        std::lock_guard<std::recursive_mutex> lk(state_lock);
        if (!sync_ev_pool.empty())
        {
            hipEvent_t e = sync_ev_pool.back();
            sync_ev_pool.pop_back();
            return e;
        }
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
}

void hbhip_ctx::sync_ev_put(hipEvent_t e)
{
    if (!e) return;
    std::lock_guard<std::recursive_mutex> lk(state_lock);
    sync_ev_pool.push_back(e);
}

void IdleMark::record_now()
{
    std::lock_guard<std::mutex> lk(lock);
    if (recorded) return;
    closed.store(true, std::memory_order_release);     // whoever still sees it open attached before the record below
    (void)hipEventRecord(ev, stream);
    recorded = true;
}

std::shared_ptr<IdleMark> hbhip_ctx::mark()
{
    std::lock_guard<std::recursive_mutex> lk(state_lock);
    if (open_mark && open_mark->closed.load(std::memory_order_acquire)) open_mark.reset();   // recorded through a picture
    if (!open_mark)
    {
        std::shared_ptr<IdleMark> m = std::make_shared<IdleMark>();
        if (hipEventCreateWithFlags(&m->ev, hipEventDisableTiming) != hipSuccess)
        {
            (void)hipGetLastError();
            return nullptr;
        }
        m->stream = stream;
        open_mark = m;
        has_open_mark.store(true, std::memory_order_release);
    }
    return open_mark;
}

void hbhip_ctx::close_mark()
{
    std::shared_ptr<IdleMark> m;
    {
        std::lock_guard<std::recursive_mutex> lk(state_lock);
        m.swap(open_mark);
        has_open_mark.store(false, std::memory_order_release);
    }
    if (m) m->record_now();
}

void hbhip_pic_mark_idle(hbhip_ctx *ctx, DevPicture *p)
{
    if (!ctx || !p) return;
    p->idle = ctx->mark();
    // no event to be had: the next user could not wait for this one's work, so it is waited out here
    if (!p->idle) (void)hipStreamSynchronize(ctx->stream);
}

bool hbhip_pic_idle_done(DevPicture *p)
{
    if (!p || !p->idle) return true;
    p->idle->record_now();
    if (hipEventQuery(p->idle->ev) == hipSuccess) return true;
    (void)hipGetLastError();                           // hipErrorNotReady
    return false;
}

hipError_t hbhip_pic_wait_idle(hipStream_t stream, DevPicture *p)
{
    if (!p || !p->idle || p->idle->stream == stream) return hipSuccess;
    p->idle->record_now();
    return hipStreamWaitEvent(stream, p->idle->ev, 0);
}

int hbhip_ctx::prof_name(const char *name)
{
    for (size_t i = 0; i < prof_stats.size(); i++)
        if (prof_stats[i].name == name) return (int)i;
    hbhip_prof_stat s;
    s.name = name;
    prof_stats.push_back(s);
    return (int)prof_stats.size() - 1;
}

void hbhip_ctx::prof_begin(const char *name)
{
    std::lock_guard<std::recursive_mutex> lk(state_lock);
    if (prof_pending.size() >= 8192) prof_resolve();
    hbhip_prof_pending p;
    p.name_idx = prof_name(name);
    p.ev0 = ev_get();
    p.ev1 = ev_get();
    if (p.ev0) (void)hipEventRecord(p.ev0, stream);
    prof_pending.push_back(p);
}

void hbhip_ctx::prof_end()
{
    std::lock_guard<std::recursive_mutex> lk(state_lock);
    if (prof_pending.empty()) return;
    hbhip_prof_pending &p = prof_pending.back();
    if (p.ev1) (void)hipEventRecord(p.ev1, stream);
}

void hbhip_ctx::prof_resolve()
{
    std::lock_guard<std::recursive_mutex> lk(state_lock);
    if (prof_pending.empty()) return;
    (void)hipStreamSynchronize(stream);
    for (auto &p : prof_pending)
    {
        float ms = 0.f;
        if (p.ev0 && p.ev1 && hipEventElapsedTime(&ms, p.ev0, p.ev1) == hipSuccess)
        {
            prof_stats[p.name_idx].launches++;
            prof_stats[p.name_idx].total_ms += ms;
        }
        if (p.ev0) ev_pool.push_back(p.ev0);
        if (p.ev1) ev_pool.push_back(p.ev1);
    }
    prof_pending.clear();
}

// ---------------------------------------------------------------- pool
PicturePool::~PicturePool()
{
    for (DevPicture *p : all_)
    {
        if (p->base) (void)hipFree(p->base);
        delete p;
    }
}

void PicturePool::configure(hbhip_ctx *ctx, const PicGeometry &g, int pitch_align, int pad_rows)
{
    ctx_ = ctx;
    geo_ = g;
    pitch_align_ = pitch_align;
    pad_rows_ = pad_rows;
}

DevPicture *PicturePool::acquire()
{
    if (frames_)
    {
        hbhip_frame *fr = nullptr;
        if (hbhip_frame_alloc(ctx_, geo_.width, geo_.height, geo_.depth, geo_.log2_cw, geo_.log2_ch, &fr) != HBHIP_OK) return nullptr;
        return &fr->pic;
    }
    if (!free_.empty())
    {
        // Most recently released first (warm in L2) when its last user ran on this pool's stream.  One last used on
        // ANOTHER stream (a later stage of a chain with a stream per stage) is only taken once that user is done:
        // waiting for it here would chain this stage's next batch behind the later stage's current one.  While there
        // is no such picture the pool grows (to a bound), after that the longest-released one is waited for.
        for (size_t i = free_.size(); i-- > 0;)
        {
            DevPicture *p = free_[i];
            if (p->idle && p->idle->stream != ctx_->stream && !hbhip_pic_idle_done(p)) continue;
            free_.erase(free_.begin() + (ptrdiff_t)i);
            return p;
        }
        if (all_.size() >= max_pictures_)
        {
            DevPicture *p = free_.front();
            free_.erase(free_.begin());
            (void)hbhip_pic_wait_idle(ctx_->stream, p);
            return p;
        }
    }
    DevPicture *p = new (std::nothrow) DevPicture();
    if (!p) return nullptr;
    size_t off[3], total = 0;
    for (int c = 0; c < 3; c++)
    {
        p->width[c] = geo_.pw[c];
        p->height[c] = geo_.ph[c];
        p->pitch[c] = hbhip_align_up(geo_.pw[c] * geo_.bps, pitch_align_);
        off[c] = total;
        total += (size_t)p->pitch[c] * (geo_.ph[c] + pad_rows_);
        total = (total + 255) & ~(size_t)255;
    }
    p->bps = geo_.bps;
    p->bytes = total;
    if (hipMalloc((void **)&p->base, total) != hipSuccess)
    {
        (void)hipGetLastError();
        delete p;
        return nullptr;
    }
    (void)hipMemsetAsync(p->base, 0, total, ctx_->stream);
    hbhip_pic_mark_idle(ctx_, p);          // an upload (on the copy stream) must not overtake the clearing
    for (int c = 0; c < 3; c++) p->plane[c] = p->base + off[c];
    p->owner = this;
    all_.push_back(p);
    return p;
}

void PicturePool::release(DevPicture *p, hbhip_ctx *last_user)
{
    if (!p) return;
    if (p->frame) { hbhip_frame_release(p->frame); return; }      // a frame's picture, whoever made it
    hbhip_pic_mark_idle(last_user ? last_user : ctx_, p);   // an upload into the recycled picture waits for this (hbhip_copy_h2d)
    free_.push_back(p);
}

// ---------------------------------------------------------------- copies
// Uploads carry the caller's row padding too (up to our pitch): a few reference
// filters read into it (lapsharp.c:145-157 when stride > width).
// The host frame and the device picture have the same strides and their planes lie one behind the other in both:
// the whole frame is one contiguous block on either side (hbhip_frame_alloc lays its frames out like hb_frame_buffer_init).
static bool same_layout(const DevPicture *p, uint8_t *const plane[3], const int stride[3])
{
    for (int c = 0; c < 3; c++)
    {
        if (stride[c] != p->pitch[c]) return false;
        if (c && plane[c] - plane[c - 1] != p->plane[c] - p->plane[c - 1]) return false;
        if (c && p->plane[c] - p->plane[c - 1] != (ptrdiff_t)p->pitch[c - 1] * p->height[c - 1]) return false;
    }
    return true;
}
static size_t layout_bytes(const DevPicture *p)
{
    return (size_t)(p->plane[2] - p->plane[0]) + (size_t)p->pitch[2] * p->height[2];
}

// the H2D copy of a frame queued on the context's upload stream, and an event behind it (from the context's pool: the
// caller gives it back with sync_ev_put once it has fired)
static int hbhip_copy_h2d_queue(hbhip_ctx *ctx, DevPicture *dst, const hbhip_host_frame *src, hipEvent_t *done_out)
{
    for (int c = 0; c < 3; c++)
        if (src->plane[c] == nullptr || src->stride[c] < dst->width[c] * dst->bps) return HBHIP_ERR_ARG;
    hipEvent_t done = ctx->sync_ev_get();
    if (!done) return ctx->fail(hipErrorOutOfMemory, "hipEventCreate(upload)");
    auto fail = [&](hipError_t e, const char *what) { ctx->sync_ev_put(done); return ctx->fail(e, what); };
    // whatever still reads the picture's previous contents was queued ahead of its idle mark when it was recycled
    hipError_t e = hbhip_pic_wait_idle(ctx->up(), dst);
    if (e != hipSuccess) return fail(e, "upload: wait for the picture's last reader");
    if (same_layout(dst, src->plane, src->stride))
        e = hipMemcpyAsync(dst->plane[0], src->plane[0], layout_bytes(dst), hipMemcpyHostToDevice, ctx->up());
    else
        for (int c = 0; c < 3 && e == hipSuccess; c++)
        {
            const size_t row = (size_t)std::min(src->stride[c], dst->pitch[c]);
            e = hipMemcpy2DAsync(dst->plane[c], dst->pitch[c], src->plane[c], src->stride[c],
                                 row, dst->height[c], hipMemcpyHostToDevice, ctx->up());
        }
    if (e == hipSuccess) e = hipEventRecord(done, ctx->up());
    if (e != hipSuccess)
    {
        (void)hipStreamSynchronize(ctx->up());                // planes already queued must not outlive the call
        return fail(e, "hipMemcpyAsync(upload)");
    }
    *done_out = done;
    return HBHIP_OK;
}

int hbhip_copy_h2d(hbhip_ctx *ctx, DevPicture *dst, const hbhip_host_frame *src)
{
    hipEvent_t done = nullptr;
    const int rc = hbhip_copy_h2d_queue(ctx, dst, src, &done);
    if (rc != HBHIP_OK) return rc;
    // The caller may free or reuse its planes as soon as we return (filter_loop closes buf_in, work.c:2566-2569):
    // wait for THIS copy - not for the kernels other filters have queued.  Once it has completed, work launched
    // on any stream afterwards sees the data.
    const hipError_t e = hipEventSynchronize(done);
    ctx->sync_ev_put(done);
    if (e != hipSuccess) return ctx->fail(e, "hipEventSynchronize(upload)");
    return HBHIP_OK;
}

int hbhip_copy_d2h(hbhip_ctx *ctx, const hbhip_host_frame *dst, const DevPicture *src)
{
    for (int c = 0; c < 3; c++)
        if (dst->plane[c] == nullptr || dst->stride[c] < src->width[c] * src->bps) return HBHIP_ERR_ARG;
    hipEvent_t ev = ctx->sync_ev_get();
    if (!ev) return ctx->fail(hipErrorOutOfMemory, "hipEventCreate(download)");
    // the picture's producers are on ctx->stream, all queued by now
    HBHIP_CHECK(ctx, hipEventRecord(ev, ctx->stream));
    HBHIP_CHECK(ctx, hipStreamWaitEvent(ctx->down(), ev, 0));
    for (int c = 0; c < 3; c++)
    {
        const size_t row = (size_t)src->width[c] * src->bps;
        HBHIP_CHECK(ctx, hipMemcpy2DAsync(dst->plane[c], dst->stride[c], src->plane[c], src->pitch[c],
                                          row, src->height[c], hipMemcpyDeviceToHost, ctx->down()));
    }
    HBHIP_CHECK(ctx, hipEventRecord(ev, ctx->down()));
    const hipError_t e = hipEventSynchronize(ev);
    ctx->sync_ev_put(ev);
    if (e != hipSuccess) return ctx->fail(e, "hipEventSynchronize(download)");
    return HBHIP_OK;
}

// Device-to-device copies of a picture's three planes as ONE kernel launch: hipMemcpy2DAsync makes a blit kernel per
// plane and brackets it with system-scope barrier packets - 25 us of idle queue per picture in the kernel trace of the
// NLMeans workload (tools/trace_gaps.py).
namespace {
struct PlaneCopy3
{
    const uint8_t *src[3];
    uint8_t       *dst[3];
    int spitch[3], dpitch[3], row_bytes[3], rows[3];
};

__global__ void __launch_bounds__(256) plane_copy3_kernel(PlaneCopy3 a)
{
    const int pl = blockIdx.z, y = blockIdx.y;
    if (y >= a.rows[pl]) return;
    const int x = (blockIdx.x * 256 + threadIdx.x) * 16;
    const int rb = a.row_bytes[pl];
    if (x >= rb) return;
    const uint8_t *s = a.src[pl] + (size_t)y * a.spitch[pl] + x;
    uint8_t *d = a.dst[pl] + (size_t)y * a.dpitch[pl] + x;
    if (x + 16 <= rb && (((uintptr_t)s | (uintptr_t)d) & 15) == 0)
        *reinterpret_cast<uint4 *>(d) = *reinterpret_cast<const uint4 *>(s);
    else
        for (int i = 0; i < 16 && x + i < rb; i++) d[i] = s[i];
}

int plane_copy3(hbhip_ctx *ctx, const PlaneCopy3 &a)
{
    int maxrow = 0, maxrows = 0;
    for (int c = 0; c < 3; c++) { maxrow = std::max(maxrow, a.row_bytes[c]); maxrows = std::max(maxrows, a.rows[c]); }
    HBHIP_LAUNCH(ctx, "copy_planes", plane_copy3_kernel, dim3((maxrow + 4095) / 4096, maxrows, 3), dim3(256), 0, a);
    HBHIP_CHECK(ctx, hipGetLastError());
    return HBHIP_OK;
}
} // namespace

int hbhip_copy_d2d_in(hbhip_ctx *ctx, DevPicture *dst, const hbhip_dev_frame *src)
{
    PlaneCopy3 a;
    for (int c = 0; c < 3; c++)
    {
        const size_t vis = (size_t)dst->width[c] * dst->bps;
        if (src->plane[c] == nullptr || src->stride[c] < (int)vis) return HBHIP_ERR_ARG;
        a.src[c] = (const uint8_t *)src->plane[c]; a.dst[c] = dst->plane[c];
        a.spitch[c] = src->stride[c]; a.dpitch[c] = dst->pitch[c];
        a.row_bytes[c] = std::min(src->stride[c], dst->pitch[c]);      // the caller's row padding travels too (up to our pitch)
        a.rows[c] = dst->height[c];
    }
    return plane_copy3(ctx, a);
}

int hbhip_copy_d2d_out(hbhip_ctx *ctx, const hbhip_dev_frame *dst, const DevPicture *src)
{
    PlaneCopy3 a;
    for (int c = 0; c < 3; c++)
    {
        const size_t row = (size_t)src->width[c] * src->bps;
        if (dst->plane[c] == nullptr || dst->stride[c] < (int)row) return HBHIP_ERR_ARG;
        a.src[c] = src->plane[c]; a.dst[c] = (uint8_t *)dst->plane[c];
        a.spitch[c] = src->pitch[c]; a.dpitch[c] = dst->stride[c];
        a.row_bytes[c] = (int)row; a.rows[c] = src->height[c];
    }
    return plane_copy3(ctx, a);
}

int hbhip_filter::process_dev_batch(const hbhip_dev_frame *in, int n_in, int64_t tag0,
                                    const hbhip_dev_frame *out, int out_cap, int *n_out)
{
    int produced = 0;
    for (int i = 0; i < n_in; i++)
    {
        int rc = hbhip_filter_push_dev(this, &in[i], tag0 + i);
        if (rc != HBHIP_OK) return rc;
        while (produced < out_cap && pending() > 0)
        {
            rc = hbhip_filter_pull_dev(this, &out[produced], nullptr);
            if (rc != HBHIP_OK) return rc;
            produced++;
        }
    }
    *n_out = produced;
    return HBHIP_OK;
}

// ---------------------------------------------------------------- C ABI
// ---- the on-box copy ceiling ("measure an on-box ... ceiling and report against both") -----------------
// A float4 grid-stride copy, the way /opt/skills/guides/MI355X_MICROARCH.md measures its 6.29 TB/s: 16 bytes per lane and
// trip, enough workgroups to fill every CU several times over, buffers far past the 256 MB Infinity Cache.
typedef float f32x4 __attribute__((ext_vector_type(4)));
// U float4 per lane in flight and trip; NT: non-temporal loads and stores (neither buffer is touched again)
template <int U, bool NT>
__global__ __launch_bounds__(1024) void copy_f4_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (size_t)(U - 1) * stride < n; i += (size_t)U * stride)
    {
        f32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = NT ? __builtin_nontemporal_load(&src[i + (size_t)k * stride]) : src[i + (size_t)k * stride];
#pragma unroll
        for (int k = 0; k < U; k++)
        {
            if (NT) __builtin_nontemporal_store(v[k], &dst[i + (size_t)k * stride]);
            else dst[i + (size_t)k * stride] = v[k];
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

extern "C" {

int hbhip_abi_version(void) { return HBHIP_ABI_VERSION; }

int hbhip_host_alloc(size_t bytes, void **out)
{
    if (!out) return HBHIP_ERR_ARG;
    *out = nullptr;
    if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess)
    {
        (void)hipGetLastError();
        *out = nullptr;
        return HBHIP_ERR_NOMEM;
    }
    return HBHIP_OK;
}

void hbhip_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int hbhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
    {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char *hbhip_strerror(int code)
{
    switch (code)
    {
        case HBHIP_OK:              return "ok";
        case HBHIP_AGAIN:           return "no frame ready yet";
        case HBHIP_ERR_NODEVICE:    return "no usable HIP device";
        case HBHIP_ERR_HIP:         return "HIP runtime error (see hbhip_ctx_last_error)";
        case HBHIP_ERR_ARG:         return "invalid argument";
        case HBHIP_ERR_NOMEM:       return "out of (device) memory";
        case HBHIP_ERR_UNSUPPORTED: return "settings not supported by the HIP path";
        case HBHIP_ERR_STATE:       return "call not valid in this state";
        default:                    return "unknown hbhip error";
    }
}

// the copy streams, made when first asked for (hbhip_internal.h); without one the copies go to the context's own stream
static hipStream_t lazy_stream(hbhip_ctx *ctx, std::once_flag &once, hipStream_t &slot)
{
    std::call_once(once, [&] {
        (void)hipSetDevice(ctx->device);
        if (hipStreamCreateWithFlags(&slot, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); slot = nullptr; }
    });
    return slot ? slot : ctx->stream;
}
hipStream_t hbhip_ctx::up() { return lazy_stream(this, up_once, up_stream); }
hipStream_t hbhip_ctx::down() { return lazy_stream(this, down_once, down_stream); }

static int ctx_create_common(int device, void *stream, bool adopt, hbhip_ctx **out)
{
    if (out == nullptr) return HBHIP_ERR_ARG;
    *out = nullptr;
    int n = hbhip_device_count();
    if (n <= 0 || device < 0 || device >= n) return HBHIP_ERR_NODEVICE;
    if (hipSetDevice(device) != hipSuccess)
    {
        (void)hipGetLastError();
        return HBHIP_ERR_NODEVICE;
    }
    hbhip_ctx *ctx = new (std::nothrow) hbhip_ctx();
    if (!ctx) return HBHIP_ERR_NOMEM;
    ctx->device = device;
    if (adopt)
    {
        ctx->stream = (hipStream_t)stream;
        ctx->own_stream = false;
    }
    else if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess)
    {
        (void)hipGetLastError();
        delete ctx;
        return HBHIP_ERR_HIP;
    }
#ifndef HBHIP_LAZY_COPY_STREAMS
    // The copy streams with the context.  (Made at their first use instead - -DHBHIP_LAZY_COPY_STREAMS - a chain whose
    // frames never leave the device has two streams fewer per context, which NLMeans alone likes (37.8 -> 38.7 k fps) and
    // the chain does not notice; but the host path's download stream, made last then, lands on a hardware queue it shares
    // with a busy compute stream: 4 070 -> 3 710 fps PCIe-inclusive, 56.9 -> 51.9 GB/s.  profiles/r6Z_streams_and_queues.log)
    (void)ctx->up(); (void)ctx->down();
#endif
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess)
        snprintf(ctx->dev_name, sizeof(ctx->dev_name), "%s (%s, %d CUs)", prop.name,
                 prop.gcnArchName, prop.multiProcessorCount);
    *out = ctx;
    return HBHIP_OK;
}

int hbhip_ctx_create(int device, hbhip_ctx **out)
{
    return ctx_create_common(device, nullptr, false, out);
}

int hbhip_ctx_create_on_stream(int device, void *hip_stream, hbhip_ctx **out)
{
    return ctx_create_common(device, hip_stream, true, out);
}

void hbhip_ctx_destroy(hbhip_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    ctx->close_mark();
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->up_stream) { (void)hipStreamSynchronize(ctx->up_stream); (void)hipStreamDestroy(ctx->up_stream); }
    if (ctx->down_stream) { (void)hipStreamSynchronize(ctx->down_stream); (void)hipStreamDestroy(ctx->down_stream); }
    for (hipEvent_t e : ctx->sync_ev_pool) (void)hipEventDestroy(e);
    for (auto &p : ctx->prof_pending)
    {
        if (p.ev0) (void)hipEventDestroy(p.ev0);
        if (p.ev1) (void)hipEventDestroy(p.ev1);
    }
    for (hbhip_frame *fr : ctx->frame_pool)
    {
        if (fr->pic.base) (void)hipFree(fr->pic.base);
        delete fr;
    }
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < HBHIP_MAX_MARKS; i++)
        if (ctx->marks[i]) (void)hipEventDestroy(ctx->marks[i]);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int hbhip_ctx_sync(hbhip_ctx *ctx)
{
    if (!ctx) return HBHIP_ERR_ARG;
    HBHIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return HBHIP_OK;
}

const char *hbhip_ctx_last_error(hbhip_ctx *ctx)
{
    if (!ctx) return "";
    // a copy per calling thread: the string itself may be rewritten by another filter's thread
    static thread_local std::string copy;
    std::lock_guard<std::recursive_mutex> lk(ctx->state_lock);
    copy = ctx->last_error;
    return copy.c_str();
}

int hbhip_ctx_device_name(hbhip_ctx *ctx, char *buf, int len)
{
    if (!ctx || !buf || len <= 0) return HBHIP_ERR_ARG;
    snprintf(buf, len, "%s", ctx->dev_name);
    return HBHIP_OK;
}

int hbhip_ctx_copy_bandwidth(hbhip_ctx *ctx, size_t bytes, int iters, double *gbps)
{
    if (!ctx || !gbps || bytes < (1u << 20) || iters < 1) return HBHIP_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    float4 *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const size_t n = bytes / sizeof(float4);
    int rc = HBHIP_OK;
    double best_ms = 0.0;
    if (hipMalloc((void **)&a, n * sizeof(float4)) != hipSuccess || hipMalloc((void **)&b, n * sizeof(float4)) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        rc = ctx->fail(hipErrorOutOfMemory, "hbhip_ctx_copy_bandwidth: buffers");
    if (rc == HBHIP_OK && hipMemsetAsync(a, 1, n * sizeof(float4), ctx->stream) != hipSuccess) rc = HBHIP_ERR_HIP;
    // the ceiling is whatever the best of a few shapes reaches: 1 / 4 / 8 float4 in flight per lane, plain or non-temporal
    // accesses, 256 or 1024 lanes per workgroup, 8 or 32 workgroups per CU - and the runtime's own device-to-device copy
    struct Shape { int u, nt, block, per_cu; };
    static const Shape shapes[] = { {1, 0, 256, 8}, {4, 0, 256, 8}, {4, 1, 256, 8}, {8, 1, 256, 8}, {4, 0, 256, 32}, {4, 1, 256, 32},
                                    {4, 1, 1024, 2}, {8, 1, 1024, 2}, {8, 0, 512, 4}, {0, 0, 0, 0} /* hipMemcpyAsync */ };
    for (const Shape &sh : shapes)
        for (int i = 0; rc == HBHIP_OK && i < iters + 1; i++)                    // the first pass of a shape warms up
        {
            (void)hipEventRecord(e0, ctx->stream);
            if (sh.u == 0)
                (void)hipMemcpyAsync(b, a, n * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream);
            else
            {
                const dim3 grid(256 * sh.per_cu), block(sh.block);
                const f32x4 *pa = (const f32x4 *)a;
                f32x4 *pb = (f32x4 *)b;
                if (sh.u == 1)                copy_f4_kernel<1, false><<<grid, block, 0, ctx->stream>>>(pa, pb, n);
                else if (sh.u == 4 && !sh.nt) copy_f4_kernel<4, false><<<grid, block, 0, ctx->stream>>>(pa, pb, n);
                else if (sh.u == 4)           copy_f4_kernel<4, true><<<grid, block, 0, ctx->stream>>>(pa, pb, n);
                else if (!sh.nt)              copy_f4_kernel<8, false><<<grid, block, 0, ctx->stream>>>(pa, pb, n);
                else                          copy_f4_kernel<8, true><<<grid, block, 0, ctx->stream>>>(pa, pb, n);
            }
            (void)hipEventRecord(e1, ctx->stream);
            if (hipEventSynchronize(e1) != hipSuccess) { rc = ctx->fail(hipGetLastError(), "copy_f4_kernel"); break; }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (i > 0 && (best_ms == 0.0 || ms < best_ms)) best_ms = ms;
        }
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (rc == HBHIP_OK) *gbps = best_ms > 0.0 ? 2.0 * (double)(n * sizeof(float4)) / (best_ms * 1e-3) / 1e9 : 0.0;   // read + write
    return rc;
}

int hbhip_ctx_device_index(hbhip_ctx *ctx)
{
    return ctx ? ctx->device : -1;
}

int hbhip_ctx_profile_enable(hbhip_ctx *ctx, int on)
{
    if (!ctx) return HBHIP_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(ctx->state_lock);
    if (!on) ctx->prof_resolve();
    ctx->profile = on != 0;
    return HBHIP_OK;
}

int hbhip_ctx_profile_reset(hbhip_ctx *ctx)
{
    if (!ctx) return HBHIP_ERR_ARG;
    ctx->prof_resolve();
    ctx->prof_stats.clear();
    return HBHIP_OK;
}

int hbhip_ctx_profile_count(hbhip_ctx *ctx)
{
    if (!ctx) return HBHIP_ERR_ARG;
    ctx->prof_resolve();
    return (int)ctx->prof_stats.size();
}

int hbhip_ctx_profile_get(hbhip_ctx *ctx, int idx, char *name, int name_len,
                          int64_t *launches, double *total_ms)
{
    if (!ctx || idx < 0 || idx >= (int)ctx->prof_stats.size()) return HBHIP_ERR_ARG;
    const hbhip_prof_stat &s = ctx->prof_stats[idx];
    if (name && name_len > 0) snprintf(name, name_len, "%s", s.name.c_str());
    if (launches) *launches = s.launches;
    if (total_ms) *total_ms = s.total_ms;
    return HBHIP_OK;
}

int hbhip_ctx_mark(hbhip_ctx *ctx, int slot)
{
    if (!ctx || slot < 0 || slot >= HBHIP_MAX_MARKS) return HBHIP_ERR_ARG;
    if (!ctx->marks[slot]) HBHIP_CHECK(ctx, hipEventCreate(&ctx->marks[slot]));
    HBHIP_CHECK(ctx, hipEventRecord(ctx->marks[slot], ctx->stream));
    return HBHIP_OK;
}

int hbhip_ctx_elapsed_ms(hbhip_ctx *ctx, int a, int b, double *ms)
{
    if (!ctx || !ms || a < 0 || b < 0 || a >= HBHIP_MAX_MARKS || b >= HBHIP_MAX_MARKS ||
        !ctx->marks[a] || !ctx->marks[b])
        return HBHIP_ERR_ARG;
    HBHIP_CHECK(ctx, hipEventSynchronize(ctx->marks[b]));
    float f = 0.f;
    HBHIP_CHECK(ctx, hipEventElapsedTime(&f, ctx->marks[a], ctx->marks[b]));
    *ms = f;
    return HBHIP_OK;
}

int hbhip_dev_alloc(hbhip_ctx *ctx, size_t bytes, void **out)
{
    if (!ctx || !out) return HBHIP_ERR_ARG;
    HBHIP_CHECK(ctx, hipSetDevice(ctx->device));
    HBHIP_CHECK(ctx, hipMalloc(out, bytes));
    return HBHIP_OK;
}

int hbhip_dev_free(hbhip_ctx *ctx, void *p)
{
    if (!ctx) return HBHIP_ERR_ARG;
    HBHIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    HBHIP_CHECK(ctx, hipFree(p));
    return HBHIP_OK;
}

int hbhip_dev_upload(hbhip_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx) return HBHIP_ERR_ARG;
    HBHIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HBHIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return HBHIP_OK;
}

int hbhip_dev_download(hbhip_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx) return HBHIP_ERR_ARG;
    HBHIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HBHIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return HBHIP_OK;
}

// ---- device-resident frames -----------------------------------------------------
hbhip_ctx *hbhip_frame_context(hbhip_frame *fr)
{
    return fr ? fr->ctx : nullptr;
}

int hbhip_frame_alloc(hbhip_ctx *ctx, int width, int height, int depth, int lcw, int lch, hbhip_frame **out)
{
    if (!ctx || !out || width < 1 || height < 1) return HBHIP_ERR_ARG;
    *out = nullptr;
    (void)hipSetDevice(ctx->device);
    std::unique_lock<std::mutex> lk(ctx->frame_lock);
    {
        // a frame whose last user has finished (its idle event has fired) can be filled at once; one that is still
        // being read would hold an upload back until the GPU gets there - take such a frame only when the pool has
        // grown to its bound (the oldest: the likeliest to be free soonest)
        int same = 0, pick = -1, oldest = -1;
        for (size_t i = 0; i < ctx->frame_pool.size(); i++)
        {
            hbhip_frame *fr = ctx->frame_pool[i];
            if (fr->width != width || fr->height != height || fr->depth != depth || fr->lcw != lcw || fr->lch != lch) continue;
            same++;
            if (oldest < 0) oldest = (int)i;
            if (hbhip_pic_idle_done(&fr->pic)) { pick = (int)i; break; }
        }
        (void)hipGetLastError();                                  // hipErrorNotReady of the queries
        if (pick < 0 && same >= 48) pick = oldest;
        if (pick >= 0)
        {
            hbhip_frame *fr = ctx->frame_pool[pick];
            ctx->frame_pool.erase(ctx->frame_pool.begin() + pick);
            fr->refs = 1;
            fr->ready.reset();
            fr->last_user = nullptr;
            fr->complete = false;
            fr->pic.frame = fr;
            fr->pic.refs = 0; fr->pic.flags = 0; fr->pic.combed = 0; fr->pic.aux = 0; fr->pic.tag = 0;
            // taken although its last reader may still be running (the pool is at its bound): whoever fills it on this
            // context's stream does so behind that reader, which may sit on another context's stream (hbhip_frame_use_on)
            if (fr->pic.idle && fr->pic.idle->stream != ctx->stream) (void)hbhip_pic_wait_idle(ctx->stream, &fr->pic);
            *out = fr;
            return HBHIP_OK;
        }
    }
    lk.unlock();
    hbhip_frame *fr = new (std::nothrow) hbhip_frame();
    if (!fr) return HBHIP_ERR_NOMEM;
    fr->ctx = ctx; fr->width = width; fr->height = height; fr->depth = depth; fr->lcw = lcw; fr->lch = lch;
    PicGeometry g;
    g.set(width, height, depth, lcw, lch);
    // laid out exactly like hb_frame_buffer_init lays a host frame out (fifo.c:820-881: stride = the row rounded up to
    // 64 bytes, the planes one behind the other), so that a frame moves between the two in ONE 1-D copy
    size_t off[3], total = 0;
    for (int c = 0; c < 3; c++)
    {
        fr->pic.width[c] = g.pw[c];
        fr->pic.height[c] = g.ph[c];
        fr->pic.pitch[c] = hbhip_align_up(g.pw[c] * g.bps, 64);
        off[c] = total;
        total += (size_t)fr->pic.pitch[c] * g.ph[c];
    }
    total = (total + 255) & ~(size_t)255;
    fr->pic.bps = g.bps;
    fr->pic.bytes = total;
    if (hipMalloc((void **)&fr->pic.base, total) != hipSuccess)
    {
        (void)hipGetLastError();
        delete fr;
        return HBHIP_ERR_NOMEM;
    }
    (void)hipMemsetAsync(fr->pic.base, 0, total, ctx->stream);
    hbhip_pic_mark_idle(ctx, &fr->pic);    // an upload (on the copy stream) must not overtake the clearing
    for (int c = 0; c < 3; c++) fr->pic.plane[c] = fr->pic.base + off[c];
    fr->pic.frame = fr;
    *out = fr;
    return HBHIP_OK;
}

void hbhip_frame_retain(hbhip_frame *fr)
{
    if (!fr) return;
    std::lock_guard<std::mutex> lk(fr->ctx->frame_lock);
    fr->refs++;
}

void hbhip_frame_release(hbhip_frame *fr)
{
    if (!fr) return;
    std::lock_guard<std::mutex> lk(fr->ctx->frame_lock);
    if (--fr->refs > 0) return;
    (void)hipSetDevice(fr->ctx->device);
    // its users are all queued by now: on the context's stream, or on the stream of the context that read it last
    hbhip_pic_mark_idle(fr->last_user ? fr->last_user : fr->ctx, &fr->pic);
    fr->ctx->frame_pool.push_back(fr);        // reuse is ordered by that event (uploads) or by the stream itself
}

// A filter on another context of the same GPU is about to queue work that reads the frame (a job whose filters run on
// more than one HIP stream: libhb/hbhip_registry.c, hbhip_host_ctx_for_role).  Orders `ctx`'s stream behind the frame's
// producer - its ready mark, nothing else of the owner's stream: the point of a second stream is not to wait for what
// the first has queued since - and notes `ctx` as the stream the frame goes idle behind.  A frame that a second foreign
// context (or its owner again) reads after the first is ordered behind that first reader's stream as it stands, so
// that one idle mark still covers every reader.  No-op on the owner's context while nobody else has read the frame.
int hbhip_frame_use_on(hbhip_frame *fr, hbhip_ctx *ctx)
{
    if (!fr || !ctx || ctx->device != fr->ctx->device) return HBHIP_ERR_ARG;
    std::lock_guard<std::mutex> lk(fr->ctx->frame_lock);
    (void)hipSetDevice(ctx->device);
    bool waited_ready = false;
    if (fr->ready && fr->ready->stream != ctx->stream)
    {
        // the producer sits on another stream: the owner's (a foreign reader), or the upload stream (every reader, the
        // owner's context included: hbhip_frame_upload_async)
        fr->ready->record_now();
        const hipError_t e = hipStreamWaitEvent(ctx->stream, fr->ready->ev, 0);
        if (e != hipSuccess) return ctx->fail(e, "use_on: order behind the frame's producer");
        waited_ready = true;
    }
    hbhip_ctx *prev = fr->last_user ? fr->last_user : fr->ctx;
    if (prev == ctx) return HBHIP_OK;
    auto behind = [&](hbhip_ctx *of) -> int {          // ctx->stream behind everything queued on of->stream so far
        hipEvent_t ev = of->sync_ev_get();
        if (!ev) return ctx->fail(hipErrorOutOfMemory, "hipEventCreate(use_on)");
        hipError_t e = hipEventRecord(ev, of->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ev, 0);
        of->sync_ev_put(ev);                              // (the wait holds the event's state as recorded here)
        return e == hipSuccess ? HBHIP_OK : ctx->fail(e, "use_on: order behind the frame's earlier users");
    };
    int rc = HBHIP_OK;
    if (fr->last_user == nullptr || fr->last_user == fr->ctx)
    {
        // (the owner as the last user: it has written the frame again - the compositor - and marked it ready behind that)
        if (!waited_ready && !fr->ready && !fr->complete) rc = behind(fr->ctx);     // no mark to go by: the owner's stream as it stands
    }
    else
        rc = behind(prev);
    if (rc == HBHIP_OK) fr->last_user = ctx;
    return rc;
}

// The writability test of a device picture (fifo.c:624-639 asks av_buffer_is_writable the same thing): a holder that
// sees 1 is the only one, and nobody else can take a reference except through it.
int hbhip_frame_refs(hbhip_frame *fr)
{
    if (!fr) return 0;
    std::lock_guard<std::mutex> lk(fr->ctx->frame_lock);
    return fr->refs;
}

int hbhip_frame_describe(hbhip_frame *fr, hbhip_dev_frame *out, int *width, int *height)
{
    if (!fr || !out) return HBHIP_ERR_ARG;
    for (int c = 0; c < 3; c++)
    {
        out->plane[c] = fr->pic.plane[c];
        out->stride[c] = fr->pic.pitch[c];
    }
    if (width) *width = fr->width;
    if (height) *height = fr->height;
    return HBHIP_OK;
}

int hbhip_frame_copy(hbhip_frame *dst, hbhip_frame *src)
{
    if (!dst || !src || dst->ctx != src->ctx) return HBHIP_ERR_ARG;
    if (dst->width != src->width || dst->height != src->height || dst->depth != src->depth ||
        dst->lcw != src->lcw || dst->lch != src->lch)
        return HBHIP_ERR_ARG;
    (void)hipSetDevice(dst->ctx->device);
    if (src->ready && src->ready->stream != dst->ctx->stream)              // (a frame whose upload is still in flight)
    {
        src->ready->record_now();
        const hipError_t e = hipStreamWaitEvent(dst->ctx->stream, src->ready->ev, 0);
        if (e != hipSuccess) return dst->ctx->fail(e, "frame_copy: order behind the source's producer");
    }
    hbhip_dev_frame d;
    for (int c = 0; c < 3; c++) { d.plane[c] = src->pic.plane[c]; d.stride[c] = src->pic.pitch[c]; }
    return hbhip_copy_d2d_in(dst->ctx, &dst->pic, &d);
}

int hbhip_frame_upload(hbhip_frame *fr, const hbhip_host_frame *src)
{
    if (!fr || !src) return HBHIP_ERR_ARG;
    (void)hipSetDevice(fr->ctx->device);
    const int rc = hbhip_copy_h2d(fr->ctx, &fr->pic, src);      // returns when the copy has finished
    if (rc == HBHIP_OK && !fr->ready) fr->complete = true;
    return rc;
}

// The pipelined H2D (the upload adapter keeps a few in flight, as the download adapter does with its copies): the copy is
// queued on the upload stream and the call returns; `src` must stay valid until hbhip_ctx_upload_done(token) says so.  The
// frame's ready mark is the copy's event, so whoever reads the frame - on any context: hbhip_frame_use_on,
// hbhip_frame_copy, a download - waits for the copy and for nothing else.
int hbhip_frame_upload_async(hbhip_frame *fr, const hbhip_host_frame *src, void **token)
{
    if (!fr || !src || !token) return HBHIP_ERR_ARG;
    *token = nullptr;
    hbhip_ctx *ctx = fr->ctx;
    (void)hipSetDevice(ctx->device);
    std::shared_ptr<IdleMark> m = std::make_shared<IdleMark>();
    if (hipEventCreateWithFlags(&m->ev, hipEventDisableTiming) != hipSuccess)
    {
        (void)hipGetLastError();
        return ctx->fail(hipErrorOutOfMemory, "hipEventCreate(upload ready)");
    }
    hipEvent_t done = nullptr;
    const int rc = hbhip_copy_h2d_queue(ctx, &fr->pic, src, &done);
    if (rc != HBHIP_OK) return rc;
    m->stream = ctx->up();
    const hipError_t e = hipEventRecord(m->ev, ctx->up());
    if (e != hipSuccess)
    {
        (void)hipEventSynchronize(done);
        ctx->sync_ev_put(done);
        return ctx->fail(e, "hipEventRecord(upload ready)");
    }
    m->recorded = true;
    m->closed.store(true, std::memory_order_release);
    fr->ready = m;
    fr->complete = false;
    *token = done;
    return HBHIP_OK;
}

// has the copy behind `token` finished (block != 0: wait for it)?  HBHIP_OK: yes, the source planes are free again and the
// token is spent; HBHIP_AGAIN: not yet (block == 0 only)
int hbhip_ctx_upload_done(hbhip_ctx *ctx, void *token, int block)
{
    if (!ctx || !token) return HBHIP_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    hipEvent_t ev = (hipEvent_t)token;
    hipError_t e = block ? hipEventSynchronize(ev) : hipEventQuery(ev);
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return HBHIP_AGAIN; }
    ctx->sync_ev_put(ev);
    return e == hipSuccess ? HBHIP_OK : ctx->fail(e, "upload: wait for the copy");
}

int hbhip_frame_download(hbhip_frame *fr, const hbhip_host_frame *dst)
{
    if (!fr || !dst) return HBHIP_ERR_ARG;
    void *token = nullptr;
    const int rc = hbhip_frame_download_async(fr, dst, &token);
    return rc != HBHIP_OK ? rc : hbhip_frame_download_wait(fr, token);
}

int hbhip_frame_mark_ready(hbhip_frame *fr)
{
    if (!fr) return HBHIP_ERR_ARG;
    hbhip_ctx *ctx = fr->ctx;
    (void)hipSetDevice(ctx->device);
    fr->ready = ctx->mark();
    if (!fr->ready) return ctx->fail(hipErrorOutOfMemory, "hipEventCreate(ready)");
    return HBHIP_OK;
}

// D2H on the download stream behind the frame's ready point (or, without one, behind everything queued on the
// context's stream so far).  Several in flight keep the bus busy while the kernels of later frames run.
int hbhip_frame_download_async(hbhip_frame *fr, const hbhip_host_frame *dst, void **token)
{
    if (!fr || !dst || !token) return HBHIP_ERR_ARG;
    *token = nullptr;
    hbhip_ctx *ctx = fr->ctx;
    (void)hipSetDevice(ctx->device);
    const DevPicture *src = &fr->pic;
    for (int c = 0; c < 3; c++)
        if (dst->plane[c] == nullptr || dst->stride[c] < src->width[c] * src->bps) return HBHIP_ERR_ARG;
    hipEvent_t ev = ctx->sync_ev_get();
    if (!ev) return ctx->fail(hipErrorOutOfMemory, "hipEventCreate(download)");
    auto fail = [&](hipError_t e, const char *what) { ctx->sync_ev_put(ev); return ctx->fail(e, what); };
    hipError_t e;
    if (fr->ready)
    {
        fr->ready->record_now();
        e = hipStreamWaitEvent(ctx->down(), fr->ready->ev, 0);
    }
    else
    {
        e = hipEventRecord(ev, ctx->stream);           // the picture's producers are on ctx->stream, all queued by now
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->down(), ev, 0);
    }
    if (e != hipSuccess) return fail(e, "download: order behind the frame's producers");
    if (same_layout(src, dst->plane, dst->stride))
    {
        e = hipMemcpyAsync(dst->plane[0], src->plane[0], layout_bytes(src), hipMemcpyDeviceToHost, ctx->down());
        if (e != hipSuccess) return fail(e, "hipMemcpyAsync(download)");
    }
    else
        for (int c = 0; c < 3; c++)
        {
            const size_t row = (size_t)src->width[c] * src->bps;
            e = hipMemcpy2DAsync(dst->plane[c], dst->stride[c], src->plane[c], src->pitch[c], row, src->height[c],
                                 hipMemcpyDeviceToHost, ctx->down());
            if (e != hipSuccess)
            {
                (void)hipStreamSynchronize(ctx->down());      // planes already queued must not outlive the call
                return fail(e, "hipMemcpy2DAsync(download)");
            }
        }
    e = hipEventRecord(ev, ctx->down());
    if (e != hipSuccess) { (void)hipStreamSynchronize(ctx->down()); return fail(e, "hipEventRecord(download)"); }
    *token = ev;
    return HBHIP_OK;
}

int hbhip_frame_download_wait(hbhip_frame *fr, void *token)
{
    if (!fr || !token) return HBHIP_ERR_ARG;
    hbhip_ctx *ctx = fr->ctx;
    (void)hipSetDevice(ctx->device);
    hipEvent_t ev = (hipEvent_t)token;
    const hipError_t e = hipEventSynchronize(ev);
    ctx->sync_ev_put(ev);
    return e == hipSuccess ? HBHIP_OK : ctx->fail(e, "hipEventSynchronize(download)");
}

// ---- generic filter surface -------------------------------------------------
int hbhip_filter_push(hbhip_filter *f, const hbhip_host_frame *in, int64_t tag)
{
    if (!f || !in) return HBHIP_ERR_ARG;
    (void)hipSetDevice(f->ctx->device);
    DevPicture *pic = f->acquire_input();
    if (!pic) return HBHIP_ERR_NOMEM;
    pic->tag = tag;
    for (int c = 0; c < 3; c++) f->in_stride[c] = in->stride[c];
    f->in_is_dev = false;
    int rc = hbhip_copy_h2d(f->ctx, pic, in);          // returns when `in` has been consumed
    if (rc != HBHIP_OK)
    {
        f->abandon_input(pic);             // not submitted: back to its pool
        return rc;
    }
    return f->submit(pic);
}

int hbhip_filter_push_dev(hbhip_filter *f, const hbhip_dev_frame *in, int64_t tag)
{
    if (!f || !in) return HBHIP_ERR_ARG;
    (void)hipSetDevice(f->ctx->device);
    DevPicture *pic = f->acquire_input();
    if (!pic) return HBHIP_ERR_NOMEM;
    pic->tag = tag;
    for (int c = 0; c < 3; c++) f->in_stride[c] = in->stride[c];
    f->in_is_dev = true;
    int rc = hbhip_copy_d2d_in(f->ctx, pic, in);
    if (rc != HBHIP_OK)
    {
        f->abandon_input(pic);
        return rc;
    }
    return f->submit(pic);
}

// ---- frames in, frames out: the drop-ins of a device-resident run ---------------------------------------------
int hbhip_filter_use_frames(hbhip_filter *f)
{
    if (!f) return HBHIP_ERR_ARG;
    return f->use_frames();
}

// The frame becomes the filter's input picture - no copy - when the filter works on frames, the caller is the frame's
// only holder (a frame vfr has duplicated stays shared: it is copied, as before) and the geometry is the filter's; the
// filter keeps a reference of its own until it is done with the picture.
int hbhip_filter_push_frame(hbhip_filter *f, hbhip_frame *fr, int64_t tag)
{
    if (!f || !fr) return HBHIP_ERR_ARG;
    (void)hipSetDevice(f->ctx->device);
    int rc = hbhip_frame_use_on(fr, f->ctx);
    if (rc != HBHIP_OK) return rc;
    const PicGeometry &g = f->in_geo;
    const bool fits = fr->width == g.width && fr->height == g.height && fr->depth == g.depth && fr->lcw == g.log2_cw && fr->lch == g.log2_ch;
    if (!f->frames_mode || !fits || hbhip_frame_refs(fr) != 1)
    {
        hbhip_dev_frame d;
        hbhip_frame_describe(fr, &d, nullptr, nullptr);
        return hbhip_filter_push_dev(f, &d, tag);
    }
    hbhip_frame_retain(fr);
    DevPicture *pic = &fr->pic;
    pic->tag = tag;
    f->adopt_input(pic);
    for (int c = 0; c < 3; c++) f->in_stride[c] = pic->pitch[c];
    f->in_is_dev = true;
    return f->submit(pic);
}

// The next output as a frame: the picture itself when the filter works on frames (its reference passes to the caller), else
// a fresh frame the picture is copied into.  HBHIP_AGAIN: none pending.
int hbhip_filter_pull_frame(hbhip_filter *f, hbhip_frame **out, int64_t *tag)
{
    if (!f || !out) return HBHIP_ERR_ARG;
    *out = nullptr;
    (void)hipSetDevice(f->ctx->device);
    DevPicture *pic = f->pop_output();
    if (!pic) return HBHIP_AGAIN;
    if (tag) *tag = pic->tag;
    if (pic->frame)
    {
        *out = pic->frame;
        return HBHIP_OK;
    }
    const PicGeometry &g = f->out_geo;
    hbhip_frame *fr = nullptr;
    int rc = hbhip_frame_alloc(f->ctx, g.width, g.height, g.depth, g.log2_cw, g.log2_ch, &fr);
    if (rc == HBHIP_OK)
    {
        hbhip_dev_frame d;
        hbhip_frame_describe(fr, &d, nullptr, nullptr);
        rc = hbhip_copy_d2d_out(f->ctx, &d, pic);
        if (rc != HBHIP_OK) { hbhip_frame_release(fr); fr = nullptr; }
    }
    f->recycle_output(pic);
    *out = fr;
    return rc;
}

int hbhip_filter_pull(hbhip_filter *f, const hbhip_host_frame *out, int64_t *tag)
{
    if (!f || !out) return HBHIP_ERR_ARG;
    DevPicture *pic = f->pop_output();
    if (!pic) return HBHIP_AGAIN;
    if (tag) *tag = pic->tag;
    int rc = hbhip_copy_d2h(f->ctx, out, pic);         // returns when `out` is filled
    f->recycle_output(pic);
    return rc;
}

int hbhip_filter_pull_dev(hbhip_filter *f, const hbhip_dev_frame *out, int64_t *tag)
{
    if (!f || !out) return HBHIP_ERR_ARG;
    DevPicture *pic = f->pop_output();
    if (!pic) return HBHIP_AGAIN;
    if (tag) *tag = pic->tag;
    int rc = hbhip_copy_d2d_out(f->ctx, out, pic);
    f->recycle_output(pic);
    return rc;
}

int hbhip_filter_process_dev(hbhip_filter *f, const hbhip_dev_frame *in, int n_in, int64_t tag0,
                             const hbhip_dev_frame *out, int out_cap, int *n_out)
{
    if (!f || (n_in > 0 && !in) || !n_out) return HBHIP_ERR_ARG;
    (void)hipSetDevice(f->ctx->device);
    return f->process_dev_batch(in, n_in, tag0, out, out_cap, n_out);
}

// ---- pipelined host path ---------------------------------------------------------------
// push + pull of a one-in / one-out filter as ONE asynchronous submission: upload on the context's upload stream,
// the filter's kernels on the compute stream behind an event, the download on the download stream behind another.
// Nothing waits on the host; hbhip_filter_wait() later blocks on the oldest submission's last event only.  With two
// submissions in flight the H2D of frame n, the kernels of frame n-1 and the D2H of frame n-2 overlap - what the
// reference gets from keeping `threads` frames in flight (nlmeans.c:464-597, mt_frame_filter.c:45-237).
int hbhip_filter_submit_async(hbhip_filter *f, const hbhip_host_frame *in, const hbhip_host_frame *out, int64_t tag)
{
    if (!f || !in || !out) return HBHIP_ERR_ARG;
    hbhip_ctx *ctx = f->ctx;
    (void)hipSetDevice(ctx->device);
    DevPicture *pic = f->acquire_input();
    if (!pic) return HBHIP_ERR_NOMEM;
    DevPicture *o = f->acquire_output();
    if (!o)
    {
        f->abandon_input(pic);
        return HBHIP_ERR_UNSUPPORTED;              // not a one-in / one-out filter: use push / pull
    }
    hipEvent_t ev = ctx->sync_ev_get(), done = ctx->sync_ev_get();
    int rc = (ev && done) ? HBHIP_OK : HBHIP_ERR_NOMEM;
    for (int c = 0; rc == HBHIP_OK && c < 3; c++)
        if (!in->plane[c] || in->stride[c] < pic->width[c] * pic->bps || !out->plane[c] ||
            out->stride[c] < o->width[c] * o->bps)
            rc = HBHIP_ERR_ARG;
    auto fail = [&](int code) {
        // copies of the caller's `in` / `out` may be in flight: nothing of them may outlive this call
        if (ctx->up_stream) (void)hipStreamSynchronize(ctx->up_stream);
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->down_stream) (void)hipStreamSynchronize(ctx->down_stream);
        if (pic) f->abandon_input(pic);            // (already handed back once the kernels were queued)
        f->recycle_output(o);
        ctx->sync_ev_put(ev);
        ctx->sync_ev_put(done);
        return code;
    };
    if (rc != HBHIP_OK) return fail(rc);
#define ASYNC_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(ctx->fail(e_, #expr)); } while (0)
    pic->tag = tag;
    for (int c = 0; c < 3; c++) f->in_stride[c] = in->stride[c];
    f->in_is_dev = false;
    ASYNC_CHECK(hbhip_pic_wait_idle(ctx->up(), pic));
    for (int c = 0; c < 3; c++)
        ASYNC_CHECK(hipMemcpy2DAsync(pic->plane[c], pic->pitch[c], in->plane[c], in->stride[c],
                                     (size_t)std::min(in->stride[c], pic->pitch[c]), pic->height[c],
                                     hipMemcpyHostToDevice, ctx->up()));
    ASYNC_CHECK(hipEventRecord(ev, ctx->up()));
    ASYNC_CHECK(hipStreamWaitEvent(ctx->stream, ev, 0));
    rc = f->process_pair(pic, o);
    if (rc != HBHIP_OK) return fail(rc);
    hbhip_pic_release(pic, f->ctx);                // idle event behind the kernels that read it
    pic = nullptr;
    ASYNC_CHECK(hipEventRecord(ev, ctx->stream));
    ASYNC_CHECK(hipStreamWaitEvent(ctx->down(), ev, 0));
    for (int c = 0; c < 3; c++)
        ASYNC_CHECK(hipMemcpy2DAsync(out->plane[c], out->stride[c], o->plane[c], o->pitch[c], (size_t)o->width[c] * o->bps,
                                     o->height[c], hipMemcpyDeviceToHost, ctx->down()));
    ASYNC_CHECK(hipEventRecord(done, ctx->down()));
#undef ASYNC_CHECK
    ctx->sync_ev_put(ev);
    f->async_q.push_back({o, done, tag});
    return HBHIP_OK;
}

int hbhip_filter_wait(hbhip_filter *f, int64_t *tag)
{
    if (!f) return HBHIP_ERR_ARG;
    if (f->async_q.empty()) return HBHIP_AGAIN;
    (void)hipSetDevice(f->ctx->device);
    hbhip_filter::AsyncSlot s = f->async_q.front();
    f->async_q.pop_front();
    const hipError_t e = hipEventSynchronize(s.done);
    f->ctx->sync_ev_put(s.done);
    f->recycle_output(s.out);
    if (tag) *tag = s.tag;
    return e == hipSuccess ? HBHIP_OK : f->ctx->fail(e, "hipEventSynchronize(wait)");
}

int hbhip_filter_inflight(hbhip_filter *f)
{
    return f ? (int)f->async_q.size() : 0;
}

int hbhip_filter_flush(hbhip_filter *f)
{
    if (!f) return HBHIP_ERR_ARG;
    (void)hipSetDevice(f->ctx->device);
    return f->flush();
}

int hbhip_filter_pending(hbhip_filter *f)
{
    return f ? f->pending() : 0;
}

int hbhip_filter_defer(hbhip_filter *f, int on)
{
    if (!f) return HBHIP_ERR_ARG;
    (void)hipSetDevice(f->ctx->device);
    f->defer_launches(on != 0);
    return HBHIP_OK;
}

int hbhip_filter_kick(hbhip_filter *f)
{
    if (!f) return HBHIP_ERR_ARG;
    (void)hipSetDevice(f->ctx->device);
    return f->kick();
}

void hbhip_filter_destroy(hbhip_filter *f)
{
    if (!f) return;
    (void)hipSetDevice(f->ctx->device);
    while (!f->async_q.empty()) (void)hbhip_filter_wait(f, nullptr);
    (void)hipStreamSynchronize(f->ctx->stream);
    delete f;
}

hbhip_ctx *hbhip_filter_context(hbhip_filter *f)
{
    return f ? f->ctx : nullptr;
}

int hbhip_filter_out_geometry(hbhip_filter *f, int *width, int *height)
{
    if (!f) return HBHIP_ERR_ARG;
    if (width) *width = f->out_geo.width;
    if (height) *height = f->out_geo.height;
    return HBHIP_OK;
}

} // extern "C"
