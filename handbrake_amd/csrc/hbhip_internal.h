// hbhip_internal.h — shared plumbing of libhbhip.so (context, stream, launch
// bookkeeping, device frame pool, filter base class).  Not part of the ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "hbhip.h"

#define HBHIP_MAX_MARKS 8

struct hbhip_prof_pending
{
    int        name_idx;
    hipEvent_t ev0, ev1;
};

struct hbhip_prof_stat
{
    std::string name;
    int64_t     launches = 0;
    double      total_ms = 0.0;
};

// A point on a stream behind which some pictures' last users have all run: the pictures released between two launches
// on a stream share ONE event, recorded right before the stream's next launch (or when somebody needs it first).  An
// event per released picture is what this replaces: every record is a barrier packet the command processor works off
// in about 4.5 us, and a chain step released 80 pictures (32 + 32 + 16 at its stage boundaries: three 143 us bubbles
// in the kernel trace, 6 % of a step).
struct IdleMark
{
    hipEvent_t  ev = nullptr;
    hipStream_t stream = nullptr;
    std::atomic<bool> closed{false};                 // set ahead of the record: nothing attaches to the mark from then on
    bool        recorded = false;                    // (lock)
    std::mutex  lock;
    ~IdleMark() { if (ev) (void)hipEventDestroy(ev); }
    void record_now();                               // idempotent; any thread
};

struct hbhip_ctx
{
    int         device = 0;
    hipStream_t stream = nullptr;
    bool        own_stream = true;
    // Transfers of host frames run on two streams of their own, so that a filter thread moving a frame waits for
    // ITS copy and not for every kernel the job's other filters have queued on `stream` (libhb runs each filter on
    // its own thread, work.c:2527-2600), and so that copies overlap the kernels.  Ordering against `stream` is by
    // events: a picture's `idle` event (recorded when it goes back to its pool) gates the next upload into it, an
    // event recorded on `stream` at download time gates the copy out.
    // Reached through up() / down().  HIP spreads its streams over a handful of hardware queues in the order they are made,
    // so when these two are made decides which streams share a queue: with the context (the product: ctx_create_common
    // has the measurement) or at their first use (-DHBHIP_LAZY_COPY_STREAMS).
    hipStream_t up_stream = nullptr, down_stream = nullptr;
    std::once_flag up_once, down_once;
    hipStream_t up();                               // the upload stream; `stream` itself should it not be had
    hipStream_t down();
    std::shared_ptr<IdleMark> open_mark;            // the mark pictures released right now attach to (state_lock)
    std::atomic<bool> has_open_mark{false};
    std::shared_ptr<IdleMark> mark();               // the open mark (made if there is none); null: no event to be had
    void close_mark();                              // record it: called ahead of every launch on `stream`
    std::vector<hipEvent_t> sync_ev_pool;           // hipEventDisableTiming events (state_lock)
    hipEvent_t sync_ev_get();
    void       sync_ev_put(hipEvent_t e);
    std::string last_error;
    char        dev_name[256] = {0};

    bool                            profile = false;
    std::vector<hbhip_prof_stat>    prof_stats;
    std::vector<hbhip_prof_pending> prof_pending;
    std::vector<hipEvent_t>         ev_pool;
    hipEvent_t                      marks[HBHIP_MAX_MARKS] = {};

    // pool of device-resident frames handed between filters (hbhip_frame_*)
    std::vector<struct hbhip_frame *> frame_pool;
    std::mutex frame_lock;          // filters of one job run on different host threads
    // last_error and the profiler's bookkeeping are written from whichever filter thread fails / launches:
    // libhb runs every filter of a job on its own thread (work.c:2527-2600) and they share this context
    std::recursive_mutex state_lock;

    int  fail(hipError_t e, const char *what);
    int  prof_name(const char *name);
    void prof_begin(const char *name);
    void prof_end();
    void prof_resolve();
    hipEvent_t ev_get();
};

#define HBHIP_CHECK(ctx, expr)                                                  \
    do {                                                                        \
        hipError_t _e = (expr);                                                 \
        if (_e != hipSuccess) return (ctx)->fail(_e, #expr);                    \
    } while (0)

// Development builds only (make dev: -DHBHIP_DEV; `make product` never defines it):
//  * hbhip_dev_int("NAME", default): a tuning knob read from the environment - in a product build the default, a constant;
//  * what-if profiling: HBHIP_SKIP_KERNELS=<substring>[,<substring>...] drops the launches whose name matches (results
//    are wrong then, by design): how much of a workload's wall time hangs on a kernel is then a measurement.
#ifdef HBHIP_DEV
static inline int hbhip_dev_int(const char *name, int def)
{
    const char *e = getenv(name);
    return e && *e ? atoi(e) : def;
}
static inline bool hbhip_skip_launch(const char *name)
{
    static const char *list = getenv("HBHIP_SKIP_KERNELS");
    if (!list || !*list) return false;
    for (const char *p = list; *p; )
    {
        const char *q = strchr(p, ',');
        const size_t n = q ? (size_t)(q - p) : strlen(p);
        if (n && n < 96) { char sub[96]; memcpy(sub, p, n); sub[n] = 0; if (strstr(name, sub)) return true; }
        p = q ? q + 1 : p + n;
    }
    return false;
}
#else
#define hbhip_dev_int(name, def) (def)
#define hbhip_skip_launch(name) false
#endif

// Workgroups are handed to the eight XCDs round robin in dispatch order (x fastest).  A grid with a multiple of 8
// workgroups per row pins every column of tiles to one XCD for the whole launch - which is what a 1920-sample row cut into
// 256-sample tiles gives - and that measured 20 - 25 % slower than letting the columns rotate over the XCDs (lapsharp 1080p:
// 50.9 us per 16 frames with 8 workgroups per row, 39.9 us with 9; the other way round, 2160p padded from 15 to 16: 121 ->
// 165 us; profiles/r5x_*, r5y_*).  The counters say it is balance, not caching (profiles/r5N_*): the grids are sized for the
// widest plane and the chroma planes' right-hand tiles return at once, so with column k on XCD k half of the XCDs get no
// chroma work.  With an ODD count a column visits all eight XCDs (gcd(count, 8) = 1), with an even one
// 8 / gcd of them: hbhip_grid_x() makes the count odd by adding a workgroup per row where it is even (2 -> 3 for the
// 1024-sample tiles of fill_gaps / the lattice candidates: 18 - 22 % faster on dense masks, profiles/r5E_*).  Every kernel
// launched through it returns at once for a tile that starts beyond its plane (their grids are sized for the widest plane
// of the launch anyway).  HBHIP_GRID_ROTATE: 0 = grids as computed, 1 = only multiples of 8 bumped, 2 = always odd.
#ifndef HBHIP_GRID_ROTATE
#define HBHIP_GRID_ROTATE 2
#endif
static inline unsigned hbhip_grid_x(unsigned gx) { return HBHIP_GRID_ROTATE == 2 ? (gx | 1u) : (HBHIP_GRID_ROTATE && gx && (gx & 7u) == 0u) ? gx + 1u : gx; }

// Launch a kernel on the context's stream; when profiling is on the launch is
// bracketed by two events whose delta is accumulated under `name`.
#define HBHIP_LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                \
    do {                                                                        \
        if (hbhip_skip_launch(name)) break;                                     \
        if ((ctx)->has_open_mark.load(std::memory_order_acquire)) (ctx)->close_mark(); \
        if ((ctx)->profile)                                                     \
        {                                                                       \
            /* launch and its two events as one unit: other filter threads share the stream */ \
            std::lock_guard<std::recursive_mutex> _plk((ctx)->state_lock);      \
            (ctx)->prof_begin(name);                                            \
            hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__); \
            (ctx)->prof_end();                                                  \
        }                                                                       \
        else                                                                    \
            hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__); \
    } while (0)

// The same on a stream of the caller's choosing: the context's own (then it IS HBHIP_LAUNCH) or a side stream that the
// caller has ordered against it with events (no profiling brackets, no idle marks: neither lives on a side stream).
#define HBHIP_LAUNCH_ON(ctx, strm, name, kernel, grid, block, shmem, ...)       \
    do {                                                                        \
        if ((strm) == (ctx)->stream) HBHIP_LAUNCH(ctx, name, kernel, grid, block, shmem, __VA_ARGS__); \
        else if (!hbhip_skip_launch(name)) hipLaunchKernelGGL(kernel, grid, block, shmem, (strm), __VA_ARGS__); \
    } while (0)

static inline int hbhip_align_up(int v, int a) { return (v + a - 1) / a * a; }

// One planar picture in HBM (a single allocation, planes 256-byte aligned).
struct DevPicture
{
    uint8_t *base = nullptr;
    uint8_t *plane[3] = {nullptr, nullptr, nullptr};
    int      pitch[3] = {0, 0, 0};
    int      width[3] = {0, 0, 0};   // in pixels
    int      height[3] = {0, 0, 0};
    int      bps = 1;
    int64_t  tag = 0;
    size_t   bytes = 0;
    int      refs = 0;       // used by filters that keep a picture in several ring slots
    int      flags = 0;      // PIC_FLAG_* of the source buffer (decomb / comb detect)
    int      combed = 0;     // HB_COMB_* of the source buffer
    int      aux = 0;        // filter specific (decomb: which field of a bob pair)
    std::shared_ptr<IdleMark> idle;  // set when the picture goes back to its pool: everything that used it has been queued
                                     // on idle->stream ahead of that mark (null: never used)
    class PicturePool *owner = nullptr;   // the pool the picture goes back to (hbhip_pic_release)
    struct hbhip_frame *frame = nullptr;  // the picture IS a frame's (hbhip_frame::pic): it goes back through hbhip_frame_release
};

// A reference-counted device picture that travels between filters inside an hb_buffer_t.
struct hbhip_frame
{
    hbhip_ctx  *ctx = nullptr;
    DevPicture  pic;
    int         width = 0, height = 0, depth = 8, lcw = 1, lch = 1;
    int         refs = 1;
    // hbhip_frame_mark_ready: the point of the context's stream behind which the frame's contents are complete -
    // a download waits for this point, not for whatever other filter threads have queued since
    std::shared_ptr<IdleMark> ready;      // (shared by the frames marked between two launches, like `idle`)
    // A job may run its filters on more than one context of a GPU (hbhip_frame_use_on): the context whose stream the
    // frame's newest reader sits on, if that is not the owner's - the frame goes idle behind THAT stream's work
    // (frame_lock of the owner)
    hbhip_ctx  *last_user = nullptr;
    bool        complete = false;         // filled by a copy that has finished (hbhip_frame_upload): no producer to wait for
};

// Geometry of a planar YUV picture.
struct PicGeometry
{
    int width = 0, height = 0, depth = 8, bps = 1;
    int log2_cw = 1, log2_ch = 1;
    int pw[3], ph[3];
    void set(int w, int h, int d, int lcw, int lch)
    {
        width = w; height = h; depth = d; bps = d > 8 ? 2 : 1;
        log2_cw = lcw; log2_ch = lch;
        pw[0] = w; ph[0] = h;
        pw[1] = pw[2] = -((-w) >> lcw);
        ph[1] = ph[2] = -((-h) >> lch);
    }
};

// Pool of equally-shaped device pictures.  Reuse is stream-ordered: a picture handed back by a user on the pool's
// own stream needs nothing; one handed back from another context's stream (a fused chain whose stages run on
// streams of their own) makes the pool's stream wait for that user's `idle` event when it is taken out again.
class PicturePool
{
public:
    PicturePool() = default;
    ~PicturePool();
    void configure(hbhip_ctx *ctx, const PicGeometry &g, int pitch_align = 256, int pad_rows = 0);
    // Frame-backed: acquire() hands out the picture of a fresh hbhip_frame of the context's frame pool (laid out like
    // hb_frame_buffer_init lays a host frame out) and release() gives the frame back.  What a drop-in inside a
    // device-resident run wants: the pictures a filter produces leave it AS frames (hbhip_filter_pull_frame) and the frames
    // it is pushed become its input pictures (hbhip_filter_push_frame) - nothing is copied at either end.
    void use_frames(bool on) { frames_ = on; }
    bool frames() const { return frames_; }
    DevPicture *acquire();            // nullptr on allocation failure
    void        release(DevPicture *p, hbhip_ctx *last_user = nullptr);   // last_user: the context that read it (default: the pool's)
    const PicGeometry &geometry() const { return geo_; }
private:
    hbhip_ctx *ctx_ = nullptr;
    PicGeometry geo_;
    int pitch_align_ = 256;
    int pad_rows_ = 0;
    bool frames_ = false;
    std::vector<DevPicture *> all_;
    std::vector<DevPicture *> free_;
    size_t max_pictures_ = 96;            // growth bound of the cross-stream case (acquire): three batches of 32 - a chain split over
                                          // two streams is there after its warm-up; every new picture is a hipMalloc and a memset
};

// Give a picture back to the pool it came from.  Filters release their INPUT pictures through this,
// so that a picture one filter produced can be handed to the next filter of a fused chain
// (hbhip_chain, chain.hip) without a copy: whoever finishes with it returns it to its producer's pool.
void hbhip_frame_release(struct hbhip_frame *fr);
inline void hbhip_pic_release(DevPicture *p, hbhip_ctx *last_user = nullptr)
{
    if (p && p->frame) hbhip_frame_release(p->frame);        // (its reader's stream is on record: hbhip_frame_use_on)
    else if (p && p->owner) p->owner->release(p, last_user);
}

// Host <-> device: on the context's copy streams, synchronous for the CALLER only (see hbhip_ctx::up_stream):
// upload returns when `src` has been consumed, download when `dst` is filled.
int hbhip_copy_h2d(hbhip_ctx *ctx, DevPicture *dst, const hbhip_host_frame *src);
int hbhip_copy_d2h(hbhip_ctx *ctx, const hbhip_host_frame *dst, const DevPicture *src);
// attach `p` to the context's open mark (the picture is being recycled; its users are all queued on ctx->stream)
void hbhip_pic_mark_idle(hbhip_ctx *ctx, DevPicture *p);
// has everything that used `p` run? (records the mark if it still is open)
bool hbhip_pic_idle_done(DevPicture *p);
// make `stream` wait for everything that used `p` (nothing to do when that was queued on `stream` itself)
hipError_t hbhip_pic_wait_idle(hipStream_t stream, DevPicture *p);
// Device <-> device copies (2-D, any pitch on either side) on ctx->stream.
int hbhip_copy_d2d_in(hbhip_ctx *ctx, DevPicture *dst, const hbhip_dev_frame *src);
int hbhip_copy_d2d_out(hbhip_ctx *ctx, const hbhip_dev_frame *dst, const DevPicture *src);

// Base class of every filter instance behind the C ABI.
struct hbhip_filter
{
    hbhip_ctx  *ctx = nullptr;
    PicGeometry in_geo, out_geo;
    // byte stride the CALLER's planes had at the last push (some reference
    // filters derive an edge rule from it, e.g. lapsharp.c:145)
    int         in_stride[3] = {0, 0, 0};
    bool        in_is_dev = false;      // last push came from a device-resident frame
    explicit hbhip_filter(hbhip_ctx *c) : ctx(c) {}
    virtual ~hbhip_filter() {}

    // Input side: the subclass gets a device picture already filled.
    virtual DevPicture *acquire_input() = 0;
    virtual int submit(DevPicture *pic) = 0;       // takes ownership
    // give back a picture from acquire_input() that was never submitted (a failed upload)
    virtual void abandon_input(DevPicture *pic) { hbhip_pic_release(pic, ctx); }
    virtual int flush() = 0;
    // Output side.
    virtual int pending() = 0;
    virtual DevPicture *pop_output() = 0;          // nullptr when none
    virtual void recycle_output(DevPicture *pic) = 0;
    // Device-resident batch: default = push_dev/pull_dev per frame; filters may override
    // with a zero-copy path that reads the caller's frames and writes its outputs in place.
    virtual int process_dev_batch(const hbhip_dev_frame *in, int n_in, int64_t tag0,
                                  const hbhip_dev_frame *out, int out_cap, int *n_out);
    // ---- fused chains (hbhip_chain) ----
    // Launch whatever is ready now (filters that wait for a batch to fill up).
    virtual int kick() { return HBHIP_OK; }
    // true: the filter gathers frames until kick() instead of launching on its own batch threshold
    virtual void defer_launches(bool) {}
    // One frame in, written straight into the caller's picture `out` (a view, not pool-owned);
    // takes ownership of `in`.  HBHIP_ERR_UNSUPPORTED when the filter has no such path.
    virtual int submit_to(DevPicture *, const DevPicture *) { return HBHIP_ERR_UNSUPPORTED; }
    virtual bool can_submit_to() const { return false; }
    // ---- frames in, frames out (hbhip_filter_use_frames: the drop-ins of a device-resident run) ----
    bool frames_mode = false;
    virtual int use_frames() { return HBHIP_ERR_UNSUPPORTED; }      // switch the filter's pools to frame-backed pictures
    virtual void adopt_input(DevPicture *pic) { pic->refs = 0; }      // a pushed frame's picture becomes an input picture
    // Several pictures at once (a chain batch).  out_views != nullptr: write into those pictures (only when
    // can_submit_to()); default = one by one.
    virtual int submit_many(DevPicture *const *pics, int n, const DevPicture *out_views)
    {
        for (int i = 0; i < n; i++)
        {
            int rc = out_views ? submit_to(pics[i], &out_views[i]) : submit(pics[i]);
            if (rc != HBHIP_OK) return rc;
        }
        return HBHIP_OK;
    }
    // ---- pipelined host path (hbhip_filter_submit_async): one-in / one-out filters ----
    virtual DevPicture *acquire_output() { return nullptr; }
    virtual int process_pair(DevPicture *, DevPicture *) { return HBHIP_ERR_UNSUPPORTED; }   // in -> out on ctx->stream
    struct AsyncSlot { DevPicture *out; hipEvent_t done; int64_t tag; };
    std::deque<AsyncSlot> async_q;
};

// A stateless one-frame-in / one-frame-out filter: subclasses implement process().
struct SimpleFilter : hbhip_filter
{
    PicturePool in_pool, out_pool;
    std::deque<DevPicture *> outq;
    explicit SimpleFilter(hbhip_ctx *c) : hbhip_filter(c) {}

    void configure(const PicGeometry &gin, const PicGeometry &gout)
    {
        in_geo = gin;
        out_geo = gout;
        in_pool.configure(ctx, gin);
        out_pool.configure(ctx, gout);
    }
    virtual int process(DevPicture *in, DevPicture *out) = 0;
    // n frames at once: filters whose kernels take several frames per launch override this (single planes are
    // too small to fill the GPU or to hide a launch)
    virtual int process_many(DevPicture *const *ins, DevPicture *const *outs, int n)
    {
        for (int i = 0; i < n; i++)
        {
            int rc = process(ins[i], outs[i]);
            if (rc != HBHIP_OK) return rc;
        }
        return HBHIP_OK;
    }
    // fused chains: the pictures of a batch in one go; outs == nullptr -> own output pictures (queued), else the
    // caller's pictures (views).  Takes ownership of the inputs.
    int submit_many(DevPicture *const *pics, int n, const DevPicture *out_views) override
    {
        std::vector<DevPicture *> outs(n, nullptr);
        std::vector<DevPicture> views;
        if (out_views)
        {
            views.assign(out_views, out_views + n);
            for (int i = 0; i < n; i++) { views[i].tag = pics[i]->tag; outs[i] = &views[i]; }
        }
        else
            for (int i = 0; i < n; i++)
            {
                outs[i] = out_pool.acquire();
                if (!outs[i])
                {
                    for (int k = 0; k < i; k++) out_pool.release(outs[k]);
                    return HBHIP_ERR_NOMEM;
                }
                outs[i]->tag = pics[i]->tag;
            }
        int rc = process_many(pics, outs.data(), n);
        for (int i = 0; i < n; i++) hbhip_pic_release(pics[i], ctx);
        if (rc != HBHIP_OK)
        {
            if (!out_views) for (DevPicture *o : outs) out_pool.release(o);
            return rc;
        }
        if (!out_views) for (DevPicture *o : outs) outq.push_back(o);
        return HBHIP_OK;
    }

    DevPicture *acquire_input() override { return in_pool.acquire(); }
    int submit(DevPicture *pic) override
    {
        DevPicture *o = out_pool.acquire();
        if (!o) return HBHIP_ERR_NOMEM;
        o->tag = pic->tag;
        int rc = process(pic, o);
        hbhip_pic_release(pic, ctx);       // stream-ordered reuse (back to whichever pool made it)
        if (rc != HBHIP_OK)
        {
            out_pool.release(o);
            return rc;
        }
        outq.push_back(o);
        return HBHIP_OK;
    }
    int flush() override { return HBHIP_OK; }
    int use_frames() override { in_pool.use_frames(true); out_pool.use_frames(true); frames_mode = true; return HBHIP_OK; }
    int pending() override { return (int)outq.size(); }
    DevPicture *pop_output() override
    {
        if (outq.empty()) return nullptr;
        DevPicture *p = outq.front();
        outq.pop_front();
        return p;
    }
    void recycle_output(DevPicture *p) override { out_pool.release(p); }
    DevPicture *acquire_output() override { return out_pool.acquire(); }
    int process_pair(DevPicture *in, DevPicture *out) override { out->tag = in->tag; return process(in, out); }
    bool can_submit_to() const override { return outq.empty(); }
    int submit_to(DevPicture *pic, const DevPicture *out) override
    {
        DevPicture vo = *out;
        vo.tag = pic->tag;
        int rc = process(pic, &vo);
        hbhip_pic_release(pic, ctx);
        return rc;
    }

    // Zero-copy: a stateless filter can read the caller's device frame and write the caller's
    // output frame directly (no pool pictures, no device-to-device copies).  Falls back to the
    // copying path when a plane is not 16-byte aligned (some kernels use vector loads).
    int process_dev_batch(const hbhip_dev_frame *in, int n_in, int64_t tag0,
                          const hbhip_dev_frame *out, int out_cap, int *n_out) override
    {
        bool direct = outq.empty() && n_in <= out_cap;
        for (int i = 0; direct && i < n_in; i++)
            for (int c = 0; c < 3; c++)
                if ((in[i].stride[c] & 15) || ((uintptr_t)in[i].plane[c] & 15) ||
                    (out[i].stride[c] & 15) || ((uintptr_t)out[i].plane[c] & 15))
                    direct = false;
        if (!direct) return hbhip_filter::process_dev_batch(in, n_in, tag0, out, out_cap, n_out);
        std::vector<DevPicture> vi(n_in), vo(n_in);
        std::vector<DevPicture *> pi(n_in), po(n_in);
        for (int i = 0; i < n_in; i++)
        {
            for (int c = 0; c < 3; c++)
            {
                vi[i].plane[c] = (uint8_t *)in[i].plane[c];  vi[i].pitch[c] = in[i].stride[c];
                vi[i].width[c] = in_geo.pw[c];               vi[i].height[c] = in_geo.ph[c];
                vo[i].plane[c] = (uint8_t *)out[i].plane[c]; vo[i].pitch[c] = out[i].stride[c];
                vo[i].width[c] = out_geo.pw[c];              vo[i].height[c] = out_geo.ph[c];
                in_stride[c] = in[i].stride[c];
            }
            vi[i].bps = in_geo.bps; vo[i].bps = out_geo.bps;
            vi[i].tag = vo[i].tag = tag0 + i;
            pi[i] = &vi[i]; po[i] = &vo[i];
        }
        in_is_dev = true;
        int rc = process_many(pi.data(), po.data(), n_in);
        if (rc != HBHIP_OK) return rc;
        *n_out = n_in;
        return HBHIP_OK;
    }
};
