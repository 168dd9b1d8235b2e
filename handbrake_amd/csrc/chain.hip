// chain.hip — a run of adjacent HIP filters fused into one object (hbhip_chain).
//
// libhb gives every filter its own thread and fifo (work.c:2527-2600) — except for runs of
// libavfilter-backed filters, which hb_avfilter_combine (hbavfilter.c:510-622) merges into ONE
// filter object whose work() pushes a frame through the whole graph.  hbhip_chain is that idea
// for a run of HIP filters: one caller, one stream, and pictures handed from one stage to the
// next by pointer (hbhip_pic_release returns each to the pool of the stage that produced it),
// so a frame that enters the chain is read from HBM once per stage and never copied between
// stages.  A batch of input frames is walked stage by stage: stage k sees all the frames of the
// batch before stage k+1 starts, which lets batching stages (NLMeans) cover them in one launch.
// (Round 6 tried the other order for the stateless tail of a chain - parts of ~100 MB of intermediate walked through
// scaler and lapsharp while they are hot in the 256 MB Infinity Cache, because sixteen 2160p frames are 199 MB at 8 bits
// and 398 MB at 10 and lapsharp takes 279 us per 16 frames behind the scaler against 211 on cache-resident input.  What
// lapsharp gained the scaler lost on launches a quarter the size: 10 bits 1114 -> 896 and 827 -> 954 us per two steps,
// 8 bits both slower - profiles/r6h_bench_chain*_chunked.json.  Not kept.)
//
// The last stage writes straight into the caller's output frames when it can (stateless filters);
// the first stage gets the caller's frames through one 3-plane copy launch (stateful filters keep
// input pictures beyond the call, the caller's frames are only borrowed for its duration).
//
// Stages may live on contexts (HIP streams) of their own - the counterpart of libhb's one thread per
// filter: the chain then orders stage k+1 behind stage k with an event per hand-over, nothing else,
// so while the last stages work on one batch the first ones already run the next.  Pictures find
// their way back across streams through their `idle` event (PicturePool).  The caller's input frames
// must be complete in the chain context's stream order, and may be rewritten in that order as soon as
// the call has returned (the chain's stream waits for the copy-in); the output frames are written behind
// whatever the chain context's stream held when the call was made, and are complete once
// hbhip_chain_sync() returns (or the last stage's context has been synchronized).
#include "hbhip_internal.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <new>
#include <vector>

namespace {

// The copy-in (or copy-out) of a whole batch: up to COPY_FRAMES frames, three planes each, in one launch (blockIdx.z = 3 * frame +
// plane).  All frames of a call share the plane geometry; source and destination pitches are per call too (the
// caller's frames of one call share their strides by contract, the pool's pictures by construction).
constexpr int COPY_FRAMES = 16;
struct CopyBatch
{
    const uint8_t *src[COPY_FRAMES][3];
    uint8_t       *dst[COPY_FRAMES][3];
    int spitch[3], dpitch[3], row_bytes[3], rows[3];
};

__global__ void __launch_bounds__(256) copy3_batch_kernel(CopyBatch a)
{
    const int fr = blockIdx.z / 3, pl = blockIdx.z - 3 * fr;
    const int y = blockIdx.y;
    if (y >= a.rows[pl]) return;
    const int x = (blockIdx.x * 256 + threadIdx.x) * 16;
    const int rb = a.row_bytes[pl];
    if (x >= rb) return;
    const uint8_t *s = a.src[fr][pl] + (size_t)y * a.spitch[pl] + x;
    uint8_t *d = a.dst[fr][pl] + (size_t)y * a.dpitch[pl] + x;
    if (x + 16 <= rb && (((uintptr_t)s | (uintptr_t)d) & 15) == 0)
        *reinterpret_cast<uint4 *>(d) = *reinterpret_cast<const uint4 *>(s);
    else
        for (int i = 0; i < 16 && x + i < rb; i++) d[i] = s[i];
}

int copy3_batch(hbhip_ctx *ctx, const CopyBatch &a, int frames)
{
    int maxrow = 0, maxrows = 0;
    for (int c = 0; c < 3; c++) { maxrow = std::max(maxrow, a.row_bytes[c]); maxrows = std::max(maxrows, a.rows[c]); }
    dim3 grid((maxrow + 4095) / 4096, maxrows, 3 * frames);
    HBHIP_LAUNCH(ctx, "copy_planes", copy3_batch_kernel, grid, dim3(256), 0, a);
    HBHIP_CHECK(ctx, hipGetLastError());
    return HBHIP_OK;
}

} // namespace

struct hbhip_chain
{
    hbhip_ctx *ctx = nullptr;
    std::vector<hbhip_filter *> st;
    std::deque<DevPicture *> held;          // finished pictures the caller had no room for yet
    std::vector<hipEvent_t> ev;             // ev[s]: stage s has been given everything of the current batch; ev[n]: the caller's inputs;
                                            // ev[n+1]: the copy-in is done; ev[n+2]: the caller's earlier use of the output frames
    bool split = false;                     // stages on more than one context
    std::vector<double> host_ms;            // HBHIP_CHAIN_TIMING: host time spent per stage (+ copy-in at the end)
    bool timing = hbhip_dev_int("HBHIP_CHAIN_TIMING", 0) != 0;   // development builds only

    // make `to`'s stream wait for what `from`'s stream holds right now
    int order(hbhip_ctx *from, hbhip_ctx *to, hipEvent_t e)
    {
        if (from == to) return HBHIP_OK;
        HBHIP_CHECK(ctx, hipEventRecord(e, from->stream));
        HBHIP_CHECK(ctx, hipStreamWaitEvent(to->stream, e, 0));
        return HBHIP_OK;
    }

    // the caller's frames in[0..n) into the first stage's pictures dst[0..n): one launch per COPY_FRAMES frames that share
    // their pitches (a frame whose pitches differ from the one before it starts a new launch)
    int copy_in(DevPicture *const *dst, const hbhip_dev_frame *src, int n)
    {
        for (int i = 0; i < n; i++)
            for (int c = 0; c < 3; c++)
                if (src[i].plane[c] == nullptr || src[i].stride[c] < dst[i]->width[c] * dst[i]->bps) return HBHIP_ERR_ARG;
        for (int i0 = 0; i0 < n;)
        {
            CopyBatch a;
            for (int c = 0; c < 3; c++)
            {
                a.spitch[c] = src[i0].stride[c]; a.dpitch[c] = dst[i0]->pitch[c];
                a.row_bytes[c] = dst[i0]->width[c] * dst[i0]->bps; a.rows[c] = dst[i0]->height[c];
            }
            int k = 0;
            for (; i0 + k < n && k < COPY_FRAMES; k++)
            {
                bool same = true;
                for (int c = 0; c < 3; c++)
                    same &= src[i0 + k].stride[c] == a.spitch[c] && dst[i0 + k]->pitch[c] == a.dpitch[c];
                if (!same) break;
                for (int c = 0; c < 3; c++) { a.src[k][c] = (const uint8_t *)src[i0 + k].plane[c]; a.dst[k][c] = dst[i0 + k]->plane[c]; }
            }
            int rc = copy3_batch(st.front()->ctx, a, k);
            if (rc != HBHIP_OK) return rc;
            i0 += k;
        }
        return HBHIP_OK;
    }
    // Deliver held / freshly finished pictures into the caller's frames: one copy launch per COPY_FRAMES pictures that
    // share their pitches.  A picture whose output frame cannot take it is given up (recycled) and the call fails, after
    // the pictures ahead of it have been delivered.
    int deliver(const hbhip_dev_frame *out, int64_t *tags, int cap, int &produced)
    {
        hbhip_filter *last = st.back();
        while (!held.empty() && produced < cap)
        {
            CopyBatch a;
            const DevPicture *p0 = held.front();
            for (int c = 0; c < 3; c++)
            {
                a.spitch[c] = p0->pitch[c]; a.dpitch[c] = out[produced].stride[c];
                a.row_bytes[c] = p0->width[c] * p0->bps; a.rows[c] = p0->height[c];
            }
            int k = 0;
            bool bad = false;
            for (; k < COPY_FRAMES && k < (int)held.size() && produced + k < cap; k++)
            {
                const DevPicture *p = held[k];
                const hbhip_dev_frame &d = out[produced + k];
                bool same = true;
                for (int c = 0; c < 3; c++)
                {
                    bad |= d.plane[c] == nullptr || d.stride[c] < a.row_bytes[c];
                    same &= p->pitch[c] == a.spitch[c] && d.stride[c] == a.dpitch[c];
                }
                if (bad || !same) break;
                for (int c = 0; c < 3; c++) { a.src[k][c] = p->plane[c]; a.dst[k][c] = (uint8_t *)d.plane[c]; }
            }
            if (k == 0)
            {
                // `bad` (a frame always shares its pitches with itself)
                DevPicture *p = held.front();
                held.pop_front();
                last->recycle_output(p);
                return HBHIP_ERR_ARG;
            }
            const int rc = copy3_batch(last->ctx, a, k);
            for (int i = 0; i < k; i++)
            {
                DevPicture *p = held.front();
                held.pop_front();
                if (tags) tags[produced + i] = p->tag;
                last->recycle_output(p);
            }
            if (rc != HBHIP_OK) return rc;
            produced += k;
        }
        return HBHIP_OK;
    }

    int run(const hbhip_dev_frame *in, const int *flags, const int *combed, int n_in, int64_t tag0, bool flush,
            const hbhip_dev_frame *out, int64_t *tags, int cap, int *n_out)
    {
        int produced = 0;
        int rc = deliver(out, tags, cap, produced);
        if (rc != HBHIP_OK) return rc;

        std::vector<DevPicture *> cur, next;
        hbhip_filter *first = st.front();
        auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        if (timing && host_ms.empty()) host_ms.assign(st.size() + 1, 0.0);
        double t_mark = timing ? now() : 0.0;
        if (n_in > 0) { rc = order(ctx, first->ctx, ev[st.size()]); if (rc != HBHIP_OK) return rc; }
        // whoever writes the caller's output frames (the last stage, direct or by copy-out) does so behind what the
        // caller's stream has queued on them so far - also on a flush, or when only delayed frames come out
        if (cap > 0) { rc = order(ctx, st.back()->ctx, ev[st.size() + 2]); if (rc != HBHIP_OK) return rc; }
        for (int i = 0; i < n_in; i++)
        {
            DevPicture *p = first->acquire_input();
            if (!p)
            {
                for (DevPicture *q : cur) first->abandon_input(q);
                return HBHIP_ERR_NOMEM;
            }
            p->tag = tag0 + i;
            p->refs = 0;
            if (flags) p->flags = flags[i];
            if (combed) p->combed = combed[i];
            for (int c = 0; c < 3; c++) first->in_stride[c] = in[i].stride[c];
            first->in_is_dev = true;
            cur.push_back(p);
        }
        if (n_in > 0)
        {
            rc = copy_in(cur.data(), in, n_in);
            if (rc != HBHIP_OK)
            {
                for (DevPicture *q : cur) first->abandon_input(q);
                return rc;
            }
        }
        // the caller may refill its input frames in its stream's order from here on
        if (n_in > 0) { rc = order(first->ctx, ctx, ev[st.size() + 1]); if (rc != HBHIP_OK) return rc; }
        if (timing) { const double t = now(); host_ms[st.size()] += t - t_mark; t_mark = t; }
        for (size_t s = 0; s < st.size(); s++)
        {
            hbhip_filter *f = st[s];
            const bool is_last = s + 1 == st.size();
            // the last stage writes the caller's frames itself when every frame it is about to
            // make has a slot and nothing older is waiting
            bool direct = is_last && held.empty() && f->can_submit_to() && produced + (int)cur.size() <= cap;
            std::vector<DevPicture> views;
            if (direct)
            {
                views.resize(cur.size());
                for (size_t i = 0; i < cur.size() && direct; i++)
                {
                    DevPicture &vo = views[i];
                    for (int c = 0; c < 3; c++)
                    {
                        vo.plane[c] = (uint8_t *)out[produced + i].plane[c]; vo.pitch[c] = out[produced + i].stride[c];
                        vo.width[c] = f->out_geo.pw[c]; vo.height[c] = f->out_geo.ph[c];
                        if (vo.plane[c] == nullptr || vo.pitch[c] < vo.width[c] * f->out_geo.bps ||
                            (vo.pitch[c] & 15) || ((uintptr_t)vo.plane[c] & 15))
                            direct = false;
                    }
                    vo.bps = f->out_geo.bps;
                }
            }
            if (!cur.empty())
            {
                if (s > 0)
                {
                    for (int c = 0; c < 3; c++) f->in_stride[c] = cur[0]->pitch[c];
                    f->in_is_dev = true;
                    rc = order(st[s - 1]->ctx, f->ctx, ev[s - 1]);        // the pictures of `cur` are complete on the previous stage's stream
                    if (rc != HBHIP_OK) return rc;
                }
                if (direct && tags) for (size_t i = 0; i < cur.size(); i++) tags[produced + i] = cur[i]->tag;
                rc = f->submit_many(cur.data(), (int)cur.size(), direct ? views.data() : nullptr);
                if (rc != HBHIP_OK) return rc;
                if (direct) produced += (int)cur.size();
            }
            rc = flush ? f->flush() : f->kick();
            if (rc != HBHIP_OK) return rc;
            next.clear();
            while (f->pending() > 0)
            {
                DevPicture *o = f->pop_output();
                if (!o) break;
                o->refs = 0;
                next.push_back(o);
            }
            cur.swap(next);
            if (timing) { const double t = now(); host_ms[s] += t - t_mark; t_mark = t; }
        }
        for (DevPicture *p : cur) held.push_back(p);
        rc = deliver(out, tags, cap, produced);
        if (rc != HBHIP_OK) return rc;
        *n_out = produced;
        return HBHIP_OK;
    }
};

extern "C" {

int hbhip_chain_create(hbhip_ctx *ctx, hbhip_filter *const *stages, int n_stages, hbhip_chain **out)
{
    if (!ctx || !stages || n_stages < 1 || !out) return HBHIP_ERR_ARG;
    *out = nullptr;
    for (int i = 0; i < n_stages; i++)
    {
        if (!stages[i] || !stages[i]->ctx || stages[i]->ctx->device != ctx->device) return HBHIP_ERR_ARG;
        if (i > 0)
        {
            const PicGeometry &a = stages[i - 1]->out_geo, &b = stages[i]->in_geo;
            if (a.width != b.width || a.height != b.height || a.depth != b.depth ||
                a.log2_cw != b.log2_cw || a.log2_ch != b.log2_ch)
                return HBHIP_ERR_ARG;
        }
    }
    hbhip_chain *c = new (std::nothrow) hbhip_chain();
    if (!c) return HBHIP_ERR_NOMEM;
    c->ctx = ctx;
    c->st.assign(stages, stages + n_stages);
    (void)hipSetDevice(ctx->device);
    for (int i = 0; i <= n_stages + 2; i++)
    {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
        {
            for (hipEvent_t x : c->ev) (void)hipEventDestroy(x);
            delete c;
            return HBHIP_ERR_NOMEM;
        }
        c->ev.push_back(e);
    }
    for (hbhip_filter *f : c->st) { f->defer_launches(true); c->split |= f->ctx != ctx; }
    *out = c;
    return HBHIP_OK;
}

int hbhip_chain_process_dev(hbhip_chain *c, const hbhip_dev_frame *in, const int *pic_flags, const int *combed,
                            int n_in, int64_t tag0, const hbhip_dev_frame *out, int64_t *out_tags, int out_cap,
                            int *n_out)
{
    if (!c || (n_in > 0 && !in) || (out_cap > 0 && !out) || !n_out || n_in < 0 || out_cap < 0) return HBHIP_ERR_ARG;
    (void)hipSetDevice(c->ctx->device);
    return c->run(in, pic_flags, combed, n_in, tag0, false, out, out_tags, out_cap, n_out);
}

int hbhip_chain_flush_dev(hbhip_chain *c, const hbhip_dev_frame *out, int64_t *out_tags, int out_cap, int *n_out)
{
    if (!c || (out_cap > 0 && !out) || !n_out || out_cap < 0) return HBHIP_ERR_ARG;
    (void)hipSetDevice(c->ctx->device);
    return c->run(nullptr, nullptr, nullptr, 0, 0, true, out, out_tags, out_cap, n_out);
}

int hbhip_chain_pending(hbhip_chain *c)
{
    return c ? (int)c->held.size() : 0;
}

int hbhip_chain_sync(hbhip_chain *c)
{
    if (!c) return HBHIP_ERR_ARG;
    (void)hipSetDevice(c->ctx->device);
    HBHIP_CHECK(c->ctx, hipStreamSynchronize(c->ctx->stream));
    for (hbhip_filter *f : c->st)
        if (f->ctx != c->ctx) HBHIP_CHECK(c->ctx, hipStreamSynchronize(f->ctx->stream));
    return HBHIP_OK;
}

void hbhip_chain_destroy(hbhip_chain *c)
{
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hbhip_chain_sync(c);
    for (DevPicture *p : c->held) c->st.back()->recycle_output(p);
    for (hbhip_filter *f : c->st) f->defer_launches(false);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    if (c->timing && !c->host_ms.empty())
    {
        fprintf(stderr, "hbhip_chain host ms: copy-in %.2f", c->host_ms.back());
        for (size_t i = 0; i + 1 < c->host_ms.size(); i++) fprintf(stderr, ", stage %zu %.2f", i, c->host_ms[i]);
        fprintf(stderr, "\n");
    }
    delete c;
}

} // extern "C"
