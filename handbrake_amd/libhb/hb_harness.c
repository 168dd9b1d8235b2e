/* hb_harness.c — a single-threaded stand-in for the part of work.c that owns a
 * video filter chain: it instantiates filter objects the way do_job() does
 * (copy the registered object, attach settings, call init() with the running
 * hb_filter_init_t, work.c:1840-1870), then plays filter_loop() (work.c:2527-2600)
 * for every stage in order: work(&in, &out), close `in` unless the filter took
 * it, push the `->next` chain of `out` to the next stage, stop a stage at
 * HB_FILTER_DONE.
 *
 * It is deliberately agnostic of WHICH filter objects it drives: the parity
 * tests run the reference's own objects (oracle/_ref/libhbref.so) and the HIP
 * drop-ins (libhbhip_filters.so) through exactly the same calls.
 */
#include "hbhip_libhb.h"
#include "hb_harness.h"

#include <pthread.h>
#include <time.h>

#define HBH_MAX_STAGES 32

/* threaded mode: a fifo in front of every stage, one thread per stage (filter_loop, work.c:2527-2600) */
/* bounded like libhb's: filter fifos hold FIFO_MINI = 4 buffers (work.c:46, 2113, 2181); a producer waits until
 * there is room and then pushes its whole output list (hb_fifo_full_wait + hb_fifo_push, work.c:2577-2586) */
#define HBH_FIFO_CAP 4
typedef struct
{
    pthread_mutex_t lock;
    pthread_cond_t  cond, room;
    hb_buffer_t    *head, *tail;
    int             count;
} hbh_fifo_t;

struct hbh_chain_s
{
    int                 threaded;
    hbh_fifo_t          fifo[HBH_MAX_STAGES];
    pthread_t           thread[HBH_MAX_STAGES];
    int                 thread_live[HBH_MAX_STAGES];
    pthread_mutex_t     out_lock;
    int                 discard, produced;  /* threaded mode: drop finished frames / how many the last stage has made */
    double              busy_ms[HBH_MAX_STAGES];   /* threaded mode: time each stage's thread spent inside work() */
    int                 nstages;
    hb_filter_object_t *stage[HBH_MAX_STAGES];
    int                 done[HBH_MAX_STAGES];
    hb_filter_init_t    init;        /* state after the last stage's init()   */
    hb_filter_init_t    init_in;     /* what the first stage was given        */
    hb_buffer_list_t    out;         /* frames that left the last stage       */
    int                 eof_seen;
    int                 failed;
    hb_job_t           *job;         /* hbh_job_open: what init->job points to */
    hb_subtitle_t      *subtitle;    /*               the job's burn-in track, if any */
    hb_title_t          title;
};

static hb_filter_object_t *clone_filter(const hb_filter_object_t *proto, const char *settings)
{
    /* hb_filter_copy (common.c:5247-5258): shallow copy + own settings dict */
    hb_filter_object_t *f = malloc(sizeof(*f));
    if (f == NULL) return NULL;
    memcpy(f, proto, sizeof(*f));
    f->settings = hbhip_dict_from_string(settings ? settings : "");
    f->private_data = NULL;
    f->sub_filter = NULL;
    return f;
}

static int g_src_color[4] = { 1, 1, 1, 1 };
static int g_threaded = 0;
static int g_discard = 0;            /* threaded mode: count and drop the last stage's frames (a consumer that keeps up) */

/* Chains opened from now on run every stage on a thread of its own, as libhb does (work.c:2527-2600): distinct
 * filters then call into the shared device context concurrently.  Frames pushed are queued; hbh_chain_push_eof()
 * returns when every stage has finished, so output is complete once it returns. */
static int g_job_device = -1;        /* job->hw_device_index of jobs opened from now on (common.h:991; -1 = not set) */
void hbh_set_job_device(int index) { g_job_device = index; }
static int g_job_subtitle = -1;      /* jobs opened from now on carry one subtitle track of this source marked for burn-in */
void hbh_set_job_subtitle(int source) { g_job_subtitle = source; }
void hbh_set_threaded(int on) { g_threaded = on; }
void hbh_set_discard_output(int on) { g_discard = on; }

static void fifo_put(hbh_fifo_t *q, hb_buffer_t *list)        /* a buffer or a ->next list of them */
{
    pthread_mutex_lock(&q->lock);
    while (q->count >= HBH_FIFO_CAP) pthread_cond_wait(&q->room, &q->lock);
    if (q->tail) q->tail->next = list; else q->head = list;
    for (hb_buffer_t *b = list; b != NULL; b = b->next) { q->tail = b; q->count++; }
    pthread_cond_signal(&q->cond);
    pthread_mutex_unlock(&q->lock);
}

static hb_buffer_t *fifo_get(hbh_fifo_t *q)
{
    pthread_mutex_lock(&q->lock);
    while (q->head == NULL) pthread_cond_wait(&q->cond, &q->lock);
    hb_buffer_t *b = q->head;
    q->head = b->next;
    if (q->head == NULL) q->tail = NULL;
    b->next = NULL;
    if (--q->count < HBH_FIFO_CAP) pthread_cond_signal(&q->room);
    pthread_mutex_unlock(&q->lock);
    return b;
}

typedef struct { hbh_chain_t *c; int s; } stage_arg_t;

static void *stage_loop(void *pv)                 /* filter_loop (work.c:2527-2600) */
{
    stage_arg_t *a = pv;
    hbh_chain_t *c = a->c;
    const int s = a->s;
    free(a);
    hb_filter_object_t *f = c->stage[s];
    for (;;)
    {
        hb_buffer_t *in = fifo_get(&c->fifo[s]), *out = NULL;
        const int eof_in = (in->s.flags & HB_BUF_FLAG_EOF) != 0;
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        int status = f->work(f, &in, &out);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        c->busy_ms[s] += (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
        if (in != NULL) hb_buffer_close(&in);
        if (status == HB_FILTER_FAILED)
        {
            c->failed = 1;
            if (out == NULL) out = hb_buffer_eof_init();      /* let the stages behind finish */
        }
        if (out != NULL && s + 1 < c->nstages)
        {
            fifo_put(&c->fifo[s + 1], out);
            out = NULL;
        }
        while (out != NULL)
        {
            hb_buffer_t *next = out->next;
            out->next = NULL;
            {
                pthread_mutex_lock(&c->out_lock);
                if (!(out->s.flags & HB_BUF_FLAG_EOF)) c->produced++;
                if (c->discard && !(out->s.flags & HB_BUF_FLAG_EOF)) hb_buffer_close(&out);
                else hb_buffer_list_append(&c->out, out);
                pthread_mutex_unlock(&c->out_lock);
            }
            out = next;
        }
        if (status == HB_FILTER_DONE || status == HB_FILTER_FAILED || (eof_in && status != HB_FILTER_DELAY)) break;
    }
    return NULL;
}

static void start_threads(hbh_chain_t *c)
{
    if (!g_threaded) return;
    c->threaded = 1;
    c->discard = g_discard;
    pthread_mutex_init(&c->out_lock, NULL);
    for (int s = 0; s < c->nstages; s++)
    {
        pthread_mutex_init(&c->fifo[s].lock, NULL);
        pthread_cond_init(&c->fifo[s].cond, NULL);
        pthread_cond_init(&c->fifo[s].room, NULL);
    }
    for (int s = 0; s < c->nstages; s++)
    {
        stage_arg_t *a = malloc(sizeof(*a));
        a->c = c; a->s = s;
        c->thread_live[s] = pthread_create(&c->thread[s], NULL, stage_loop, a) == 0;
    }
}

static void join_threads(hbh_chain_t *c)
{
    for (int s = 0; s < c->nstages; s++)
        if (c->thread_live[s])
        {
            pthread_join(c->thread[s], NULL);
            c->thread_live[s] = 0;
        }
}

/* what work.c calls directly inside libhb (hip_common.h); here the filter library registers them when it loads */
static void (*g_hip_setup)(hb_job_t *) = NULL;
static int  (*g_hip_init_failed)(hb_job_t *, int, hb_filter_init_t *) = NULL;
static void (*g_hip_job_close)(hb_job_t *) = NULL;

void hbhip_rt_set_job_hooks(void (*setup)(hb_job_t *), int (*init_failed)(hb_job_t *, int, hb_filter_init_t *),
                            void (*job_close)(hb_job_t *))
{
    g_hip_setup = setup;
    g_hip_init_failed = init_failed;
    g_hip_job_close = job_close;
}

/* the hb_job_t of a chain opened with hbh_job_open (NULL for hbh_chain_open) */
void *hbh_chain_job(hbh_chain_t *c) { return c != NULL ? c->job : NULL; }

void hbh_set_source_color(int prim, int transfer, int matrix, int range)
{
    g_src_color[0] = prim; g_src_color[1] = transfer; g_src_color[2] = matrix; g_src_color[3] = range;
}

static void source_init(hb_filter_init_t *init, int pix_fmt, int width, int height, int vrate_num, int vrate_den);

/* A job the way do_job() prepares its video filters (work.c:1820-1870): the filter list is built from the
 * REGISTERED (CPU) objects by id - hb_filter_init + hb_add_filter_dict, which orders by id (hb.c:1701-1713) -, then
 * sanitize_filter_list_post's hook swaps in the HIP drop-ins (hb_hip_setup_hw_filters), then every filter is
 * initialised with the running hb_filter_init_t; a filter whose init fails is dropped (:1861-1868) unless the hook
 * puts its CPU filter back (hb_hip_filter_init_failed).  use_hip = 0 leaves the list as registered. */
hbh_chain_t *hbh_job_open(int nfilters, const int *ids, const char *const *settings, int pix_fmt, int width, int height,
                          int vrate_num, int vrate_den, int use_hip)
{
    hbh_chain_t *c = calloc(1, sizeof(*c));
    if (c == NULL) return NULL;
    /* the filters keep init->job (vfr.c:413, :546), so the job lives as long as the chain */
    hb_job_t *pjob = calloc(1, sizeof(*pjob));
    if (pjob == NULL) { free(c); return NULL; }
    c->job = pjob;
    pjob->hw_pix_fmt = AV_PIX_FMT_NONE;
    pjob->input_pix_fmt = pix_fmt;
    pjob->hw_device_index = g_job_device;
    pjob->list_filter = hb_list_init();
    if (g_job_subtitle >= 0)
    {
        /* a track marked for burn-in (rendersub.c:1199-1209 looks for config.dest == RENDERSUB); its decoder's output is
         * what hbh_chain_push_subtitle() puts into fifo_out */
        hb_subtitle_t *sub = calloc(1, sizeof(*sub));
        c->title.geometry.width = width; c->title.geometry.height = height;
        c->title.geometry.par.num = c->title.geometry.par.den = 1;
        pjob->title = &c->title;
        pjob->list_subtitle = hb_list_init();
        pjob->list_attachment = hb_list_init();
        if (sub != NULL)
        {
            sub->source = g_job_subtitle;
            sub->format = PICTURESUB;
            sub->config.dest = RENDERSUB;
            sub->width = width; sub->height = height;
            sub->fifo_out = hb_fifo_init(8, 1);
            hb_list_add(pjob->list_subtitle, sub);
            c->subtitle = sub;
        }
    }
    for (int i = 0; i < nfilters; i++)
    {
        hb_filter_object_t *f = hb_filter_init(ids[i]);
        if (f == NULL)
        {
            hb_error("hbh_job_open: no filter registered for id %d", ids[i]);
            continue;
        }
        hb_dict_t *d = hbhip_dict_from_string(settings && settings[i] ? settings[i] : "");
        hb_add_filter_dict(pjob->list_filter, f, d);
        hb_dict_free(&d);
    }
    if (use_hip && g_hip_setup != NULL) g_hip_setup(pjob);

    hb_filter_init_t init;
    source_init(&init, pix_fmt, width, height, vrate_num, vrate_den);
    init.job = pjob;
    c->init_in = init;
    for (int i = 0; i < hb_list_count(pjob->list_filter);)
    {
        hb_filter_object_t *f = hb_list_item(pjob->list_filter, i);
        f->private_data = NULL;
        if (f->init != NULL && f->init(f, &init))
        {
            const int back = (use_hip && g_hip_init_failed != NULL) ? g_hip_init_failed(pjob, i, &init) : 0;
            if (back > 0)
            {
                i -= back - 1;
                continue;
            }
            hb_log("Failure to initialise filter '%s', disabling", f->name);
            hb_list_rem(pjob->list_filter, f);
            hb_filter_close(&f);
            continue;
        }
        i++;
    }
    /* post_init: the filters learn the final job (work.c:1891-1898); one that fails it is dropped like a failed init */
    for (int i = 0; i < hb_list_count(pjob->list_filter);)
    {
        hb_filter_object_t *f = hb_list_item(pjob->list_filter, i);
        if (f->post_init != NULL && f->post_init(f, pjob))
        {
            hb_log("Failure to initialise filter '%s', disabling", f->name);
            hb_list_rem(pjob->list_filter, f);
            if (f->close != NULL) f->close(f);
            hb_filter_close(&f);
            continue;
        }
        i++;
    }
    for (int i = 0; i < hb_list_count(pjob->list_filter) && c->nstages < HBH_MAX_STAGES; i++)
        c->stage[c->nstages++] = hb_list_item(pjob->list_filter, i);
    hb_list_close(&pjob->list_filter);
    c->init = init;
    start_threads(c);
    return c;
}

/* "name|name|..." of the stages, for tests that check what the swap / fallback left in the list */
int hbh_chain_describe(hbh_chain_t *c, char *buf, int len)
{
    if (c == NULL || buf == NULL || len < 1) return -1;
    int n = 0;
    buf[0] = 0;
    for (int i = 0; i < c->nstages; i++)
        n += snprintf(buf + n, n < len ? len - n : 0, "%s%s", i ? "|" : "", c->stage[i]->name);
    return c->nstages;
}

hbh_chain_t *hbh_chain_open(int nstages, void *const *protos, const char *const *settings,
                            int pix_fmt, int width, int height,
                            int vrate_num, int vrate_den)
{
    if (nstages < 1 || nstages > HBH_MAX_STAGES) return NULL;
    hbh_chain_t *c = calloc(1, sizeof(*c));
    if (c == NULL) return NULL;

    hb_filter_init_t init;
    source_init(&init, pix_fmt, width, height, vrate_num, vrate_den);
    c->init_in = init;

    init.pix_fmt = pix_fmt;
    init.hw_pix_fmt = AV_PIX_FMT_NONE;
    init.geometry.width = width;
    init.geometry.height = height;
    init.geometry.par.num = 1;
    init.geometry.par.den = 1;
    init.vrate.num = vrate_num;
    init.vrate.den = vrate_den;
    init.time_base.num = 1;
    init.time_base.den = 90000;
    init.color_prim = g_src_color[0];
    init.color_transfer = g_src_color[1];
    init.color_matrix = g_src_color[2];
    init.color_range = g_src_color[3];
    init.chroma_location = 1;
    c->init_in = init;

    for (int i = 0; i < nstages; i++)
    {
        hb_filter_object_t *f = clone_filter((const hb_filter_object_t *)protos[i], settings ? settings[i] : NULL);
        if (f == NULL || f->init == NULL || f->init(f, &init) != 0)
        {
            /* work.c:1861-1868 would drop the filter and continue; a test
             * harness wants to know instead. */
            hb_error("hbh_chain_open: init of stage %d (%s) failed", i, f ? f->name : "?");
            if (f) { hb_dict_free(&f->settings); free(f); }
            hbh_chain_close(c);
            return NULL;
        }
        c->stage[c->nstages++] = f;
    }
    c->init = init;
    start_threads(c);
    return c;
}

static void source_init(hb_filter_init_t *pinit, int pix_fmt, int width, int height, int vrate_num, int vrate_den)
{
    hb_filter_init_t init;
    memset(&init, 0, sizeof(init));
    init.pix_fmt = pix_fmt;
    init.hw_pix_fmt = AV_PIX_FMT_NONE;
    init.geometry.width = width;
    init.geometry.height = height;
    init.geometry.par.num = 1;
    init.geometry.par.den = 1;
    init.vrate.num = vrate_num;
    init.vrate.den = vrate_den;
    init.time_base.num = 1;
    init.time_base.den = 90000;
    init.color_prim = g_src_color[0];
    init.color_transfer = g_src_color[1];
    init.color_matrix = g_src_color[2];
    init.color_range = g_src_color[3];
    init.chroma_location = 1;
    *pinit = init;
}

/* Run `in` (one buffer, not a chain) through stage s and everything after it. */
static void run_from(hbh_chain_t *c, int s, hb_buffer_t *in)
{
    if (s >= c->nstages)
    {
        hb_buffer_list_append(&c->out, in);
        return;
    }
    hb_filter_object_t *f = c->stage[s];
    if (c->done[s])
    {
        /* stage already finished (saw EOF): filter_loop has exited, drop */
        hb_buffer_close(&in);
        return;
    }
    hb_buffer_t *out = NULL;
    int status = f->work(f, &in, &out);
    if (in != NULL)
        hb_buffer_close(&in);
    if (status == HB_FILTER_FAILED)
        c->failed = 1;
    if (status == HB_FILTER_DONE)
        c->done[s] = 1;
    while (out != NULL)
    {
        hb_buffer_t *next = out->next;
        out->next = NULL;
        run_from(c, s + 1, out);
        out = next;
    }
}

int hbh_chain_push(hbh_chain_t *c, const uint8_t *const plane[3], const int stride[3],
                   int64_t start, int64_t stop, int flags, int combed)
{
    if (c == NULL || c->eof_seen) return -1;
    hb_buffer_t *b = hb_frame_buffer_init(c->init_in.pix_fmt, c->init_in.geometry.width,
                                          c->init_in.geometry.height);
    if (b == NULL) return -1;
    for (int p = 0; p <= b->f.max_plane; p++)
    {
        const int row = MIN(stride[p], b->plane[p].stride);
        for (int y = 0; y < b->plane[p].height; y++)
            memcpy(b->plane[p].data + (size_t)y * b->plane[p].stride,
                   plane[p] + (size_t)y * stride[p], row);
    }
    b->s.start = start;
    b->s.stop = stop;
    b->s.duration = (double)(stop - start);
    b->s.flags = (uint16_t)flags;
    b->s.combed = (uint8_t)combed;
    b->f.color_prim = c->init_in.color_prim;
    b->f.color_transfer = c->init_in.color_transfer;
    b->f.color_matrix = c->init_in.color_matrix;
    b->f.color_range = c->init_in.color_range;
    b->f.chroma_location = c->init_in.chroma_location;
    if (c->threaded) fifo_put(&c->fifo[0], b);
    else             run_from(c, 0, b);
    return c->failed ? -2 : 0;
}

/* A source that keeps up: `count` frames, cycling through `n_unique` prepared pictures (plane[3 * k + p]), made by
 * `nthreads` filler threads - each allocates an hb_buffer_t and copies a picture into it, what the decoder's threads do in
 * libhb - and pushed IN ORDER by the caller's thread.  hbh_chain_push from Python tops out near 2 k frames / s at 1080p (one
 * thread's memcpy and the interpreter): lists lighter than the bus then measure their source, not the filters. */
typedef struct
{
    hbh_chain_t *c;
    const uint8_t *const *plane;
    const int *stride;
    int n_unique, first, count, flags, ring;
    int64_t duration;
    pthread_mutex_t lock;
    pthread_cond_t  cond;
    int next;                    /* next sequence number to be made */
    int pushed;                  /* sequence numbers below this have been pushed */
    hb_buffer_t **slot;          /* [ring]: slot[i % ring] = frame i once it is made */
    int failed;
} feed_t;

static void *feed_fill(void *arg)
{
    feed_t *f = arg;
    for (;;)
    {
        pthread_mutex_lock(&f->lock);
        const int i = f->next < f->count ? f->next++ : -1;
        while (i >= 0 && i >= f->pushed + f->ring && !f->failed) pthread_cond_wait(&f->cond, &f->lock);   /* its slot is still taken */
        const int stop = i < 0 || f->failed;
        pthread_mutex_unlock(&f->lock);
        if (stop) return NULL;
        hbh_chain_t *c = f->c;
        hb_buffer_t *b = hb_frame_buffer_init(c->init_in.pix_fmt, c->init_in.geometry.width, c->init_in.geometry.height);
        if (b != NULL)
        {
            const int k = (f->first + i) % f->n_unique;
            for (int p = 0; p <= b->f.max_plane; p++)
            {
                const uint8_t *src = f->plane[3 * k + p];
                const int row = MIN(f->stride[p], b->plane[p].stride);
                if (f->stride[p] == b->plane[p].stride)
                    memcpy(b->plane[p].data, src, (size_t)row * b->plane[p].height);
                else
                    for (int y = 0; y < b->plane[p].height; y++)
                        memcpy(b->plane[p].data + (size_t)y * b->plane[p].stride, src + (size_t)y * f->stride[p], row);
            }
            const int64_t n = f->first + i;
            b->s.start = n * f->duration;
            b->s.stop = (n + 1) * f->duration;
            b->s.duration = (double)f->duration;
            b->s.flags = (uint16_t)f->flags;
            b->f.color_prim = c->init_in.color_prim;
            b->f.color_transfer = c->init_in.color_transfer;
            b->f.color_matrix = c->init_in.color_matrix;
            b->f.color_range = c->init_in.color_range;
            b->f.chroma_location = c->init_in.chroma_location;
        }
        pthread_mutex_lock(&f->lock);
        if (b == NULL) f->failed = 1;
        f->slot[i % f->ring] = b;
        pthread_cond_broadcast(&f->cond);
        pthread_mutex_unlock(&f->lock);
    }
}

int hbh_chain_feed(hbh_chain_t *c, const uint8_t *const *plane, const int stride[3], int n_unique, int first, int count,
                   int64_t duration, int flags, int nthreads)
{
    if (c == NULL || c->eof_seen || plane == NULL || n_unique < 1 || count < 0) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 16) nthreads = 16;
    feed_t f;
    memset(&f, 0, sizeof(f));
    f.c = c; f.plane = plane; f.stride = stride; f.n_unique = n_unique; f.first = first; f.count = count;
    f.flags = flags; f.duration = duration; f.ring = 2 * nthreads;
    f.slot = calloc((size_t)f.ring, sizeof(*f.slot));
    if (f.slot == NULL) return -1;
    pthread_mutex_init(&f.lock, NULL);
    pthread_cond_init(&f.cond, NULL);
    pthread_t th[16];
    int started = 0;
    for (; started < nthreads; started++)
        if (pthread_create(&th[started], NULL, feed_fill, &f) != 0) break;
    for (int i = 0; i < count && started > 0; i++)
    {
        pthread_mutex_lock(&f.lock);
        while (f.slot[i % f.ring] == NULL && !f.failed) pthread_cond_wait(&f.cond, &f.lock);
        hb_buffer_t *b = f.slot[i % f.ring];
        f.slot[i % f.ring] = NULL;
        pthread_mutex_unlock(&f.lock);
        if (b == NULL) break;
        if (c->threaded) fifo_put(&c->fifo[0], b);       /* may wait for room: the fillers run ahead meanwhile */
        else             run_from(c, 0, b);
        pthread_mutex_lock(&f.lock);
        f.pushed = i + 1;
        if (c->failed) f.failed = 1;
        pthread_cond_broadcast(&f.cond);
        pthread_mutex_unlock(&f.lock);
        if (c->failed) break;
    }
    pthread_mutex_lock(&f.lock);
    if (f.pushed < count) f.failed = 1;                    /* (an error: let the fillers go) */
    pthread_cond_broadcast(&f.cond);
    pthread_mutex_unlock(&f.lock);
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    for (int r = 0; r < f.ring; r++) if (f.slot[r] != NULL) hb_buffer_close(&f.slot[r]);
    free(f.slot);
    pthread_mutex_destroy(&f.lock);
    pthread_cond_destroy(&f.cond);
    return (started == 0 || f.pushed < count || c->failed) ? -2 : 0;
}

int hbh_chain_push_eof(hbh_chain_t *c)
{
    if (c == NULL || c->eof_seen) return -1;
    c->eof_seen = 1;
    if (c->threaded)
    {
        fifo_put(&c->fifo[0], hb_buffer_eof_init());
        join_threads(c);                         /* every stage has seen EOF and returned: the output is complete */
    }
    else
        run_from(c, 0, hb_buffer_eof_init());
    return c->failed ? -2 : 0;
}

double hbh_chain_stage_busy_ms(hbh_chain_t *c, int stage)
{
    return (c == NULL || stage < 0 || stage >= c->nstages) ? 0.0 : c->busy_ms[stage];
}

int hbh_chain_produced(hbh_chain_t *c)
{
    if (c == NULL || !c->threaded) return 0;
    pthread_mutex_lock(&c->out_lock);
    const int n = c->produced;
    pthread_mutex_unlock(&c->out_lock);
    return n;
}

int hbh_chain_pending(hbh_chain_t *c)
{
    if (c == NULL) return 0;
    if (c->threaded && !c->eof_seen) return 0;      /* stages are still running: collect after hbh_chain_push_eof() */
    return hb_buffer_list_count(&c->out);
}

int hbh_chain_peek(hbh_chain_t *c, hbh_frame_info_t *info)
{
    hb_buffer_t *b = c ? hb_buffer_list_head(&c->out) : NULL;
    if (b == NULL) return -1;
    memset(info, 0, sizeof(*info));
    info->is_eof = !!(b->s.flags & HB_BUF_FLAG_EOF);
    info->start = b->s.start;
    info->stop = b->s.stop;
    info->flags = b->s.flags;
    info->combed = b->s.combed;
    info->width = b->f.width;
    info->height = b->f.height;
    info->fmt = b->f.fmt;
    info->nplanes = b->size ? b->f.max_plane + 1 : 0;
    for (int p = 0; p < info->nplanes; p++)
    {
        info->plane_width[p] = b->plane[p].width;
        info->plane_height[p] = b->plane[p].height;
        info->plane_stride[p] = b->plane[p].stride;
    }
    return 0;
}

int hbh_chain_pop(hbh_chain_t *c, uint8_t *const plane[3], const int stride[3])
{
    hb_buffer_t *b = c ? hb_buffer_list_rem_head(&c->out) : NULL;
    if (b == NULL) return -1;
    if (b->size && plane != NULL)
    {
        for (int p = 0; p <= b->f.max_plane; p++)
        {
            if (plane[p] == NULL) continue;
            const int row = MIN(stride[p], b->plane[p].stride);
            for (int y = 0; y < b->plane[p].height; y++)
                memcpy(plane[p] + (size_t)y * stride[p],
                       b->plane[p].data + (size_t)y * b->plane[p].stride, row);
        }
    }
    hb_buffer_close(&b);
    return 0;
}

void hbh_chain_output_geometry(hbh_chain_t *c, int *width, int *height, int *vrate_num, int *vrate_den)
{
    if (width)     *width = c->init.geometry.width;
    if (height)    *height = c->init.geometry.height;
    if (vrate_num) *vrate_num = c->init.vrate.num;
    if (vrate_den) *vrate_den = c->init.vrate.den;
}

void hbh_chain_close(hbh_chain_t *c)
{
    if (c == NULL) return;
    if (c->threaded && !c->eof_seen)
    {
        fifo_put(&c->fifo[0], hb_buffer_eof_init());
        join_threads(c);
    }
    for (int i = 0; i < c->nstages; i++)
    {
        hb_filter_object_t *f = c->stage[i];
        if (f->close) f->close(f);
        hb_dict_free(&f->settings);
        free(f);
    }
    hb_buffer_list_close(&c->out);
    if (c->subtitle != NULL)
    {
        hb_fifo_close(&c->subtitle->fifo_out);
        free(c->subtitle);
    }
    if (c->job != NULL)
    {
        if (g_hip_job_close != NULL) g_hip_job_close(c->job);      /* do_job's clean-up (hip_common.h) */
        hb_list_close(&c->job->list_subtitle);
        hb_list_close(&c->job->list_attachment);
    }
    free(c->job);
    free(c);
}

/* A decoded bitmap subtitle (what decpgssub / decvobsub hand on): a YUVA 4:4:4 picture at (x, y) of a window_w x window_h
 * canvas, shown from start to stop (90 kHz; stop < 0: until the next one).  Goes into the burn-in track's fifo_out, where
 * rendersub picks it up with the next frame. */
int hbh_chain_push_subtitle(hbh_chain_t *c, const hbh_overlay_t *ov, int64_t start, int64_t stop, int window_w, int window_h)
{
    if (c == NULL || c->subtitle == NULL || ov == NULL) return -1;
    hb_buffer_t *o = hb_frame_buffer_init(AV_PIX_FMT_YUVA444P, ov->width, ov->height);
    if (o == NULL) return -1;
    for (int p = 0; p <= o->f.max_plane; p++)
        for (int y = 0; y < o->plane[p].height; y++)
            memcpy(o->plane[p].data + (size_t)y * o->plane[p].stride, ov->plane[p] + (size_t)y * ov->stride[p], o->plane[p].width);
    o->f.x = ov->x;
    o->f.y = ov->y;
    o->f.window_width = window_w;
    o->f.window_height = window_h;
    o->s.start = start;
    o->s.stop = stop < 0 ? AV_NOPTS_VALUE : stop;
    hb_fifo_push(c->subtitle->fifo_out, o);
    return 0;
}

/* ---- compositor objects (hb_blend_object_t, handbrake/common.h:1813-1828) ------------------
 * Plays rendersub.c's part (:1129-1161, :467): copy the prototype, init, hand `work` the frame and
 * the list of rendered overlays.  `passes` > 1 repeats work() on a fresh copy of the frame with
 * changed = 0 (the overlays of the previous call are still valid), as rendersub does between
 * subtitle changes; the frame written back is the last pass's. */
int hbh_blend_run(const void *proto, int pix_fmt, int width, int height, int chroma_location, int overlay_fmt,
                  uint8_t *const plane[3], const int stride[3], int n_overlays, const hbh_overlay_t *ov, int passes)
{
    hb_blend_object_t blend = *(const hb_blend_object_t *)proto;
    if (blend.init(&blend, width, height, pix_fmt, chroma_location, 1, overlay_fmt) != 0) return -1;
    hb_buffer_list_t list;
    memset(&list, 0, sizeof(list));
    for (int i = 0; i < n_overlays; i++)
    {
        hb_buffer_t *o = hb_frame_buffer_init(overlay_fmt, ov[i].width, ov[i].height);
        if (o == NULL) return -1;
        for (int p = 0; p <= o->f.max_plane; p++)
            for (int y = 0; y < o->plane[p].height; y++)
                memcpy(o->plane[p].data + (size_t)y * o->plane[p].stride, ov[i].plane[p] + (size_t)y * ov[i].stride[p],
                       o->plane[p].width);
        o->f.x = ov[i].x;
        o->f.y = ov[i].y;
        hb_buffer_list_append(&list, o);
    }
    int rc = 0;
    for (int pass = 0; pass < passes && rc == 0; pass++)
    {
        hb_buffer_t *b = hb_frame_buffer_init(pix_fmt, width, height);
        if (b == NULL) { rc = -1; break; }
        for (int p = 0; p <= b->f.max_plane; p++)
            for (int y = 0; y < b->plane[p].height; y++)
                memcpy(b->plane[p].data + (size_t)y * b->plane[p].stride, plane[p] + (size_t)y * stride[p],
                       MIN(stride[p], b->plane[p].stride));
        hb_buffer_t *out = blend.work(&blend, b, &list, pass == 0);
        if (out == NULL) { rc = -2; break; }
        if (pass == passes - 1)
            for (int p = 0; p <= out->f.max_plane; p++)
                for (int y = 0; y < out->plane[p].height; y++)
                    memcpy(plane[p] + (size_t)y * stride[p], out->plane[p].data + (size_t)y * out->plane[p].stride,
                           MIN(stride[p], out->plane[p].stride));
        hb_buffer_close(&out);
    }
    hb_buffer_list_close(&list);
    blend.close(&blend);
    return rc;
}

/* ---- frame-difference metric objects (hb_motion_metric_object_t) ---------------------------
 * vfr.c's part (:76-108, :380): copy the prototype, init with the stream's hb_filter_init_t, then
 * work(previous frame, current frame).  Only luma is looked at. */
int hbh_motion_metric_run(const void *proto, int pix_fmt, int width, int height,
                          const uint8_t *luma_a, int stride_a, const uint8_t *luma_b, int stride_b, float *out)
{
    hb_motion_metric_object_t metric = *(const hb_motion_metric_object_t *)proto;
    hb_filter_init_t init;
    memset(&init, 0, sizeof(init));
    init.pix_fmt = pix_fmt;
    init.hw_pix_fmt = AV_PIX_FMT_NONE;
    init.geometry.width = width;
    init.geometry.height = height;
    if (metric.init(&metric, &init) != 0) return -1;
    hb_buffer_t *a = hb_frame_buffer_init(pix_fmt, width, height), *b = hb_frame_buffer_init(pix_fmt, width, height);
    if (a == NULL || b == NULL) return -1;
    for (int y = 0; y < height; y++)
    {
        memcpy(a->plane[0].data + (size_t)y * a->plane[0].stride, luma_a + (size_t)y * stride_a, MIN(stride_a, a->plane[0].stride));
        memcpy(b->plane[0].data + (size_t)y * b->plane[0].stride, luma_b + (size_t)y * stride_b, MIN(stride_b, b->plane[0].stride));
    }
    *out = metric.work(&metric, a, b);
    hb_buffer_close(&a);
    hb_buffer_close(&b);
    metric.close(&metric);
    return 0;
}
