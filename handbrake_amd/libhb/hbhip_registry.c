/* hbhip_registry.c — registry of the HIP drop-in filter objects and the shared
 * device context they run on.
 *
 * hbhip_filter_get() is the counterpart of hb_filter_get() (common.c:5331-5495)
 * for the HIP-backed objects: same ids, so a job's filter list does not change.
 * Inside libhb the hook is the one the reference already has for its own GPU
 * filters: sanitize_filter_list_post() swaps CPU objects for GPU ones in place
 * (work.c:1515-1523, platform/macosx/vt_common.c:486-540); INTEGRATION.md shows
 * the equivalent hb_hip_setup_hw_filters().
 */
#include "hbhip_host.h"
#include "hip_common.h"

#include <pthread.h>

/* Contexts (a GPU, a HIP stream, the pools behind them) are created on first use and kept for the life of the process
 * (HIP tears them down at exit).  The filters of ONE job share one: they share a stream, which is what lets adjacent HIP
 * filters hand frames over in HBM without extra synchronisation.  Two jobs that are live at the same time on one GPU (a
 * GUI's queue beside a preview encode: one hb_handle_t each) get a context EACH - up to HBHIP_CTX_SLOTS per GPU -, so their
 * kernels sit on different streams instead of queueing behind each other: a job leases a slot at its first
 * hbhip_host_ctx_for() and hb_hip_job_close() gives it back, so that jobs run one after the other keep reusing slot 0
 * and its pools.  A job that is never closed costs nothing but its slot: once all are leased, later jobs share them
 * (by job address, so that all filters of a job still agree).
 * WHICH GPU is the job's business: hb_job_t carries hw_device_index (common.h:991; the QSV / NVENC paths read it the same
 * way, hwaccel.c:230-257, qsv_common.c:2238-2244), so a process that runs several jobs puts stream k on GPU k.  A job
 * that does not say (index < 0) - or a caller without an init, such as rendersub's compositor object - gets the process
 * default: HBHIP_DEVICE from the environment, else 0, and a caller without a job slot 0 of it. */
#define HBHIP_MAX_DEVICES 64
#define HBHIP_CTX_SLOTS   4
typedef struct { hbhip_ctx *ctx; const hb_job_t *job; int failed; hbhip_ctx *aux; int aux_failed; } ctx_slot_t;
static pthread_mutex_t g_ctx_lock = PTHREAD_MUTEX_INITIALIZER;
static ctx_slot_t      g_slot[HBHIP_MAX_DEVICES][HBHIP_CTX_SLOTS];

int hbhip_host_default_device(void)
{
    const char *env = getenv("HBHIP_DEVICE");
    const int device = env != NULL ? atoi(env) : 0;
    return device < 0 ? 0 : device;
}

/* hw_device_index is an ADAPTER index owned by whichever hardware path the job uses: QSV sets it to a DX11 / VA adapter
 * (qsv_common.c:2238-2244, also when only the encoder is QSV and hw_pix_fmt stays NONE), NVDEC / NVENC to a CUDA ordinal
 * (hwaccel.c:230-257).  Only when no other vendor's decoder or encoder is in the job - software codecs, or AMD's own
 * AMF decoder / VCE encoders, whose adapter is this GPU - does the number name a HIP device. */
int hbhip_host_job_index_is_hip(const hb_job_t *job)
{
    if (job == NULL || job->hw_device_index < 0) return 0;
    if (job->hw_decode & (HB_DECODE_QSV | HB_DECODE_NVDEC | HB_DECODE_VIDEOTOOLBOX | HB_DECODE_MF)) return 0;
    if (job->vcodec & (HB_VCODEC_QSV_MASK | HB_VCODEC_VT_MASK)) return 0;
    if (job->vcodec & HB_VCODEC_FFMPEG_MASK)
    {
        const int n = job->vcodec & 0xff;                     /* common.h:747-756: 0x20-0x22 Media Foundation, 0x30-0x35 NVENC */
        if ((n >= 0x20 && n <= 0x22) || (n >= 0x30 && n <= 0x35)) return 0;
    }
    return 1;
}

int hbhip_host_device_for(const hb_filter_init_t *init)
{
    if (init != NULL && hbhip_host_job_index_is_hip(init->job))
        return init->job->hw_device_index;
    return hbhip_host_default_device();
}

/* g_ctx_lock held */
static hbhip_ctx *slot_ctx(int device, int k)
{
    ctx_slot_t *s = &g_slot[device][k];
    if (s->ctx == NULL && !s->failed)
    {
        int rc = hbhip_ctx_create(device, &s->ctx);
        if (rc != HBHIP_OK)
        {
            hb_error("hbhip: cannot create device context on GPU %d: %s", device, hbhip_strerror(rc));
            s->ctx = NULL;
            s->failed = hbhip_device_count() > 0;      /* a GPU that is not there stays not there; no GPU at all may change (tests) */
        }
        else
        {
            char name[256];
            hbhip_ctx_device_name(s->ctx, name, sizeof(name));
            hb_log("hbhip: using GPU %d: %s%s", device, name, k ? " (a further job's stream)" : "");
        }
    }
    return s->ctx;
}

hbhip_ctx *hbhip_host_ctx_on(int device)
{
    if (device < 0 || device >= HBHIP_MAX_DEVICES) return NULL;
    pthread_mutex_lock(&g_ctx_lock);
    hbhip_ctx *c = slot_ctx(device, 0);
    pthread_mutex_unlock(&g_ctx_lock);
    return c;
}

hbhip_ctx *hbhip_host_ctx_for(const hb_filter_init_t *init)
{
    const int device = hbhip_host_device_for(init);
    const hb_job_t *job = init != NULL ? init->job : NULL;
    if (job == NULL) return hbhip_host_ctx_on(device);
    if (device < 0 || device >= HBHIP_MAX_DEVICES) return NULL;
    pthread_mutex_lock(&g_ctx_lock);
    int k = -1;
    for (int i = 0; i < HBHIP_CTX_SLOTS && k < 0; i++)
        if (g_slot[device][i].job == job) k = i;                              /* the job's lease */
    for (int i = 0; i < HBHIP_CTX_SLOTS && k < 0; i++)
        if (g_slot[device][i].job == NULL) { g_slot[device][i].job = job; k = i; }   /* the lowest free slot */
    if (k < 0) k = (int)(((uintptr_t)job >> 6) % HBHIP_CTX_SLOTS);                /* all leased: share, the same one for every filter of the job */
    hbhip_ctx *c = slot_ctx(device, k);
    if (c == NULL && k > 0) c = slot_ctx(device, 0);                          /* (no further stream to be had: the shared one) */
    pthread_mutex_unlock(&g_ctx_lock);
    return c;
}

/* A SECOND stream inside a job (the default; HBHIP_JOB_STREAMS=1 turns it off): the deinterlacing side of a job's filter
 * list - comb detect, decomb, yadif, bwdif: ids 4 - 9, always at the head of a device-resident run - gets a context of its
 * own beside the job's, so that EEDI2 of the next frames runs beside NLMeans / scaler / sharpen of the current ones the way
 * bench.py's device-resident line runs them (--stage-streams 2).  Frames then cross contexts: every consumer of a device
 * frame orders its stream behind the frame's producer (hbhip_frame_use_on) - behind the producer only, not behind what
 * the other stream has queued since - and the frame goes idle behind its last reader's stream.  Measured through the
 * plugin surface (python -m handbrake_amd.hostpath): a list that ends in a 2160p download is bound by the bus
 * either way (3 900 fps); the same list at 1080p out runs at 5 957 fps on one stream and 7 111 on two. */
#define HBHIP_ROLE_MAIN        0
#define HBHIP_ROLE_DEINTERLACE 1
static int job_streams(void)
{
    const char *e = getenv("HBHIP_JOB_STREAMS");
    return e != NULL && atoi(e) == 1 ? 1 : 2;
}

hbhip_ctx *hbhip_host_ctx_for_role(const hb_filter_init_t *init, int role)
{
    hbhip_ctx *main_ctx = hbhip_host_ctx_for(init);
    if (main_ctx == NULL || role != HBHIP_ROLE_DEINTERLACE || job_streams() < 2) return main_ctx;
    hbhip_ctx *aux = NULL;
    pthread_mutex_lock(&g_ctx_lock);
    for (int d = 0; d < HBHIP_MAX_DEVICES && aux == NULL; d++)
        for (int i = 0; i < HBHIP_CTX_SLOTS; i++)
        {
            ctx_slot_t *sl = &g_slot[d][i];
            if (sl->ctx != main_ctx) continue;
            if (sl->aux == NULL && !sl->aux_failed && hbhip_ctx_create(d, &sl->aux) != HBHIP_OK)
            {
                sl->aux = NULL;
                sl->aux_failed = 1;
            }
            aux = sl->aux;
            break;
        }
    pthread_mutex_unlock(&g_ctx_lock);
    return aux != NULL ? aux : main_ctx;
}

/* a filter on `ctx` is about to queue work that reads `in`: see above (nothing to do for host frames / one stream) */
int hbhip_host_use_frame(hbhip_ctx *ctx, const hb_buffer_t *in)
{
    hbhip_frame *fr = in != NULL ? hbhip_host_frame_of(in) : NULL;
    if (fr == NULL || ctx == NULL) return HBHIP_OK;
    return hbhip_frame_use_on(fr, ctx);
}

/* the job's lease, if it has one (tests) */
hbhip_ctx *hbhip_host_job_ctx(const hb_job_t *job)
{
    hbhip_ctx *c = NULL;
    pthread_mutex_lock(&g_ctx_lock);
    for (int d = 0; d < HBHIP_MAX_DEVICES && c == NULL && job != NULL; d++)
        for (int i = 0; i < HBHIP_CTX_SLOTS; i++)
            if (g_slot[d][i].job == job) { c = g_slot[d][i].ctx; break; }
    pthread_mutex_unlock(&g_ctx_lock);
    return c;
}

/* work.c, when a job's filters have been closed: the job's stream goes back to the pool (hip_common.h) */
void hb_hip_job_close(hb_job_t *job)
{
    if (job == NULL) return;
    pthread_mutex_lock(&g_ctx_lock);
    for (int d = 0; d < HBHIP_MAX_DEVICES; d++)
        for (int i = 0; i < HBHIP_CTX_SLOTS; i++)
            if (g_slot[d][i].job == job) g_slot[d][i].job = NULL;
    pthread_mutex_unlock(&g_ctx_lock);
}

hbhip_ctx *hbhip_host_ctx(void)                             { return hbhip_host_ctx_on(hbhip_host_default_device()); }
void       hbhip_host_ctx_release(void)                     { /* contexts live as long as the process */ }

hb_filter_object_t *hbhip_filter_get(int filter_id)
{
    switch (filter_id)
    {
        case HB_FILTER_NLMEANS:       return &hb_filter_nlmeans_hip;
        case HB_FILTER_LAPSHARP:      return &hb_filter_lapsharp_hip;
        case HB_FILTER_UNSHARP:       return &hb_filter_unsharp_hip;
        case HB_FILTER_CHROMA_SMOOTH: return &hb_filter_chroma_smooth_hip;
        case HB_FILTER_DENOISE:       return &hb_filter_denoise_hip;
        case HB_FILTER_CROP_SCALE:    return &hb_filter_crop_scale_hip;
        case HB_FILTER_GRAYSCALE:     return &hb_filter_grayscale_hip;
        case HB_FILTER_ROTATE:        return &hb_filter_rotate_hip;
        case HB_FILTER_COLORSPACE:    return &hb_filter_colorspace_hip;
        case HB_FILTER_PAD:           return &hb_filter_pad_hip;
        case HB_FILTER_YADIF:         return &hb_filter_yadif_hip;
        case HB_FILTER_BWDIF:         return &hb_filter_bwdif_hip;
        case HB_FILTER_FORMAT:        return &hb_filter_format_hip;
        case HB_FILTER_DECOMB:        return &hb_filter_decomb_hip;
        case HB_FILTER_COMB_DETECT:   return &hb_filter_comb_detect_hip;
        case HB_FILTER_HIP_UPLOAD:    return &hb_filter_hip_upload;
        case HB_FILTER_HIP_DOWNLOAD:  return &hb_filter_hip_download;
        default:                      return NULL;
    }
}

/* Address of the shared context, for the bench/test bindings. */
void *hbhip_host_ctx_ptr(void)
{
    return hbhip_host_ctx();
}

/* ---- device-resident buffers ----------------------------------------------------------- */
static void storage_retain(void *p)  { hbhip_frame_retain((hbhip_frame *)p); }
static void storage_release(void *p) { hbhip_frame_release((hbhip_frame *)p); }

/* ---- page-locked frame buffers ------------------------------------------------------------
 * hb_buffer payloads of frame size come from a recycling pool of hipHostMalloc'd blocks, so the
 * uploads / downloads of the filters are DMA transfers.  (hipHostMalloc itself is far too slow to
 * call per frame.)  Blocks are kept per exact size; at most PIN_KEEP idle blocks are retained.
 * HBHIP_PINNED=0 turns the pool off. */
#define PIN_KEEP 256
static pthread_mutex_t g_pin_lock = PTHREAD_MUTEX_INITIALIZER;
static struct { void *p; size_t size; } g_pin_idle[PIN_KEEP];
static int g_pin_count = 0;

static void *pinned_alloc(size_t size)
{
    void *p = NULL;
    pthread_mutex_lock(&g_pin_lock);
    for (int i = 0; i < g_pin_count; i++)
        if (g_pin_idle[i].size == size)
        {
            p = g_pin_idle[i].p;
            g_pin_idle[i] = g_pin_idle[--g_pin_count];
            break;
        }
    pthread_mutex_unlock(&g_pin_lock);
    if (p != NULL) return p;
    if (hbhip_device_count() <= 0) return NULL;               /* no GPU: plain malloc in the caller */
    if (hbhip_host_alloc(size, &p) != HBHIP_OK) return NULL;
    return p;
}

static void pinned_release(void *p, size_t size)
{
    pthread_mutex_lock(&g_pin_lock);
    if (g_pin_count < PIN_KEEP)
    {
        g_pin_idle[g_pin_count].p = p;
        g_pin_idle[g_pin_count].size = size;
        g_pin_count++;
        p = NULL;
    }
    pthread_mutex_unlock(&g_pin_lock);
    if (p != NULL) hbhip_host_free(p);
}

__attribute__((constructor)) static void hbhip_host_register_hooks(void)
{
#ifndef HBHIP_IN_LIBHB
    hbhip_rt_set_storage_hooks(storage_retain, storage_release);
    hbhip_rt_set_job_hooks(hb_hip_setup_hw_filters, hb_hip_filter_init_failed, hb_hip_job_close);   /* inside libhb work.c calls them itself */
    /* inside libhb: `case AV_PIX_FMT_HBHIP:` in vfr.c:76-108 and rendersub.c:1129-1161 */
    hbhip_rt_register_hw_helper(0, AV_PIX_FMT_HBHIP, &hb_motion_metric_hip);
    hbhip_rt_register_hw_helper(1, AV_PIX_FMT_HBHIP, &hb_blend_hip);
    const char *e = getenv("HBHIP_PINNED");
    if (e == NULL || atoi(e) != 0)
        hbhip_rt_set_alloc_hooks(pinned_alloc, pinned_release);
#endif
}

hb_buffer_t *hbhip_host_wrap_frame(hbhip_frame *fr, const hb_filter_init_t *o, int width, int height)
{
    hb_buffer_t *b = hb_buffer_init(0);
    if (b == NULL)
    {
        hbhip_frame_release(fr);
        return NULL;
    }
    const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(o->pix_fmt);
    b->s.type = FRAME_BUF;
    b->f.fmt = o->pix_fmt;
    b->f.width = width;
    b->f.height = height;
    b->f.max_plane = 2;
    b->f.color_prim = o->color_prim;
    b->f.color_transfer = o->color_transfer;
    b->f.color_matrix = o->color_matrix;
    b->f.color_range = o->color_range;
    b->f.chroma_location = o->chroma_location;
    hbhip_dev_frame d;
    hbhip_frame_describe(fr, &d, NULL, NULL);
    for (int p = 0; p < 3; p++)
    {
        b->plane[p].data = NULL;                       /* not host-addressable */
        b->plane[p].stride = d.stride[p];
        b->plane[p].width = hb_image_width(o->pix_fmt, width, p);
        b->plane[p].height = hb_image_height(o->pix_fmt, height, p);
        b->plane[p].size = 0;
    }
    (void)desc;
    b->storage = fr;
    b->storage_type = HBHIP_DEVICE;
    return b;
}

int hbhip_host_push(hbhip_filter *dev, const hb_buffer_t *in, int64_t tag)
{
    hbhip_frame *fr = hbhip_host_frame_of(in);
    if (fr != NULL)
        return hbhip_filter_push_frame(dev, fr, tag);      /* the frame itself where the filter works on frames, else a copy of it */
    hbhip_host_frame hf;
    hbhip_host_frame_from_buf(&hf, in);
    return hbhip_filter_push(dev, &hf, tag);
}

hb_buffer_t *hbhip_host_pull(hbhip_filter *dev, const hb_filter_init_t *o, int width, int height,
                             int dev_io, int64_t *tag)
{
    int64_t t = 0;
    if (dev_io)
    {
        hbhip_frame *fr = NULL;
        if (hbhip_filter_pull_frame(dev, &fr, &t) != HBHIP_OK || fr == NULL)     /* the filter's own picture, or a copy of it */
            return NULL;
        hbhip_frame_mark_ready(fr);                    /* what fills it is queued: a download waits for this point only */
        if (tag) *tag = t;
        return hbhip_host_wrap_frame(fr, o, width, height);
    }
    hb_buffer_t *out = hbhip_host_alloc_out(o, width, height);
    if (out == NULL) return NULL;
    hbhip_host_frame hf;
    hbhip_host_frame_from_buf(&hf, out);
    if (hbhip_filter_pull(dev, &hf, &t) != HBHIP_OK)
    {
        hb_buffer_close(&out);
        return NULL;
    }
    if (tag) *tag = t;
    return out;
}

/* host frames: two submissions kept in flight (hbhip_filter_submit_async), so that the upload of this frame, the
 * kernels of the previous one and the download of the one before overlap.  The price is the reference's own: the
 * filter answers HB_FILTER_DELAY until the pipe is full (nlmeans.c:548 does the same for `threads` frames). */
#define HBHIP_PIPE_DEPTH 2
typedef struct { hb_buffer_t *in, *out; } pipe_pair_t;

static hb_buffer_t *pipe_collect(hbhip_filter *dev)
{
    int64_t tag = 0;
    if (hbhip_filter_wait(dev, &tag) != HBHIP_OK) return NULL;
    pipe_pair_t *pp = (pipe_pair_t *)(intptr_t)tag;
    hb_buffer_t *out = pp->out;
    hb_buffer_copy_props(out, pp->in);
    hb_buffer_close(&pp->in);
    free(pp);
    return out;
}

/* device frames: a stateless filter inside a device-resident run gathers a burst of frames and makes them in one launch
 * per burst (hbhip_filter_process_dev with n frames), the way decomb_hip and nlmeans_hip gather theirs: a launch per frame
 * costs such a filter 8-20 us of a 1080p frame where the batched launch takes 2-4 (profiles/r6X_kernel_rooflines.json).  It
 * answers HB_FILTER_DELAY while it gathers and emits the burst as a buffer list - what the filters in front of it do with
 * their bursts of eight anyway.  HBHIP_STATELESS_BATCH: frames per burst (default 8, 1 = a launch per frame). */
#define SB_MAX 16
typedef struct hold_s { struct hold_s *next; hbhip_filter *dev; int n; hb_buffer_t *in[SB_MAX]; } hold_t;
static hold_t *g_holds;
static pthread_mutex_t g_hold_lock = PTHREAD_MUTEX_INITIALIZER;

static int burst_frames(void)
{
    const char *e = getenv("HBHIP_STATELESS_BATCH");       /* (read per call: the tests switch it inside a process) */
    const int v = e != NULL ? atoi(e) : 8;
    return v < 1 ? 1 : (v > SB_MAX ? SB_MAX : v);
}

static hold_t *hold_of(hbhip_filter *dev, int create)
{
    pthread_mutex_lock(&g_hold_lock);
    hold_t *h = g_holds;
    while (h != NULL && h->dev != dev) h = h->next;
    if (h == NULL && create && (h = calloc(1, sizeof(*h))) != NULL)
    {
        h->dev = dev;
        h->next = g_holds;
        g_holds = h;
    }
    pthread_mutex_unlock(&g_hold_lock);
    return h;
}

static void hold_drop(hbhip_filter *dev)
{
    pthread_mutex_lock(&g_hold_lock);
    for (hold_t **pp = &g_holds; *pp != NULL; pp = &(*pp)->next)
        if ((*pp)->dev == dev)
        {
            hold_t *h = *pp;
            *pp = h->next;
            for (int i = 0; i < h->n; i++) hb_buffer_close(&h->in[i]);
            free(h);
            break;
        }
    pthread_mutex_unlock(&g_hold_lock);
}

/* the gathered frames through the filter in one call; the results as a buffer list appended to `list` */
static int burst_run(hbhip_filter *dev, const hb_filter_init_t *output, const char *who, hold_t *h, hb_buffer_list_t *list)
{
    const int n = h->n;
    if (n == 0) return HBHIP_OK;
    int ow = h->in[0]->f.width, oh = h->in[0]->f.height;
    hbhip_filter_out_geometry(dev, &ow, &oh);
    const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(output->pix_fmt);
    hbhip_frame *dst[SB_MAX] = { NULL };
    hbhip_dev_frame di[SB_MAX], dd[SB_MAX];
    int rc = desc == NULL ? HBHIP_ERR_ARG : HBHIP_OK, made = 0;
    for (int i = 0; rc == HBHIP_OK && i < n; i++)
    {
        hbhip_frame *src = hbhip_host_frame_of(h->in[i]);
        rc = src == NULL ? HBHIP_ERR_ARG
                         : hbhip_frame_alloc(hbhip_filter_context(dev), ow, oh, desc->comp[0].depth,
                                             desc->log2_chroma_w, desc->log2_chroma_h, &dst[i]);
        if (rc != HBHIP_OK) break;
        hbhip_frame_describe(src, &di[i], NULL, NULL);
        hbhip_frame_describe(dst[i], &dd[i], NULL, NULL);
        rc = hbhip_frame_use_on(src, hbhip_filter_context(dev));
    }
    if (rc == HBHIP_OK) rc = hbhip_filter_process_dev(dev, di, n, 0, dd, n, &made);
    if (rc == HBHIP_OK && made != n) rc = HBHIP_ERR_ARG;
    if (rc != HBHIP_OK)
    {
        for (int i = 0; i < n; i++) if (dst[i] != NULL) hbhip_frame_release(dst[i]);
        hb_error("%s(hip): burst of %d frames failed (%s)", who, n, hbhip_strerror(rc));
        return rc;
    }
    for (int i = 0; i < n; i++)
    {
        hbhip_frame_mark_ready(dst[i]);
        hb_buffer_t *out = hbhip_host_wrap_frame(dst[i], output, ow, oh);      /* takes the reference */
        if (out == NULL)
        {
            for (int k = i; k < n; k++) hbhip_frame_release(dst[k]);
            for (int k = i; k < n; k++) hb_buffer_close(&h->in[k]);
            h->n = 0;
            return HBHIP_ERR_NOMEM;
        }
        hb_buffer_copy_props(out, h->in[i]);
        hb_buffer_close(&h->in[i]);
        hb_buffer_list_append(list, out);
    }
    h->n = 0;
    return HBHIP_OK;
}

/* close(): frames still in the pipe (a cancelled job) are dropped with their buffers */
void hbhip_host_simple_destroy(hbhip_filter *dev)
{
    if (dev == NULL) return;
    hold_drop(dev);
    while (hbhip_filter_inflight(dev) > 0)
    {
        hb_buffer_t *o = pipe_collect(dev);
        if (o == NULL) break;
        hb_buffer_close(&o);
    }
    hbhip_filter_destroy(dev);
}

int hbhip_host_simple_work(hbhip_filter *dev, const hb_filter_init_t *output, const char *who,
                           int dev_io, hb_buffer_t **buf_in, hb_buffer_t **buf_out)
{
    hb_buffer_t *in = *buf_in;
    if (in->s.flags & HB_BUF_FLAG_EOF)
    {
        hb_buffer_list_t list;
        hb_buffer_list_clear(&list);
        hold_t *held = hold_of(dev, 0);
        if (held != NULL && burst_run(dev, output, who, held, &list) != HBHIP_OK)       /* the frames gathered so far */
        {
            hb_buffer_list_close(&list);
            return HB_FILTER_FAILED;
        }
        while (hbhip_filter_inflight(dev) > 0)                 /* drain the pipe in order */
        {
            hb_buffer_t *o = pipe_collect(dev);
            if (o == NULL) { hb_buffer_list_close(&list); return HB_FILTER_FAILED; }
            hb_buffer_list_append(&list, o);
        }
        hb_buffer_list_append(&list, in);
        *buf_out = hb_buffer_list_clear(&list);
        *buf_in = NULL;
        return HB_FILTER_DONE;
    }
    int ow = in->f.width, oh = in->f.height;
    hbhip_filter_out_geometry(dev, &ow, &oh);
    int rc;
    hb_buffer_t *out = NULL;
    hbhip_frame *src = hbhip_host_frame_of(in);
    if (src != NULL && dev_io && burst_frames() > 1)
    {
        /* device frames in, device frames out, a burst per launch */
        hold_t *h = hold_of(dev, 1);
        if (h == NULL) { hb_error("%s(hip): out of memory", who); return HB_FILTER_FAILED; }
        h->in[h->n++] = in;
        *buf_in = NULL;
        if (h->n < burst_frames()) return HB_FILTER_DELAY;
        hb_buffer_list_t list;
        hb_buffer_list_clear(&list);
        if (burst_run(dev, output, who, h, &list) != HBHIP_OK)
        {
            hb_buffer_list_close(&list);
            return HB_FILTER_FAILED;
        }
        *buf_out = hb_buffer_list_clear(&list);
        return HB_FILTER_OK;
    }
    if (src != NULL && dev_io)
    {
        /* device frame in, device frame out: the filter reads and writes the two frames in place */
        const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(output->pix_fmt);
        hbhip_frame *dst = NULL;
        rc = desc == NULL ? HBHIP_ERR_ARG
                          : hbhip_frame_alloc(hbhip_filter_context(dev), ow, oh, desc->comp[0].depth,
                                              desc->log2_chroma_w, desc->log2_chroma_h, &dst);
        if (rc == HBHIP_OK)
        {
            hbhip_dev_frame di, dd;
            int n = 0;
            hbhip_frame_describe(src, &di, NULL, NULL);
            hbhip_frame_describe(dst, &dd, NULL, NULL);
            rc = hbhip_frame_use_on(src, hbhip_filter_context(dev));
            if (rc == HBHIP_OK) rc = hbhip_filter_process_dev(dev, &di, 1, 0, &dd, 1, &n);
            if (rc == HBHIP_OK && n == 1)
            {
                hbhip_frame_mark_ready(dst);
                out = hbhip_host_wrap_frame(dst, output, ow, oh);      /* takes the reference */
            }
            else
                hbhip_frame_release(dst);
        }
    }
    else if (src == NULL && !dev_io)
    {
        /* host frame in, host frame out: pipelined */
        out = hbhip_host_alloc_out(output, ow, oh);
        pipe_pair_t *pp = out != NULL ? malloc(sizeof(*pp)) : NULL;
        if (pp == NULL)
        {
            hb_buffer_close(&out);
            hb_error("%s(hip): out of memory", who);
            return HB_FILTER_FAILED;
        }
        pp->in = in;
        pp->out = out;
        hbhip_host_frame hi, ho;
        hbhip_host_frame_from_buf(&hi, in);
        hbhip_host_frame_from_buf(&ho, out);
        rc = hbhip_filter_submit_async(dev, &hi, &ho, (int64_t)(intptr_t)pp);
        if (rc == HBHIP_OK)
        {
            *buf_in = NULL;                                         /* ours until its submission has finished */
            if (hbhip_filter_inflight(dev) <= HBHIP_PIPE_DEPTH)
                return HB_FILTER_DELAY;
            out = pipe_collect(dev);
            if (out == NULL) { hb_error("%s(hip): wait failed", who); return HB_FILTER_FAILED; }
            *buf_out = out;
            return HB_FILTER_OK;
        }
        free(pp);
        hb_buffer_close(&out);
        if (rc != HBHIP_ERR_UNSUPPORTED)
        {
            hb_error("%s(hip): submit failed (%s)", who, hbhip_strerror(rc));
            return HB_FILTER_FAILED;
        }
        rc = hbhip_host_push(dev, in, 0);                           /* not a one-in / one-out device filter */
        out = rc == HBHIP_OK ? hbhip_host_pull(dev, output, ow, oh, dev_io, NULL) : NULL;
    }
    else
    {
        rc = hbhip_host_push(dev, in, 0);
        out = rc == HBHIP_OK ? hbhip_host_pull(dev, output, ow, oh, dev_io, NULL) : NULL;
    }
    if (out == NULL)
    {
        hb_error("%s(hip): push/pull failed (%s)", who, hbhip_strerror(rc));
        return HB_FILTER_FAILED;
    }
    hb_buffer_copy_props(out, in);
    *buf_out = out;
    return HB_FILTER_OK;
}

/* ---- adapters: host <-> device at the ends of a run of HIP filters ----------------------
 * (the reference's analogue is HB_FILTER_ADAPTER_VT, platform/macosx/adapter_vt.c) */
/* The download adapter keeps DL_DEPTH copies in flight on the context's download stream: the D2H of frame n runs
 * while the kernels of the frames behind it do, and the filter thread waits for the oldest copy only - the frames in
 * flight of the reference's own threaded filters (nlmeans.c:464-597, mt_frame_filter.c:45-237); like them it answers
 * HB_FILTER_DELAY until its pipe is full. */
#define DL_DEPTH 6
typedef struct { hb_buffer_t *in, *out; void *token; } dl_slot_t;
/* The upload adapter keeps UL_DEPTH copies in flight on the context's upload stream: the device frame goes downstream at
 * once - its readers wait for the copy, hbhip_frame_use_on - while the host buffer stays with the adapter until the copy
 * has finished (a synchronous copy per frame held the thread for a bus round trip each: 0.27 ms of a 0.32 ms frame
 * period on a list that the GPU, not the download, bounds; with four in flight the thread was 90 % busy waiting for the
 * oldest at 7.7 k output fps, with eight 11 % at 8.2 k). */
#ifndef UL_DEPTH
#define UL_DEPTH 8
#endif
typedef struct { hb_buffer_t *in; void *token; } ul_slot_t;

struct hb_filter_private_s
{
    hb_filter_init_t input;
    hb_filter_init_t output;
    int              depth, lcw, lch;
    hbhip_ctx       *ctx;                 /* the job's GPU (hbhip_host_ctx_for) */
    dl_slot_t        dl[DL_DEPTH + 1];
    int              dl_head, dl_count;
    ul_slot_t        ul[UL_DEPTH];
    int              ul_head, ul_count;
};

static int adapter_init(hb_filter_object_t *filter, hb_filter_init_t *init, int to_device)
{
    hb_filter_private_t *pv = calloc(1, sizeof(*pv));
    if (pv == NULL) return 1;
    const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(init->pix_fmt);
    pv->ctx = hbhip_host_ctx_for(init);
    if (desc == NULL || (desc->comp[0].depth != 8 && desc->comp[0].depth != 10 && desc->comp[0].depth != 12) ||
        pv->ctx == NULL)
    {
        free(pv);
        return 1;
    }
    pv->depth = desc->comp[0].depth;
    pv->lcw = desc->log2_chroma_w;
    pv->lch = desc->log2_chroma_h;
    pv->input = *init;
    init->hw_pix_fmt = to_device ? AV_PIX_FMT_HBHIP : AV_PIX_FMT_NONE;
    pv->output = *init;
    filter->private_data = pv;
    return 0;
}

static int ul_retire(hb_filter_private_t *pv, int all);
static int upload_init(hb_filter_object_t *f, hb_filter_init_t *init)   { return adapter_init(f, init, 1); }
static int download_init(hb_filter_object_t *f, hb_filter_init_t *init) { return adapter_init(f, init, 0); }

/* oldest copy in flight -> its host buffer (NULL on error); the device frame goes back to its pool */
static hb_buffer_t *dl_collect(hb_filter_private_t *pv)
{
    dl_slot_t s = pv->dl[pv->dl_head];
    pv->dl_head = (pv->dl_head + 1) % (DL_DEPTH + 1);
    pv->dl_count--;
    const int rc = hbhip_frame_download_wait(hbhip_host_frame_of(s.in), s.token);
    hb_buffer_copy_props(s.out, s.in);
    hb_buffer_close(&s.in);
    if (rc != HBHIP_OK) hb_buffer_close(&s.out);
    return s.out;
}

static void adapter_close(hb_filter_object_t *filter)
{
    hb_filter_private_t *pv = filter->private_data;
    if (pv == NULL) return;
    (void)ul_retire(pv, 1);                            /* copies still in flight read the host buffers: wait them out */
    while (pv->dl_count > 0)                           /* a cancelled job: drop what is still in the pipe */
    {
        hb_buffer_t *o = dl_collect(pv);
        if (o != NULL) hb_buffer_close(&o);
    }
    free(pv);
    filter->private_data = NULL;
}

/* give back the host buffers whose copies have finished (all of them when `all` - then the oldest ones are waited for) */
static int ul_retire(hb_filter_private_t *pv, int all)
{
    int rc = HBHIP_OK;
    while (pv->ul_count > 0)
    {
        ul_slot_t *s = &pv->ul[pv->ul_head];
        const int d = hbhip_ctx_upload_done(pv->ctx, s->token, all || pv->ul_count >= UL_DEPTH);
        if (d == HBHIP_AGAIN) break;
        if (d != HBHIP_OK) rc = d;                      /* (the buffer is released all the same: nothing reads it any more) */
        hb_buffer_close(&s->in);
        pv->ul_head = (pv->ul_head + 1) % UL_DEPTH;
        pv->ul_count--;
    }
    return rc;
}

static int upload_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out)
{
    hb_filter_private_t *pv = filter->private_data;
    hb_buffer_t *in = *buf_in;
    if ((in->s.flags & HB_BUF_FLAG_EOF) || hbhip_host_frame_of(in) != NULL)
    {
        if (ul_retire(pv, (in->s.flags & HB_BUF_FLAG_EOF) != 0) != HBHIP_OK) return HB_FILTER_FAILED;
        *buf_out = in;
        *buf_in = NULL;
        return (in->s.flags & HB_BUF_FLAG_EOF) ? HB_FILTER_DONE : HB_FILTER_OK;
    }
    if (ul_retire(pv, 0) != HBHIP_OK) return HB_FILTER_FAILED;       /* makes room: waits for the oldest copy when the ring is full */
    hbhip_frame *fr = NULL;
    hbhip_host_frame hf;
    hbhip_host_frame_from_buf(&hf, in);
    if (hbhip_frame_alloc(pv->ctx, in->f.width, in->f.height, pv->depth, pv->lcw, pv->lch, &fr) != HBHIP_OK)
        return HB_FILTER_FAILED;
    void *token = NULL;
    if (hbhip_frame_upload_async(fr, &hf, &token) != HBHIP_OK)
    {
        hbhip_frame_release(fr);
        return HB_FILTER_FAILED;
    }
    hb_buffer_t *out = hbhip_host_wrap_frame(fr, &pv->output, in->f.width, in->f.height);
    if (out == NULL)
    {
        (void)hbhip_ctx_upload_done(pv->ctx, token, 1);            /* the copy reads `in`, which the caller closes */
        hbhip_frame_release(fr);
        return HB_FILTER_FAILED;
    }
    hb_buffer_copy_props(out, in);
    ul_slot_t *s = &pv->ul[(pv->ul_head + pv->ul_count) % UL_DEPTH];
    s->in = in; s->token = token;
    pv->ul_count++;
    *buf_in = NULL;                                     /* ours until its copy has finished */
    *buf_out = out;
    return HB_FILTER_OK;
}

static int download_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out)
{
    hb_filter_private_t *pv = filter->private_data;
    hb_buffer_t *in = *buf_in;
    hbhip_frame *fr = hbhip_host_frame_of(in);
    if ((in->s.flags & HB_BUF_FLAG_EOF) || fr == NULL)
    {
        /* drain the pipe in order, then the buffer itself (EOF, or a frame that never was on the device) */
        hb_buffer_list_t list;
        hb_buffer_list_clear(&list);
        while (pv->dl_count > 0)
        {
            hb_buffer_t *o = dl_collect(pv);
            if (o == NULL) { hb_buffer_list_close(&list); return HB_FILTER_FAILED; }
            hb_buffer_list_append(&list, o);
        }
        hb_buffer_list_append(&list, in);
        *buf_out = hb_buffer_list_clear(&list);
        *buf_in = NULL;
        return (in->s.flags & HB_BUF_FLAG_EOF) ? HB_FILTER_DONE : HB_FILTER_OK;
    }
#ifndef HBHIP_IN_LIBHB
    hbhip_rt_next_buffer_uninitialised();               /* the copy fills it (libhb's own hb_buffer_init never clears) */
#endif
    hb_buffer_t *out = hbhip_host_alloc_out(&pv->output, in->f.width, in->f.height);
    if (out == NULL) return HB_FILTER_FAILED;
    hbhip_host_frame hf;
    hbhip_host_frame_from_buf(&hf, out);
    void *token = NULL;
    if (hbhip_frame_download_async(fr, &hf, &token) != HBHIP_OK)
    {
        hb_buffer_close(&out);
        return HB_FILTER_FAILED;
    }
    dl_slot_t *s = &pv->dl[(pv->dl_head + pv->dl_count) % (DL_DEPTH + 1)];
    s->in = in; s->out = out; s->token = token;
    pv->dl_count++;
    *buf_in = NULL;                                     /* ours until its copy has finished */
    if (pv->dl_count <= DL_DEPTH) return HB_FILTER_DELAY;
    out = dl_collect(pv);
    if (out == NULL) return HB_FILTER_FAILED;
    *buf_out = out;
    return HB_FILTER_OK;
}

hb_filter_object_t hb_filter_hip_upload =
{
    .id            = HB_FILTER_HIP_UPLOAD,
    .enforce_order = 0,
    .name          = "HIP upload adapter",
    .short_name    = "hipupload",
    .init          = upload_init,
    .work          = upload_work,
    .close         = adapter_close,
};

hb_filter_object_t hb_filter_hip_download =
{
    .id            = HB_FILTER_HIP_DOWNLOAD,
    .enforce_order = 0,
    .name          = "HIP download adapter",
    .short_name    = "hipdownload",
    .init          = download_init,
    .work          = download_work,
    .close         = adapter_close,
};
