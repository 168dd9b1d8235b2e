/* hb_runtime.c — standalone stand-in for the libhb L1 services a video filter
 * uses (buffers, buffer lists, settings dictionary, locks/conds/threads,
 * logging, pixel-format descriptors).
 *
 * This file exists only so that filter objects — ours AND the reference's own,
 * compiled in place by oracle/Makefile — can be driven OUTSIDE of libhb by the
 * test harness.  Inside libhb the real fifo.c / ports.c / hb_dict.c provide
 * these symbols.  Semantics follow, re-implemented:
 *   fifo.c:358-457   hb_buffer_init (64 B tail padding, NOPTS timestamps)
 *   fifo.c:618-622   hb_buffer_copy_props
 *   fifo.c:820-881   hb_buffer_init_planes / hb_frame_buffer_init
 *   fifo.c:883-958   blank_stride / mirror_stride (incl. the 8-bit dispatch quirk)
 *   common.c:4002-4235 hb_buffer_list_*
 * One deliberate difference: frame buffers are always zero-filled (the
 * reference recycles pool buffers with stale content; SURVEY §6b hazard 10) so
 * that the oracle is deterministic.
 */
#include "hbhip_libhb.h"

#include <pthread.h>
#include <stdarg.h>
#include <unistd.h>

/* ------------------------------------------------------------------ logging */
static int g_log_level = 0;

void hbhip_set_log_level(int level) { g_log_level = level; }

void hb_log(const char *fmt, ...)
{
    if (g_log_level < 1) return;
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
}

void hb_deep_log(int level, const char *fmt, ...)
{
    if (g_log_level < level + 1) return;
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
}

void hb_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    fputs("ERROR: ", stderr);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
}

/* ------------------------------------------------------- pixel descriptors */
static const AVPixFmtDescriptor k_desc_yuv420p = {
    "yuv420p", 3, 1, 1, 0,
    { {0, 1, 0, 0, 8}, {1, 1, 0, 0, 8}, {2, 1, 0, 0, 8}, {0, 0, 0, 0, 0} } };
static const AVPixFmtDescriptor k_desc_yuv422p = {
    "yuv422p", 3, 1, 0, 0,
    { {0, 1, 0, 0, 8}, {1, 1, 0, 0, 8}, {2, 1, 0, 0, 8}, {0, 0, 0, 0, 0} } };
static const AVPixFmtDescriptor k_desc_yuv444p = {
    "yuv444p", 3, 0, 0, 0,
    { {0, 1, 0, 0, 8}, {1, 1, 0, 0, 8}, {2, 1, 0, 0, 8}, {0, 0, 0, 0, 0} } };
static const AVPixFmtDescriptor k_desc_gray8 = {
    "gray", 1, 0, 0, 0,
    { {0, 1, 0, 0, 8}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0} } };
static const AVPixFmtDescriptor k_desc_yuva420p = {
    "yuva420p", 4, 1, 1, 0,
    { {0, 1, 0, 0, 8}, {1, 1, 0, 0, 8}, {2, 1, 0, 0, 8}, {3, 1, 0, 0, 8} } };
static const AVPixFmtDescriptor k_desc_yuva422p = {
    "yuva422p", 4, 1, 0, 0,
    { {0, 1, 0, 0, 8}, {1, 1, 0, 0, 8}, {2, 1, 0, 0, 8}, {3, 1, 0, 0, 8} } };
static const AVPixFmtDescriptor k_desc_yuva444p = {
    "yuva444p", 4, 0, 0, 0,
    { {0, 1, 0, 0, 8}, {1, 1, 0, 0, 8}, {2, 1, 0, 0, 8}, {3, 1, 0, 0, 8} } };
static const AVPixFmtDescriptor k_desc_yuv420p10 = {
    "yuv420p10le", 3, 1, 1, 0,
    { {0, 2, 0, 0, 10}, {1, 2, 0, 0, 10}, {2, 2, 0, 0, 10}, {0, 0, 0, 0, 0} } };
static const AVPixFmtDescriptor k_desc_yuv420p12 = {
    "yuv420p12le", 3, 1, 1, 0,
    { {0, 2, 0, 0, 12}, {1, 2, 0, 0, 12}, {2, 2, 0, 0, 12}, {0, 0, 0, 0, 0} } };

#define DESC16(sym, nm, lcw, lch, depth) static const AVPixFmtDescriptor sym = { nm, 3, lcw, lch, 0, \
    { {0, 2, 0, 0, depth}, {1, 2, 0, 0, depth}, {2, 2, 0, 0, depth}, {0, 0, 0, 0, 0} } }
DESC16(k_desc_yuv422p10, "yuv422p10le", 1, 0, 10);
DESC16(k_desc_yuv444p10, "yuv444p10le", 0, 0, 10);
DESC16(k_desc_yuv422p12, "yuv422p12le", 1, 0, 12);
DESC16(k_desc_yuv444p12, "yuv444p12le", 0, 0, 12);

const AVPixFmtDescriptor *av_pix_fmt_desc_get(int pix_fmt)
{
    switch (pix_fmt)
    {
        case AV_PIX_FMT_YUV420P:     return &k_desc_yuv420p;
        case AV_PIX_FMT_YUV422P:     return &k_desc_yuv422p;
        case AV_PIX_FMT_YUV444P:     return &k_desc_yuv444p;
        case AV_PIX_FMT_GRAY8:       return &k_desc_gray8;
        case AV_PIX_FMT_YUVA420P:    return &k_desc_yuva420p;
        case AV_PIX_FMT_YUVA422P:    return &k_desc_yuva422p;
        case AV_PIX_FMT_YUVA444P:    return &k_desc_yuva444p;
        case AV_PIX_FMT_YUV420P10LE: return &k_desc_yuv420p10;
        case AV_PIX_FMT_YUV420P12LE: return &k_desc_yuv420p12;
        case AV_PIX_FMT_YUV422P10LE: return &k_desc_yuv422p10;
        case AV_PIX_FMT_YUV444P10LE: return &k_desc_yuv444p10;
        case AV_PIX_FMT_YUV422P12LE: return &k_desc_yuv422p12;
        case AV_PIX_FMT_YUV444P12LE: return &k_desc_yuv444p12;
        default:                     return NULL;
    }
}

int av_get_pix_fmt(const char *name)
{
    static const int known[] = { AV_PIX_FMT_YUV420P, AV_PIX_FMT_YUV422P, AV_PIX_FMT_YUV444P, AV_PIX_FMT_GRAY8,
                                 AV_PIX_FMT_YUVA420P, AV_PIX_FMT_YUVA422P, AV_PIX_FMT_YUVA444P,
                                 AV_PIX_FMT_YUV420P10LE, AV_PIX_FMT_YUV420P12LE, AV_PIX_FMT_YUV422P10LE, AV_PIX_FMT_YUV444P10LE,
                                 AV_PIX_FMT_YUV422P12LE, AV_PIX_FMT_YUV444P12LE };
    if (name == NULL) return AV_PIX_FMT_NONE;
    for (size_t i = 0; i < sizeof(known) / sizeof(known[0]); i++)
    {
        const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(known[i]);
        if (strcmp(d->name, name) == 0) return known[i];
        /* native-endian aliases: "yuv420p10" == "yuv420p10le" here */
        const size_t n = strlen(name);
        if (strlen(d->name) == n + 2 && strncmp(d->name, name, n) == 0 && strcmp(d->name + n, "le") == 0) return known[i];
    }
    return AV_PIX_FMT_NONE;
}

int av_pix_fmt_count_planes(int pix_fmt)
{
    const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(pix_fmt);
    if (d == NULL) return -1;
    int n = 0;
    for (int i = 0; i < d->nb_components; i++)
        if (d->comp[i].plane + 1 > n) n = d->comp[i].plane + 1;
    return n;
}

int av_image_get_linesize(int pix_fmt, int width, int plane)
{
    const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(pix_fmt);
    if (d == NULL) return -1;
    int w = width;
    if (plane == 1 || plane == 2)
        w = -((-width) >> d->log2_chroma_w);
    return w * d->comp[plane].step;
}

void *av_malloc(size_t size)
{
    void *p = NULL;
    if (posix_memalign(&p, 64, size ? size : 1) != 0) return NULL;
    return p;
}

void av_freep(void *arg)
{
    void **pp = (void **)arg;
    free(*pp);
    *pp = NULL;
}

int av_get_cpu_flags(void)
{
#if defined(__x86_64__)
    return AV_CPU_FLAG_SSE2;
#else
    return 0;
#endif
}

/* ----------------------------------------------------------------- threads */
struct hb_lock_s   { pthread_mutex_t m; };
struct hb_cond_s   { pthread_cond_t c; };
struct hb_thread_s { pthread_t t; thread_func_t *fn; void *arg; };

static int g_cpu_override = 0;
void hbhip_set_cpu_count(int n) { g_cpu_override = n; }

int hb_get_cpu_count(void)
{
    if (g_cpu_override > 0) return g_cpu_override;
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    if (n < 1) n = 1;
    if (n > 128) n = 128;
    return (int)n;
}

hb_lock_t *hb_lock_init(void)
{
    hb_lock_t *l = calloc(1, sizeof(*l));
    if (l) pthread_mutex_init(&l->m, NULL);
    return l;
}
void hb_lock_close(hb_lock_t **l)
{
    if (l == NULL || *l == NULL) return;
    pthread_mutex_destroy(&(*l)->m);
    free(*l);
    *l = NULL;
}
void hb_lock(hb_lock_t *l)   { pthread_mutex_lock(&l->m); }
void hb_unlock(hb_lock_t *l) { pthread_mutex_unlock(&l->m); }

hb_cond_t *hb_cond_init(void)
{
    hb_cond_t *c = calloc(1, sizeof(*c));
    if (c) pthread_cond_init(&c->c, NULL);
    return c;
}
void hb_cond_wait(hb_cond_t *c, hb_lock_t *l) { pthread_cond_wait(&c->c, &l->m); }
void hb_cond_signal(hb_cond_t *c)    { pthread_cond_signal(&c->c); }
void hb_cond_broadcast(hb_cond_t *c) { pthread_cond_broadcast(&c->c); }
void hb_cond_close(hb_cond_t **c)
{
    if (c == NULL || *c == NULL) return;
    pthread_cond_destroy(&(*c)->c);
    free(*c);
    *c = NULL;
}

static void *thread_trampoline(void *p)
{
    hb_thread_t *t = p;
    t->fn(t->arg);
    return NULL;
}

hb_thread_t *hb_thread_init(const char *name, thread_func_t *function, void *arg, int priority)
{
    (void)name; (void)priority;
    hb_thread_t *t = calloc(1, sizeof(*t));
    if (t == NULL) return NULL;
    t->fn = function;
    t->arg = arg;
    if (pthread_create(&t->t, NULL, thread_trampoline, t) != 0)
    {
        free(t);
        return NULL;
    }
    return t;
}

void hb_thread_close(hb_thread_t **t)
{
    if (t == NULL || *t == NULL) return;
    pthread_join((*t)->t, NULL);
    free(*t);
    *t = NULL;
}

/* -------------------------------------------------------------- dictionary */
typedef struct kv_s { char *k; char *v; struct kv_s *next; } kv_t;
struct hbhip_dict_s { kv_t *head; kv_t *tail; };

hb_dict_t *hb_dict_init(void) { return calloc(1, sizeof(hb_dict_t)); }

void hb_dict_free(hb_dict_t **pd)
{
    if (pd == NULL || *pd == NULL) return;
    kv_t *e = (*pd)->head;
    while (e)
    {
        kv_t *n = e->next;
        free(e->k); free(e->v); free(e);
        e = n;
    }
    free(*pd);
    *pd = NULL;
}

static kv_t *dict_find(const hb_dict_t *d, const char *key)
{
    if (d == NULL) return NULL;
    for (kv_t *e = d->head; e; e = e->next)
        if (!strcmp(e->k, key)) return e;
    return NULL;
}

void hbhip_dict_set(hb_dict_t *d, const char *key, const char *value)
{
    kv_t *e = dict_find(d, key);
    if (e)
    {
        free(e->v);
        e->v = strdup(value);
        return;
    }
    e = calloc(1, sizeof(*e));
    e->k = strdup(key);
    e->v = strdup(value);
    if (d->tail) d->tail->next = e; else d->head = e;
    d->tail = e;
}

hb_dict_t *hbhip_dict_from_string(const char *settings)
{
    hb_dict_t *d = hb_dict_init();
    if (settings == NULL || d == NULL) return d;
    char *copy = strdup(settings), *save = NULL;
    for (char *tok = strtok_r(copy, ":", &save); tok; tok = strtok_r(NULL, ":", &save))
    {
        char *eq = strchr(tok, '=');
        if (eq == NULL) continue;
        *eq = 0;
        hbhip_dict_set(d, tok, eq + 1);
    }
    free(copy);
    return d;
}

int hb_dict_extract_int(int *dst, const hb_dict_t *dict, const char *key)
{
    kv_t *e = dict_find(dict, key);
    if (e == NULL || dst == NULL) return 0;
    char *end = NULL;
    double v = strtod(e->v, &end);
    if (end == e->v)
    {
        if (!strcasecmp(e->v, "true") || !strcasecmp(e->v, "yes")) { *dst = 1; return 1; }
        if (!strcasecmp(e->v, "false") || !strcasecmp(e->v, "no")) { *dst = 0; return 1; }
        return 0;
    }
    *dst = (int)v;
    return 1;
}

int hb_dict_extract_double(double *dst, const hb_dict_t *dict, const char *key)
{
    kv_t *e = dict_find(dict, key);
    if (e == NULL || dst == NULL) return 0;
    char *end = NULL;
    double v = strtod(e->v, &end);
    if (end == e->v) return 0;
    *dst = v;
    return 1;
}

int hb_dict_extract_bool(int *dst, const hb_dict_t *dict, const char *key)
{
    int v;
    if (!hb_dict_extract_int(&v, dict, key)) return 0;
    *dst = !!v;
    return 1;
}

int hb_dict_extract_string(char **dst, const hb_dict_t *dict, const char *key)
{
    kv_t *e = dict_find(dict, key);
    if (e == NULL || dst == NULL) return 0;
    *dst = strdup(e->v);
    return 1;
}

/* hb_dict.c:607-662, the string form ("30000/1001"): both parts entirely numbers, else 0 */
int hb_dict_extract_rational(hb_rational_t *dst, const hb_dict_t *dict, const char *key)
{
    kv_t *e = dict_find(dict, key);
    if (e == NULL || dst == NULL) return 0;
    const char *slash = strchr(e->v, '/');
    if (slash == NULL || slash == e->v || slash[1] == 0) return 0;
    if (e->v[0] < '0' || e->v[0] > '9' || slash[1] < '0' || slash[1] > '9') return 0;
    char *num_end = NULL, *den_end = NULL;
    const long num = strtol(e->v, &num_end, 0);
    const long den = strtol(slash + 1, &den_end, 0);
    if (num_end != slash || den_end[0] != 0) return 0;
    dst->num = (int)num;
    dst->den = (int)den;
    return 1;
}

/* ----------------------------------------------------------------- buffers */
#define HBHIP_BUF_PADDING 64

/* Frame-sized buffers can come from an allocator supplied by the device library (page-locked
 * memory, recycled): what a HIP build of libhb would do in fifo.c's buffer pools. */
static void *(*g_big_alloc)(size_t) = NULL;
static void (*g_big_free)(void *, size_t) = NULL;
#define HBHIP_BIG_BUFFER (64 * 1024)

void hbhip_rt_set_alloc_hooks(void *(*alloc)(size_t), void (*release)(void *, size_t))
{
    g_big_alloc = alloc;
    g_big_free = release;
}

/* libhb's hb_buffer_init does not clear its payload (fifo.c: av_malloc); this stand-in does, so that whatever the
 * reference's filters read from row padding is the same from run to run.  A caller that is about to overwrite the whole
 * payload (the download adapter: a DMA fills it) says so for its next buffer and saves a pass over the memory. */
static __thread int t_skip_clear = 0;
void hbhip_rt_next_buffer_uninitialised(void) { t_skip_clear = 1; }

hb_buffer_t *hb_buffer_init(int size)
{
    hb_buffer_t *b = calloc(1, sizeof(*b));
    if (b == NULL) return NULL;
    b->size = size;
    b->alloc = size ? size + HBHIP_BUF_PADDING : 0;
    if (size)
    {
        if (g_big_alloc != NULL && b->alloc >= HBHIP_BIG_BUFFER)
        {
            b->data = g_big_alloc(b->alloc);
            if (b->data != NULL) b->hooked_alloc = 1;
        }
        if (b->data == NULL) b->data = av_malloc(b->alloc);
        if (b->data == NULL) { free(b); return NULL; }
        if (!t_skip_clear) memset(b->data, 0, b->alloc);
    }
    t_skip_clear = 0;
    b->s.start = AV_NOPTS_VALUE;
    b->s.stop = AV_NOPTS_VALUE;
    b->s.renderOffset = AV_NOPTS_VALUE;
    b->s.scr_sequence = -1;
    return b;
}

hb_buffer_t *hb_buffer_eof_init(void)
{
    hb_buffer_t *b = hb_buffer_init(0);
    if (b) b->s.flags = HB_BUF_FLAG_EOF;
    return b;
}

void hb_buffer_init_planes(hb_buffer_t *b)
{
    uint8_t *p = b->data;
    for (int pp = 0; pp <= b->f.max_plane; pp++)
    {
        b->plane[pp].data   = p;
        b->plane[pp].stride = hb_image_stride(b->f.fmt, b->f.width, pp);
        b->plane[pp].width  = hb_image_width(b->f.fmt, b->f.width, pp);
        b->plane[pp].height = hb_image_height(b->f.fmt, b->f.height, pp);
        b->plane[pp].size   = b->plane[pp].stride * b->plane[pp].height;
        p += b->plane[pp].size;
    }
}

hb_buffer_t *hb_frame_buffer_init(int pix_fmt, int width, int height)
{
    const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(pix_fmt);
    if (desc == NULL) return NULL;

    int max_plane = 0, size = 0;
    uint8_t seen[4] = {0, 0, 0, 0};
    for (int i = 0; i < desc->nb_components; i++)
    {
        int pp = desc->comp[i].plane;
        if (pp > max_plane) max_plane = pp;
        if (!seen[pp])
        {
            seen[pp] = 1;
            size += hb_image_stride(pix_fmt, width, pp) * hb_image_height(pix_fmt, height, pp);
        }
    }
    hb_buffer_t *b = hb_buffer_init(size);
    if (b == NULL) return NULL;
    b->f.max_plane = max_plane;
    b->s.type = FRAME_BUF;
    b->f.width = width;
    b->f.height = height;
    b->f.fmt = pix_fmt;
    hb_buffer_init_planes(b);
    return b;
}

/* fifo.c:624-639 (the stand-in runtime only has STANDARD and device buffers) */
int hb_buffer_is_writable(const hb_buffer_t *buf)
{
    return buf->storage_type == STANDARD;
}

/* common.c:7054-7091.  The position of the chroma sample inside its 2x2 / 2x1 luma block picks a
 * window into the kernel 1 3 9 27 9 3 1; an even offset averages two neighbouring taps. */
void hb_compute_chroma_smoothing_coefficient(uint32_t chroma_coeffs[2][4], int pix_fmt, int chroma_location)
{
    const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(pix_fmt);
    int wx = 4 - (1 << desc->log2_chroma_w), wy = 4 - (1 << desc->log2_chroma_h);
    const int left = chroma_location == AVCHROMA_LOC_TOPLEFT || chroma_location == AVCHROMA_LOC_LEFT ||
                     chroma_location == AVCHROMA_LOC_BOTTOMLEFT;
    /* the reference's switch falls through TOPLEFT -> TOP and BOTTOMLEFT -> BOTTOM -> CENTER */
    const int vertical = chroma_location == AVCHROMA_LOC_TOPLEFT || chroma_location == AVCHROMA_LOC_TOP ||
                         chroma_location == AVCHROMA_LOC_BOTTOMLEFT || chroma_location == AVCHROMA_LOC_BOTTOM;
    if (left) wx += (1 << desc->log2_chroma_w) - 1;
    if (vertical) wy += (1 << desc->log2_chroma_h) - 1;
    static const uint32_t base[] = { 1, 3, 9, 27, 9, 3, 1 };
    for (int i = 0; i < 4; i++)
    {
        chroma_coeffs[0][i] = (base[i + wx] + base[i + wx + !(wx & 1)]) >> 1;
        chroma_coeffs[1][i] = (base[i + wy] + base[i + wy + !(wy & 1)]) >> 1;
    }
}

void hb_frame_buffer_blank_stride(hb_buffer_t *buf)
{
    for (int pp = 0; pp <= buf->f.max_plane; pp++)
    {
        uint8_t *d = buf->plane[pp].data;
        if (d == NULL) continue;
        for (int y = 0; y < buf->plane[pp].height; y++)
            memset(d + (size_t)y * buf->plane[pp].stride + buf->plane[pp].width, 0x80,
                   buf->plane[pp].stride - buf->plane[pp].width);
    }
}

/* fifo.c:906-958.  The reference computes depth = (bits > 8 ? 2 : 1) and then
 * switches on `case 8`, so every format goes down the 16-bit branch; keep that. */
static void mirror_stride_u16(uint8_t *data, int width, int height, int stride)
{
    uint16_t *px = (uint16_t *)data;
    stride /= 2;
    const int margin = stride - width;
    const int front = margin / 2;
    const int back = margin - front;
    for (int y = 0; y < height; y++)
    {
        int pos = y * stride + width;
        for (int i = 0; i < back; i++)
            px[pos + i] = px[pos - i - 1];
        pos = (y + 1) * stride - 1;
        for (int i = 0; i < front; i++)
            px[pos - i] = px[pos + i + 1];
    }
}

void hb_frame_buffer_mirror_stride(hb_buffer_t *buf)
{
    for (int pp = 0; pp <= buf->f.max_plane; pp++)
    {
        if (buf->plane[pp].data == NULL) continue;
        mirror_stride_u16(buf->plane[pp].data, buf->plane[pp].width,
                          buf->plane[pp].height, buf->plane[pp].stride);
    }
}

/* Device-resident buffers (storage_type HBHIP_DEVICE): the runtime does not know the
 * device library; it is told how to drop / share a storage handle.  Inside libhb this is
 * one more case in fifo.c's free_buffer_resources / hb_buffer_shallow_dup, next to COREMEDIA. */
static void (*g_storage_release)(void *) = NULL;
static void (*g_storage_retain)(void *) = NULL;

void hbhip_rt_set_storage_hooks(void (*retain)(void *), void (*release)(void *))
{
    g_storage_retain = retain;
    g_storage_release = release;
}

void hb_buffer_close(hb_buffer_t **pb)
{
    if (pb == NULL) return;
    hb_buffer_t *b = *pb;
    while (b)
    {
        hb_buffer_t *next = b->next;
        if (b->storage_type == HBHIP_DEVICE && b->storage && g_storage_release)
            g_storage_release(b->storage);
        if (b->data && b->storage_type == STANDARD)
        {
            if (b->hooked_alloc && g_big_free) g_big_free(b->data, b->alloc);
            else free(b->data);
        }
        free(b);
        b = next;
    }
    *pb = NULL;
}

void hb_buffer_copy_props(hb_buffer_t *dst, const hb_buffer_t *src)
{
    dst->s = src->s;
}

int hb_buffer_copy(hb_buffer_t *dst, const hb_buffer_t *src)
{
    if (src == NULL || dst == NULL || dst->size < src->size) return -1;
    memcpy(dst->data, src->data, src->size);
    dst->f = src->f;
    hb_buffer_copy_props(dst, src);
    if (dst->s.type == FRAME_BUF)
        hb_buffer_init_planes(dst);
    return 0;
}

hb_buffer_t *hb_buffer_dup(const hb_buffer_t *src)
{
    if (src == NULL) return NULL;
    if (src->storage_type == HBHIP_DEVICE)
    {
        /* share the device picture, like av_frame_ref in the reference's AVFRAME case (fifo.c:668-714) */
        hb_buffer_t *b = hb_buffer_init(0);
        if (b == NULL) return NULL;
        b->s = src->s;
        b->f = src->f;
        memcpy(b->plane, src->plane, sizeof(b->plane));
        b->storage = src->storage;
        b->storage_type = HBHIP_DEVICE;
        if (g_storage_retain) g_storage_retain(b->storage);
        return b;
    }
    hb_buffer_t *b = hb_buffer_init(src->size);
    if (b == NULL) return NULL;
    if (src->size) memcpy(b->data, src->data, src->size);
    b->s = src->s;
    b->f = src->f;
    if (b->s.type == FRAME_BUF)
        hb_buffer_init_planes(b);
    return b;
}

hb_buffer_t *hb_buffer_shallow_dup(const hb_buffer_t *src)
{
    return hb_buffer_dup(src);
}

/* ------------------------------------------------------------ buffer lists */
static hb_buffer_t *chain_end(hb_buffer_t *b, int *count, int *size)
{
    *count = 1;
    *size = b->size;
    while (b->next)
    {
        b = b->next;
        (*count)++;
        *size += b->size;
    }
    return b;
}

void hb_buffer_list_append(hb_buffer_list_t *l, hb_buffer_t *buf)
{
    if (buf == NULL) return;
    int n, sz;
    hb_buffer_t *end = chain_end(buf, &n, &sz);
    if (l->tail == NULL) l->head = buf; else l->tail->next = buf;
    l->tail = end;
    l->count += n;
    l->size += sz;
}

void hb_buffer_list_prepend(hb_buffer_list_t *l, hb_buffer_t *buf)
{
    if (buf == NULL) return;
    int n, sz;
    hb_buffer_t *end = chain_end(buf, &n, &sz);
    if (l->tail == NULL) l->tail = end; else end->next = l->head;
    l->head = buf;
    l->count += n;
    l->size += sz;
}

hb_buffer_t *hb_buffer_list_head(hb_buffer_list_t *l) { return l ? l->head : NULL; }
hb_buffer_t *hb_buffer_list_tail(hb_buffer_list_t *l) { return l ? l->tail : NULL; }

hb_buffer_t *hb_buffer_list_rem_head(hb_buffer_list_t *l)
{
    if (l == NULL || l->head == NULL) return NULL;
    hb_buffer_t *h = l->head;
    if (l->head == l->tail) l->tail = NULL;
    l->head = h->next;
    l->count--;
    l->size -= h->size;
    h->next = NULL;
    return h;
}

hb_buffer_t *hb_buffer_list_rem_tail(hb_buffer_list_t *l)
{
    if (l == NULL || l->tail == NULL) return NULL;
    hb_buffer_t *t = l->tail;
    if (l->head == t)
    {
        l->head = l->tail = NULL;
        l->count = 0;
        l->size = 0;
    }
    else
    {
        hb_buffer_t *p = l->head;
        while (p->next != t) p = p->next;
        p->next = NULL;
        l->tail = p;
        l->count--;
        l->size -= t->size;
    }
    t->next = NULL;
    return t;
}

hb_buffer_t *hb_buffer_list_rem(hb_buffer_list_t *l, hb_buffer_t *b)
{
    if (l == NULL) return NULL;
    if (b == l->head) return hb_buffer_list_rem_head(l);
    hb_buffer_t *a = l->head;
    while (a && a->next != b) a = a->next;
    if (a == NULL) return NULL;
    a->next = b->next;
    if (l->tail == b) l->tail = a;
    l->count--;
    l->size -= b->size;
    b->next = NULL;
    return b;
}

hb_buffer_t *hb_buffer_list_clear(hb_buffer_list_t *l)
{
    if (l == NULL) return NULL;
    hb_buffer_t *h = l->head;
    l->head = l->tail = NULL;
    l->count = 0;
    l->size = 0;
    return h;
}

hb_buffer_t *hb_buffer_list_set(hb_buffer_list_t *l, hb_buffer_t *buf)
{
    if (l == NULL) return NULL;
    hb_buffer_t *old = l->head;
    l->head = buf;
    l->tail = NULL;
    l->count = 0;
    l->size = 0;
    if (buf)
        l->tail = chain_end(buf, &l->count, &l->size);
    return old;
}

void hb_buffer_list_close(hb_buffer_list_t *l)
{
    hb_buffer_t *b = hb_buffer_list_clear(l);
    hb_buffer_close(&b);
}

int hb_buffer_list_count(hb_buffer_list_t *l) { return l ? l->count : 0; }
int hb_buffer_list_size(hb_buffer_list_t *l)  { return l ? l->size : 0; }


/* ------------------------------------------------------------ fifos (fifo.c:1194-1557)
 * What a filter's own queues need: a locked singly linked list that never blocks. */
struct hb_fifo_s
{
    pthread_mutex_t lock;
    hb_buffer_t    *first, *last;
    int             size;
};

hb_fifo_t *hb_fifo_init(int capacity, int thresh)
{
    (void)capacity; (void)thresh;
    hb_fifo_t *f = calloc(1, sizeof(*f));
    if (f != NULL) pthread_mutex_init(&f->lock, NULL);
    return f;
}

int hb_fifo_size(hb_fifo_t *f)
{
    pthread_mutex_lock(&f->lock);
    const int n = f->size;
    pthread_mutex_unlock(&f->lock);
    return n;
}

hb_buffer_t *hb_fifo_get(hb_fifo_t *f)
{
    pthread_mutex_lock(&f->lock);
    hb_buffer_t *b = f->size < 1 ? NULL : f->first;
    if (b != NULL)
    {
        f->first = b->next;
        b->next = NULL;
        f->size--;
    }
    pthread_mutex_unlock(&f->lock);
    return b;
}

hb_buffer_t *hb_fifo_see(hb_fifo_t *f)
{
    pthread_mutex_lock(&f->lock);
    hb_buffer_t *b = f->size < 1 ? NULL : f->first;
    pthread_mutex_unlock(&f->lock);
    return b;
}

void hb_fifo_push(hb_fifo_t *f, hb_buffer_t *b)
{
    if (b == NULL) return;
    pthread_mutex_lock(&f->lock);
    if (f->size > 0) f->last->next = b; else f->first = b;
    f->last = b;
    f->size++;
    while (f->last->next != NULL)                  /* a ->next list goes in whole (fifo.c:1455-1460) */
    {
        f->size++;
        f->last = f->last->next;
    }
    pthread_mutex_unlock(&f->lock);
}

void hb_fifo_flush(hb_fifo_t *f)
{
    hb_buffer_t *b;
    while ((b = hb_fifo_get(f)) != NULL)
        hb_buffer_close(&b);
}

void hb_fifo_close(hb_fifo_t **pf)
{
    if (pf == NULL || *pf == NULL) return;
    hb_fifo_flush(*pf);
    pthread_mutex_destroy(&(*pf)->lock);
    free(*pf);
    *pf = NULL;
}

/* handbrake.h:136 / hb.c: one hb_interjob_t per hb_handle_t; the stand-in has one handle */
hb_interjob_t *hb_interjob_get(hb_handle_t *h)
{
    static hb_interjob_t interjob;
    (void)h;
    return &interjob;
}

/* the helper objects a hw pipeline registers for its hw_pix_fmt (see include/hbhip_libhb.h) */
#define HW_HELPER_MAX 8
static struct { int kind, fmt; void *obj; } g_hw_helper[HW_HELPER_MAX];
static int g_hw_helpers = 0;

void hbhip_rt_register_hw_helper(int kind, int hw_pix_fmt, void *object)
{
    for (int i = 0; i < g_hw_helpers; i++)
        if (g_hw_helper[i].kind == kind && g_hw_helper[i].fmt == hw_pix_fmt)
        {
            g_hw_helper[i].obj = object;
            return;
        }
    if (g_hw_helpers < HW_HELPER_MAX)
    {
        g_hw_helper[g_hw_helpers].kind = kind;
        g_hw_helper[g_hw_helpers].fmt = hw_pix_fmt;
        g_hw_helper[g_hw_helpers++].obj = object;
    }
}

void *hbhip_rt_hw_helper(int kind, int hw_pix_fmt)
{
    for (int i = 0; i < g_hw_helpers; i++)
        if (g_hw_helper[i].kind == kind && g_hw_helper[i].fmt == hw_pix_fmt)
            return g_hw_helper[i].obj;
    return NULL;
}

/* ---------------------------------------------------------------- lists, filter registry, filter lists
 * (common.c:2489-2700, :5247-5540; hb.c:1676-1723) */
struct hb_list_s { void **items; int count, cap; };

hb_list_t *hb_list_init(void) { return calloc(1, sizeof(hb_list_t)); }
int hb_list_count(const hb_list_t *l) { return l ? l->count : 0; }
void *hb_list_item(const hb_list_t *l, int i) { return (l == NULL || i < 0 || i >= l->count) ? NULL : l->items[i]; }

void hb_list_insert(hb_list_t *l, int pos, void *p)
{
    if (l == NULL || p == NULL) return;
    if (l->count == l->cap)
    {
        l->cap = l->cap ? 2 * l->cap : 8;
        l->items = realloc(l->items, sizeof(void *) * l->cap);
    }
    if (pos < 0) pos = 0;
    if (pos > l->count) pos = l->count;
    memmove(&l->items[pos + 1], &l->items[pos], sizeof(void *) * (l->count - pos));
    l->items[pos] = p;
    l->count++;
}

void hb_list_add(hb_list_t *l, void *p) { hb_list_insert(l, hb_list_count(l), p); }

void hb_list_rem(hb_list_t *l, void *p)
{
    if (l == NULL) return;
    for (int i = 0; i < l->count; i++)
        if (l->items[i] == p)
        {
            memmove(&l->items[i], &l->items[i + 1], sizeof(void *) * (l->count - i - 1));
            l->count--;
            return;
        }
}

void hb_list_close(hb_list_t **pl)
{
    if (pl == NULL || *pl == NULL) return;
    free((*pl)->items);
    free(*pl);
    *pl = NULL;
}

hb_dict_t *hb_value_dup(const hb_dict_t *d)
{
    hb_dict_t *c = hb_dict_init();
    if (d != NULL && c != NULL)
        for (kv_t *e = d->head; e; e = e->next) hbhip_dict_set(c, e->k, e->v);
    return c;
}

#define RT_MAX_FILTER_ID 64
static hb_filter_object_t *g_registry[RT_MAX_FILTER_ID];

void hbhip_rt_register_filter(int filter_id, hb_filter_object_t *proto)
{
    if (filter_id >= 0 && filter_id < RT_MAX_FILTER_ID) g_registry[filter_id] = proto;
}

hb_filter_object_t *hb_filter_get(int filter_id)
{
    return (filter_id >= 0 && filter_id < RT_MAX_FILTER_ID) ? g_registry[filter_id] : NULL;
}

hb_filter_object_t *hb_filter_copy(hb_filter_object_t *filter)            /* common.c:5247-5258 */
{
    if (filter == NULL) return NULL;
    hb_filter_object_t *c = malloc(sizeof(*c));
    if (c == NULL) return NULL;
    memcpy(c, filter, sizeof(*c));
    if (filter->settings) c->settings = hb_value_dup(filter->settings);
    c->sub_filter = hb_filter_copy(filter->sub_filter);
    return c;
}

hb_filter_object_t *hb_filter_init(int filter_id)                         /* common.c:5497-5522, without the mt_frame wrap */
{
    return hb_filter_copy(hb_filter_get(filter_id));
}

void hb_filter_close(hb_filter_object_t **pf)                             /* common.c:5524-5537 */
{
    if (pf == NULL || *pf == NULL) return;
    hb_filter_close(&(*pf)->sub_filter);
    hb_dict_free(&(*pf)->settings);
    free(*pf);
    *pf = NULL;
}

hb_filter_object_t *hb_filter_find(const hb_list_t *list, int filter_id)  /* common.c:5305-5323 */
{
    for (int i = 0; i < hb_list_count(list); i++)
    {
        hb_filter_object_t *f = hb_list_item(list, i);
        if (f->id == filter_id) return f;
    }
    return NULL;
}

void hb_add_filter_dict(hb_list_t *list, hb_filter_object_t *filter, const hb_dict_t *settings_in)   /* hb.c:1676-1723 */
{
    if (filter == NULL) return;
    hb_dict_free(&filter->settings);
    filter->settings = settings_in ? hb_value_dup(settings_in) : hb_dict_init();
    if (filter->enforce_order)
        for (int i = 0; i < hb_list_count(list); i++)
        {
            hb_filter_object_t *f = hb_list_item(list, i);
            if (f->id > filter->id) { hb_list_insert(list, i, filter); return; }
            if (f->id == filter->id) { hb_filter_close(&filter); return; }         /* no filter twice */
        }
    hb_list_add(list, filter);
}
