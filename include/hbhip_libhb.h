/* hbhip_libhb.h — the slice of libhb's L1/L3 interface that a video filter
 * plugin touches, written from scratch so the HIP-backed filter objects (and
 * the test oracle) can be built OUTSIDE of libhb.
 *
 * When these filters are compiled inside libhb this header is
 * not used at all: define HBHIP_IN_LIBHB and "handbrake/handbrake.h" supplies
 * the real declarations.  Every declaration below names the reference
 * declaration it stands in for (path relative to /root/reference/libhb):
 *
 *   hb_filter_object_t, hb_filter_init_t, HB_FILTER_*   handbrake/common.h:1628-1780
 *   hb_buffer_t, hb_buffer_settings_t, hb_image_format_t handbrake/internal.h:65-165
 *   hb_image_stride/width/height, hb_image_copy_plane   handbrake/internal.h:220-275
 *   hb_buffer_list_t                                    handbrake/common.h:115-134
 *   hb_lock/hb_cond/hb_thread                           handbrake/ports.h:162-210
 *   hb_dict_extract_*                                   handbrake/hb_dict.h:54-69
 *   HB_*_REG                                            handbrake/common.h:1887-1893
 *
 * Only fields a filter reads or writes are present; the struct layouts are our
 * own (both the oracle wrapper TUs and our filters compile against THIS file,
 * so they agree with each other by construction).
 */
#ifndef HBHIP_LIBHB_H
#define HBHIP_LIBHB_H

#ifdef HBHIP_IN_LIBHB
#include "handbrake/handbrake.h"
#else

#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <math.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __LIBHB__
#define __LIBHB__ 1
#endif

#if defined(__x86_64__) && !defined(ARCH_X86)
#define ARCH_X86 1
#define ARCH_X86_64 1
#endif

/* ---- small macros (common.h:59-80) ------------------------------------ */
#ifndef MIN
#define MIN(a, b) (((a) < (b)) ? (a) : (b))
#endif
#ifndef MAX
#define MAX(a, b) (((a) > (b)) ? (a) : (b))
#endif
#ifndef ABS
#define ABS(a) ((a) > 0 ? (a) : (-(a)))
#endif
#define MULTIPLE_MOD_UP(a, b) (((b) * (int)(((a) + ((b) - 1)) / (b))))
#define HB_ALIGN(x, a) (((x) + (a) - 1) & ~((a) - 1))

#define HB_FLOAT_REG    "(([0-9]+([.,][0-9]+)?)|([.,][0-9]+))"
#define HB_INT_REG      "([0-9]+)"
#define HB_RATIONAL_REG "([0-9]+/[0-9]+)"
#define HB_BOOL_REG     "(yes|no|true|false|[01])"
#define HB_ALL_REG      "(.*)"

/* ---- libavutil stand-ins ------------------------------------------------ */
#define AV_NOPTS_VALUE ((int64_t)UINT64_C(0x8000000000000000))
#define FFMIN(a, b) ((a) > (b) ? (b) : (a))
#define FFMAX(a, b) ((a) > (b) ? (a) : (b))
#define AV_CEIL_RSHIFT(a, b) (-((-(a)) >> (b)))

enum hbhip_pix_fmt
{
    AV_PIX_FMT_NONE        = -1,
    AV_PIX_FMT_YUV420P     = 0,
    AV_PIX_FMT_YUV422P     = 4,
    AV_PIX_FMT_YUV444P     = 5,
    AV_PIX_FMT_GRAY8       = 8,
    AV_PIX_FMT_YUVA420P    = 33,   /* subtitle overlays (rendersub.c: pix_fmt_alpha) */
    AV_PIX_FMT_YUVA422P    = 78,
    AV_PIX_FMT_YUVA444P    = 79,
    AV_PIX_FMT_YUV420P10LE = 62,
    AV_PIX_FMT_YUV420P10   = 62,
    AV_PIX_FMT_YUV420P12LE = 123,
    AV_PIX_FMT_YUV420P12   = 123,
    /* the other 10 / 12-bit layouts hb_av_can_use_zscale lists (hbffmpeg.c:893-909), FFmpeg's numbers */
    AV_PIX_FMT_YUV422P10LE = 64,  AV_PIX_FMT_YUV422P10 = 64,
    AV_PIX_FMT_YUV444P10LE = 68,  AV_PIX_FMT_YUV444P10 = 68,
    AV_PIX_FMT_YUV422P12LE = 127, AV_PIX_FMT_YUV422P12 = 127,
    AV_PIX_FMT_YUV444P12LE = 131, AV_PIX_FMT_YUV444P12 = 131
};

typedef struct AVComponentDescriptor
{
    int plane;
    int step;
    int offset;
    int shift;
    int depth;
} AVComponentDescriptor;

typedef struct AVPixFmtDescriptor
{
    const char *name;
    uint8_t     nb_components;
    uint8_t     log2_chroma_w;
    uint8_t     log2_chroma_h;
    uint64_t    flags;
    AVComponentDescriptor comp[4];
} AVPixFmtDescriptor;

const AVPixFmtDescriptor *av_pix_fmt_desc_get(int pix_fmt);
int av_get_pix_fmt(const char *name);                 /* libavutil/pixdesc.h; AV_PIX_FMT_NONE when unknown */
int  av_pix_fmt_count_planes(int pix_fmt);
enum { AVCHROMA_LOC_UNSPECIFIED = 0, AVCHROMA_LOC_LEFT, AVCHROMA_LOC_CENTER, AVCHROMA_LOC_TOPLEFT,
       AVCHROMA_LOC_TOP, AVCHROMA_LOC_BOTTOMLEFT, AVCHROMA_LOC_BOTTOM };
int  av_image_get_linesize(int pix_fmt, int width, int plane);
void *av_malloc(size_t size);
void  av_freep(void *ptr);
#define AV_CPU_FLAG_SSE2 0x0010
int  av_get_cpu_flags(void);

typedef struct AVChannelLayout { int nb_channels; } AVChannelLayout;

/* ---- opaque handles ------------------------------------------------------ */
typedef struct hb_job_s            hb_job_t;
typedef struct hb_fifo_s           hb_fifo_t;
typedef struct hb_subtitle_s       hb_subtitle_t;
typedef struct hb_lock_s           hb_lock_t;
typedef struct hb_cond_s           hb_cond_t;
typedef struct hb_thread_s         hb_thread_t;
typedef struct hb_filter_private_s hb_filter_private_t;
typedef struct hb_filter_object_s  hb_filter_object_t;
typedef struct hb_blend_private_s  hb_blend_private_t;
typedef struct hb_motion_metric_private_s hb_motion_metric_private_t;
typedef struct hb_motion_metric_object_s  hb_motion_metric_object_t;
typedef struct hb_blend_object_s   hb_blend_object_t;
typedef struct hb_buffer_s         hb_buffer_t;
typedef struct hb_buffer_list_s    hb_buffer_list_t;
typedef struct hbhip_dict_s        hb_dict_t;
typedef struct hbhip_dict_s        hb_value_t;

typedef struct hb_rational_s { int num; int den; } hb_rational_t;
typedef struct hb_geometry_s { int width; int height; hb_rational_t par; } hb_geometry_t;

/* ---- logging (handbrake/internal.h:23-32) ------------------------------- */
void hb_log(const char *fmt, ...);
void hb_deep_log(int level, const char *fmt, ...);
void hb_error(const char *fmt, ...);

/* ---- ports.h:162-210 ------------------------------------------------------ */
#define HB_LOW_PRIORITY    0
#define HB_NORMAL_PRIORITY 0
typedef void (thread_func_t)(void *);
hb_thread_t *hb_thread_init(const char *name, thread_func_t *function, void *arg, int priority);
void         hb_thread_close(hb_thread_t **);
hb_lock_t   *hb_lock_init(void);
void         hb_lock_close(hb_lock_t **);
void         hb_lock(hb_lock_t *);
void         hb_unlock(hb_lock_t *);
hb_cond_t   *hb_cond_init(void);
void         hb_cond_wait(hb_cond_t *, hb_lock_t *);
void         hb_cond_signal(hb_cond_t *);
void         hb_cond_broadcast(hb_cond_t *);
void         hb_cond_close(hb_cond_t **);
int          hb_get_cpu_count(void);
/* test/bench hook (ours): override what hb_get_cpu_count() reports; 0 = real. */
void         hbhip_set_cpu_count(int n);

/* ---- settings dictionary (hb_dict.h:54-69) -------------------------------
 * Stand-in: an ordered list of key/value strings.  hb_dict_extract_* return 1
 * when the key exists (and was convertible), 0 otherwise, like the reference. */
hb_dict_t *hb_dict_init(void);
void       hb_dict_free(hb_dict_t **);
void       hbhip_dict_set(hb_dict_t *, const char *key, const char *value);
/* "key=value:key=value" (the CLI / settings_template form). */
hb_dict_t *hbhip_dict_from_string(const char *settings);
int hb_dict_extract_int(int *dst, const hb_dict_t *dict, const char *key);
int hb_dict_extract_double(double *dst, const hb_dict_t *dict, const char *key);
int hb_dict_extract_bool(int *dst, const hb_dict_t *dict, const char *key);
int hb_dict_extract_string(char **dst, const hb_dict_t *dict, const char *key);
int hb_dict_extract_rational(hb_rational_t *dst, const hb_dict_t *dict, const char *key);   /* "num/den", hb_dict.c:607-662 */

/* ---- buffers (internal.h:65-165) ------------------------------------------ */
#define PIC_FLAG_TOP_FIELD_FIRST    0x0008
#define PIC_FLAG_PROGRESSIVE_FRAME  0x0010
#define PIC_FLAG_REPEAT_FIRST_FIELD 0x0100
#define PIC_FLAG_REPEAT_FRAME       0x0200
#define HB_BUF_FLAG_EOF             0x0400
#define HB_BUF_FLAG_EOS             0x0800

#define HB_COMB_NONE  0
#define HB_COMB_LIGHT 1
#define HB_COMB_HEAVY 2

typedef struct hb_buffer_settings_s
{
    enum { OTHER_BUF, AUDIO_BUF, VIDEO_BUF, SUBTITLE_BUF, FRAME_BUF } type;
    int      id;
    int64_t  start;
    double   duration;
    int64_t  stop;
    int64_t  renderOffset;
    int64_t  pcr;
    int      scr_sequence;
    int      split;
    uint8_t  discontinuity;
    int      new_chap;
    uint8_t  frametype;
    uint16_t flags;
    uint8_t  combed;
} hb_buffer_settings_t;

typedef struct hb_image_format_s
{
    int x, y;
    int width, height;
    int fmt;
    int color_prim, color_transfer, color_matrix, color_range;
    int chroma_location;
    int max_plane;
    int window_width, window_height;
} hb_image_format_t;

struct hb_buffer_s
{
    int      size;
    int      alloc;
    uint8_t *data;
    int      offset;

    hb_buffer_settings_t s;
    hb_image_format_t    f;

    struct buffer_plane
    {
        uint8_t *data;
        int      stride;
        int      width;
        int      height;
        int      size;
    } plane[4];

    void *storage;
    enum { STANDARD, AVFRAME, COREMEDIA, HBHIP_DEVICE } storage_type;

    hb_buffer_t *palette;
    void       **side_data;
    int          nb_side_data;

    hb_buffer_t *next;
#ifndef HBHIP_IN_LIBHB
    int          hooked_alloc;   /* stand-in runtime only: `data` came from hbhip_rt_set_alloc_hooks' allocator */
#endif
};

struct hb_buffer_list_s
{
    hb_buffer_t *head;
    hb_buffer_t *tail;
    int count;
    int size;
};

hb_buffer_t *hb_buffer_init(int size);
hb_buffer_t *hb_buffer_eof_init(void);
hb_buffer_t *hb_frame_buffer_init(int pix_fmt, int w, int h);
void         hb_frame_buffer_blank_stride(hb_buffer_t *buf);
void         hb_frame_buffer_mirror_stride(hb_buffer_t *buf);
void         hb_buffer_init_planes(hb_buffer_t *b);
void         hb_buffer_close(hb_buffer_t **);
hb_buffer_t *hb_buffer_dup(const hb_buffer_t *src);
hb_buffer_t *hb_buffer_shallow_dup(const hb_buffer_t *src);
int          hb_buffer_copy(hb_buffer_t *dst, const hb_buffer_t *src);
void         hb_buffer_copy_props(hb_buffer_t *dst, const hb_buffer_t *src);
int          hb_buffer_is_writable(const hb_buffer_t *buf);       /* fifo.c:624-639 */
/* common.c:7054-7091: weights of the up-to-4 overlay samples under one chroma sample, per axis */
void         hb_compute_chroma_smoothing_coefficient(uint32_t chroma_coeffs[2][4], int pix_fmt, int chroma_location);
/* stand-in runtime only: how to share / drop an HBHIP_DEVICE storage handle */
void         hbhip_rt_set_storage_hooks(void (*retain)(void *), void (*release)(void *));
/* stand-in runtime only: allocator for frame-sized buffer payloads (page-locked pool) */
void         hbhip_rt_set_alloc_hooks(void *(*alloc)(size_t), void (*release)(void *, size_t));
void         hbhip_rt_next_buffer_uninitialised(void);   /* the next hb_buffer_init on this thread leaves its payload as it is */

/* fifo.c:1194-1557 - the part a filter uses for its own queues (vfr.c's delay queue, rendersub.c's subtitle fifo):
 * unbounded, never blocking; hb_fifo_push takes a ->next list, hb_fifo_get returns NULL when empty */
hb_fifo_t   *hb_fifo_init(int capacity, int thresh);
void         hb_fifo_close(hb_fifo_t **);
int          hb_fifo_size(hb_fifo_t *);
hb_buffer_t *hb_fifo_get(hb_fifo_t *);
hb_buffer_t *hb_fifo_see(hb_fifo_t *);
void         hb_fifo_push(hb_fifo_t *, hb_buffer_t *);
void         hb_fifo_flush(hb_fifo_t *);

void         hb_buffer_list_append(hb_buffer_list_t *list, hb_buffer_t *buf);
void         hb_buffer_list_prepend(hb_buffer_list_t *list, hb_buffer_t *buf);
hb_buffer_t *hb_buffer_list_head(hb_buffer_list_t *list);
hb_buffer_t *hb_buffer_list_rem_head(hb_buffer_list_t *list);
hb_buffer_t *hb_buffer_list_tail(hb_buffer_list_t *list);
hb_buffer_t *hb_buffer_list_rem_tail(hb_buffer_list_t *list);
hb_buffer_t *hb_buffer_list_rem(hb_buffer_list_t *list, hb_buffer_t *b);
hb_buffer_t *hb_buffer_list_clear(hb_buffer_list_t *list);
hb_buffer_t *hb_buffer_list_set(hb_buffer_list_t *list, hb_buffer_t *buf);
void         hb_buffer_list_close(hb_buffer_list_t *list);
int          hb_buffer_list_count(hb_buffer_list_t *list);
int          hb_buffer_list_size(hb_buffer_list_t *list);

/* internal.h:220-275 */
static inline int hb_image_stride(int pix_fmt, int width, int plane)
{
    int linesize = av_image_get_linesize(pix_fmt, width, plane);
    return MULTIPLE_MOD_UP(linesize, 64);
}

static inline int hb_image_width(int pix_fmt, int width, int plane)
{
    const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(pix_fmt);
    if (desc != NULL && (plane == 1 || plane == 2))
        width = -((-width) >> desc->log2_chroma_w);
    return width;
}

static inline int hb_image_height(int pix_fmt, int height, int plane)
{
    const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(pix_fmt);
    if (desc != NULL && (plane == 1 || plane == 2))
        height = -((-height) >> desc->log2_chroma_h);
    return height;
}

static inline void hb_image_copy_plane(uint8_t *dst, const uint8_t *src,
                                       const int stride_dst, const int stride_src,
                                       const int height)
{
    if (src == dst)
        return;
    if (stride_src == stride_dst)
    {
        memcpy(dst, src, (size_t)stride_dst * height);
        return;
    }
    const int n = stride_src < stride_dst ? ABS(stride_src) : stride_dst;
    for (int y = 0; y < height; y++)
        memcpy(dst + (size_t)y * stride_dst, src + (ptrdiff_t)y * stride_src, n);
}

/* ---- filter plugin surface (common.h:1628-1780) ---------------------------- */
#define HB_FILTER_OK     0
#define HB_FILTER_DELAY  1
#define HB_FILTER_FAILED 2
#define HB_FILTER_DROP   3
#define HB_FILTER_DONE   4

typedef struct hb_filter_init_s
{
    hb_job_t     *job;
    int           pix_fmt;
    int           hw_pix_fmt;
    void         *hw_frames_ctx;
    int           color_prim;
    int           color_transfer;
    int           color_matrix;
    int           color_range;
    int           chroma_location;
    hb_geometry_t geometry;
    int           crop[4];
    int           grayscale;
    hb_rational_t vrate;
    int           cfr;
    hb_rational_t time_base;
    int           samplerate;
    int           sample_fmt;
    AVChannelLayout ch_layout;
} hb_filter_init_t;

typedef struct hb_filter_info_s
{
    char            *human_readable_desc;
    hb_filter_init_t output;
} hb_filter_info_t;

struct hb_filter_object_s
{
    int         id;
    int         enforce_order;
    int         skip;
    int         aliased;
    char       *name;
    char       *short_name;
    hb_dict_t  *settings;

    int  (*init)(hb_filter_object_t *, hb_filter_init_t *);
    int  (*init_thread)(hb_filter_object_t *, int);
    int  (*post_init)(hb_filter_object_t *, hb_job_t *);
    int  (*work)(hb_filter_object_t *, hb_buffer_t **, hb_buffer_t **);
    int  (*work_thread)(hb_filter_object_t *, hb_buffer_t **, hb_buffer_t **, int);
    void (*close)(hb_filter_object_t *);
    hb_filter_info_t *(*info)(hb_filter_object_t *);

    const char *settings_template;

    hb_fifo_t *fifo_in;
    hb_fifo_t *fifo_out;
    hb_subtitle_t *subtitle;
    hb_filter_private_t *private_data;
    hb_thread_t *thread;
    volatile int *done;
    int status;
    int chapter_val;
    int64_t chapter_time;
    hb_filter_object_t *sub_filter;
};

/* Numeric ids are the reference's (common.h:1729-1778). */
enum
{
    HB_FILTER_INVALID = 0,
    HB_FILTER_FIRST = 1,
    HB_FILTER_ADAPTER_VT,
    HB_FILTER_DETELECINE,
    HB_FILTER_COMB_DETECT,
    HB_FILTER_COMB_DETECT_VT,
    HB_FILTER_DECOMB,
    HB_FILTER_YADIF,
    HB_FILTER_YADIF_VT,
    HB_FILTER_BWDIF,
    HB_FILTER_BWDIF_VT,
    HB_FILTER_VFR,
    HB_FILTER_DEBLOCK,
    HB_FILTER_DEBAND,
    HB_FILTER_DENOISE,
    HB_FILTER_HQDN3D = HB_FILTER_DENOISE,
    HB_FILTER_BM3D,
    HB_FILTER_NLMEANS,
    HB_FILTER_CHROMA_SMOOTH,
    HB_FILTER_CHROMA_SMOOTH_VT,
    HB_FILTER_ROTATE,
    HB_FILTER_ROTATE_VT,
    HB_FILTER_RENDER_SUB,
    HB_FILTER_CROP_SCALE,
    HB_FILTER_CROP_SCALE_VT,
    HB_FILTER_LAPSHARP,
    HB_FILTER_LAPSHARP_VT,
    HB_FILTER_UNSHARP,
    HB_FILTER_UNSHARP_VT,
    HB_FILTER_GRAYSCALE,
    HB_FILTER_GRAYSCALE_VT,
    HB_FILTER_PAD,
    HB_FILTER_PAD_VT,
    HB_FILTER_COLORSPACE,
    HB_FILTER_FORMAT,
    HB_FILTER_RPU,
    HB_FILTER_AVFILTER,
    HB_FILTER_LAST,
    HB_FILTER_MT_FRAME,
    /* appended, never inserted (saved job JSON and the C# interop carry the numbers above) */
    HB_FILTER_HIP_UPLOAD,
    HB_FILTER_HIP_DOWNLOAD
};

/* ---- the job's filter list: hb_list (common.c:2489-2700), hb_filter_get / _init / _copy / _close / _find
 *      (common.c:5247-5540), hb_add_filter_dict (hb.c:1676-1723).  Stand-in: only what the swap of CPU filters
 *      for HIP ones touches (handbrake_amd/libhb/hip_common.c; the reference's precedent is
 *      platform/macosx/vt_common.c:486-540, called from work.c:1515-1523). -------------------------------- */
typedef struct hb_list_s hb_list_t;
hb_list_t *hb_list_init(void);
int        hb_list_count(const hb_list_t *);
void      *hb_list_item(const hb_list_t *, int);
void       hb_list_add(hb_list_t *, void *);
void       hb_list_insert(hb_list_t *, int pos, void *);
void       hb_list_rem(hb_list_t *, void *);
void       hb_list_close(hb_list_t **);
hb_dict_t *hb_value_dup(const hb_dict_t *);                         /* hb_dict.h: deep copy of a settings dict */
typedef struct hb_handle_s hb_handle_t;
/* handbrake.h:121-136: what vfr.c leaves behind for a second pass */
typedef struct hb_interjob_s
{
    int     sequence_id;
    int     frame_count;
    int     out_frame_count;
    int64_t total_time;
    hb_rational_t vrate;
    hb_subtitle_t *select_subtitle;
    void *context;
    int   context_size;
} hb_interjob_t;
hb_interjob_t *hb_interjob_get(hb_handle_t *);
/* what rendersub.c reads of a subtitle track (common.h:1254-1332) and of an attachment (:1339-1345) */
typedef struct hb_data_s { uint8_t *bytes; size_t size; } hb_data_t;
typedef struct hb_subtitle_config_s
{
    enum subdest { RENDERSUB, PASSTHRUSUB } dest;
    int      force, default_track, external_filename_set;
    int64_t  offset;
} hb_subtitle_config_t;
struct hb_subtitle_s
{
    int  id, track, out_track;
    hb_subtitle_config_t config;
    enum subtype { PICTURESUB, TEXTSUB } format;
    enum subsource { VOBSUB, CC608SUB, CC708SUB, UTF8SUB, TX3GSUB, SSASUB, PGSSUB, IMPORTSRT, IMPORTSSA, DVBSUB,
                     SRTSUB = IMPORTSRT } source;
    uint32_t     palette[16];
    uint8_t      palette_set;
    int          width, height;
    hb_data_t   *extradata;
    hb_fifo_t   *fifo_in, *fifo_raw, *fifo_sync, *fifo_out;          /* rendersub takes its bitmaps from fifo_out */
};
typedef struct hb_attachment_s
{
    enum attachtype { FONT_TTF_ATTACH, FONT_OTF_ATTACH, HB_ART_ATTACH } type;
    char *name, *data;
    int   size;
} hb_attachment_t;
typedef struct hb_title_s { hb_geometry_t geometry; } hb_title_t;
struct hb_job_s                                                     /* the fields of hb_job_t the swap and the filters read */
{
    hb_list_t   *list_filter;
    int          hw_pix_fmt;                                        /* AV_PIX_FMT_NONE unless a hw decoder set it */
    int          input_pix_fmt;
    int          hw_device_index;                                   /* common.h:991: which adapter the job runs on; -1 = default */
    hb_handle_t *h;
    volatile int done;
    int          crop[4];                                           /* rendersub keeps subtitles inside the picture that is left */
    hb_title_t  *title;
    hb_list_t   *list_subtitle, *list_attachment;
    int          vcodec;                                            /* common.h:707-770: whose encoder */
    int          hw_decode;                                         /* common.h:990, 1433-1440: whose decoder */
};
/* whose hardware a job's decoder / encoder is (common.h:715-756, 1433-1440): hw_device_index is THEIR adapter index then */
#define HB_VCODEC_VT_MASK            0x00080000
#define HB_VCODEC_QSV_MASK           0x00040000
#define HB_VCODEC_FFMPEG_MASK        0x00010000
#define HB_DECODE_QSV                0x02
#define HB_DECODE_NVDEC              0x04
#define HB_DECODE_VIDEOTOOLBOX       0x08
#define HB_DECODE_MF                 0x10
#define HB_DECODE_AMFDEC             0x20
hb_filter_object_t *hb_filter_get(int filter_id);                   /* the registered CPU prototype, or NULL */
hb_filter_object_t *hb_filter_init(int filter_id);                  /* a copy of it, ready for settings */
hb_filter_object_t *hb_filter_copy(hb_filter_object_t *);
void                hb_filter_close(hb_filter_object_t **);
hb_filter_object_t *hb_filter_find(const hb_list_t *, int filter_id);
void                hb_add_filter_dict(hb_list_t *, hb_filter_object_t *, const hb_dict_t *settings);
/* stand-in only: what hb_filter_get's switch holds inside libhb (tests register the reference's own objects) */
void                hbhip_rt_register_filter(int filter_id, hb_filter_object_t *proto);
/* stand-in only: the harness's do_job() calls hip_common.c through these (inside libhb work.c calls it directly) */
void                hbhip_rt_set_job_hooks(void (*setup)(hb_job_t *), int (*init_failed)(hb_job_t *, int, hb_filter_init_t *),
                                           void (*job_close)(hb_job_t *));

/* ---- the frame-difference metric plugin type vfr.c uses (handbrake/common.h:1799-1811) -- */
struct hb_motion_metric_object_s
{
    char *name;
    int   (*init)(hb_motion_metric_object_t *, hb_filter_init_t *);
    float (*work)(hb_motion_metric_object_t *, hb_buffer_t *, hb_buffer_t *);
    void  (*close)(hb_motion_metric_object_t *);
    hb_motion_metric_private_t *private_data;
};

extern hb_motion_metric_object_t hb_motion_metric;                 /* motion_metric.c:306-312 (tests: from oracle/_ref) */
extern hb_blend_object_t         hb_blend;                         /* blend.c:40-46 (tests: from oracle/_ref) */
/* stand-in only: the helper object a hw pipeline supplies for hw_pix_fmt (inside libhb: one more `case` in
 * vfr.c:76-108 / rendersub.c:1129-1161).  kind 0 = motion metric, 1 = blend.  NULL when none. */
void        hbhip_rt_register_hw_helper(int kind, int hw_pix_fmt, void *object);
void       *hbhip_rt_hw_helper(int kind, int hw_pix_fmt);

/* ---- the subtitle compositor plugin type (handbrake/common.h:1813-1828) ---------------- */
struct hb_blend_object_s
{
    char *name;
    int          (*init)(hb_blend_object_t *, int in_width, int in_height, int in_pix_fmt,
                         int in_chroma_location, int in_color_range, int overlay_pix_fmt);
    hb_buffer_t *(*work)(hb_blend_object_t *, hb_buffer_t *, hb_buffer_list_t *, int changed);
    void         (*close)(hb_blend_object_t *);
    hb_blend_private_t *private_data;
};

#ifdef __cplusplus
}
#endif

#endif /* HBHIP_IN_LIBHB */
#endif /* HBHIP_LIBHB_H */
