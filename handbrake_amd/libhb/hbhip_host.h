/* hbhip_host.h — what the HIP-backed hb_filter_object_t implementations share:
 * the libhb types (real ones inside libhb, include/hbhip_libhb.h outside), the
 * C ABI of the device library, and a few helpers to move hb_buffer_t planes
 * through it.
 */
#ifndef HBHIP_HOST_H
#define HBHIP_HOST_H

#include "hbhip_libhb.h"
#include "hbhip.h"

/* One process-wide device context per GPU index, created on first use.  Filters of one job share it, i.e. they share
 * one stream, which is what lets adjacent HIP filters hand frames over in HBM without extra synchronisation.  The GPU
 * is the job's: init->job->hw_device_index (common.h:991) when it is >= 0, else the process default (HBHIP_DEVICE in
 * the environment, else 0) - libhb/hbhip_registry.c. */
int        hbhip_host_default_device(void);
int        hbhip_host_device_for(const hb_filter_init_t *init);
int        hbhip_host_job_index_is_hip(const hb_job_t *job);     /* hw_device_index names a HIP device (no other vendor's hw path in the job) */
hbhip_ctx *hbhip_host_ctx_on(int device);
hbhip_ctx *hbhip_host_ctx_for(const hb_filter_init_t *init);     /* what a drop-in's init() uses: the job's own stream on the job's GPU */
hbhip_ctx *hbhip_host_job_ctx(const hb_job_t *job);              /* the context a live job has leased, or NULL */
/* role 1 = the deinterlacing side of a job (comb detect, decomb, yadif, bwdif): a second context of the job's GPU unless
 * HBHIP_JOB_STREAMS=1, then the job's own; hbhip_host_use_frame: called by every consumer of a device frame in front of the
 * work that reads it (libhb/hbhip_registry.c) */
hbhip_ctx *hbhip_host_ctx_for_role(const hb_filter_init_t *init, int role);
int        hbhip_host_use_frame(hbhip_ctx *ctx, const hb_buffer_t *in);
hbhip_ctx *hbhip_host_ctx(void);                                 /* the process default's context */
void       hbhip_host_ctx_release(void);

static inline void hbhip_host_frame_from_buf(hbhip_host_frame *f, const hb_buffer_t *b)
{
    for (int p = 0; p < 3; p++)
    {
        f->plane[p]  = b->plane[p].data;
        f->stride[p] = b->plane[p].stride;
    }
}

/* Allocate an output frame the way every reference filter does
 * (e.g. lapsharp.c:334-339): hb_frame_buffer_init + colour properties. */
static inline hb_buffer_t *hbhip_host_alloc_out(const hb_filter_init_t *o, int width, int height)
{
    hb_buffer_t *out = hb_frame_buffer_init(o->pix_fmt, width, height);
    if (out == NULL) return NULL;
    out->f.color_prim      = o->color_prim;
    out->f.color_transfer  = o->color_transfer;
    out->f.color_matrix    = o->color_matrix;
    out->f.color_range     = o->color_range;
    out->f.chroma_location = o->chroma_location;
    return out;
}

/* ---- device-resident hand-off -------------------------------------
 * Between hb_filter_hip_upload and hb_filter_hip_download the chain's frames stay in HBM:
 * init->hw_pix_fmt is set to AV_PIX_FMT_HBHIP by the upload adapter (as the reference's
 * VideoToolbox path sets hw_pix_fmt for its Metal filters), every HIP filter that sees it
 * takes and produces hb_buffer_t whose storage_type is HBHIP_DEVICE and whose `storage` is
 * an hbhip_frame*, and the download adapter restores host buffers. */
#define AV_PIX_FMT_HBHIP 0x48495001
static inline int hbhip_host_dev_io(const hb_filter_init_t *init) { return init->hw_pix_fmt == AV_PIX_FMT_HBHIP; }
static inline hbhip_frame *hbhip_host_frame_of(const hb_buffer_t *b)
{
    return b->storage_type == HBHIP_DEVICE ? (hbhip_frame *)b->storage : NULL;
}
/* HBHIP_ZERO_COPY=0: the drop-ins of a device-resident run copy frames into and out of pictures of their own again
 * (hbhip_filter_push_dev / pull_dev) instead of adopting them (hbhip_filter_use_frames) - for A/B measurements */
static inline int hbhip_host_zero_copy(void) { const char *e = getenv("HBHIP_ZERO_COPY"); return e == NULL || atoi(e) != 0; }
/* hb_buffer_t shell around a device frame (takes over the caller's reference). */
hb_buffer_t *hbhip_host_wrap_frame(hbhip_frame *fr, const hb_filter_init_t *o, int width, int height);
/* Feed `in` (host planes or device frame) to a device filter. */
int hbhip_host_push(hbhip_filter *dev, const hb_buffer_t *in, int64_t tag);
/* Next finished frame of a device filter as a fresh buffer: device-resident when dev_io,
 * else hb_frame_buffer_init() + download.  NULL on error. */
hb_buffer_t *hbhip_host_pull(hbhip_filter *dev, const hb_filter_init_t *o, int width, int height,
                             int dev_io, int64_t *tag);

/* destroy a device filter driven by hbhip_host_simple_work (drops what is still in its pipe) */
void hbhip_host_simple_destroy(hbhip_filter *dev);

extern hb_filter_object_t hb_filter_hip_upload;
extern hb_filter_object_t hb_filter_hip_download;

/* Shared body of the stateless (one in, one out) HIP filters: EOF is forwarded
 * (e.g. lapsharp.c:326-331), otherwise the frame goes through the device filter
 * and comes back in a fresh hb_frame_buffer_init() buffer carrying the input's
 * properties (lapsharp.c:334-353). */
int hbhip_host_simple_work(hbhip_filter *dev, const hb_filter_init_t *output, const char *who,
                           int dev_io, hb_buffer_t **buf_in, hb_buffer_t **buf_out);

/* The HIP drop-ins registered by this library (ids = the CPU filters' ids,
 * SURVEY Appendix D). */
extern hb_filter_object_t hb_filter_nlmeans_hip;
extern hb_filter_object_t hb_filter_lapsharp_hip;
extern hb_filter_object_t hb_filter_unsharp_hip;
extern hb_filter_object_t hb_filter_chroma_smooth_hip;
extern hb_filter_object_t hb_filter_denoise_hip;
extern hb_filter_object_t hb_filter_crop_scale_hip;
extern hb_filter_object_t hb_filter_grayscale_hip;
extern hb_filter_object_t hb_filter_rotate_hip;
extern hb_filter_object_t hb_filter_colorspace_hip;
extern hb_filter_object_t hb_filter_pad_hip;
extern hb_filter_object_t hb_filter_yadif_hip;
extern hb_filter_object_t hb_filter_bwdif_hip;
extern hb_filter_object_t hb_filter_format_hip;
/* the subtitle compositor object rendersub.c can use in place of hb_blend (blend.c:40-46) */
extern hb_blend_object_t  hb_blend_hip;
/* the frame-difference metric object vfr.c can use in place of hb_motion_metric (motion_metric.c:306-312) */
extern hb_motion_metric_object_t hb_motion_metric_hip;
extern hb_filter_object_t hb_filter_decomb_hip;
extern hb_filter_object_t hb_filter_comb_detect_hip;

#ifndef HBHIP_IN_LIBHB
void hbhip_nlmeans_params_from_settings(const char *settings, int depth, hbhip_nlmeans_params *p);   /* bench / tests only */
#endif

/* hb_filter_get() analogue (common.c:5331-5495) for the HIP drop-ins. */
hb_filter_object_t *hbhip_filter_get(int filter_id);

#endif
