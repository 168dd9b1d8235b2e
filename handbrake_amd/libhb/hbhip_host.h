/* hbhip_host.h — what the HIP-backed hb_filter_object_t implementations share:
 * the libhb types (real ones inside libhb, include/hbhip_libhb.h outside), the
 * C ABI of the device library, and a few helpers to move hb_buffer_t planes
 * through it.
 */
#ifndef HBHIP_HOST_H
#define HBHIP_HOST_H

#include "hbhip_libhb.h"
#include "hbhip.h"

/* One process-wide device context per GPU index, created on first use
 * (HBHIP_DEVICE env selects the index; default 0).  Filters of one job share
 * it, i.e. they share one stream, which is what lets adjacent HIP filters
 * hand frames over in HBM without extra synchronisation. */
hbhip_ctx *hbhip_host_ctx(void);
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

/* Shared body of the stateless (one in, one out) HIP filters: EOF is forwarded
 * (e.g. lapsharp.c:326-331), otherwise the frame goes through the device filter
 * and comes back in a fresh hb_frame_buffer_init() buffer carrying the input's
 * properties (lapsharp.c:334-353). */
int hbhip_host_simple_work(hbhip_filter *dev, const hb_filter_init_t *output, const char *who,
                           hb_buffer_t **buf_in, hb_buffer_t **buf_out);

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
extern hb_filter_object_t hb_filter_decomb_hip;
extern hb_filter_object_t hb_filter_comb_detect_hip;

void hbhip_nlmeans_params_from_settings(const char *settings, int depth, hbhip_nlmeans_params *p);

/* hb_filter_get() analogue (common.c:5331-5495) for the HIP drop-ins. */
hb_filter_object_t *hbhip_filter_get(int filter_id);

#endif
