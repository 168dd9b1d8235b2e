/* nlmeans_hip.c — HIP-backed drop-in for libhb's NLMeans filter object.
 *
 * Same plugin surface, settings keys, cascade/sanitise rules and output as the
 * reference's hb_filter_nlmeans (libhb/nlmeans.c:190-213 object + template,
 * :223-419 init, :666-694 work, :421-462 close); the pixel work runs in
 * libhbhip.so (csrc/nlmeans.hip) through the C ABI of include/hbhip.h.
 *
 * Differences that are NOT visible in the output frames:
 *   - no taskset / frame threads: `threads` is accepted and ignored.  The
 *     reference emits frames in bursts of `threads` once max_frames+threads are
 *     buffered (nlmeans.c:546-596); we emit frame t as soon as frame
 *     t+nframes-1 has arrived.  Frame content does not depend on that
 *     (SURVEY §6b hazard 8).
 *   - at EOF the remaining frames are filtered with the same shrinking
 *     temporal window as nlmeans_filter_flush (nlmeans.c:599-664).
 * If the device path cannot take the settings (>8-bit, patch size outside
 * 3/5/7/9) init() returns non-zero so libhb keeps its CPU filter
 * (work.c:1861-1868); there is no CPU code in here.
 */
#include "hbhip_host.h"

#define NLM_FRAMES_MAX 32   /* nlmeans.c:87 */
#define NLM_EXPSIZE    128  /* nlmeans.c:88 */

struct hb_filter_private_s
{
    hbhip_nlmeans_params par;
    hbhip_filter        *dev;
    hb_buffer_list_t     props;      /* one zero-size buffer per queued frame, holds `s` */
    int64_t              next_tag;
    int                  dev_io;
    hb_filter_init_t     input;
    hb_filter_init_t     output;
};

static int  nlmeans_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init);
static int  nlmeans_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out);
static void nlmeans_hip_close(hb_filter_object_t *filter);

static const char nlmeans_hip_template[] =
    "y-strength=^"HB_FLOAT_REG"$:y-origin-tune=^"HB_FLOAT_REG"$:"
    "y-patch-size=^"HB_INT_REG"$:y-range=^"HB_INT_REG"$:"
    "y-frame-count=^"HB_INT_REG"$:y-prefilter=^"HB_INT_REG"$:"
    "cb-strength=^"HB_FLOAT_REG"$:cb-origin-tune=^"HB_FLOAT_REG"$:"
    "cb-patch-size=^"HB_INT_REG"$:cb-range=^"HB_INT_REG"$:"
    "cb-frame-count=^"HB_INT_REG"$:cb-prefilter=^"HB_INT_REG"$:"
    "cr-strength=^"HB_FLOAT_REG"$:cr-origin-tune=^"HB_FLOAT_REG"$:"
    "cr-patch-size=^"HB_INT_REG"$:cr-range=^"HB_INT_REG"$:"
    "cr-frame-count=^"HB_INT_REG"$:cr-prefilter=^"HB_INT_REG"$:"
    "threads=^"HB_INT_REG"$";

hb_filter_object_t hb_filter_nlmeans_hip =
{
    .id                = HB_FILTER_NLMEANS,
    .enforce_order     = 1,
    .name              = "Denoise (nlmeans, HIP)",
    .short_name        = "nlmeans",
    .settings          = NULL,
    .init              = nlmeans_hip_init,
    .work              = nlmeans_hip_work,
    .close             = nlmeans_hip_close,
    .settings_template = nlmeans_hip_template,
};

/* Read, cascade and sanitise the settings exactly as nlmeans.c:266-343 does,
 * then build the weight tables of :345-358 with host libm. */
static void nlmeans_hip_params(hb_dict_t *dict, int depth, hbhip_nlmeans_params *p)
{
    static const char *pfx[3] = { "y", "cb", "cr" };
    char key[32];

    for (int c = 0; c < 3; c++)
    {
        p->strength[c] = p->origin_tune[c] = -1;
        p->patch_size[c] = p->range[c] = p->nframes[c] = p->prefilter[c] = -1;
        if (dict == NULL) continue;
        snprintf(key, sizeof(key), "%s-strength", pfx[c]);    hb_dict_extract_double(&p->strength[c], dict, key);
        snprintf(key, sizeof(key), "%s-origin-tune", pfx[c]); hb_dict_extract_double(&p->origin_tune[c], dict, key);
        snprintf(key, sizeof(key), "%s-patch-size", pfx[c]);  hb_dict_extract_int(&p->patch_size[c], dict, key);
        snprintf(key, sizeof(key), "%s-range", pfx[c]);       hb_dict_extract_int(&p->range[c], dict, key);
        snprintf(key, sizeof(key), "%s-frame-count", pfx[c]); hb_dict_extract_int(&p->nframes[c], dict, key);
        snprintf(key, sizeof(key), "%s-prefilter", pfx[c]);   hb_dict_extract_int(&p->prefilter[c], dict, key);
    }
    /* Cr inherits Cb inherits Y (nlmeans.c:306-316) */
    for (int c = 1; c < 3; c++)
    {
        if (p->strength[c] == -1)    p->strength[c]    = p->strength[c - 1];
        if (p->origin_tune[c] == -1) p->origin_tune[c] = p->origin_tune[c - 1];
        if (p->patch_size[c] == -1)  p->patch_size[c]  = p->patch_size[c - 1];
        if (p->range[c] == -1)       p->range[c]       = p->range[c - 1];
        if (p->nframes[c] == -1)     p->nframes[c]     = p->nframes[c - 1];
        if (p->prefilter[c] == -1)   p->prefilter[c]   = p->prefilter[c - 1];
    }
    for (int c = 0; c < 3; c++)
    {
        /* defaults 6:1:7:3:2:0 for every channel (nlmeans.c:58-69, 321-326) */
        if (p->strength[c] == -1)    p->strength[c]    = 6;
        if (p->origin_tune[c] == -1) p->origin_tune[c] = 1;
        if (p->patch_size[c] == -1)  p->patch_size[c]  = 7;
        if (p->range[c] == -1)       p->range[c]       = 3;
        if (p->nframes[c] == -1)     p->nframes[c]     = 2;
        if (p->prefilter[c] == -1)   p->prefilter[c]   = 0;

        /* sanitise (nlmeans.c:329-338) */
        if (p->strength[c] < 0)         p->strength[c] = 0;
        if (p->origin_tune[c] < 0.01)   p->origin_tune[c] = 0.01;
        if (p->origin_tune[c] > 1)      p->origin_tune[c] = 1;
        if (p->patch_size[c] % 2 == 0)  p->patch_size[c]--;
        if (p->patch_size[c] < 1)       p->patch_size[c] = 1;
        if (p->range[c] % 2 == 0)       p->range[c]--;
        if (p->range[c] < 1)            p->range[c] = 1;
        if (p->nframes[c] < 1)          p->nframes[c] = 1;
        if (p->nframes[c] > NLM_FRAMES_MAX) p->nframes[c] = NLM_FRAMES_MAX;
        if (p->prefilter[c] < 0)        p->prefilter[c] = 0;

        /* strength scales with bit depth (nlmeans.c:343) */
        p->strength[c] *= depth > 8 ? (depth - 8) * (depth - 8) : 1;

        /* weight LUT (nlmeans.c:345-358): float/double mix kept as written there */
        const float weight_factor       = 1.0 / p->patch_size[c] / p->patch_size[c] /
                                          (p->strength[c] * p->strength[c]);
        const float min_weight_in_table = 0.0005;
        const float stretch             = NLM_EXPSIZE / (-log(min_weight_in_table));
        p->weight_fact_table[c] = weight_factor * stretch;
        p->diff_max[c]          = NLM_EXPSIZE / p->weight_fact_table[c];
        for (int i = 0; i < NLM_EXPSIZE; i++)
            p->exptable[c][i] = exp(-i / stretch);
        p->exptable[c][NLM_EXPSIZE - 1] = 0;
    }
}

#ifndef HBHIP_IN_LIBHB
/* Same parameter/table derivation from a "key=value:..." string, for callers that drive the C ABI directly
 * (bench.py, device-resident tests).  Uses the stand-in runtime's string parser, so it does not exist in a build
 * inside libhb (where nothing needs it). */
void hbhip_nlmeans_params_from_settings(const char *settings, int depth, hbhip_nlmeans_params *p)
{
    hb_dict_t *d = hbhip_dict_from_string(settings);
    nlmeans_hip_params(d, depth, p);
    hb_dict_free(&d);
}
#endif

static int nlmeans_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init)
{
    hb_filter_private_t *pv = calloc(1, sizeof(*pv));
    if (pv == NULL)
    {
        hb_error("nlmeans(hip): calloc failed");
        return -1;
    }
    filter->private_data = pv;
    pv->input = *init;
    pv->dev_io = hbhip_host_dev_io(init);

    const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(init->pix_fmt);
    if (desc == NULL)
        goto fail;
    const int depth = desc->comp[0].depth;

    nlmeans_hip_params(filter->settings, depth, &pv->par);

    hbhip_ctx *ctx = hbhip_host_ctx_for(init);
    if (ctx == NULL)
    {
        hb_error("nlmeans(hip): no HIP device context");
        goto fail;
    }
    int rc = hbhip_nlmeans_create(ctx, &pv->par, init->geometry.width, init->geometry.height,
                                  depth, desc->log2_chroma_w, desc->log2_chroma_h, &pv->dev);
    if (rc != HBHIP_OK)
    {
        hb_error("nlmeans(hip): %s", hbhip_strerror(rc));
        goto fail;
    }
    /* inside a device-resident run frames come in and go out AS frames: nothing is copied at either end (include/hbhip.h) */
    if (pv->dev_io && hbhip_host_zero_copy()) hbhip_filter_use_frames(pv->dev);
    /* frames per launch: the kernel needs several frames' tiles to fill the GPU.  Like the reference, which works on
     * `threads` frames at a time (nlmeans.c:548-571), the filter then emits bursts. */
    const char *env = getenv("HBHIP_NLMEANS_BATCH");
    hbhip_nlmeans_set_batch(pv->dev, env != NULL && atoi(env) > 0 ? atoi(env) : 8);

    hb_buffer_list_clear(&pv->props);
    pv->output = *init;
    return 0;

fail:
    free(pv);
    filter->private_data = NULL;
    return -1;
}

static void nlmeans_hip_close(hb_filter_object_t *filter)
{
    hb_filter_private_t *pv = filter->private_data;
    if (pv == NULL)
        return;
    hbhip_filter_destroy(pv->dev);
    hb_buffer_list_close(&pv->props);
    free(pv);
    filter->private_data = NULL;
}

/* Pull every frame the device has finished into a ->next chain. */
static int nlmeans_hip_collect(hb_filter_private_t *pv, hb_buffer_list_t *list)
{
    while (hbhip_filter_pending(pv->dev) > 0)
    {
        hb_buffer_t *out = hbhip_host_pull(pv->dev, &pv->output, pv->input.geometry.width,
                                           pv->input.geometry.height, pv->dev_io, NULL);
        if (out == NULL)
        {
            hb_error("nlmeans(hip): pull failed");
            return -1;
        }
        hb_buffer_t *props = hb_buffer_list_rem_head(&pv->props);
        if (props != NULL)
        {
            hb_buffer_copy_props(out, props);
            hb_buffer_close(&props);
        }
        hb_buffer_list_append(list, out);
    }
    return 0;
}

static int nlmeans_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out)
{
    hb_filter_private_t *pv = filter->private_data;
    hb_buffer_t *in = *buf_in;
    hb_buffer_list_t list;
    hb_buffer_list_clear(&list);

    if (in->s.flags & HB_BUF_FLAG_EOF)
    {
        /* drain with a shrinking look-ahead, then forward the EOF buffer last
         * (nlmeans.c:673-688) */
        int rc = hbhip_filter_flush(pv->dev);
        if (rc != HBHIP_OK || nlmeans_hip_collect(pv, &list) != 0)
        {
            hb_buffer_list_close(&list);
            return HB_FILTER_FAILED;
        }
        hb_buffer_list_append(&list, in);
        *buf_out = hb_buffer_list_clear(&list);
        *buf_in = NULL;
        return HB_FILTER_DONE;
    }

    int rc = hbhip_host_push(pv->dev, in, pv->next_tag++);
    if (rc != HBHIP_OK)
    {
        hb_error("nlmeans(hip): push: %s", hbhip_strerror(rc));
        return HB_FILTER_FAILED;
    }
    /* remember timestamps/flags of this frame (nlmeans.c:540-541) */
    hb_buffer_t *props = hb_buffer_init(0);
    hb_buffer_copy_props(props, in);
    hb_buffer_list_append(&pv->props, props);

    if (nlmeans_hip_collect(pv, &list) != 0)
    {
        hb_buffer_list_close(&list);
        return HB_FILTER_FAILED;
    }
    *buf_out = hb_buffer_list_clear(&list);
    return HB_FILTER_OK;
}
