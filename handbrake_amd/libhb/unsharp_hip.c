/* unsharp_hip.c — HIP-backed drop-ins for libhb's unsharp and chroma-smooth
 * filter objects (libhb/unsharp.c:68-87, :175-272 init, :331-390 work;
 * libhb/chroma_smooth.c:67-85, :174-283 init, :345-407 work).  Same keys,
 * cascades, defaults and clamps; the reference's per-thread scratch rows
 * (init_thread / work_thread for mt_frame_filter.c) have no equivalent: one
 * device instance serves every frame.  Pixel work: csrc/sharpen.hip. */
#include "hbhip_host.h"

struct hb_filter_private_s
{
    hbhip_blur_params par;
    hbhip_filter     *dev;
    hb_filter_init_t  input;
    hb_filter_init_t  output;
    int               dev_io;
};

static int  unsharp_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init);
static int  chroma_smooth_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init);
static int  blur_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out);
static void blur_hip_close(hb_filter_object_t *filter);

static const char unsharp_hip_template[] =
    "y-strength=^"HB_FLOAT_REG"$:y-size=^"HB_INT_REG"$:"
    "cb-strength=^"HB_FLOAT_REG"$:cb-size=^"HB_INT_REG"$:"
    "cr-strength=^"HB_FLOAT_REG"$:cr-size=^"HB_INT_REG"$";

static const char chroma_smooth_hip_template[] =
    "cb-strength=^"HB_FLOAT_REG"$:cb-size=^"HB_INT_REG"$:"
    "cr-strength=^"HB_FLOAT_REG"$:cr-size=^"HB_INT_REG"$";

hb_filter_object_t hb_filter_unsharp_hip =
{
    .id                = HB_FILTER_UNSHARP,
    .enforce_order     = 1,
    .name              = "Sharpen (unsharp, HIP)",
    .short_name        = "unsharp",
    .settings          = NULL,
    .init              = unsharp_hip_init,
    .work              = blur_hip_work,
    .close             = blur_hip_close,
    .settings_template = unsharp_hip_template,
};

hb_filter_object_t hb_filter_chroma_smooth_hip =
{
    .id                = HB_FILTER_CHROMA_SMOOTH,
    .enforce_order     = 1,
    .name              = "Chroma Smooth (HIP)",
    .short_name        = "chromasmooth",
    .settings          = NULL,
    .init              = chroma_smooth_hip_init,
    .work              = blur_hip_work,
    .close             = blur_hip_close,
    .settings_template = chroma_smooth_hip_template,
};

static int blur_hip_init_common(hb_filter_object_t *filter, hb_filter_init_t *init,
                                int first_plane, double max_strength, int chroma_only)
{
    hb_filter_private_t *pv = calloc(1, sizeof(*pv));
    if (pv == NULL) return -1;
    filter->private_data = pv;
    pv->input = *init;
    pv->dev_io = hbhip_host_dev_io(init);

    const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(init->pix_fmt);
    if (desc == NULL) goto fail;

    static const char *pfx[3] = { "y", "cb", "cr" };
    double strength[3] = { -1, -1, -1 };
    int    size[3] = { -1, -1, -1 };
    char   key[32];
    for (int c = first_plane; c < 3 && filter->settings != NULL; c++)
    {
        snprintf(key, sizeof(key), "%s-strength", pfx[c]);
        hb_dict_extract_double(&strength[c], filter->settings, key);
        snprintf(key, sizeof(key), "%s-size", pfx[c]);
        hb_dict_extract_int(&size[c], filter->settings, key);
    }
    /* cascade: unsharp Y -> Cb -> Cr (unsharp.c:213-220); chroma smooth Cb -> Cr only
     * (chroma_smooth.c:214-221) */
    for (int c = first_plane + 1; c < 3; c++)
    {
        if (strength[c] == -1) strength[c] = strength[c - 1];
        if (size[c] == -1)     size[c] = size[c - 1];
    }
    for (int c = 0; c < 3; c++)
    {
        if (strength[c] == -1) strength[c] = 0.25;        /* unsharp.c:13-16, chroma_smooth.c:13 */
        if (size[c] == -1)     size[c] = 7;
        if (strength[c] < 0) strength[c] = 0;
        if (strength[c] > max_strength) strength[c] = max_strength;
        if (size[c] % 2 == 0) size[c]--;
        if (size[c] < 3)  size[c] = 3;
        if (size[c] > 15) size[c] = 15;
        pv->par.amount[c] = strength[c] * 65536.0;        /* unsharp.c:258 */
        pv->par.size[c] = size[c];
    }
    if (chroma_only)
        pv->par.amount[0] = 0;                             /* luma is copied, chroma_smooth.c:262-269 */

    hbhip_ctx *ctx = hbhip_host_ctx_for(init);
    if (ctx == NULL) goto fail;
    int rc = chroma_only
        ? hbhip_chroma_smooth_create(ctx, &pv->par, init->geometry.width, init->geometry.height,
                                     desc->comp[0].depth, desc->log2_chroma_w, desc->log2_chroma_h, &pv->dev)
        : hbhip_unsharp_create(ctx, &pv->par, init->geometry.width, init->geometry.height,
                               desc->comp[0].depth, desc->log2_chroma_w, desc->log2_chroma_h, &pv->dev);
    if (rc != HBHIP_OK)
    {
        hb_error("%s(hip): %s", filter->short_name, hbhip_strerror(rc));
        goto fail;
    }
    pv->output = *init;
    return 0;
fail:
    free(pv);
    filter->private_data = NULL;
    return -1;
}

static int unsharp_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init)
{
    return blur_hip_init_common(filter, init, 0, 1.5, 0);   /* unsharp.c:247-248 */
}

static int chroma_smooth_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init)
{
    return blur_hip_init_common(filter, init, 1, 3.0, 1);   /* chroma_smooth.c:247-248 */
}

static void blur_hip_close(hb_filter_object_t *filter)
{
    hb_filter_private_t *pv = filter->private_data;
    if (pv == NULL) return;
    hbhip_host_simple_destroy(pv->dev);
    free(pv);
    filter->private_data = NULL;
}

static int blur_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out)
{
    hb_filter_private_t *pv = filter->private_data;
    return hbhip_host_simple_work(pv->dev, &pv->output, filter->short_name, pv->dev_io, buf_in, buf_out);
}
