/* lapsharp_hip.c — HIP-backed drop-in for libhb's lapsharp filter object
 * (libhb/lapsharp.c:107-123 template/object, :190-311 init, :326-358 work).
 * Same keys (y/cb/cr-strength, y/cb/cr-kernel), cascade Y -> Cb -> Cr, defaults
 * 0.2 / isolap, clamp 0..1.5.  Pixel work: csrc/sharpen.hip via include/hbhip.h. */
#include "hbhip_host.h"

struct hb_filter_private_s
{
    hbhip_lapsharp_params par;
    hbhip_filter         *dev;
    hb_filter_init_t      input;
    hb_filter_init_t      output;
    int                   dev_io;
};

static int  lapsharp_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init);
static int  lapsharp_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out);
static void lapsharp_hip_close(hb_filter_object_t *filter);

static const char lapsharp_hip_template[] =
    "y-strength=^"HB_FLOAT_REG"$:y-kernel=^"HB_ALL_REG"$:"
    "cb-strength=^"HB_FLOAT_REG"$:cb-kernel=^"HB_ALL_REG"$:"
    "cr-strength=^"HB_FLOAT_REG"$:cr-kernel=^"HB_ALL_REG"$";

hb_filter_object_t hb_filter_lapsharp_hip =
{
    .id                = HB_FILTER_LAPSHARP,
    .enforce_order     = 1,
    .name              = "Sharpen (lapsharp, HIP)",
    .short_name        = "lapsharp",
    .settings          = NULL,
    .init              = lapsharp_hip_init,
    .work              = lapsharp_hip_work,
    .close             = lapsharp_hip_close,
    .settings_template = lapsharp_hip_template,
};

static int kernel_id(const char *s)
{
    static const char *names[4] = { "lap", "isolap", "log", "isolog" };   /* lapsharp.c:237-252 */
    if (s == NULL) return -1;
    for (int i = 0; i < 4; i++)
        if (!strcasecmp(s, names[i])) return i;
    return -1;
}

static int lapsharp_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init)
{
    hb_filter_private_t *pv = calloc(1, sizeof(*pv));
    if (pv == NULL) return -1;
    filter->private_data = pv;
    pv->input = *init;
    pv->dev_io = hbhip_host_dev_io(init);

    const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(init->pix_fmt);
    if (desc == NULL) goto fail;

    static const char *pfx[3] = { "y", "cb", "cr" };
    char key[32];
    for (int c = 0; c < 3; c++)
    {
        char *ks = NULL;
        pv->par.strength[c] = -1;
        pv->par.kernel[c] = -1;
        if (filter->settings != NULL)
        {
            snprintf(key, sizeof(key), "%s-strength", pfx[c]);
            hb_dict_extract_double(&pv->par.strength[c], filter->settings, key);
            snprintf(key, sizeof(key), "%s-kernel", pfx[c]);
            hb_dict_extract_string(&ks, filter->settings, key);
        }
        pv->par.kernel[c] = kernel_id(ks);
        free(ks);
    }
    for (int c = 1; c < 3; c++)
    {
        if (pv->par.strength[c] == -1) pv->par.strength[c] = pv->par.strength[c - 1];
        if (pv->par.kernel[c] == -1)   pv->par.kernel[c]   = pv->par.kernel[c - 1];
    }
    for (int c = 0; c < 3; c++)
    {
        if (pv->par.strength[c] == -1) pv->par.strength[c] = 0.2;          /* lapsharp.c:12-13 */
        if (pv->par.kernel[c] == -1)   pv->par.kernel[c] = 2;              /* LAPSHARP_KERNEL_*_DEFAULT, :16-17 */
        if (pv->par.strength[c] < 0)   pv->par.strength[c] = 0;
        if (pv->par.strength[c] > 1.5) pv->par.strength[c] = 1.5;
        if (pv->par.kernel[c] < 0 || pv->par.kernel[c] >= 4) pv->par.kernel[c] = 2;
    }

    hbhip_ctx *ctx = hbhip_host_ctx_for(init);
    if (ctx == NULL) goto fail;
    int rc = hbhip_lapsharp_create(ctx, &pv->par, init->geometry.width, init->geometry.height,
                                   desc->comp[0].depth, desc->log2_chroma_w, desc->log2_chroma_h, &pv->dev);
    if (rc != HBHIP_OK)
    {
        hb_error("lapsharp(hip): %s", hbhip_strerror(rc));
        goto fail;
    }
    pv->output = *init;
    return 0;
fail:
    free(pv);
    filter->private_data = NULL;
    return -1;
}

static void lapsharp_hip_close(hb_filter_object_t *filter)
{
    hb_filter_private_t *pv = filter->private_data;
    if (pv == NULL) return;
    hbhip_host_simple_destroy(pv->dev);
    free(pv);
    filter->private_data = NULL;
}

static int lapsharp_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out)
{
    hb_filter_private_t *pv = filter->private_data;
    if (!((*buf_in)->s.flags & HB_BUF_FLAG_EOF) && hbhip_host_frame_of(*buf_in) == NULL)
        hb_frame_buffer_mirror_stride(*buf_in);                             /* lapsharp.c:333 */
    return hbhip_host_simple_work(pv->dev, &pv->output, "lapsharp", pv->dev_io, buf_in, buf_out);
}
