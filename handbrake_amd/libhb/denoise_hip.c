/* denoise_hip.c — HIP-backed drop-in for libhb's hqdn3d filter object
 * (libhb/denoise.c:59-76 template/object, :203-265 init, :315-371 work).
 * Same keys and the same defaulting chain of strengths (:228-256); the six LUTs are
 * built here with host libm exactly as hqdn3d_precalc_coef does (:78-94) and handed
 * to the device.  Pixel work: csrc/hqdn3d.hip. */
#include "hbhip_host.h"

#define HQDN3D_LUT_BITS 4

struct hb_filter_private_s
{
    hbhip_hqdn3d_params par;
    hbhip_filter       *dev;
    hb_filter_init_t    input;
    hb_filter_init_t    output;
    int                 dev_io;
};

static int  denoise_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init);
static int  denoise_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out);
static void denoise_hip_close(hb_filter_object_t *filter);

static const char denoise_hip_template[] =
    "y-spatial=^"HB_FLOAT_REG"$:cb-spatial=^"HB_FLOAT_REG"$:"
    "cr-spatial=^"HB_FLOAT_REG"$:"
    "y-temporal=^"HB_FLOAT_REG"$:cb-temporal=^"HB_FLOAT_REG"$:"
    "cr-temporal=^"HB_FLOAT_REG"$";

hb_filter_object_t hb_filter_denoise_hip =
{
    .id                = HB_FILTER_DENOISE,
    .enforce_order     = 1,
    .name              = "Denoise (hqdn3d, HIP)",
    .short_name        = "hqdn3d",
    .settings          = NULL,
    .init              = denoise_hip_init,
    .work              = denoise_hip_work,
    .close             = denoise_hip_close,
    .settings_template = denoise_hip_template,
};

/* denoise.c:78-94; LUT_BITS is 4 for every depth below 16 (denoise.c:31), so the table is the
 * same for 8, 10 and 12-bit samples */
static void precalc_coef(int16_t *ct, double dist25)
{
    const double gamma = log(0.25) / log(1.0 - FFMIN(dist25, 252.0) / 255.0 - 0.00001);
    for (int i = -(256 << HQDN3D_LUT_BITS); i < 256 << HQDN3D_LUT_BITS; i++)
    {
        const double f = (i * (1 << (9 - HQDN3D_LUT_BITS)) + (1 << (8 - HQDN3D_LUT_BITS)) - 1) / 512.0;
        const double simil = FFMAX(0, 1.0 - fabs(f) / 255.0);
        ct[(256 << HQDN3D_LUT_BITS) + i] = lrint(pow(simil, gamma) * 256.0 * f);
    }
    ct[0] = !!dist25;
}

static int denoise_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init)
{
    hb_filter_private_t *pv = calloc(1, sizeof(*pv));
    if (pv == NULL) return -1;
    filter->private_data = pv;
    pv->input = *init;
    pv->dev_io = hbhip_host_dev_io(init);

    const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(init->pix_fmt);
    if (desc == NULL || (desc->comp[0].depth != 8 && desc->comp[0].depth != 10 && desc->comp[0].depth != 12)) goto fail;

    double sy, scb, scr, ty, tcb, tcr;                      /* denoise.c:228-256 */
    if (!hb_dict_extract_double(&sy, filter->settings, "y-spatial"))    sy = 4.0;
    if (!hb_dict_extract_double(&scb, filter->settings, "cb-spatial"))  scb = 3.0 * sy / 4.0;
    if (!hb_dict_extract_double(&scr, filter->settings, "cr-spatial"))  scr = scb;
    if (!hb_dict_extract_double(&ty, filter->settings, "y-temporal"))   ty = 6.0 * sy / 4.0;
    if (!hb_dict_extract_double(&tcb, filter->settings, "cb-temporal")) tcb = ty * scb / sy;
    if (!hb_dict_extract_double(&tcr, filter->settings, "cr-temporal")) tcr = tcb;
    precalc_coef(pv->par.coef[0], sy);
    precalc_coef(pv->par.coef[1], ty);
    precalc_coef(pv->par.coef[2], scb);
    precalc_coef(pv->par.coef[3], tcb);
    precalc_coef(pv->par.coef[4], scr);
    precalc_coef(pv->par.coef[5], tcr);

    hbhip_ctx *ctx = hbhip_host_ctx_for(init);
    if (ctx == NULL) goto fail;
    int rc = hbhip_hqdn3d_create(ctx, &pv->par, init->geometry.width, init->geometry.height,
                                 desc->comp[0].depth, desc->log2_chroma_w, desc->log2_chroma_h, &pv->dev);
    if (rc != HBHIP_OK)
    {
        hb_error("hqdn3d(hip): %s", hbhip_strerror(rc));
        goto fail;
    }
    pv->output = *init;
    return 0;
fail:
    free(pv);
    filter->private_data = NULL;
    return -1;
}

static void denoise_hip_close(hb_filter_object_t *filter)
{
    hb_filter_private_t *pv = filter->private_data;
    if (pv == NULL) return;
    hbhip_host_simple_destroy(pv->dev);
    free(pv);
    filter->private_data = NULL;
}

static int denoise_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out)
{
    hb_filter_private_t *pv = filter->private_data;
    return hbhip_host_simple_work(pv->dev, &pv->output, "hqdn3d", pv->dev_io, buf_in, buf_out);
}
