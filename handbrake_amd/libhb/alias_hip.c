/* alias_hip.c — HIP-backed drop-ins for the reference's libavfilter "alias" filters:
 *   hb_filter_crop_scale_hip  (libhb/cropscale.c:21-185)
 *   hb_filter_grayscale_hip   (libhb/grayscale.c:13-68)
 *   hb_filter_rotate_hip      (libhb/rotate.c:15-270)
 *
 * In the reference these objects have .skip = 1 and no work(): their init() only
 * builds settings for FFmpeg's crop/zscale, monochrome and transpose/hflip/vflip,
 * which hb_avfilter_combine later merges into one HB_FILTER_AVFILTER graph
 * (hbavfilter.c:510-622).  Here they are real filters (skip = 0, own work), exactly
 * as the reference's own HB_FILTER_*_VT GPU variants are; they must therefore be
 * left out of hb_avfilter_combine's switch.  Same settings keys,
 * same mutation of init->geometry / PAR / crop for the filters downstream.
 * The pixel arithmetic is FFmpeg's / zimg's in the reference and is NOT in its tree:
 * parity is pinned to oracle/alias_oracle.c only ("parity unpinned").
 */
#include "hbhip_host.h"

struct hb_filter_private_s
{
    hbhip_filter    *dev;
    hb_filter_init_t input;
    hb_filter_init_t output;
    int              dev_io;
};

static void alias_hip_close(hb_filter_object_t *filter)
{
    hb_filter_private_t *pv = filter->private_data;
    if (pv == NULL) return;
    hbhip_host_simple_destroy(pv->dev);
    free(pv);
    filter->private_data = NULL;
}

static int alias_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out)
{
    hb_filter_private_t *pv = filter->private_data;
    return hbhip_host_simple_work(pv->dev, &pv->output, filter->short_name, pv->dev_io, buf_in, buf_out);
}

static hb_filter_private_t *alias_begin(hb_filter_object_t *filter, hb_filter_init_t *init,
                                        const AVPixFmtDescriptor **desc)
{
    hb_filter_private_t *pv = calloc(1, sizeof(*pv));
    filter->private_data = pv;
    if (pv == NULL) return NULL;
    pv->input = *init;
    pv->dev_io = hbhip_host_dev_io(init);
    *desc = av_pix_fmt_desc_get(init->pix_fmt);
    if (*desc == NULL)
    {
        free(pv);
        filter->private_data = NULL;
        return NULL;
    }
    return pv;
}

static int alias_fail(hb_filter_object_t *filter, int rc)
{
    hb_error("%s(hip): %s", filter->short_name, hbhip_strerror(rc));
    free(filter->private_data);
    filter->private_data = NULL;
    return 1;
}

/* ---- crop + scale ------------------------------------------------------------------ */
static int crop_scale_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init);

static const char crop_scale_hip_template[] =
    "width=^"HB_INT_REG"$:height=^"HB_INT_REG"$:"
    "crop-top=^"HB_INT_REG"$:crop-bottom=^"HB_INT_REG"$:"
    "crop-left=^"HB_INT_REG"$:crop-right=^"HB_INT_REG"$";

hb_filter_object_t hb_filter_crop_scale_hip =
{
    .id                = HB_FILTER_CROP_SCALE,
    .enforce_order     = 1,
    .name              = "Crop and Scale (HIP)",
    .short_name        = "cropscale",
    .settings          = NULL,
    .init              = crop_scale_hip_init,
    .work              = alias_hip_work,
    .close             = alias_hip_close,
    .settings_template = crop_scale_hip_template,
};

#ifdef HBHIP_IN_LIBHB
#define limit_rational hb_limit_rational
#else
/* hb_limit_rational (common.c:3912-3938), the same arithmetic: reduce by the gcd (hb_reduce64, :3947-3968); if a term
 * still reaches the limit, the larger one becomes the limit and the other is scaled by a double and truncated */
static void limit_rational(int *x, int *y, int64_t num, int64_t den, int limit)
{
    int64_t n = num, d = den;
    while (d) { int64_t t = d; d = n % d; n = t; }
    if (n) { num /= n; den /= n; }
    if (num < limit && den < limit)
    {
        *x = (int)num;
        *y = (int)den;
        return;
    }
    if (num > den)
    {
        double div = (double)limit / num;
        num = limit;
        den *= div;
    }
    else
    {
        double div = (double)limit / den;
        den = limit;
        num *= div;
    }
    *x = (int)num;
    *y = (int)den;
}
#endif

static int crop_scale_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init)
{
    const AVPixFmtDescriptor *desc;
    hb_filter_private_t *pv = alias_begin(filter, init, &desc);
    if (pv == NULL) return 1;

    hbhip_cropscale_params p;
    memset(&p, 0, sizeof(p));
    hb_dict_extract_int(&p.crop_top, filter->settings, "crop-top");          /* cropscale.c:68-71 */
    hb_dict_extract_int(&p.crop_bottom, filter->settings, "crop-bottom");
    hb_dict_extract_int(&p.crop_left, filter->settings, "crop-left");
    hb_dict_extract_int(&p.crop_right, filter->settings, "crop-right");
    const int cropped_width  = init->geometry.width - p.crop_left - p.crop_right;
    const int cropped_height = init->geometry.height - p.crop_top - p.crop_bottom;
    p.width = cropped_width;
    p.height = cropped_height;
    hb_dict_extract_int(&p.width, filter->settings, "width");                /* :93-94 */
    hb_dict_extract_int(&p.height, filter->settings, "height");
    /* crop_scale_init has two scalers (cropscale.c:97-165): zscale=filter=lanczos where hb_av_can_use_zscale() agrees
     * (hbffmpeg.c:870-915: every dimension even, a planar YUV format), else swscale `lanczos+accurate_rnd`, which computes
     * something else.  The drop-in follows the same rule with the restatement of the same library (both parity unpinned:
     * neither library is in the reference tree).  The swscale form covers 8-bit planes (hScale8To15 / yuv2planeX_8) and
     * 10 / 12-bit planes (hScale16To15 / yuv2planeX_10, _12). */
    const int odd = (cropped_width & 1) || (cropped_height & 1) || (p.width & 1) || (p.height & 1);
    /* The swscale restatement has never met a real libswscale (vf_scale's chroma positions from chroma_location, its
     * field handling and its cascaded contexts for large ratios are what could differ silently), and unlike the zimg form
     * it has no second implementation to be held against: it is OPT-IN (HBHIP_SWSCALE=1).  Without it an odd size
     * declines here and the job keeps the reference's CPU filter (hb_hip_filter_init_failed). */
    if (odd)
    {
        const char *sws = getenv("HBHIP_SWSCALE");
        if (sws == NULL || atoi(sws) == 0) return alias_fail(filter, HBHIP_ERR_UNSUPPORTED);
    }

    hbhip_ctx *ctx = hbhip_host_ctx_for(init);
    if (ctx == NULL) return alias_fail(filter, HBHIP_ERR_NODEVICE);
    int rc = odd ? hbhip_cropscale_sws_create(ctx, &p, init->geometry.width, init->geometry.height, desc->comp[0].depth,
                                              desc->log2_chroma_w, desc->log2_chroma_h, &pv->dev)
                 : hbhip_cropscale_create(ctx, &p, init->geometry.width, init->geometry.height, desc->comp[0].depth,
                                          desc->log2_chroma_w, desc->log2_chroma_h, &pv->dev);
    if (rc != HBHIP_OK) return alias_fail(filter, rc);

    init->crop[0] = p.crop_top;                                              /* :168-178 */
    init->crop[1] = p.crop_bottom;
    init->crop[2] = p.crop_left;
    init->crop[3] = p.crop_right;
    limit_rational(&init->geometry.par.num, &init->geometry.par.den,
                   (int64_t)init->geometry.par.num * p.height * cropped_width,
                   (int64_t)init->geometry.par.den * p.width * cropped_height, 65535);
    init->geometry.width = p.width;
    init->geometry.height = p.height;
    pv->output = *init;
    return 0;
}

/* ---- grayscale ----------------------------------------------------------------------- */
static int grayscale_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init);

static const char grayscale_hip_template[] =
    "cb=^"HB_FLOAT_REG"$:cr=^"HB_FLOAT_REG"$:size=^"HB_FLOAT_REG"$:high=^"HB_FLOAT_REG"$";

hb_filter_object_t hb_filter_grayscale_hip =
{
    .id                = HB_FILTER_GRAYSCALE,
    .enforce_order     = 1,
    .name              = "Grayscale (HIP)",
    .short_name        = "grayscale",
    .settings          = NULL,
    .init              = grayscale_hip_init,
    .work              = alias_hip_work,
    .close             = alias_hip_close,
    .settings_template = grayscale_hip_template,
};

static int grayscale_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init)
{
    const AVPixFmtDescriptor *desc;
    hb_filter_private_t *pv = alias_begin(filter, init, &desc);
    if (pv == NULL) return 1;
    double cb = 0, cr = 0, size = 1, high = 0;                               /* grayscale.c:43-48 */
    hb_dict_extract_double(&cb, filter->settings, "cb");
    hb_dict_extract_double(&cr, filter->settings, "cr");
    hb_dict_extract_double(&size, filter->settings, "size");
    hb_dict_extract_double(&high, filter->settings, "high");
    hbhip_ctx *ctx = hbhip_host_ctx_for(init);
    if (ctx == NULL) return alias_fail(filter, HBHIP_ERR_NODEVICE);
    int rc = hbhip_grayscale_create(ctx, cb, cr, size, high, init->geometry.width, init->geometry.height,
                                    desc->comp[0].depth, desc->log2_chroma_w, desc->log2_chroma_h, &pv->dev);
    if (rc != HBHIP_OK) return alias_fail(filter, rc);
    pv->output = *init;
    return 0;
}

/* ---- rotate -------------------------------------------------------------------------- */
static int rotate_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init);

static const char rotate_hip_template[] =
    "angle=^(0|90|180|270)$:hflip=^"HB_BOOL_REG"$:disable=^"HB_BOOL_REG"$";

hb_filter_object_t hb_filter_rotate_hip =
{
    .id                = HB_FILTER_ROTATE,
    .enforce_order     = 1,
    .name              = "Rotate (HIP)",
    .short_name        = "rotate",
    .settings          = NULL,
    .init              = rotate_hip_init,
    .work              = alias_hip_work,
    .close             = alias_hip_close,
    .settings_template = rotate_hip_template,
};

static int rotate_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init)
{
    const AVPixFmtDescriptor *desc;
    hb_filter_private_t *pv = alias_begin(filter, init, &desc);
    if (pv == NULL) return 1;
    int angle = 0, flip = 0;
    hb_dict_extract_int(&angle, filter->settings, "angle");                  /* rotate.c:166-167 */
    hb_dict_extract_bool(&flip, filter->settings, "hflip");
    hbhip_ctx *ctx = hbhip_host_ctx_for(init);
    if (ctx == NULL) return alias_fail(filter, HBHIP_ERR_NODEVICE);
    int rc = hbhip_rotate_create(ctx, angle, flip, init->geometry.width, init->geometry.height,
                                 desc->comp[0].depth, desc->log2_chroma_w, desc->log2_chroma_h, &pv->dev);
    if (rc != HBHIP_OK) return alias_fail(filter, rc);
    if (angle == 90 || angle == 270)                                         /* rotate.c:195-214, 261-263 */
    {
        const int w = init->geometry.width, n = init->geometry.par.num;
        init->geometry.width = init->geometry.height;
        init->geometry.height = w;
        init->geometry.par.num = init->geometry.par.den;
        init->geometry.par.den = n;
    }
    pv->output = *init;
    return 0;
}

/* ---- format (libhb/format.c:13-111) -------------------------------------------------------------
 * In the reference this object has .skip = 1: format_init only writes `format=pix_fmts=<name>` for
 * libavfilter (:97-100) and sets init->pix_fmt (:107); the conversion itself is the `scale` filter
 * libavfilter auto-inserts, i.e. libswscale.  work.c adds it when the encoder wants another pixel format
 * than the pipeline's (work.c:1530-1549), typically 8 <-> 10 bits.  Here it is a real filter for what that
 * use needs - planar YUV depth changes with the subsampling unchanged (csrc/alias.hip:format_kernel,
 * parity unpinned); any other target (different subsampling, semi-planar, RGB) makes init() fail, which
 * keeps the CPU filter (work.c:1861-1868). */
static int format_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init);
static int format_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out);

static const char format_hip_template[] = "format=^"HB_ALL_REG"$";

hb_filter_object_t hb_filter_format_hip =
{
    .id                = HB_FILTER_FORMAT,
    .enforce_order     = 1,
    .name              = "Format (HIP)",
    .short_name        = "format",
    .settings          = NULL,
    .init              = format_hip_init,
    .work              = format_hip_work,
    .close             = alias_hip_close,
    .settings_template = format_hip_template,
};

static int format_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init)
{
    const AVPixFmtDescriptor *desc;
    hb_filter_private_t *pv = alias_begin(filter, init, &desc);
    if (pv == NULL) return 1;
    char *format = NULL;
    hb_dict_extract_string(&format, filter->settings, "format");              /* format.c:46-52 */
    if (format == NULL)
    {
        pv->output = *init;                                                   /* nothing to do: frames pass through */
        return 0;
    }
    const int dst_fmt = av_get_pix_fmt(format);                               /* :107 */
    free(format);
    const AVPixFmtDescriptor *dd = av_pix_fmt_desc_get(dst_fmt);
    if (dd == NULL || dd->nb_components != desc->nb_components || desc->nb_components < 3 ||
        dd->log2_chroma_w != desc->log2_chroma_w || dd->log2_chroma_h != desc->log2_chroma_h)
        return alias_fail(filter, HBHIP_ERR_UNSUPPORTED);
    hbhip_ctx *ctx = hbhip_host_ctx_for(init);
    if (ctx == NULL) return alias_fail(filter, HBHIP_ERR_NODEVICE);
    int rc = hbhip_format_create(ctx, init->geometry.width, init->geometry.height, desc->comp[0].depth,
                                 dd->comp[0].depth, desc->log2_chroma_w, desc->log2_chroma_h,
                                 init->color_range == 2 /* AVCOL_RANGE_JPEG */, &pv->dev);
    if (rc != HBHIP_OK) return alias_fail(filter, rc);
    init->pix_fmt = dst_fmt;
    pv->output = *init;
    return 0;
}

static int format_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out)
{
    hb_filter_private_t *pv = filter->private_data;
    if (pv->dev == NULL)
    {
        *buf_out = *buf_in;
        *buf_in = NULL;
        return ((*buf_out)->s.flags & HB_BUF_FLAG_EOF) ? HB_FILTER_DONE : HB_FILTER_OK;
    }
    return hbhip_host_simple_work(pv->dev, &pv->output, filter->short_name, pv->dev_io, buf_in, buf_out);
}
