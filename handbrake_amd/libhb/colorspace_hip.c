/* colorspace_hip.c — HIP-backed drop-in for hb_filter_colorspace (libhb/colorspace.c:15-207).
 *
 * In the reference this object has .skip = 1: colorspace_init only assembles settings for
 * FFmpeg's zscale / format / tonemap, which hb_avfilter_combine folds into HB_FILTER_AVFILTER.
 * Here it is a real filter (own work()), like the reference's HB_FILTER_*_VT variants, and has
 * to be left out of hb_avfilter_combine's switch.  Same settings keys, same
 * early-outs (nothing asked for / nothing changes => the filter does nothing), same rewrite of
 * init->color_* for the filters downstream.  Pixel arithmetic: csrc/colorspace.hip, pinned to
 * oracle/colorspace_oracle.c only ("parity unpinned").
 */
#include "hbhip_host.h"

#include <math.h>
#include <string.h>

struct hb_filter_private_s
{
    hbhip_filter    *dev;         /* NULL: pass-through (colorspace.c:87-90, 122-126) */
    hb_filter_init_t input;
    hb_filter_init_t output;
    int              dev_io;
};

static int colorspace_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init);
static int colorspace_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out);
static void colorspace_hip_close(hb_filter_object_t *filter);

static const char colorspace_hip_template[] =
    "primaries=^"HB_ALL_REG"$:transfer=^"HB_ALL_REG"$:matrix=^"HB_ALL_REG"$:range=^"HB_ALL_REG"$:"
    "tonemap=^"HB_ALL_REG"$:param=^"HB_FLOAT_REG"$:desat=^"HB_FLOAT_REG"$:npl=^"HB_FLOAT_REG"$";

hb_filter_object_t hb_filter_colorspace_hip =
{
    .id                = HB_FILTER_COLORSPACE,
    .enforce_order     = 1,
    .name              = "Colorspace (HIP)",
    .short_name        = "colorspace",
    .settings          = NULL,
    .init              = colorspace_hip_init,
    .work              = colorspace_hip_work,
    .close             = colorspace_hip_close,
    .settings_template = colorspace_hip_template,
};

/* libavutil's names (av_color_primaries_from_name & co., pixdesc.c), by AVCOL_* number */
typedef struct { const char *name; int id; } name_id_t;
static const name_id_t primaries_names[] = {
    { "bt709", 1 }, { "unknown", 2 }, { "bt470m", 4 }, { "bt470bg", 5 }, { "smpte170m", 6 }, { "smpte240m", 7 },
    { "film", 8 }, { "bt2020", 9 }, { "smpte428", 10 }, { "smpte431", 11 }, { "smpte432", 12 }, { "ebu3213", 22 },
    { NULL, 0 } };
static const name_id_t transfer_names[] = {
    { "bt709", 1 }, { "unknown", 2 }, { "gamma22", 4 }, { "gamma28", 5 }, { "smpte170m", 6 }, { "smpte240m", 7 },
    { "linear", 8 }, { "log100", 9 }, { "log316", 10 }, { "iec61966-2-4", 11 }, { "bt1361e", 12 },
    { "iec61966-2-1", 13 }, { "bt2020-10", 14 }, { "bt2020-12", 15 }, { "smpte2084", 16 }, { "smpte428", 17 },
    { "arib-std-b67", 18 }, { NULL, 0 } };
static const name_id_t matrix_names[] = {
    { "gbr", 0 }, { "bt709", 1 }, { "unknown", 2 }, { "fcc", 4 }, { "bt470bg", 5 }, { "smpte170m", 6 },
    { "smpte240m", 7 }, { "ycgco", 8 }, { "bt2020nc", 9 }, { "bt2020c", 10 }, { "smpte2085", 11 },
    { "chroma-derived-nc", 12 }, { "chroma-derived-c", 13 }, { "ictcp", 14 }, { NULL, 0 } };
static const name_id_t range_names[] = {
    { "unknown", 0 }, { "tv", 1 }, { "pc", 2 }, { "mpeg", 1 }, { "jpeg", 2 }, { "limited", 1 }, { "full", 2 },
    { NULL, 0 } };
static const name_id_t tonemap_names[] = {
    { "none", HBHIP_TONEMAP_NONE }, { "linear", HBHIP_TONEMAP_LINEAR }, { "gamma", HBHIP_TONEMAP_GAMMA },
    { "clip", HBHIP_TONEMAP_CLIP }, { "reinhard", HBHIP_TONEMAP_REINHARD }, { "hable", HBHIP_TONEMAP_HABLE },
    { "mobius", HBHIP_TONEMAP_MOBIUS }, { NULL, 0 } };

/* av_color_*_from_name: negative when unknown */
static int from_name(const name_id_t *t, const char *name)
{
    for (; t->name != NULL; t++)
        if (strcmp(t->name, name) == 0)
            return t->id;
    return -1;
}

#define REFERENCE_WHITE 100.0

/* determine_signal_peak (colorspace.c:37-49) */
static double signal_peak(const hb_filter_init_t *init)
{
    double peak = 0;
#ifdef HBHIP_IN_LIBHB
    if (init->job != NULL)
    {
        peak = init->job->coll.max_cll / REFERENCE_WHITE;
        if (!peak && init->job->mastering.has_luminance)
            peak = hb_q2d(init->job->mastering.max_luminance) / REFERENCE_WHITE;
    }
#endif
    if (!peak || peak < 1)
        peak = init->color_transfer == 16 /* HB_COLR_TRA_SMPTEST2084 */ ? 100.0 : 10.0;
    return peak;
}

static int colorspace_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init)
{
    hb_filter_private_t *pv = calloc(1, sizeof(*pv));
    filter->private_data = pv;
    if (pv == NULL) return 1;
    pv->input = *init;
    pv->output = *init;
    pv->dev_io = hbhip_host_dev_io(init);

    if (init->color_prim == 2 || init->color_transfer == 2 || init->color_matrix == 2)      /* HB_COLR_*_UNDEF, :62-68 */
    {
        hb_error("colorspace(hip): input color space undefined");
        free(pv);
        filter->private_data = NULL;
        return -1;
    }

    char *range = NULL, *primaries = NULL, *transfer = NULL, *matrix = NULL, *tonemap = NULL;
    double param = 0, desat = 0, npl = 100;
    hb_dict_extract_string(&range, filter->settings, "range");                                /* :77-84 */
    hb_dict_extract_string(&primaries, filter->settings, "primaries");
    hb_dict_extract_string(&transfer, filter->settings, "transfer");
    hb_dict_extract_string(&matrix, filter->settings, "matrix");
    hb_dict_extract_string(&tonemap, filter->settings, "tonemap");
    hb_dict_extract_double(&param, filter->settings, "param");
    hb_dict_extract_double(&desat, filter->settings, "desat");
    hb_dict_extract_double(&npl, filter->settings, "npl");

    int rc = 0;
    if (range || primaries || transfer || matrix)                                             /* :87-90 */
    {
        hbhip_colorspace_params p;
        memset(&p, 0, sizeof(p));
        p.in_prim = p.out_prim = init->color_prim;
        p.in_transfer = p.out_transfer = init->color_transfer;
        p.in_matrix = p.out_matrix = init->color_matrix;
        p.in_range = p.out_range = init->color_range;
        if (primaries) p.out_prim = from_name(primaries_names, primaries);                    /* :101-120 */
        if (transfer)  p.out_transfer = from_name(transfer_names, transfer);
        if (matrix)    p.out_matrix = from_name(matrix_names, matrix);
        if (range)     p.out_range = from_name(range_names, range);

        if (p.out_prim != p.in_prim || p.out_transfer != p.in_transfer ||
            p.out_matrix != p.in_matrix || p.out_range != p.in_range)                         /* :122-126 */
        {
            /* vf_tonemap's operator; HandBrake hands `param` over only for the operators that
             * take one and only when it is non-zero (:151-157), otherwise FFmpeg's default (NAN) */
            const char *tm = tonemap != NULL ? tonemap : "hable";
            p.tonemap = from_name(tonemap_names, tm);
            p.param = (strcmp(tm, "hable") && strcmp(tm, "none") && param != 0) ? param : NAN;
            p.desat = desat;
            p.npl = npl;
            p.peak = signal_peak(init);
            const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(init->pix_fmt);
            hbhip_ctx *ctx = desc != NULL && p.tonemap >= 0 ? hbhip_host_ctx_for(init) : NULL;
            int err = ctx == NULL ? HBHIP_ERR_NODEVICE
                                  : hbhip_colorspace_create(ctx, &p, init->geometry.width, init->geometry.height,
                                                            desc->comp[0].depth, desc->log2_chroma_w,
                                                            desc->log2_chroma_h, &pv->dev);
            if (err != HBHIP_OK)
            {
                hb_error("colorspace(hip): %s", hbhip_strerror(err));
                rc = 1;
            }
            else
            {
                init->color_prim = p.out_prim;                                                /* :195-198 */
                init->color_transfer = p.out_transfer;
                init->color_matrix = p.out_matrix;
                init->color_range = p.out_range;
                pv->output = *init;
            }
        }
    }
    free(range); free(primaries); free(transfer); free(matrix); free(tonemap);
    if (rc != 0)
    {
        free(pv);
        filter->private_data = NULL;
    }
    return rc;
}

static int colorspace_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out)
{
    hb_filter_private_t *pv = filter->private_data;
    if (pv->dev == NULL)
    {
        /* nothing to convert: the reference adds no avfilter at all in this case */
        *buf_out = *buf_in;
        *buf_in = NULL;
        return ((*buf_out)->s.flags & HB_BUF_FLAG_EOF) ? HB_FILTER_DONE : HB_FILTER_OK;
    }
    const int status = hbhip_host_simple_work(pv->dev, &pv->output, filter->short_name, pv->dev_io, buf_in, buf_out);
    if (status == HB_FILTER_OK || status == HB_FILTER_DONE)
        for (hb_buffer_t *b = *buf_out; b != NULL; b = b->next)    /* a burst, or the frames an EOF drains, come as a list */
        {
            if (b->s.flags & HB_BUF_FLAG_EOF) continue;
            b->f.color_prim = pv->output.color_prim;
            b->f.color_transfer = pv->output.color_transfer;
            b->f.color_matrix = pv->output.color_matrix;
            b->f.color_range = pv->output.color_range;
        }
    return status;
}

static void colorspace_hip_close(hb_filter_object_t *filter)
{
    hb_filter_private_t *pv = filter->private_data;
    if (pv == NULL) return;
    if (pv->dev != NULL) hbhip_host_simple_destroy(pv->dev);
    free(pv);
    filter->private_data = NULL;
}
