/* comb_detect_hip.c — HIP-backed drop-in for libhb's comb-detect filter object
 * (libhb/comb_detect.c:122-140 template/object, :1083-1450 init, :1537-1583 work,
 * :1452-1497 close).  Same keys and code defaults (mode 3, spatial-metric 2,
 * motion/spatial thresholds 3, filter-mode 2, block 40/16x16).
 *
 * The filter does not change pixels: it stamps s.combed on the frame in the
 * middle of a prev/cur/next window and forwards the very same hb_buffer_t
 * (comb_detect.c:1529-1531), holding output back until more than three buffers
 * are queued (:1579-1582).  Only the luma plane goes to the GPU.
 *
 * The debugging modes 4 (MODE_MASK) and 8 (MODE_COMPOSITE) do change pixels: a combed frame leaves as a
 * copy with the combing mask drawn on it (process_frame :1519-1526, comb_detect_template.c:21-136).  The
 * reference lets its check threads race on the box position (:205-208); here it is the one a single check
 * thread leaves.  The reference copies with hb_buffer_shallow_dup, which for decoder-backed (AVFRAME) buffers
 * shares the pixels it then draws on; the copy here is always a private one (what the reference does for
 * STANDARD buffers).
 */
#include "hbhip_host.h"

struct hb_filter_private_s
{
    hbhip_comb_detect_params par;
    hbhip_filter            *dev;
    hb_buffer_t             *ref[3];
    int                      ref_used[3];
    int                      ready;
    int                      force_exhaustive;
    int                      heavy, light, none, frames;
    hb_buffer_list_t         out_list;
    hb_filter_init_t         input;
};

static int  comb_detect_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init);
static int  comb_detect_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out);
static void comb_detect_hip_close(hb_filter_object_t *filter);

static const char comb_detect_hip_template[] =
    "mode=^"HB_INT_REG"$:spatial-metric=^([012])$:"
    "motion-thresh=^"HB_INT_REG"$:spatial-thresh=^"HB_INT_REG"$:"
    "filter-mode=^([012])$:block-thresh=^"HB_INT_REG"$:"
    "block-width=^"HB_INT_REG"$:block-height=^"HB_INT_REG"$:"
    "disable=^"HB_BOOL_REG"$";

hb_filter_object_t hb_filter_comb_detect_hip =
{
    .id                = HB_FILTER_COMB_DETECT,
    .enforce_order     = 1,
    .name              = "Comb Detect (HIP)",
    .short_name        = "combdetect",
    .settings          = NULL,
    .init              = comb_detect_hip_init,
    .work              = comb_detect_hip_work,
    .close             = comb_detect_hip_close,
    .settings_template = comb_detect_hip_template,
};

static int comb_detect_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init)
{
    hb_filter_private_t *pv = calloc(1, sizeof(*pv));
    if (pv == NULL) return -1;
    filter->private_data = pv;
    hb_buffer_list_clear(&pv->out_list);
    pv->input = *init;

    const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(init->pix_fmt);
    if (desc == NULL) goto fail;
    const int depth = desc->comp[0].depth;
    const int max_value = (1 << depth) - 1;

    hbhip_comb_detect_params *p = &pv->par;
    p->mode = 1 | 2;                       /* comb_detect.c:1118-1125 */
    p->filter_mode = 2;
    p->spatial_metric = 2;
    p->motion_threshold = 3;
    p->spatial_threshold = 3;
    p->block_threshold = 40;
    p->block_width = 16;
    p->block_height = 16;
    if (filter->settings != NULL)
    {
        hb_dict_t *d = filter->settings;
        hb_dict_extract_int(&p->mode, d, "mode");
        hb_dict_extract_int(&p->spatial_metric, d, "spatial-metric");
        hb_dict_extract_int(&p->motion_threshold, d, "motion-thresh");
        hb_dict_extract_int(&p->spatial_threshold, d, "spatial-thresh");
        hb_dict_extract_int(&p->filter_mode, d, "filter-mode");
        hb_dict_extract_int(&p->block_threshold, d, "block-thresh");
        hb_dict_extract_int(&p->block_width, d, "block-width");
        hb_dict_extract_int(&p->block_height, d, "block-height");
    }
    float *wide_lut = NULL;                /* comb_detect.c:1074-1081, host libm; 1 << depth entries */
    if (depth == 8)
    {
        for (int i = 0; i < 256; i++)
            p->gamma_lut[i] = pow(((float)i / (float)max_value), 2.2f);
    }
    else
    {
        wide_lut = malloc(sizeof(float) * (max_value + 1));
        if (wide_lut == NULL) goto fail;
        for (int i = 0; i <= max_value; i++)
            wide_lut[i] = pow(((float)i / (float)max_value), 2.2f);
    }

    pv->force_exhaustive = 1;              /* :1111 */
    hbhip_ctx *ctx = hbhip_host_ctx_for_role(init, 1);
    if (ctx == NULL) { free(wide_lut); goto fail; }
    int rc = hbhip_comb_detect_create(ctx, p, init->geometry.width, init->geometry.height, depth, &pv->dev);
    if (rc == HBHIP_OK && wide_lut != NULL)
        rc = hbhip_comb_detect_set_gamma_lut(pv->dev, wide_lut, max_value + 1);
    free(wide_lut);
    if (rc != HBHIP_OK)
    {
        hb_error("comb_detect(hip): %s", hbhip_strerror(rc));
        goto fail;
    }
    return 0;
fail:
    free(pv);
    filter->private_data = NULL;
    return -1;
}

static void comb_detect_hip_close(hb_filter_object_t *filter)
{
    hb_filter_private_t *pv = filter->private_data;
    if (pv == NULL) return;
    hb_log("comb detect(hip): heavy %i | light %i | uncombed %i | total %i",
           pv->heavy, pv->light, pv->none, pv->frames);
    hbhip_filter_destroy(pv->dev);
    hb_buffer_list_close(&pv->out_list);
    for (int i = 0; i < 3; i++)
        if (!pv->ref_used[i])
            hb_buffer_close(&pv->ref[i]);
    free(pv);
    filter->private_data = NULL;
}

/* comb_detect.c:1007-1018; dev_luma mirrors the host ring on the device */
static int store_ref(hb_filter_private_t *pv, hb_buffer_t *b, int repeat)
{
    if (!pv->ref_used[0])
        hb_buffer_close(&pv->ref[0]);
    memmove(&pv->ref[0], &pv->ref[1], sizeof(pv->ref[0]) * 2);
    memmove(&pv->ref_used[0], &pv->ref_used[1], sizeof(pv->ref_used[0]) * 2);
    pv->ref[2] = b;
    pv->ref_used[2] = 0;
    int rc;
    hbhip_frame *fr = hbhip_host_frame_of(b);
    if (repeat)
        rc = hbhip_comb_detect_store(pv->dev, NULL, 0);
    else if (fr != NULL)
    {
        hbhip_dev_frame d;
        hbhip_frame_describe(fr, &d, NULL, NULL);
        rc = hbhip_frame_use_on(fr, hbhip_filter_context(pv->dev));
        if (rc == HBHIP_OK) rc = hbhip_comb_detect_store_dev(pv->dev, d.plane[0], d.stride[0]);
    }
    else
        rc = hbhip_comb_detect_store(pv->dev, b->plane[0].data, b->plane[0].stride);
    if (rc != HBHIP_OK)
        hb_error("comb_detect(hip): store: %s", hbhip_strerror(rc));
    return rc;
}

/* a private copy of `src` with the mask drawn on it (apply_mask, comb_detect_template.c:72-136) */
static hb_buffer_t *overlay_copy(hb_filter_private_t *pv, hb_buffer_t *src)
{
    int pw[3], ph[3];
    for (int p = 0; p < 3; p++)
    {
        pw[p] = hb_image_width(pv->input.pix_fmt, src->f.width, p);
        ph[p] = hb_image_height(pv->input.pix_fmt, src->f.height, p);
    }
    hbhip_frame *fr = hbhip_host_frame_of(src);
    if (fr != NULL)
    {
        const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(pv->input.pix_fmt);
        hbhip_frame *dst = NULL;
        if (desc == NULL) return NULL;
        const int arc = hbhip_frame_alloc(hbhip_frame_context(fr), src->f.width, src->f.height, desc->comp[0].depth,
                                          desc->log2_chroma_w, desc->log2_chroma_h, &dst);
        if (arc != HBHIP_OK) return NULL;
        hbhip_dev_frame d;
        hbhip_frame_describe(dst, &d, NULL, NULL);
        /* (the copy is made on the frame's context, the mask drawn on the filter's: ordered behind the copy) */
        if (hbhip_frame_copy(dst, fr) != HBHIP_OK || hbhip_frame_use_on(dst, hbhip_filter_context(pv->dev)) != HBHIP_OK ||
            hbhip_comb_detect_overlay_dev(pv->dev, &d, pw, ph) != HBHIP_OK)
        {
            hbhip_frame_release(dst);
            return NULL;
        }
        hb_buffer_t *out = hbhip_host_wrap_frame(dst, &pv->input, src->f.width, src->f.height);
        if (out != NULL) hb_buffer_copy_props(out, src);
        return out;
    }
    hb_buffer_t *out = hb_buffer_dup(src);
    if (out == NULL) return NULL;
    hbhip_host_frame hf;
    hbhip_host_frame_from_buf(&hf, out);
    if (hbhip_comb_detect_overlay(pv->dev, &hf, pw, ph) != HBHIP_OK)
    {
        hb_buffer_close(&out);
        return NULL;
    }
    return out;
}

static int process_frame(hb_filter_private_t *pv)      /* comb_detect.c:1499-1535 */
{
    int combed = HB_COMB_NONE;
    int rc = hbhip_comb_detect_classify(pv->dev, pv->force_exhaustive, &combed);
    if (rc != HBHIP_OK)
    {
        hb_error("comb_detect(hip): classify: %s", hbhip_strerror(rc));
        return -1;
    }
    if (combed == HB_COMB_HEAVY) pv->heavy++;
    else if (combed == HB_COMB_LIGHT) pv->light++;
    else pv->none++;
    pv->frames++;
    if ((pv->par.mode & (4 | 8)) && combed)             /* MODE_MASK / MODE_COMPOSITE, :1519-1526 */
    {
        hb_buffer_t *out = overlay_copy(pv, pv->ref[1]);
        if (out == NULL)
        {
            hb_error("comb_detect(hip): mask overlay failed");
            return -1;
        }
        out->s.combed = combed;
        hb_buffer_list_append(&pv->out_list, out);
    }
    else
    {
        pv->ref_used[1] = 1;
        pv->ref[1]->s.combed = combed;
        hb_buffer_list_append(&pv->out_list, pv->ref[1]);
    }
    pv->force_exhaustive = 0;
    return 0;
}

static int comb_detect_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out)
{
    hb_filter_private_t *pv = filter->private_data;
    hb_buffer_t *in = *buf_in;
    *buf_in = NULL;                                     /* input is always consumed */

    if (in->s.flags & HB_BUF_FLAG_EOF)
    {
        /* repeat the last frame as its own successor (comb_detect.c:1548-1555) */
        if (pv->ref[2] != NULL)
        {
            if (store_ref(pv, hb_buffer_shallow_dup(pv->ref[2]), 1) != HBHIP_OK)
            {
                hb_buffer_close(&in);                   /* consumed above: nobody else will */
                return HB_FILTER_FAILED;
            }
            if (pv->ref[0] != NULL)
            {
                pv->force_exhaustive = 1;
                if (process_frame(pv) != 0)
                {
                    hb_buffer_close(&in);
                    return HB_FILTER_FAILED;
                }
            }
        }
        hb_buffer_list_append(&pv->out_list, in);
        *buf_out = hb_buffer_list_clear(&pv->out_list);
        return HB_FILTER_DONE;
    }

    if (!pv->ready)
    {
        /* no previous frame yet: the first frame stands in for it (:1561-1569) */
        if (store_ref(pv, hb_buffer_shallow_dup(in), 0) != HBHIP_OK ||
            store_ref(pv, in, 1) != HBHIP_OK)
            return HB_FILTER_FAILED;
        pv->ready = 1;
        return HB_FILTER_DELAY;
    }

    if (store_ref(pv, in, 0) != HBHIP_OK || process_frame(pv) != 0)
        return HB_FILTER_FAILED;

    if (hb_buffer_list_count(&pv->out_list) > 3)       /* :1579-1582 */
        *buf_out = hb_buffer_list_rem_head(&pv->out_list);
    return HB_FILTER_OK;
}
