/* decomb_hip.c — HIP-backed drop-in for libhb's decomb filter object
 * (libhb/decomb.c:175-193 template/object, :210-441 init, :573-612 work,
 * :443-493 close).  Same keys and defaults (mode 7, EEDI2 thresholds
 * 10/20/20/4/2/50/24/1, parity -1); bob doubles init->vrate.num (:427-430) and the
 * two output frames of a bob pair split the input's time span (:562-569).
 *
 * The frame logic that needs pixels (prev/cur/next ring, selective pass-through,
 * LIGHT -> blend, per-field EEDI2) lives with the pixels in csrc/decomb.hip; this
 * file keeps what is about hb_buffer_t: properties, timestamps, EOF, DELAY.
 */
#include "hbhip_host.h"

struct hb_filter_private_s
{
    hbhip_decomb_params par;
    hbhip_filter       *dev;
    hb_buffer_list_t    props;       /* per queued input frame: its `s` */
    int64_t             next_tag;
    int                 ready;
    int                 dev_io;
    hb_filter_init_t    input;
    hb_filter_init_t    output;
};

static int  decomb_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init);
static int  decomb_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out);
static void decomb_hip_close(hb_filter_object_t *filter);

static const char decomb_hip_template[] =
    "mode=^"HB_INT_REG"$:"
    "magnitude-thresh=^"HB_INT_REG"$:variance-thresh=^"HB_INT_REG"$:"
    "laplacian-thresh=^"HB_INT_REG"$:dilation-thresh=^"HB_INT_REG"$:"
    "erosion-thresh=^"HB_INT_REG"$:noise-thresh=^"HB_INT_REG"$:"
    "search-distance=^"HB_INT_REG"$:postproc=^([0-3])$:parity=^([01])$";

hb_filter_object_t hb_filter_decomb_hip =
{
    .id                = HB_FILTER_DECOMB,
    .enforce_order     = 1,
    .name              = "Decomb (HIP)",
    .short_name        = "decomb",
    .settings          = NULL,
    .init              = decomb_hip_init,
    .work              = decomb_hip_work,
    .close             = decomb_hip_close,
    .settings_template = decomb_hip_template,
};

#define DECOMB_EEDI2 8
#define DECOMB_BOB   16

static int decomb_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init)
{
    hb_filter_private_t *pv = calloc(1, sizeof(*pv));
    if (pv == NULL) return -1;
    filter->private_data = pv;
    pv->input = *init;
    pv->dev_io = hbhip_host_dev_io(init);
    hb_buffer_list_clear(&pv->props);

    const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(init->pix_fmt);
    if (desc == NULL) goto fail;

    hbhip_decomb_params *p = &pv->par;
    p->mode = 1 | 2 | 4;                       /* decomb.c:233 */
    p->magnitude_threshold = 10;
    p->variance_threshold = 20;
    p->laplacian_threshold = 20;
    p->dilation_threshold = 4;
    p->erosion_threshold = 2;
    p->noise_threshold = 50;
    p->maximum_search_distance = 24;
    p->post_processing = 1;
    p->parity = -1;
    if (filter->settings != NULL)
    {
        hb_dict_t *d = filter->settings;
        hb_dict_extract_int(&p->mode, d, "mode");
        hb_dict_extract_int(&p->parity, d, "parity");
        if (p->mode & DECOMB_EEDI2)            /* decomb.c:254-272 */
        {
            hb_dict_extract_int(&p->magnitude_threshold, d, "magnitude-thresh");
            hb_dict_extract_int(&p->variance_threshold, d, "variance-thresh");
            hb_dict_extract_int(&p->laplacian_threshold, d, "laplacian-thresh");
            hb_dict_extract_int(&p->dilation_threshold, d, "dilation-thresh");
            hb_dict_extract_int(&p->erosion_threshold, d, "erosion-thresh");
            hb_dict_extract_int(&p->noise_threshold, d, "noise-thresh");
            hb_dict_extract_int(&p->maximum_search_distance, d, "search-distance");
            hb_dict_extract_int(&p->post_processing, d, "postproc");
        }
    }

    hbhip_ctx *ctx = hbhip_host_ctx();
    if (ctx == NULL) goto fail;
    int rc = hbhip_decomb_create(ctx, p, init->geometry.width, init->geometry.height,
                                 desc->comp[0].depth, desc->log2_chroma_w, desc->log2_chroma_h, &pv->dev);
    if (rc != HBHIP_OK)
    {
        hb_error("decomb(hip): %s", hbhip_strerror(rc));
        goto fail;
    }
    if (p->mode & DECOMB_BOB)
        init->vrate.num *= 2;                  /* decomb.c:427-430 */
    pv->output = *init;
    return 0;
fail:
    free(pv);
    filter->private_data = NULL;
    return -1;
}

static void decomb_hip_close(hb_filter_object_t *filter)
{
    hb_filter_private_t *pv = filter->private_data;
    if (pv == NULL) return;
    hbhip_filter_destroy(pv->dev);
    hb_buffer_list_close(&pv->props);
    free(pv);
    filter->private_data = NULL;
}

/* Pull the frames made from ONE input frame: the head of pv->props describes it. */
static int decomb_hip_collect(hb_filter_private_t *pv, hb_buffer_list_t *list)
{
    hb_buffer_t *first = NULL, *second = NULL;
    hb_buffer_t *props = hb_buffer_list_rem_head(&pv->props);
    while (hbhip_filter_pending(pv->dev) > 0)
    {
        int64_t tag = 0;
        hb_buffer_t *out = hbhip_host_pull(pv->dev, &pv->output, pv->input.geometry.width,
                                           pv->input.geometry.height, pv->dev_io, &tag);
        if (out == NULL)
        {
            hb_error("decomb(hip): pull failed");
            hb_buffer_close(&props);
            return -1;
        }
        if (props != NULL)
            hb_buffer_copy_props(out, props);                  /* decomb.c:555 */
        if (tag & 1) second = out; else first = out;
        hb_buffer_list_append(list, out);
    }
    if (first != NULL && second != NULL)                       /* bob pair, decomb.c:562-569 */
    {
        first->s.stop -= (first->s.stop - first->s.start) / 2LL;
        second->s.start = first->s.stop;
        second->s.new_chap = 0;
    }
    hb_buffer_close(&props);
    return 0;
}

static int decomb_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out)
{
    hb_filter_private_t *pv = filter->private_data;
    hb_buffer_t *in = *buf_in;
    hb_buffer_list_t list;
    hb_buffer_list_clear(&list);

    if (in->s.flags & HB_BUF_FLAG_EOF)
    {
        /* the last frame is processed against a copy of itself (decomb.c:584-589) */
        if (hbhip_filter_flush(pv->dev) != HBHIP_OK || decomb_hip_collect(pv, &list) != 0)
        {
            hb_buffer_list_close(&list);
            return HB_FILTER_FAILED;
        }
        hb_buffer_list_append(&list, in);
        *buf_out = hb_buffer_list_clear(&list);
        *buf_in = NULL;
        return HB_FILTER_DONE;
    }

    int rc;
    hbhip_frame *fr = hbhip_host_frame_of(in);
    if (fr != NULL)
    {
        hbhip_dev_frame d;
        hbhip_frame_describe(fr, &d, NULL, NULL);
        rc = hbhip_decomb_push_dev(pv->dev, &d, pv->next_tag++, in->s.flags, in->s.combed);
    }
    else
    {
        hbhip_host_frame hf;
        hbhip_host_frame_from_buf(&hf, in);
        rc = hbhip_decomb_push(pv->dev, &hf, pv->next_tag++, in->s.flags, in->s.combed);
    }
    if (rc != HBHIP_OK)
    {
        hb_error("decomb(hip): push: %s", hbhip_strerror(rc));
        return HB_FILTER_FAILED;
    }
    hb_buffer_t *props = hb_buffer_init(0);
    hb_buffer_copy_props(props, in);
    hb_buffer_list_append(&pv->props, props);

    if (!pv->ready)
    {
        pv->ready = 1;
        return HB_FILTER_DELAY;                                /* decomb.c:597-605 */
    }
    if (decomb_hip_collect(pv, &list) != 0)
    {
        hb_buffer_list_close(&list);
        return HB_FILTER_FAILED;
    }
    *buf_out = hb_buffer_list_clear(&list);
    return HB_FILTER_OK;
}
