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

#define DECOMB_BATCH 8

struct hb_filter_private_s
{
    hbhip_decomb_params par;
    hbhip_filter       *dev;
    hb_buffer_list_t    props;       /* per queued input frame: its `s` */
    int64_t             next_tag;
    int                 ready;
    int                 batch;       /* input frames gathered per launch (see decomb_hip_work) */
    int                 gathered;
    int                 dev_io;
    int                 selective;   /* yadif: deint=interlaced */
    int                 ff_yadif;    /* the Deinterlace (FFmpeg yadif) object below: outputs are progressive */
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

    hbhip_ctx *ctx = hbhip_host_ctx_for_role(init, 1);
    if (ctx == NULL) goto fail;
    int rc = hbhip_decomb_create(ctx, p, init->geometry.width, init->geometry.height,
                                 desc->comp[0].depth, desc->log2_chroma_w, desc->log2_chroma_h, &pv->dev);
    if (rc != HBHIP_OK)
    {
        hb_error("decomb(hip): %s", hbhip_strerror(rc));
        goto fail;
    }
    /* inside a device-resident run frames come in and go out AS frames: nothing is copied at either end (include/hbhip.h) */
    if (pv->dev_io && hbhip_host_zero_copy()) hbhip_filter_use_frames(pv->dev);
    if (p->mode & DECOMB_BOB)
        init->vrate.num *= 2;                  /* decomb.c:427-430 */
    pv->output = *init;
    /* Frames are gathered DECOMB_BATCH at a time: EEDI2 takes the fields of a batch through every pass in one
     * launch, the blends of a batch are one launch (csrc/decomb.hip) - a frame alone leaves most of the GPU idle.  The
     * filter answers HB_FILTER_DELAY meanwhile and then emits the batch's frames as one list, the burst pattern of
     * the reference's own threaded filters (nlmeans.c:548-571). */
    pv->batch = 1;
    if (hbhip_filter_defer(pv->dev, 1) == HBHIP_OK)
    {
        const char *env = getenv("HBHIP_DECOMB_BATCH");
        pv->batch = env != NULL && atoi(env) > 0 ? atoi(env) : DECOMB_BATCH;
    }
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
    if (pv->dev != NULL) hbhip_filter_destroy(pv->dev);
    hb_buffer_list_close(&pv->props);
    free(pv);
    filter->private_data = NULL;
}

static void decomb_hip_split_pair(hb_buffer_t *first, hb_buffer_t *second)
{
    if (first != NULL && second != NULL)                       /* bob pair, decomb.c:562-569 */
    {
        first->s.stop -= (first->s.stop - first->s.start) / 2LL;
        second->s.start = first->s.stop;
        second->s.new_chap = 0;
    }
}

/* Pull every finished frame.  The frames made from one input frame carry its tag (<< 1, | 1 for the second of a bob
 * pair); the input frames' properties wait in pv->props in order. */
static int decomb_hip_collect(hb_filter_private_t *pv, hb_buffer_list_t *list)
{
    hb_buffer_t *first = NULL, *second = NULL, *props = NULL;
    int64_t cur = -1;
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
        if ((tag >> 1) != cur)                                 /* the next input frame's outputs begin */
        {
            decomb_hip_split_pair(first, second);
            first = second = NULL;
            hb_buffer_close(&props);
            props = hb_buffer_list_rem_head(&pv->props);
            cur = tag >> 1;
        }
        if (props != NULL)
            hb_buffer_copy_props(out, props);                  /* decomb.c:555 */
        if (pv->ff_yadif && (props == NULL || pv->selective == 0 || props->s.combed != 0))
        {
            /* yadif clears the interlaced flag of what it deinterlaced (yadif_common.c), which comes
             * back as PIC_FLAG_PROGRESSIVE_FRAME and no combed mark (hbffmpeg.c:151-162) */
            out->s.flags |= PIC_FLAG_PROGRESSIVE_FRAME;
            out->s.combed = HB_COMB_NONE;
        }
        if (tag & 1) second = out; else first = out;
        hb_buffer_list_append(list, out);
    }
    decomb_hip_split_pair(first, second);
    hb_buffer_close(&props);
    return 0;
}

static int decomb_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out)
{
    hb_filter_private_t *pv = filter->private_data;
    hb_buffer_t *in = *buf_in;
    hb_buffer_list_t list;
    hb_buffer_list_clear(&list);

    if (pv->dev == NULL)
    {
        /* Deinterlace with the enable bit off: the reference adds no avfilter at all (deinterlace.c:88-91) */
        *buf_out = in;
        *buf_in = NULL;
        return (in->s.flags & HB_BUF_FLAG_EOF) ? HB_FILTER_DONE : HB_FILTER_OK;
    }
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
        rc = hbhip_decomb_push_frame(pv->dev, fr, pv->next_tag++, in->s.flags, in->s.combed);   /* no copy: the frame is the input picture */
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
    if (pv->batch > 1)
    {
        if (++pv->gathered < pv->batch) return HB_FILTER_DELAY;
        pv->gathered = 0;
        if (hbhip_filter_kick(pv->dev) != HBHIP_OK)            /* the gathered frames' launches */
        {
            hb_error("decomb(hip): launch failed");
            return HB_FILTER_FAILED;
        }
    }
    if (decomb_hip_collect(pv, &list) != 0)
    {
        hb_buffer_list_close(&list);
        return HB_FILTER_FAILED;
    }
    *buf_out = hb_buffer_list_clear(&list);
    return HB_FILTER_OK;
}


/* ---- Deinterlace = FFmpeg yadif (libhb/deinterlace.c:43-143) ---------------------------------
 * In the reference hb_filter_yadif has .skip = 1: deinterlace_init only writes
 * `yadif=mode=send_frame|send_field[_nospatial][:deint=interlaced][:parity=tff|bff]` for
 * libavfilter and doubles vrate for bob.  Here it is a real filter on the decomb frame ring (same
 * first / last frame handling as yadif_common.c: the first frame's `prev` and the last frame's
 * `next` are the frame itself), with vf_yadif.c's line filter (csrc/decomb.hip:yadif_ff_kernel,
 * parity unpinned).  A bob pair splits the frame's time span, which for contiguous timestamps is
 * what yadif's pts arithmetic (cur * 2, cur + next) comes to. */
static int yadif_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init);

static const char yadif_hip_template[] = "mode=^"HB_INT_REG"$:parity=^([01])$";

hb_filter_object_t hb_filter_yadif_hip =
{
    .id                = HB_FILTER_YADIF,
    .enforce_order     = 1,
    .name              = "Deinterlace (HIP)",
    .short_name        = "deinterlace",
    .settings          = NULL,
    .init              = yadif_hip_init,
    .work              = decomb_hip_work,
    .close             = decomb_hip_close,
    .settings_template = yadif_hip_template,
};

#define YADIF_ENABLE    1      /* deinterlace.c:59-69 */
#define YADIF_SPATIAL   2
#define YADIF_BOB       4
#define YADIF_SELECTIVE 8

static int deint_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init, int is_yadif);
static int yadif_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init) { return deint_hip_init(filter, init, 1); }
static int bwdif_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init) { return deint_hip_init(filter, init, 0); }

/* ---- Bwdif = FFmpeg bwdif (libhb/deinterlace.c:46: the same macro instantiates hb_filter_bwdif, and
 * deinterlace_init :72-143 configures it like yadif except that the spatial bit only exists for yadif,
 * :98-122).  Real filter on the decomb frame ring like the yadif object above; line filters of vf_bwdif.c
 * (csrc/decomb.hip:bwdif_kernel, parity unpinned), including yadif_common.c's current_field bookkeeping:
 * the first field of the stream and - in bob mode - the last one are filtered spatially only. */
hb_filter_object_t hb_filter_bwdif_hip =
{
    .id                = HB_FILTER_BWDIF,
    .enforce_order     = 1,
    .name              = "Bwdif (HIP)",
    .short_name        = "bwdif",
    .settings          = NULL,
    .init              = bwdif_hip_init,
    .work              = decomb_hip_work,
    .close             = decomb_hip_close,
    .settings_template = yadif_hip_template,
};

static int deint_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init, int is_yadif)
{
    hb_filter_private_t *pv = calloc(1, sizeof(*pv));
    if (pv == NULL) return -1;
    filter->private_data = pv;
    pv->input = *init;
    pv->dev_io = hbhip_host_dev_io(init);
    pv->ff_yadif = 1;
    hb_buffer_list_clear(&pv->props);

    int mode = 3, parity = -1;                                   /* :82-86 */
    if (filter->settings != NULL)
    {
        hb_dict_extract_int(&mode, filter->settings, "mode");
        hb_dict_extract_int(&parity, filter->settings, "parity");
    }
    if (!(mode & YADIF_ENABLE))                                  /* :88-91: nothing to do */
    {
        pv->output = *init;
        return 0;
    }
    pv->selective = !!(mode & YADIF_SELECTIVE);
    const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(init->pix_fmt);
    hbhip_ctx *ctx = desc != NULL ? hbhip_host_ctx_for_role(init, 1) : NULL;
    int rc = ctx == NULL ? HBHIP_ERR_NODEVICE
           : is_yadif    ? hbhip_yadif_create(ctx, !!(mode & YADIF_SPATIAL), !!(mode & YADIF_BOB), pv->selective, parity,
                                              init->geometry.width, init->geometry.height, desc->comp[0].depth,
                                              desc->log2_chroma_w, desc->log2_chroma_h, &pv->dev)
                         : hbhip_bwdif_create(ctx, !!(mode & YADIF_BOB), pv->selective, parity,
                                              init->geometry.width, init->geometry.height, desc->comp[0].depth,
                                              desc->log2_chroma_w, desc->log2_chroma_h, &pv->dev);
    if (rc != HBHIP_OK)
    {
        hb_error("%s(hip): %s", is_yadif ? "deinterlace" : "bwdif", hbhip_strerror(rc));
        free(pv);
        filter->private_data = NULL;
        return -1;
    }
    if (pv->dev_io && hbhip_host_zero_copy()) hbhip_filter_use_frames(pv->dev);
    if (mode & YADIF_BOB)
        init->vrate.num *= 2;                                    /* :107 */
    pv->output = *init;
    return 0;
}
