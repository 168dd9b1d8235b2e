/* blend_hip.c — HIP-backed drop-in for the reference's subtitle compositor object `hb_blend`
 * (libhb/blend.c:40-46; type hb_blend_object_t, handbrake/common.h:1813-1828).
 *
 * rendersub.c picks the compositor in its own hb_blend_init (:1129-1161: `hb_blend_vt` for
 * VideoToolbox frames, `hb_blend` otherwise), copies the object, calls init once and then
 * work(frame, overlay list, changed) for every frame (:467, :888, :1011, :1122).  `hb_blend_hip` is a
 * third choice with the same three entry points; INTEGRATION.md shows the two-line change.
 * Same contract as hb_blend_work (:848-873): no overlays => the frame is returned untouched; a
 * frame that is not writable is duplicated first; overlays are composited in list order.
 * The bitmaps are uploaded only when `changed` says the list is new.  Frames already in HBM
 * (storage_type HBHIP_DEVICE) are composited in place without leaving the device.
 */
#include "hbhip_host.h"

struct hb_blend_private_s
{
    hbhip_blend *dev;                /* made on the first frame: on the GPU that frame lives on */
    int          have_overlays;      /* the device holds the current list */
    int          width, height, depth, lcw, lch, chroma_location, ov_lcw, ov_lch;
};

/* hb_blend_object_t.init gets no hb_filter_init_t (common.h:1813-1828), so it cannot ask which GPU the job runs on
 * (job->hw_device_index); the first frame can: a device-resident frame is composited on the context it belongs to,
 * host frames on the process default. */
static int blend_hip_device(hb_blend_private_t *pv, const hb_buffer_t *in)
{
    if (pv->dev != NULL) return HBHIP_OK;
    hbhip_frame *fr = hbhip_host_frame_of(in);
    hbhip_ctx *ctx = fr != NULL ? hbhip_frame_context(fr) : hbhip_host_ctx();
    if (ctx == NULL) return HBHIP_ERR_NODEVICE;
    return hbhip_blend_create(ctx, pv->width, pv->height, pv->depth, pv->lcw, pv->lch, pv->chroma_location,
                              pv->ov_lcw, pv->ov_lch, &pv->dev);
}

static int blend_hip_init(hb_blend_object_t *object, int in_width, int in_height, int in_pix_fmt,
                          int in_chroma_location, int in_color_range, int overlay_pix_fmt)
{
    (void)in_color_range;
    hb_blend_private_t *pv = calloc(1, sizeof(*pv));
    object->private_data = pv;
    if (pv == NULL)
    {
        hb_error("blend(hip): calloc failed");
        return -1;
    }
    const AVPixFmtDescriptor *in_desc = av_pix_fmt_desc_get(in_pix_fmt);
    const AVPixFmtDescriptor *ov_desc = av_pix_fmt_desc_get(overlay_pix_fmt);
    /* a GPU that cannot give a context (busy, out of memory) is found here, where rendersub can still take hb_blend
     * (rendersub.c:1129-1161), not at the first frame in the middle of an encode */
    int rc = in_desc == NULL || ov_desc == NULL || hbhip_device_count() <= 0 || hbhip_host_ctx() == NULL
             ? HBHIP_ERR_NODEVICE : HBHIP_OK;
    if (rc == HBHIP_OK && av_pix_fmt_count_planes(in_pix_fmt) != 3)
        rc = HBHIP_ERR_UNSUPPORTED;                     /* NV12 / P010: blend8onbi*, not built */
    if (rc == HBHIP_OK)
    {
        /* what hbhip_blend_create would refuse on the first frame is refused here, where rendersub can still react */
        const int d = in_desc->comp[0].depth;
        const int sub = in_desc->log2_chroma_w != ov_desc->log2_chroma_w || in_desc->log2_chroma_h != ov_desc->log2_chroma_h;
        if ((d != 8 && d != 10 && d != 12) || in_desc->log2_chroma_w > 1 || in_desc->log2_chroma_h > 1 ||
            (sub && (ov_desc->log2_chroma_w || ov_desc->log2_chroma_h)))
            rc = HBHIP_ERR_UNSUPPORTED;
    }
    if (rc == HBHIP_OK)
    {
        pv->width = in_width; pv->height = in_height; pv->depth = in_desc->comp[0].depth;
        pv->lcw = in_desc->log2_chroma_w; pv->lch = in_desc->log2_chroma_h; pv->chroma_location = in_chroma_location;
        pv->ov_lcw = ov_desc->log2_chroma_w; pv->ov_lch = ov_desc->log2_chroma_h;
    }
    if (rc != HBHIP_OK)
    {
        hb_error("blend(hip): %s", hbhip_strerror(rc));
        free(pv);
        object->private_data = NULL;
        return -1;
    }
    return 0;
}

static hb_buffer_t *blend_hip_work(hb_blend_object_t *object, hb_buffer_t *in, hb_buffer_list_t *overlays, int changed)
{
    hb_blend_private_t *pv = object->private_data;
    hb_buffer_t *out = in;
    const int n = hb_buffer_list_count(overlays);
    if (n == 0)
        return out;                                                                /* blend.c:856-859 */

    int rc = blend_hip_device(pv, in);
    if (rc == HBHIP_OK && (changed || !pv->have_overlays))
    {
        hbhip_overlay *ov = calloc((size_t)n, sizeof(*ov));
        if (ov == NULL) return NULL;
        int i = 0;
        for (hb_buffer_t *o = hb_buffer_list_head(overlays); o != NULL && i < n; o = o->next, i++)
        {
            for (int p = 0; p < 4; p++)
            {
                ov[i].plane[p] = o->plane[p].data;
                ov[i].stride[p] = o->plane[p].stride;
            }
            ov[i].x = o->f.x;
            ov[i].y = o->f.y;
            ov[i].width = o->f.width;
            ov[i].height = o->f.height;
        }
        rc = hbhip_blend_set_overlays(pv->dev, ov, i);
        free(ov);
        pv->have_overlays = rc == HBHIP_OK;
    }
    if (rc == HBHIP_OK)
    {
        hbhip_frame *dev_frame = hbhip_host_frame_of(in);
        if (dev_frame != NULL)
        {
            /* blend.c:861-865 for a picture in HBM: a frame somebody else holds too (vfr's CFR duplicates are
             * hb_buffer_shallow_dup's of one hbhip_frame, vfr.c:393-411) is not writable - composite on a copy of it,
             * or the overlay lands twice on the shared picture */
            if (hbhip_frame_refs(dev_frame) > 1)
            {
                hbhip_frame *copy = NULL;
                int fw = 0, fh = 0;
                hbhip_dev_frame d0;
                hbhip_frame_describe(dev_frame, &d0, &fw, &fh);
                rc = hbhip_frame_alloc(hbhip_frame_context(dev_frame), fw, fh, pv->depth, pv->lcw, pv->lch, &copy);
                if (rc == HBHIP_OK) rc = hbhip_frame_copy(copy, dev_frame);
                if (rc == HBHIP_OK && (out = hb_buffer_dup(in)) == NULL) rc = HBHIP_ERR_NOMEM;
                if (rc != HBHIP_OK)
                {
                    hbhip_frame_release(copy);
                    out = in;
                }
                else
                {
                    hbhip_frame_release(dev_frame);            /* the reference hb_buffer_dup took for `out` */
                    out->storage = copy;                       /* `out` owns the copy's one reference */
                    hb_buffer_close(&in);
                    dev_frame = copy;
                }
            }
            if (rc == HBHIP_OK)
            {
                hbhip_dev_frame d;
                hbhip_frame_describe(dev_frame, &d, NULL, NULL);
                /* the compositor writes on the frame's own context: behind whoever read it on another stream of the job */
                rc = hbhip_frame_use_on(dev_frame, hbhip_frame_context(dev_frame));
                if (rc == HBHIP_OK) rc = hbhip_blend_apply_dev(pv->dev, &d);
                if (rc == HBHIP_OK) rc = hbhip_frame_mark_ready(dev_frame);    /* complete behind the compositor now */
            }
        }
        else
        {
            if (hb_buffer_is_writable(in) == 0)                                    /* :861-865 */
            {
                out = hb_buffer_dup(in);
                hb_buffer_close(&in);
                if (out == NULL) return NULL;
            }
            hbhip_host_frame f;
            hbhip_host_frame_from_buf(&f, out);
            rc = hbhip_blend_apply(pv->dev, &f);
        }
    }
    if (rc != HBHIP_OK)
    {
        hb_error("blend(hip): %s", hbhip_strerror(rc));
        hb_buffer_close(&out);
        return NULL;
    }
    return out;
}

static void blend_hip_close(hb_blend_object_t *object)
{
    hb_blend_private_t *pv = object->private_data;
    if (pv == NULL) return;
    hbhip_blend_destroy(pv->dev);
    free(pv);
    object->private_data = NULL;
}

hb_blend_object_t hb_blend_hip =
{
    .name  = "Blend (HIP)",
    .init  = blend_hip_init,
    .work  = blend_hip_work,
    .close = blend_hip_close,
};
