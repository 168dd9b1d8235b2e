/* Compiles the reference's libhb/nlmeans.c in place (found through
 * -I$(REF)/libhb), unmodified, against include/hbhip_libhb.h, and adds test
 * entry points that reach its static plane routines.  See wrap_common.h. */
#include "wrap_common.h"
#include "nlmeans.c"

/* Run the reference's nlmeans_plane_8 (nlmeans_template.c:593-717) on one
 * plane of `nframes` consecutive frames (frame 0 = the one being filtered).
 * `settings` is the filter's own settings string; `c` selects which channel's
 * parameters (0=Y,1=Cb,2=Cr) are used.  force_scalar=1 skips the SSE2 integral. */
HBREF_EXPORT int hbref_nlmeans_plane_8(const char *settings, int c,
                                       const uint8_t *const *planes, int nframes,
                                       int w, int h, int stride,
                                       uint8_t *dst, int dst_stride, int force_scalar)
{
    hb_filter_object_t f = hb_filter_nlmeans;
    hb_filter_init_t init;
    memset(&init, 0, sizeof(init));
    init.pix_fmt = AV_PIX_FMT_YUV420P;
    init.geometry.width = w;
    init.geometry.height = h;
    f.settings = hbhip_dict_from_string(settings);
    hbhip_dict_set(f.settings, "threads", "1");
    if (f.init(&f, &init) != 0)
        return -1;
    hb_filter_private_t *pv = f.private_data;
    if (force_scalar)
        pv->functions.build_integral = build_integral_scalar_8;

    if (nframes > NLMEANS_FRAMES_MAX) nframes = NLMEANS_FRAMES_MAX;
    Frame *fr = calloc(nframes, sizeof(Frame));
    const int border = ((pv->patch_size[c] + 2) / 2 + 15) / 16 * 16;
    for (int i = 0; i < nframes; i++)
    {
        pv->nlmeans_alloc(planes[i], w, stride, h, &fr[i].plane[c], border);
        fr[i].plane[c].mutex = hb_lock_init();
    }
    int use = pv->nframes[c] < nframes ? pv->nframes[c] : nframes;
    if (pv->prefilter[c] & NLMEANS_PREFILTER_MODE_PASSTHRU)
    {
        pv->nlmeans_prefilter(&fr[0].plane[c], pv->prefilter[c]);
        pv->nlmeans_deborder(&fr[0].plane[c], dst, w, dst_stride, h);
    }
    else
    {
        pv->nlmeans_plane(&pv->functions, fr, pv->prefilter[c], c, use, dst, w, dst_stride, h,
                          pv->strength[c], pv->origin_tune[c], pv->patch_size[c], pv->range[c],
                          pv->exptable[c], pv->weight_fact_table[c], pv->diff_max[c]);
    }
    for (int i = 0; i < nframes; i++)
    {
        if (fr[i].plane[c].mem_pre != NULL && fr[i].plane[c].mem_pre != fr[i].plane[c].mem)
            free(fr[i].plane[c].mem_pre);
        free(fr[i].plane[c].mem);
        hb_lock_close(&fr[i].plane[c].mutex);
    }
    free(fr);
    f.close(&f);
    hb_dict_free(&f.settings);
    return 0;
}

/* The tables nlmeans_init builds (nlmeans.c:345-358), for pinning the oracle's. */
HBREF_EXPORT int hbref_nlmeans_tables(const char *settings, int c, float *exptable,
                                      float *weight_fact_table, int *diff_max)
{
    hb_filter_object_t f = hb_filter_nlmeans;
    hb_filter_init_t init;
    memset(&init, 0, sizeof(init));
    init.pix_fmt = AV_PIX_FMT_YUV420P;
    f.settings = hbhip_dict_from_string(settings);
    hbhip_dict_set(f.settings, "threads", "1");
    if (f.init(&f, &init) != 0)
        return -1;
    hb_filter_private_t *pv = f.private_data;
    memcpy(exptable, pv->exptable[c], sizeof(float) * NLMEANS_EXPSIZE);
    *weight_fact_table = pv->weight_fact_table[c];
    *diff_max = pv->diff_max[c];
    f.close(&f);
    hb_dict_free(&f.settings);
    return 0;
}

/* prefilter only: returns the prefiltered (de-bordered) plane */
/* The same for the 16-bit template instantiation (nlmeans_prefilter_16, nlmeans.c:253-262);
 * `stride` and `dst_stride` in bytes, samples uint16. */
HBREF_EXPORT int hbref_nlmeans_prefilter_16(const uint8_t *plane, int w, int h, int stride,
                                            int filter_type, int border, uint8_t *dst, int dst_stride)
{
    BorderedPlane bp;
    memset(&bp, 0, sizeof(bp));
    nlmeans_alloc_16(plane, w, stride / 2, h, &bp, border);      /* src_s is in samples */
    bp.mutex = hb_lock_init();
    nlmeans_prefilter_16(&bp, filter_type);
    const uint16_t *img = (const uint16_t *)bp.image_pre;
    const int bw = w + 2 * border;
    for (int y = 0; y < h; y++)
        memcpy(dst + (size_t)y * dst_stride, img + (size_t)y * bw, sizeof(uint16_t) * w);
    if (bp.mem_pre != bp.mem) free(bp.mem_pre);
    free(bp.mem);
    hb_lock_close(&bp.mutex);
    return 0;
}

HBREF_EXPORT int hbref_nlmeans_prefilter_8(const uint8_t *plane, int w, int h, int stride,
                                           int filter_type, int border, uint8_t *dst, int dst_stride)
{
    BorderedPlane bp;
    memset(&bp, 0, sizeof(bp));
    nlmeans_alloc_8(plane, w, stride, h, &bp, border);
    bp.mutex = hb_lock_init();
    nlmeans_prefilter_8(&bp, filter_type);
    const uint8_t *img = (const uint8_t *)bp.image_pre;
    const int bw = w + 2 * border;
    for (int y = 0; y < h; y++)
        memcpy(dst + (size_t)y * dst_stride, img + (size_t)y * bw, w);
    if (bp.mem_pre != bp.mem) free(bp.mem_pre);
    free(bp.mem);
    hb_lock_close(&bp.mutex);
    return 0;
}
