/* Compiles the reference's libhb/decomb.c in place (found through -I$(REF)/libhb),
 * unmodified, against include/hbhip_libhb.h.  See wrap_common.h. */
#include "wrap_common.h"
#include "decomb.c"

/* ---- test entry points: drive the reference's own eedi2_planer_8 directly --------- */
HBREF_EXPORT void *hbref_eedi2_new(int width, int height, const char *settings)
{
    hb_filter_object_t *f = calloc(1, sizeof(*f));
    *f = hb_filter_decomb;
    hb_filter_init_t init;
    memset(&init, 0, sizeof(init));
    init.pix_fmt = AV_PIX_FMT_YUV420P;
    init.geometry.width = width;
    init.geometry.height = height;
    f->settings = hbhip_dict_from_string(settings);
    if (f->init(f, &init) != 0)
    {
        free(f);
        return NULL;
    }
    /* post-processing 2/3: the derivative arrays come from malloc (decomb.c:400-403) and the
     * blur reads one element per row that nothing has written yet (eedi2_template.c:1589);
     * start from zeros, as fresh pages would */
    hb_filter_private_t *pv = f->private_data;
    if (pv->cx2 != NULL)
    {
        const size_t n = (size_t)height * hb_image_stride(init.pix_fmt, width, 0) * sizeof(int);
        memset(pv->cx2, 0, n); memset(pv->cy2, 0, n); memset(pv->cxy, 0, n); memset(pv->tmpc, 0, n);
    }
    return f;
}

/* The same object for a 10 / 12-bit pixel format (the reference then runs its _16 template functions) */
HBREF_EXPORT void *hbref_eedi2_new_fmt(int width, int height, const char *settings, int pix_fmt)
{
    hb_filter_object_t *f = calloc(1, sizeof(*f));
    *f = hb_filter_decomb;
    hb_filter_init_t init;
    memset(&init, 0, sizeof(init));
    init.pix_fmt = pix_fmt;
    init.geometry.width = width;
    init.geometry.height = height;
    f->settings = hbhip_dict_from_string(settings);
    if (f->init(f, &init) != 0)
    {
        free(f);
        return NULL;
    }
    hb_filter_private_t *pv = f->private_data;
    if (pv->cx2 != NULL)
    {
        const size_t n = (size_t)height * hb_image_stride(init.pix_fmt, width, 0) * sizeof(int);
        memset(pv->cx2, 0, n); memset(pv->cy2, 0, n); memset(pv->cxy, 0, n); memset(pv->tmpc, 0, n);
    }
    return f;
}

/* eedi2_planer_16 / plane-serial eedi2_interpolate_plane_16 on a frame of uint16 samples (strides in bytes) */
HBREF_EXPORT void hbref_eedi2_run16(void *h, const uint8_t *const plane[3], const int stride[3], int tff, int serial)
{
    hb_filter_object_t *f = h;
    hb_filter_private_t *pv = f->private_data;
    hb_buffer_t *b = hb_frame_buffer_init(pv->input.pix_fmt, pv->input.geometry.width, pv->input.geometry.height);
    for (int p = 0; p < 3; p++)
        for (int y = 0; y < b->plane[p].height; y++)
            memcpy(b->plane[p].data + (size_t)y * b->plane[p].stride, plane[p] + (size_t)y * stride[p],
                   MIN(stride[p], b->plane[p].stride));
    hb_buffer_close(&pv->ref[1]);
    pv->ref[1] = b;
    pv->tff = tff;
    if (!serial)
    {
        eedi2_planer_16(pv);
        return;
    }
    for (int p = 0; p < 3; p++)
    {
        const int src_pitch = b->plane[p].stride / 2, dst_pitch = pv->eedi_half[SRCPF]->plane[p].stride / 2;
        eedi2_fill_half_height_buffer_plane_16((uint16_t *)b->plane[p].data + src_pitch * !tff,
                                               (uint16_t *)pv->eedi_half[SRCPF]->plane[p].data, src_pitch, dst_pitch, b->plane[p].height);
    }
    for (int p = 0; p < 3; p++)
        eedi2_interpolate_plane_16(pv, p);
}

HBREF_EXPORT void hbref_eedi2_run(void *h, const uint8_t *const plane[3], const int stride[3], int tff)
{
    hb_filter_object_t *f = h;
    hb_filter_private_t *pv = f->private_data;
    hb_buffer_t *b = hb_frame_buffer_init(pv->input.pix_fmt, pv->input.geometry.width, pv->input.geometry.height);
    for (int p = 0; p < 3; p++)
        for (int y = 0; y < b->plane[p].height; y++)
            memcpy(b->plane[p].data + (size_t)y * b->plane[p].stride, plane[p] + (size_t)y * stride[p],
                   MIN(stride[p], b->plane[p].stride));
    hb_buffer_close(&pv->ref[1]);
    pv->ref[1] = b;
    pv->tff = tff;
    eedi2_planer_8(pv);
}

/* The same, but the three planes one after the other on the caller's thread instead of on the
 * three taskset threads: with post-processing 2/3 the plane threads share cx2/cy2/cxy/tmpc
 * (decomb_template.c:380-383), so only a fixed order gives a defined result to pin. */
HBREF_EXPORT void hbref_eedi2_run_serial(void *h, const uint8_t *const plane[3], const int stride[3], int tff)
{
    hb_filter_object_t *f = h;
    hb_filter_private_t *pv = f->private_data;
    hb_buffer_t *b = hb_frame_buffer_init(pv->input.pix_fmt, pv->input.geometry.width, pv->input.geometry.height);
    for (int p = 0; p < 3; p++)
        for (int y = 0; y < b->plane[p].height; y++)
            memcpy(b->plane[p].data + (size_t)y * b->plane[p].stride, plane[p] + (size_t)y * stride[p],
                   MIN(stride[p], b->plane[p].stride));
    hb_buffer_close(&pv->ref[1]);
    pv->ref[1] = b;
    pv->tff = tff;
    for (int p = 0; p < 3; p++)
    {
        const int src_pitch = b->plane[p].stride, dst_pitch = pv->eedi_half[SRCPF]->plane[p].stride;
        eedi2_fill_half_height_buffer_plane_8(b->plane[p].data + src_pitch * !tff, pv->eedi_half[SRCPF]->plane[p].data,
                                              src_pitch, dst_pitch, b->plane[p].height);
    }
    for (int p = 0; p < 3; p++)
        eedi2_interpolate_plane_8(pv, p);
}

HBREF_EXPORT const uint8_t *hbref_eedi2_plane(void *h, int buffer, int plane, int *stride, int *height)
{
    hb_filter_object_t *f = h;
    hb_filter_private_t *pv = f->private_data;
    hb_buffer_t *b = buffer < 4 ? pv->eedi_half[buffer] : pv->eedi_full[buffer - 4];
    *stride = b->plane[plane].stride;
    *height = b->plane[plane].height;
    return b->plane[plane].data;
}

HBREF_EXPORT void hbref_eedi2_free(void *h)
{
    hb_filter_object_t *f = h;
    f->close(f);
    hb_dict_free(&f->settings);
    free(f);
}
