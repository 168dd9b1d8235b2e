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
    return f;
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
