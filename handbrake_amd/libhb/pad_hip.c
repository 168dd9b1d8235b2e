/* pad_hip.c — HIP-backed drop-in for hb_filter_pad (libhb/pad.c:15-148).
 *
 * In the reference this object has .skip = 1: pad_init only assembles the settings of FFmpeg's `pad`
 * (width, height, x, y, color), which hb_avfilter_combine folds into HB_FILTER_AVFILTER.  Here it is
 * a real filter with its own work(), like the reference's pad_vt (platform/macosx/pad_vt.m), and has
 * to be left out of hb_avfilter_combine's switch.  Same settings keys, same
 * resolution of top/bottom/left/right vs width/height/x/y, same rewrite of init->geometry.
 * What vf_pad itself adds is restated: negative x / y centre the picture ((out - in) / 2), offsets
 * are rounded down to the chroma subsampling, and the RGB colour becomes Y'CbCr the way
 * drawutils.c:ff_draw_color does (oracle/alias_oracle.c:orc_pad_color; parity unpinned).
 */
#include "hbhip_host.h"

#include <string.h>
#include <strings.h>

struct hb_filter_private_s
{
    hbhip_filter    *dev;
    hb_filter_init_t input;
    hb_filter_init_t output;
    int              dev_io;
};

static int pad_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init);
static int pad_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out);
static void pad_hip_close(hb_filter_object_t *filter);

static const char pad_hip_template[] =
    "width=^"HB_INT_REG"$:height=^"HB_INT_REG"$:color=^"HB_ALL_REG"$:"
    "x=^"HB_INT_REG"$:y=^"HB_INT_REG"$:"
    "top=^"HB_INT_REG"$:bottom=^"HB_INT_REG"$:"
    "left=^"HB_INT_REG"$:right=^"HB_INT_REG"$";

hb_filter_object_t hb_filter_pad_hip =
{
    .id                = HB_FILTER_PAD,
    .enforce_order     = 1,
    .name              = "Pad (HIP)",
    .short_name        = "pad",
    .settings          = NULL,
    .init              = pad_hip_init,
    .work              = pad_hip_work,
    .close             = pad_hip_close,
    .settings_template = pad_hip_template,
};

#ifndef HBHIP_IN_LIBHB
/* Outside libhb there is no colormap.c: the handful of names anyone pads with (X11 values). */
static uint32_t hb_rgb_lookup_by_name(const char *color)
{
    static const struct { const char *name; uint32_t rgb; } names[] = {
        { "black", 0x000000 }, { "white", 0xFFFFFF }, { "red", 0xFF0000 }, { "green", 0x00FF00 }, { "blue", 0x0000FF },
        { "yellow", 0xFFFF00 }, { "cyan", 0x00FFFF }, { "magenta", 0xFF00FF }, { "gray", 0xBEBEBE }, { "grey", 0xBEBEBE },
        { "darkgray", 0xA9A9A9 }, { "lightgray", 0xD3D3D3 }, { "orange", 0xFFA500 }, { "purple", 0xA020F0 },
        { NULL, 0 } };
    for (int i = 0; names[i].name != NULL; i++)
        if (!strcasecmp(color, names[i].name)) return names[i].rgb;
    return 0;                                                /* colormap.c:710-722: unknown names are black */
}
#endif

/* drawutils.c:ff_draw_color for a planar Y'CbCr format */
static void fill_from_rgb(int rgb, int matrix, int range, int depth, int fill[3])
{
    double kr = 0.299, kb = 0.114;                           /* unspecified -> smpte170m */
    switch (matrix)
    {
        case 1: kr = 0.2126; kb = 0.0722; break;
        case 4: kr = 0.30;   kb = 0.11;   break;
        case 7: kr = 0.212;  kb = 0.087;  break;
        case 9: case 10: kr = 0.2627; kb = 0.0593; break;
    }
    const double kg = 1.0 - kr - kb;
    const double r = ((rgb >> 16) & 0xff) / 255., g = ((rgb >> 8) & 0xff) / 255., b = (rgb & 0xff) / 255.;
    double v[3];
    v[0] = kr * r + kg * g + kb * b;
    v[1] = (-kr * r - kg * g + (1.0 - kb) * b) / (2.0 * (1.0 - kb));
    v[2] = ((1.0 - kr) * r - kg * g - kb * b) / (2.0 * (1.0 - kr));
    for (int i = 0; i < 3; i++)
    {
        const int chroma = i > 0;
        if (range != 2)                                      /* AVCOL_RANGE_JPEG = 2; unspecified counts as limited */
        {
            v[i] *= (chroma ? 224. : 219.) / 255.;
            v[i] += (chroma ? 128. : 16.) / 255.;
        }
        else if (chroma)
            v[i] += 0.5;
        fill[i] = (int)(unsigned)(v[i] * ((1 << depth) - 1) + 0.5);
    }
}

static int pad_hip_init(hb_filter_object_t *filter, hb_filter_init_t *init)
{
    hb_filter_private_t *pv = calloc(1, sizeof(*pv));
    filter->private_data = pv;
    if (pv == NULL) return 1;
    pv->input = *init;
    pv->dev_io = hbhip_host_dev_io(init);

    int width = -1, height = -1, rgb = 0;
    int top = -1, bottom = -1, left = -1, right = -1, x = -1, y = -1;
    char *color = NULL;
    hb_dict_extract_int(&top, filter->settings, "top");                       /* pad.c:61-70 */
    hb_dict_extract_int(&bottom, filter->settings, "bottom");
    hb_dict_extract_int(&left, filter->settings, "left");
    hb_dict_extract_int(&right, filter->settings, "right");
    hb_dict_extract_int(&width, filter->settings, "width");
    hb_dict_extract_int(&height, filter->settings, "height");
    hb_dict_extract_string(&color, filter->settings, "color");
    hb_dict_extract_int(&x, filter->settings, "x");
    hb_dict_extract_int(&y, filter->settings, "y");
    if (x < 0) x = left;                                                      /* :72-87 */
    if (y < 0) y = top;
    if (top >= 0 && bottom >= 0 && height < 0) height = init->geometry.height + top + bottom;
    if (left >= 0 && right >= 0 && width < 0)  width = init->geometry.width + left + right;
    if (color != NULL)                                                        /* :88-99 */
    {
        char *end;
        rgb = (int)strtol(color, &end, 0);
        if (end == color) rgb = (int)hb_rgb_lookup_by_name(color);
        free(color);
    }
    if (width < init->geometry.width)   width = init->geometry.width;         /* :119-126 */
    if (height < init->geometry.height) height = init->geometry.height;

    const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(init->pix_fmt);
    hbhip_ctx *ctx = desc != NULL ? hbhip_host_ctx_for(init) : NULL;
    int rc = ctx == NULL ? HBHIP_ERR_NODEVICE : HBHIP_OK;
    if (rc == HBHIP_OK)
    {
        hbhip_pad_params p;
        memset(&p, 0, sizeof(p));
        p.width = width;
        p.height = height;
        /* vf_pad: "(out_w-in_w)/2" when unset (:101-118), then rounded down to the chroma subsampling;
         * a picture that would stick out is moved back in */
        p.x = x < 0 ? (width - init->geometry.width) / 2 : x;
        p.y = y < 0 ? (height - init->geometry.height) / 2 : y;
        p.x &= ~((1 << desc->log2_chroma_w) - 1);
        p.y &= ~((1 << desc->log2_chroma_h) - 1);
        if (p.x + init->geometry.width > width)   p.x = (width - init->geometry.width) & ~((1 << desc->log2_chroma_w) - 1);
        if (p.y + init->geometry.height > height) p.y = (height - init->geometry.height) & ~((1 << desc->log2_chroma_h) - 1);
        fill_from_rgb(rgb, init->color_matrix, init->color_range, desc->comp[0].depth, p.fill);
        rc = hbhip_pad_create(ctx, &p, init->geometry.width, init->geometry.height, desc->comp[0].depth,
                              desc->log2_chroma_w, desc->log2_chroma_h, &pv->dev);
    }
    if (rc != HBHIP_OK)
    {
        hb_error("pad(hip): %s", hbhip_strerror(rc));
        free(pv);
        filter->private_data = NULL;
        return 1;
    }
    init->geometry.width = width;                                             /* :143-145 */
    init->geometry.height = height;
    pv->output = *init;
    return 0;
}

static int pad_hip_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out)
{
    hb_filter_private_t *pv = filter->private_data;
    return hbhip_host_simple_work(pv->dev, &pv->output, filter->short_name, pv->dev_io, buf_in, buf_out);
}

static void pad_hip_close(hb_filter_object_t *filter)
{
    hb_filter_private_t *pv = filter->private_data;
    if (pv == NULL) return;
    hbhip_host_simple_destroy(pv->dev);
    free(pv);
    filter->private_data = NULL;
}
