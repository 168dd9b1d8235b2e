/* Compiles the reference's libhb/rendersub.c in place (found through -I$(REF)/libhb), unmodified, against
 * include/hbhip_libhb.h.  See wrap_common.h.  `hb_filter_render_sub` is exported for the test harness, which registers it
 * as HB_FILTER_RENDER_SUB the way hb_filter_get (common.c:5331-5495) holds it inside libhb.
 *
 * What the file needs beyond the stand-in header is declared here.  libass (SSA / text subtitles) and libswscale (a
 * bitmap subtitle whose window differs from the frame) are not in the image: their entry points are stubs that fail, and
 * the tests burn in bitmap subtitles (VOBSUB / PGS) of the frame's own size - the paths :381-476 and :1016-1127, which
 * hand every frame to the compositor object.
 *
 * rendersub.c picks that object by init.hw_pix_fmt (rendersub.c:1129-1161: hb_blend_vt for VideoToolbox frames, hb_blend
 * otherwise).  A HIP build of libhb adds one case there (`case AV_PIX_FMT_HBHIP: blend =
 * &hb_blend_hip;`); to keep the file itself untouched the same choice is made from outside, as wrap_vfr.c does for the
 * motion metric: the `default:` branch's `&hb_blend` resolves through the runtime's hw-helper table. */
#include "wrap_common.h"
#include <limits.h>
#include <inttypes.h>
#include <stdarg.h>
#define HANDBRAKE_EXTRADATA_H                       /* its prototypes are not used by rendersub.c */

/* ---- what handbrake/common.h, hbffmpeg.h and extradata.h would have declared --------------------------------------- */
enum { AV_PIX_FMT_NV12 = 23, AV_PIX_FMT_NV16 = 101, AV_PIX_FMT_NV24 = 188, AV_PIX_FMT_P010 = 158, AV_PIX_FMT_P012 = 207,
       AV_PIX_FMT_P016 = 170, AV_PIX_FMT_P210 = 197, AV_PIX_FMT_P212 = 209, AV_PIX_FMT_P216 = 199, AV_PIX_FMT_P410 = 198,
       AV_PIX_FMT_P412 = 210, AV_PIX_FMT_P416 = 200, AV_PIX_FMT_YUV420P16 = 47, AV_PIX_FMT_YUV422P16 = 49, AV_PIX_FMT_YUV444P16 = 51 };
enum { AVCOL_RANGE_MPEG = 1 };
enum { SWS_LANCZOS = 0x200, SWS_ACCURATE_RND = 0x40000, SWS_CS_DEFAULT = 5 };
struct SwsContext;
static struct SwsContext *hb_sws_get_context(int sw, int sh, int sf, int sr, int dw, int dh, int df, int dr, int flags, int cs)
{
    (void)sw; (void)sh; (void)sf; (void)sr; (void)dw; (void)dh; (void)df; (void)dr; (void)flags; (void)cs;
    hb_error("rendersub (oracle/_ref): libswscale is not in this build - subtitles must have the frame's size");
    return NULL;
}
static int sws_scale(struct SwsContext *c, const uint8_t *const s[], const int ss[], int y, int h, uint8_t *const d[], const int ds[])
{
    (void)c; (void)s; (void)ss; (void)y; (void)h; (void)d; (void)ds;
    return -1;
}
static void sws_freeContext(struct SwsContext *c) { (void)c; }
static void hb_picture_fill(uint8_t *data[], int stride[], hb_buffer_t *b)
{
    for (int p = 0; p < 4; p++) { data[p] = b->plane[p].data; stride[p] = b->plane[p].stride; }
}
typedef int (*hb_csp_convert_f)(int);
static int hb_rgb2yuv(int rgb) { return rgb; }                    /* only reached from the libass paths */
static int hb_rgb2yuv_bt709(int rgb) { return rgb; }
static hb_csp_convert_f hb_get_rgb2yuv_function(int color_matrix) { (void)color_matrix; return hb_rgb2yuv; }
static void hb_valog(int level, const char *prefix, const char *fmt, va_list args) { (void)level; (void)prefix; (void)fmt; (void)args; }

/* ---- libass: not here (shim/ass/ass.h declares it) ----------------------------------------------------------------- */
#include <ass/ass.h>
ASS_Library  *ass_library_init(void) { return NULL; }             /* ssa_post_init then reports "libass initialization failed" */
void          ass_library_done(ASS_Library *l) { (void)l; }
void          ass_set_message_cb(ASS_Library *l, void (*cb)(int, const char *, va_list, void *), void *d) { (void)l; (void)cb; (void)d; }
void          ass_set_extract_fonts(ASS_Library *l, int e) { (void)l; (void)e; }
void          ass_add_font(ASS_Library *l, const char *n, const char *d, int s) { (void)l; (void)n; (void)d; (void)s; }
void          ass_set_style_overrides(ASS_Library *l, char **o) { (void)l; (void)o; }
ASS_Renderer *ass_renderer_init(ASS_Library *l) { (void)l; return NULL; }
void          ass_renderer_done(ASS_Renderer *r) { (void)r; }
void          ass_set_use_margins(ASS_Renderer *r, int u) { (void)r; (void)u; }
void          ass_set_hinting(ASS_Renderer *r, int h) { (void)r; (void)h; }
void          ass_set_font_scale(ASS_Renderer *r, double s) { (void)r; (void)s; }
void          ass_set_line_spacing(ASS_Renderer *r, double s) { (void)r; (void)s; }
void          ass_set_fonts(ASS_Renderer *r, const char *f, const char *fam, int fc, const char *cfg, int upd) { (void)r; (void)f; (void)fam; (void)fc; (void)cfg; (void)upd; }
void          ass_set_frame_size(ASS_Renderer *r, int w, int h) { (void)r; (void)w; (void)h; }
void          ass_set_storage_size(ASS_Renderer *r, int w, int h) { (void)r; (void)w; (void)h; }
void          ass_set_pixel_aspect(ASS_Renderer *r, double p) { (void)r; (void)p; }
ASS_Track    *ass_new_track(ASS_Library *l) { (void)l; return NULL; }
void          ass_free_track(ASS_Track *t) { (void)t; }
void          ass_set_check_readorder(ASS_Track *t, int c) { (void)t; (void)c; }
void          ass_process_codec_private(ASS_Track *t, const char *d, int s) { (void)t; (void)d; (void)s; }
void          ass_process_chunk(ASS_Track *t, const char *d, int s, long long a, long long b) { (void)t; (void)d; (void)s; (void)a; (void)b; }
void          ass_process_data(ASS_Track *t, const char *d, int s) { (void)t; (void)d; (void)s; }
ASS_Image    *ass_render_frame(ASS_Renderer *r, ASS_Track *t, long long now, int *chg) { (void)r; (void)t; (void)now; if (chg) *chg = 0; return NULL; }

/* ---- the compositor choice (see the head of the file) -------------------------------------------------------------- */
static hb_blend_object_t *hbref_blend_for(int hw_pix_fmt)
{
    hb_blend_object_t *b = hbhip_rt_hw_helper(1, hw_pix_fmt);
    return b != NULL ? b : &hb_blend;
}
#define hb_blend (*hbref_blend_for(init.hw_pix_fmt))
#include "rendersub.c"
#undef hb_blend
