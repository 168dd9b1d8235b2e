/* stub of libass's <ass/ass.h> for compiling the reference's rendersub.c in place (oracle/ref_wrap/wrap_rendersub.c).
 * libass is not in the image; the SSA / text subtitle paths that call it are never taken by the tests (they burn in
 * bitmap subtitles: VOBSUB / PGS), so the functions are declared here and defined in the wrapper as stubs that fail. */
#ifndef HBREF_ASS_STUB_H
#define HBREF_ASS_STUB_H
#include <stdarg.h>
#include <stdint.h>
typedef struct ass_library ASS_Library;
typedef struct ass_renderer ASS_Renderer;
typedef struct ass_track { int YCbCrMatrix; } ASS_Track;
typedef struct ass_image
{
    int w, h, stride;
    unsigned char *bitmap;
    uint32_t color;
    int dst_x, dst_y;
    struct ass_image *next;
} ASS_Image;
enum { ASS_HINTING_NONE = 0 };
enum { YCBCR_DEFAULT = 0, YCBCR_UNKNOWN, YCBCR_NONE, YCBCR_BT601_TV, YCBCR_BT601_PC, YCBCR_BT709_TV, YCBCR_BT709_PC,
       YCBCR_SMPTE240M_TV, YCBCR_SMPTE240M_PC, YCBCR_FCC_TV, YCBCR_FCC_PC };
ASS_Library  *ass_library_init(void);
void          ass_library_done(ASS_Library *);
void          ass_set_message_cb(ASS_Library *, void (*cb)(int, const char *, va_list, void *), void *);
void          ass_set_extract_fonts(ASS_Library *, int);
void          ass_add_font(ASS_Library *, const char *, const char *, int);
void          ass_set_style_overrides(ASS_Library *, char **);
ASS_Renderer *ass_renderer_init(ASS_Library *);
void          ass_renderer_done(ASS_Renderer *);
void          ass_set_use_margins(ASS_Renderer *, int);
void          ass_set_hinting(ASS_Renderer *, int);
void          ass_set_font_scale(ASS_Renderer *, double);
void          ass_set_line_spacing(ASS_Renderer *, double);
void          ass_set_fonts(ASS_Renderer *, const char *, const char *, int, const char *, int);
void          ass_set_frame_size(ASS_Renderer *, int, int);
void          ass_set_storage_size(ASS_Renderer *, int, int);
void          ass_set_pixel_aspect(ASS_Renderer *, double);
ASS_Track    *ass_new_track(ASS_Library *);
void          ass_free_track(ASS_Track *);
void          ass_process_codec_private(ASS_Track *, const char *, int);
void          ass_process_chunk(ASS_Track *, const char *, int, long long, long long);
void          ass_process_data(ASS_Track *, const char *, int);
ASS_Image    *ass_render_frame(ASS_Renderer *, ASS_Track *, long long, int *);
#endif
