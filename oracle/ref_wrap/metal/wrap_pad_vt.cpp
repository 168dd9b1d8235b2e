// Compiles the reference's libhb/platform/macosx/shaders/pad_vt.metal in place, unmodified, as host C++ (metal_wrap.h).
#include "metal_wrap.h"
namespace {          // every shader defines its own tex2D / params / deint: keep them local to this file
#include "pad_vt.metal"
}

// one plane: the picture at (x, y) of a dw x dh plane, the rest `color` (a normalised sample value)
HBMTL_EXPORT void hbmtl_pad_plane(uint8_t *dst, int dpitch, int dw, int dh, const uint8_t *src, int spitch, int sw, int sh,
                                  int x, int y, float color)
{
    params p;
    p.plane = 0; p.channels = 1; p.color_y = color; p.color_u = color; p.color_v = color; p.x = (uint)x; p.y = (uint)y;
    texture2d<float, access::write> d(hbmtl_plane(dst, dpitch, dw, dh));
    texture2d<float, access::read> s(hbmtl_plane(src, spitch, sw, sh));
    for (int yy = 0; yy < dh; yy++)
        for (int xx = 0; xx < dw; xx++)
            pad(d, s, p, ushort2((ushort)xx, (ushort)yy));
}
