// Compiles the reference's libhb/platform/macosx/shaders/grayscale_vt.metal in place (found through -I), unmodified, as
// host C++.  See metal_wrap.h.  Its function constants are plain globals here, set before the grid loop.
#include "metal_wrap.h"
namespace {          // every shader defines its own tex2D / params / deint: keep them local to this file
#include "grayscale_vt.metal"
}

// luma plane of `monochrome` (plane 0) for planar 8-bit YUV; chroma planes are the shader's constant 0.5
HBMTL_EXPORT void hbmtl_grayscale_luma(uint8_t *dst, int dpitch, const uint8_t *y, int ypitch, const uint8_t *u, const uint8_t *v,
                                       int cpitch, int w, int h, int sub_w, int sub_h, unsigned cb_, unsigned cr_, unsigned size_,
                                       unsigned high_)
{
    plane = 0; biplanar = false; subw = (uint)sub_w; subh = (uint)sub_h; cb = cb_; cr = cr_; size = size_; high = high_;
    const int cw = (w + (1 << sub_w) - 1) >> sub_w, ch = (h + (1 << sub_h) - 1) >> sub_h;
    texture2d<float, access::write> d(hbmtl_plane(dst, dpitch, w, h));
    texture2d<float, access::read> sy(hbmtl_plane(y, ypitch, w, h)), su(hbmtl_plane(u, cpitch, cw, ch)), sv(hbmtl_plane(v, cpitch, cw, ch));
    for (int yy = 0; yy < h; yy++)
        for (int xx = 0; xx < w; xx++)
            monochrome(d, sy, su, sv, ushort2((ushort)xx, (ushort)yy));
}

HBMTL_EXPORT int hbmtl_grayscale_chroma_value(void)
{
    uint8_t px = 0;
    plane = 1;
    texture2d<float, access::write> d(hbmtl_plane(&px, 1, 1, 1));
    texture2d<float, access::read> s(hbmtl_plane(&px, 1, 1, 1));
    monochrome(d, s, s, s, ushort2(0, 0));
    return px;
}
