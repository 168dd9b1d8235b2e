// Compiles the reference's libhb/platform/macosx/shaders/yadif_vt.metal in place, unmodified, as host C++ (metal_wrap.h).
#include "metal_wrap.h"
namespace {          // every shader defines its own tex2D / params / deint: keep them local to this file
#include "yadif_vt.metal"
}

// one plane of 8-bit samples through `deint`; parity: rows y % 2 == parity are kept
HBMTL_EXPORT void hbmtl_yadif_plane(uint8_t *dst, int dpitch, const uint8_t *prev, const uint8_t *cur, const uint8_t *next, int pitch,
                                    int w, int h, int parity, int tff, int is_second_field, int skip_spatial_check)
{
    deintParams p;
    p.channels = 1; p.parity = (uint)parity; p.tff = (uint)tff; p.is_second_field = is_second_field != 0;
    p.skip_spatial_check = skip_spatial_check != 0; p.is_field_end = false;
    texture2d<float, access::write> d(hbmtl_plane(dst, dpitch, w, h));
    texture2d<float, accesstype> tp(hbmtl_plane(prev, pitch, w, h)), tc(hbmtl_plane(cur, pitch, w, h)), tn(hbmtl_plane(next, pitch, w, h));
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            deint(d, tp, tc, tn, p, uint2((uint)x, (uint)y));
}
