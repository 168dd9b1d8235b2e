/* blend_oracle.c — CPU restatement of the reference's subtitle compositor for planar frames
 * (libhb/blend.c: blend8on8 :425-509, blend8on1x :511-604, blend_subsample_8on8 :236-328,
 * blend_subsample_8on1x :48-140; selection in hb_blend_init :788-846).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Pinned: tests/test_oracle_vs_ref.py runs it against the
 * reference's own hb_blend object compiled in place (oracle/ref_wrap/wrap_blend.c).
 *
 * The 8-bit functions are the 16-bit ones with shift 0 (max 255), so one body serves both; it is
 * written per DESTINATION sample (which overlay samples land on it) instead of per overlay row,
 * with the reference's arithmetic and rounding:
 *   same subsampling      dst = (dst * (max - a) + (src << shift) * a) / max            (truncating)
 *   4:4:4 overlay on subsampled frame
 *                         luma  (dst * (max - a) + (src << shift) * a + max/2) / max    (rounding)
 *                         chroma: every overlay sample under the chroma sample blends a COPY of the
 *                         current chroma value, the copies are averaged with the chroma-location
 *                         weights of hb_compute_chroma_smoothing_coefficient (common.c:7054-7091).
 * Where an overlay sticks out of the frame to the right / bottom the reference's loops can run past
 * the row / plane (its width test compares the overlay's size with the frame's, not its position,
 * :74-75); this restatement stops at the frame edge instead — visible samples are identical.
 * One defect of the reference is NOT reproduced: with the overlay in the frame's subsampling and
 * hanging over the left / top edge by an ODD number of samples, its chroma loop starts at
 * x0 >> wshift but addresses (left >> wshift) + xx (:485-505), i.e. one sample before the row —
 * the last sample of the previous row, or of the previous plane.  Those stray writes are skipped
 * here (and on the GPU); tests use even overhangs for that path.
 */
#include "oracle.h"

#include <stddef.h>

static inline unsigned getpx(const void *plane, int stride, int x, int y, int bps)
{
    const uint8_t *row = (const uint8_t *)plane + (size_t)y * stride;
    return bps == 1 ? row[x] : ((const uint16_t *)row)[x];
}
static inline void setpx(void *plane, int stride, int x, int y, int bps, unsigned v)
{
    uint8_t *row = (uint8_t *)plane + (size_t)y * stride;
    if (bps == 1) row[x] = (uint8_t)v; else ((uint16_t *)row)[x] = (uint16_t)v;
}

/* blend8on8 / blend8on1x: the overlay has the frame's chroma subsampling */
static void blend_same(void *const plane[3], const int stride[3], int width, int height, int bps, int shift,
                       int wshift, int hshift, const orc_overlay_t *o)
{
    const int left = o->x, top = o->y;
    const int x0 = left < 0 ? -left : 0, y0 = top < 0 ? -top : 0;
    int ww = o->width, hh = o->height;
    if (o->width - x0 > width - left) ww = width - left + x0;
    if (o->height - y0 > height - top) hh = height - top + y0;
    const unsigned max = (256u << shift) - 1;
    for (int yy = y0; yy < hh && yy + top < height; yy++)
        for (int xx = x0; xx < ww && left + xx < width; xx++)
        {
            const unsigned a = (unsigned)o->plane[3][(size_t)yy * o->stride[3] + xx] << shift;
            const unsigned s = (unsigned)o->plane[0][(size_t)yy * o->stride[0] + xx] << shift;
            const unsigned d = getpx(plane[0], stride[0], left + xx, yy + top, bps);
            setpx(plane[0], stride[0], left + xx, yy + top, bps, (d * (max - a) + s * a) / max);
        }
    const int cw = wshift ? (width + 1) >> 1 : width, ch = hshift ? (height + 1) >> 1 : height;
    for (int yy = y0 >> hshift; yy < hh >> hshift; yy++)
        for (int xx = x0 >> wshift; xx < ww >> wshift; xx++)
        {
            const int dx = (left >> wshift) + xx, dy = yy + (top >> hshift);
            if (dx < 0 || dy < 0 || dx >= cw || dy >= ch) continue;
            const unsigned a = (unsigned)o->plane[3][(size_t)(yy << hshift) * o->stride[3] + (xx << wshift)] << shift;
            for (int c = 1; c < 3; c++)
            {
                const unsigned s = (unsigned)o->plane[c][(size_t)yy * o->stride[c] + xx] << shift;
                const unsigned d = getpx(plane[c], stride[c], dx, dy, bps);
                setpx(plane[c], stride[c], dx, dy, bps, (d * (max - a) + s * a) / max);
            }
        }
}

/* blend_subsample_8on8 / blend_subsample_8on1x: a 4:4:4 overlay on a chroma-subsampled frame */
static void blend_subsample(void *const plane[3], const int stride[3], int width, int height, int bps, int shift,
                            int wshift, int hshift, const uint32_t coeffs[2][4], const orc_overlay_t *o)
{
    const int x0 = o->x, y0 = o->y;
    int x0c = x0 & ~((1 << wshift) - 1), y0c = y0 & ~((1 << hshift) - 1);
    if (x0c < 0) x0c = 0;
    if (y0c < 0) y0c = 0;
    const int ow = o->width <= width ? o->width : width;           /* :74-75 with left == x0 */
    const int oh = o->height <= height ? o->height : height;
    const unsigned max = (256u << shift) - 1;

    for (int yy = y0c; yy - y0 < oh && yy < height; yy++)
        for (int xx = x0c; xx - x0 < ow && xx < width; xx++)
        {
            const int ox = xx - x0, oy = yy - y0;
            if (ox >= 0 && oy >= 0)
            {
                const unsigned a = (unsigned)o->plane[3][(size_t)oy * o->stride[3] + ox] << shift;
                const unsigned s = (unsigned)o->plane[0][(size_t)oy * o->stride[0] + ox] << shift;
                const unsigned d = getpx(plane[0], stride[0], xx, yy, bps);
                setpx(plane[0], stride[0], xx, yy, bps, (d * (max - a) + s * a + (max >> 1)) / max);
            }
            if ((yy & ((1 << hshift) - 1)) || (xx & ((1 << wshift) - 1))) continue;
            /* the chroma sample whose block starts here */
            unsigned acc[2] = { 0, 0 }, total = 0;
            const unsigned cur[2] = { getpx(plane[1], stride[1], xx >> wshift, yy >> hshift, bps),
                                      getpx(plane[2], stride[2], xx >> wshift, yy >> hshift, bps) };
            for (int yz = 0; yz < (1 << hshift) && oy + yz < oh; yz++)
                for (int xz = 0; xz < (1 << wshift) && ox + xz < ow; xz++)
                {
                    const unsigned coeff = coeffs[0][xz] * coeffs[1][yz];
                    unsigned r[2] = { cur[0], cur[1] };
                    if (ox + xz >= 0 && oy + yz >= 0)
                    {
                        const size_t at = (size_t)(oy + yz);
                        const unsigned a = (unsigned)o->plane[3][at * o->stride[3] + ox + xz] << shift;
                        for (int c = 0; c < 2; c++)
                        {
                            const unsigned s = (unsigned)o->plane[1 + c][at * o->stride[1 + c] + ox + xz] << shift;
                            r[c] = (r[c] * (max - a) + s * a + (max >> 1)) / max;
                        }
                    }
                    acc[0] += coeff * r[0];
                    acc[1] += coeff * r[1];
                    total += coeff;
                }
            if (total)
            {
                setpx(plane[1], stride[1], xx >> wshift, yy >> hshift, bps, (acc[0] + (total >> 1)) / total);
                setpx(plane[2], stride[2], xx >> wshift, yy >> hshift, bps, (acc[1] + (total >> 1)) / total);
            }
        }
}

/* the same window into 1 3 9 27 9 3 1 as hb_compute_chroma_smoothing_coefficient */
static void chroma_weights(uint32_t c[2][4], int wshift, int hshift, int loc)
{
    static const uint32_t base[] = { 1, 3, 9, 27, 9, 3, 1 };
    int wx = 4 - (1 << wshift), wy = 4 - (1 << hshift);
    if (loc == 1 || loc == 3 || loc == 5) wx += (1 << wshift) - 1;               /* left, topleft, bottomleft */
    if (loc == 3 || loc == 4 || loc == 5 || loc == 6) wy += (1 << hshift) - 1;   /* top*, bottom* (the switch falls through) */
    for (int i = 0; i < 4; i++)
    {
        c[0][i] = (base[i + wx] + base[i + wx + !(wx & 1)]) >> 1;
        c[1][i] = (base[i + wy] + base[i + wy + !(wy & 1)]) >> 1;
    }
}

int orc_blend_frame(void *const plane[3], const int stride[3], int width, int height, int depth, int wshift, int hshift,
                    int chroma_location, int overlay_wshift, int overlay_hshift, const orc_overlay_t *ov, int n)
{
    const int bps = depth > 8 ? 2 : 1, shift = depth - 8;
    const int subsample = wshift != overlay_wshift || hshift != overlay_hshift;
    if (subsample && (overlay_wshift || overlay_hshift)) return -1;      /* the reference indexes the overlay's chroma at full resolution */
    uint32_t coeffs[2][4];
    chroma_weights(coeffs, wshift, hshift, chroma_location);
    for (int i = 0; i < n; i++)                                          /* hb_blend_work :866-869: in list order */
    {
        if (subsample) blend_subsample(plane, stride, width, height, bps, shift, wshift, hshift, coeffs, &ov[i]);
        else           blend_same(plane, stride, width, height, bps, shift, wshift, hshift, &ov[i]);
    }
    return 0;
}
