/* decomb_oracle.c — CPU restatement of decomb's yadif / blend / cubic line
 * filters (8-bit and 16-bit samples).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows /root/reference/libhb/templates/decomb_template.c:
 *   :23-48   crop table (+-1024 guard) and the (-3,23,23,-3)/40 cubic
 *   :50-107  cubic_interpolate_line with its top/bottom sample doubling
 *   :279-361 blend (-1,2,6,2,-1)>>3 with its edge rows
 *   :579-712 yadif_filter_line (temporal clamp, spatial search, EEDI2 guess)
 *   :714-808 which rows are filtered / copied
 *   :810-898 mode dispatch of filter_8
 * The reference works on row segments per thread; results do not depend on
 * the segmentation (segment starts are even rows), so whole planes are walked.
 */
#include "oracle.h"

#include <stddef.h>
#include <stdlib.h>
#include <string.h>

static inline int iabs(int v) { return v < 0 ? -v : v; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin3(int a, int b, int c) { return imin(imin(a, b), c); }
static inline int imax3(int a, int b, int c) { return imax(imax(a, b), c); }

/* crop_table[v + 1024] (:23-41): 0 below 0, v inside, max_value above. */
static inline int cropv(int v, int maxv) { return v < 0 ? 0 : v > maxv ? maxv : v; }

#define PIXEL uint8_t
#define PX(n) n##_8
#include "decomb_oracle_px.h"
#undef PIXEL
#undef PX
#define PIXEL uint16_t
#define PX(n) n##_16
#include "decomb_oracle_px.h"
#undef PIXEL
#undef PX

void orc_decomb_plane(const uint8_t *prev, const uint8_t *cur, const uint8_t *next, int stride,
                      const uint8_t *guess, int guess_stride,
                      uint8_t *dst, int dst_stride, int width, int height,
                      int mode, int parity, int tff)
{
    decomb_plane_8(prev, cur, next, stride, guess, guess_stride, dst, dst_stride, width, height, mode, parity, tff, 255);
}

void orc_decomb_plane16(const uint16_t *prev, const uint16_t *cur, const uint16_t *next, int stride,
                        const uint16_t *guess, int guess_stride,
                        uint16_t *dst, int dst_stride, int width, int height,
                        int mode, int parity, int tff, int depth)
{
    decomb_plane_16(prev, cur, next, stride, guess, guess_stride, dst, dst_stride, width, height, mode, parity, tff,
                    (1 << depth) - 1);
}

/* ---- FFmpeg yadif = the reference's "Deinterlace" filter (deinterlace.c:43-143) -----------------
 * PARITY UNPINNED: the arithmetic is libavfilter/vf_yadif.c (filter_line_c / filter_edges /
 * filter_slice), which is not part of /root/reference; restated from its published source.
 * One plane: rows with ((y ^ parity) & 1) are rebuilt, the others copied from `cur`.
 * field_parity = parity ^ tff picks the same-parity neighbours (prev2 / next2); nospatial = the
 * send_*_nospatial modes (no vertical-neighbour widening of the temporal bound). */
static inline int yd_px(const void *p, long at, int bps)
{
    return bps == 1 ? ((const uint8_t *)p)[at] : ((const uint16_t *)p)[at];
}

void orc_yadif_ff_plane(const void *prev, const void *cur, const void *next, int stride, int w, int h,
                        void *dst, int dst_stride, int parity, int tff, int nospatial, int bps)
{
    const int st = stride / bps, dstp = dst_stride / bps;
    const int field_parity = parity ^ tff;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            const long at = (long)y * st + x;
            int out;
            if (!((y ^ parity) & 1))
                out = yd_px(cur, at, bps);
            else
            {
                const int prefs = y + 1 < h ? st : -st, mrefs = y ? -st : st;
                const int mode2 = nospatial || y == 1 || y + 2 == h;
                const void *prev2 = field_parity ? prev : cur, *next2 = field_parity ? cur : next;
                const int c = yd_px(cur, at + mrefs, bps), e = yd_px(cur, at + prefs, bps);
                const int d = (yd_px(prev2, at, bps) + yd_px(next2, at, bps)) >> 1;
                const int td0 = abs(yd_px(prev2, at, bps) - yd_px(next2, at, bps));
                const int td1 = (abs(yd_px(prev, at + mrefs, bps) - c) + abs(yd_px(prev, at + prefs, bps) - e)) >> 1;
                const int td2 = (abs(yd_px(next, at + mrefs, bps) - c) + abs(yd_px(next, at + prefs, bps) - e)) >> 1;
                int diff = td0 >> 1;
                if (td1 > diff) diff = td1;
                if (td2 > diff) diff = td2;
                int pred = (c + e) >> 1;
                if (x >= 3 && x < w - 3)
                {
                    int score = abs(yd_px(cur, at + mrefs - 1, bps) - yd_px(cur, at + prefs - 1, bps)) + abs(c - e) +
                                abs(yd_px(cur, at + mrefs + 1, bps) - yd_px(cur, at + prefs + 1, bps)) - 1;
                    for (int side = -1; side <= 1; side += 2)
                        for (int j = side; j == side || j == 2 * side; j += side)
                        {
                            const int s = abs(yd_px(cur, at + mrefs - 1 + j, bps) - yd_px(cur, at + prefs - 1 - j, bps)) +
                                          abs(yd_px(cur, at + mrefs + j, bps) - yd_px(cur, at + prefs - j, bps)) +
                                          abs(yd_px(cur, at + mrefs + 1 + j, bps) - yd_px(cur, at + prefs + 1 - j, bps));
                            if (s >= score) break;                       /* +-2 is only tried when +-1 improved */
                            score = s;
                            pred = (yd_px(cur, at + mrefs + j, bps) + yd_px(cur, at + prefs - j, bps)) >> 1;
                        }
                }
                if (!mode2)
                {
                    const int b = (yd_px(prev2, at + 2 * mrefs, bps) + yd_px(next2, at + 2 * mrefs, bps)) >> 1;
                    const int f = (yd_px(prev2, at + 2 * prefs, bps) + yd_px(next2, at + 2 * prefs, bps)) >> 1;
                    const int bc = b - c, fe = f - e, dc = d - c, de = d - e;
                    int mx = de > dc ? de : dc, mn = de < dc ? de : dc;
                    const int lo = bc < fe ? bc : fe, hi = bc > fe ? bc : fe;
                    if (lo > mx) mx = lo;
                    if (hi < mn) mn = hi;
                    if (mn > diff) diff = mn;
                    if (-mx > diff) diff = -mx;
                }
                if (pred > d + diff) pred = d + diff;
                else if (pred < d - diff) pred = d - diff;
                out = pred;
            }
            if (bps == 1) ((uint8_t *)dst)[(long)y * dstp + x] = (uint8_t)out;
            else          ((uint16_t *)dst)[(long)y * dstp + x] = (uint16_t)out;
        }
}
