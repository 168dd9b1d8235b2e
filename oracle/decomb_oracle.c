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
 * send_*_nospatial modes (no vertical-neighbour widening of the temporal bound).
 *
 * The reference tree does hold one restatement of this filter: platform/macosx/shaders/yadif_vt.metal (a
 * port of FFmpeg's vf_yadif_cuda, used by the VideoToolbox build).  Line by line against it:
 *   kept field copied                          metal :264-269  `pos.y % 2 == params.parity`  = rows with !((y ^ parity) & 1)
 *   prev2 / next2 choice                       metal :241-245  is_second_field ? (prev, cur | next, next) : (prev, prev | cur, next)
 *                                                               = prev2 = field_parity ? prev : cur, next2 = field_parity ? cur : next
 *   c, d, e and the three temporal differences  metal :121-131  p1 = F, p2 = (D + I) / 2, p3 = G, tdiff0..2, diff = max3
 *                                                               (here td0 is halved before the max, as filter_line_c does: `temporal_diff0 >> 1`;
 *                                                               the shader's `tdiff0 = abs(D - I)` is not - a difference of the CUDA port)
 *   vertical-neighbour widening                 metal :133-137  maxi / mini over p0..p4  = b, f, max / min below
 *   spatial prediction and the +-1, +-2 checks  metal :87-114   (d + k) / 2 then the two nested score tests each side = CHECK(-1) CHECK(-2), CHECK(1) CHECK(2)
 *   clamp to d +- diff                          metal :139
 * Where the two differ this file follows vf_yadif.c, which is what libhb's own (non-Apple) Deinterlace filter
 * runs through libavfilter: integer arithmetic with `>> 1` where the shader divides floats by 2, the first
 * spatial score reduced by 1 (`- 1` in filter_line_c, absent in the shader), and no diagonal checks within 3
 * columns of the left / right edge (filter_edges) where the shader's sampler clamps coordinates instead. */
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

/* ---- FFmpeg bwdif, the reference's "Bwdif" filter (deinterlace.c:46 -> libavfilter vf_bwdif.c; PARITY UNPINNED) --
 * libavfilter is not in the reference tree (FFmpeg 9.0.1, contrib/ffmpeg/module.defs:15-17).  This restates
 * vf_bwdif.c's C line filters (filter_intra / filter_line / filter_edge and the row dispatch of filter_slice)
 * from the published algorithm, in integers as the CPU filter computes.  The reference tree holds a second
 * restatement, platform/macosx/shaders/bwdif_vt.metal (a port of vf_bwdif_cuda), which this follows
 * operation for operation where the two agree:
 *   coefficients                         metal :67-69   = coef_lf / coef_hf / coef_sp
 *   intra (field-end) filter             metal :71-78   = FILTER_INTRA
 *   temporal differences, d, `!diff`     metal :89-99   = FILTER1
 *   vertical-neighbour widening          metal :101-107 = SPAT_CHECK
 *   high-frequency / spatial interpol    metal :110-117 = FILTER_LINE
 *   clamp to d +- diff, clip             metal :119-124 = FILTER2
 *   prev2/prev1/next1/next2 choice       metal :152-156 = `prev2 = parity ? prev : cur; next2 = parity ? cur : next`
 * and departs from it where the Metal port departs from the C filter: (i) integer >> instead of float
 * division; (ii) the last high-frequency tap is next2[prefs4] (the shader repeats next2_mrefs4, :113);
 * (iii) the C filter has explicit edge rows: y < 4 or y + 5 > h use filter_edge ((c + e) >> 1, with the
 * vertical check only where rows y +- 2 exist), and rows whose +-1 / +-3 neighbours fall outside are
 * mirrored with vf_bwdif.c's own tests, which compare against the sample SIZE df:
 * (y + df) < h, y > df - 1, (y + 3 df) < h, y > 3 df - 1.
 * field_end = yadif->current_field == YADIF_FIELD_END (first field of a stream, last field of a bob stream). */
void orc_bwdif_plane(const void *prev, const void *cur, const void *next, int stride, int w, int h,
                     void *dst, int dst_stride, int parity, int tff, int field_end, int bps, int depth)
{
    static const int coef_lf[2] = { 4309, 213 }, coef_hf[3] = { 5570, 3801, 1016 }, coef_sp[2] = { 5077, 981 };
    const int st = stride / bps, dstp = dst_stride / bps, df = bps;
    const int clip_max = (1 << depth) - 1;
    const int fp = parity ^ tff;
    const void *prev2 = fp ? prev : cur, *next2 = fp ? cur : next;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            const long at = (long)y * st + x;
            int out;
            if (!((y ^ parity) & 1))
                out = yd_px(cur, at, bps);
            else if (field_end)
            {
                const int prefs = (y + df) < h ? st : -st, mrefs = y > (df - 1) ? -st : st;
                const int prefs3 = (y + 3 * df) < h ? 3 * st : -st, mrefs3 = y > (3 * df - 1) ? -3 * st : st;
                int interpol = (coef_sp[0] * (yd_px(cur, at + mrefs, bps) + yd_px(cur, at + prefs, bps)) -
                                coef_sp[1] * (yd_px(cur, at + mrefs3, bps) + yd_px(cur, at + prefs3, bps))) >> 13;
                out = interpol < 0 ? 0 : interpol > clip_max ? clip_max : interpol;
            }
            else
            {
                const int edge = (y < 4) || ((y + 5) > h);
                const int prefs = edge ? ((y + df) < h ? st : -st) : st, mrefs = edge ? (y > (df - 1) ? -st : st) : -st;
                const int prefs2 = 2 * st, mrefs2 = -2 * st;
                const int c = yd_px(cur, at + mrefs, bps), e = yd_px(cur, at + prefs, bps);
                const int p0 = yd_px(prev2, at, bps), n0 = yd_px(next2, at, bps);
                const int d = (p0 + n0) >> 1;
                const int td0 = abs(p0 - n0);
                const int td1 = (abs(yd_px(prev, at + mrefs, bps) - c) + abs(yd_px(prev, at + prefs, bps) - e)) >> 1;
                const int td2 = (abs(yd_px(next, at + mrefs, bps) - c) + abs(yd_px(next, at + prefs, bps) - e)) >> 1;
                int diff = td0 >> 1;
                if (td1 > diff) diff = td1;
                if (td2 > diff) diff = td2;
                if (!diff)
                    out = d;
                else
                {
                    const int spat = edge ? !((y < 2) || ((y + 3) > h)) : 1;
                    if (spat)
                    {
                        const int b = ((yd_px(prev2, at + mrefs2, bps) + yd_px(next2, at + mrefs2, bps)) >> 1) - c;
                        const int f = ((yd_px(prev2, at + prefs2, bps) + yd_px(next2, at + prefs2, bps)) >> 1) - e;
                        const int dc = d - c, de = d - e;
                        int mx = de > dc ? de : dc, mn = de < dc ? de : dc;
                        const int lo = b < f ? b : f, hi = b > f ? b : f;
                        if (lo > mx) mx = lo;
                        if (hi < mn) mn = hi;
                        if (mn > diff) diff = mn;
                        if (-mx > diff) diff = -mx;
                    }
                    int interpol;
                    if (edge)
                        interpol = (c + e) >> 1;
                    else
                    {
                        const int c3 = yd_px(cur, at - 3 * st, bps) + yd_px(cur, at + 3 * st, bps);
                        if (abs(c - e) > td0)
                            interpol = (((coef_hf[0] * (p0 + n0)
                                          - coef_hf[1] * (yd_px(prev2, at + mrefs2, bps) + yd_px(next2, at + mrefs2, bps) +
                                                          yd_px(prev2, at + prefs2, bps) + yd_px(next2, at + prefs2, bps))
                                          + coef_hf[2] * (yd_px(prev2, at - 4 * st, bps) + yd_px(next2, at - 4 * st, bps) +
                                                          yd_px(prev2, at + 4 * st, bps) + yd_px(next2, at + 4 * st, bps))) >> 2)
                                        + coef_lf[0] * (c + e) - coef_lf[1] * c3) >> 13;
                        else
                            interpol = (coef_sp[0] * (c + e) - coef_sp[1] * c3) >> 13;
                    }
                    if (interpol > d + diff) interpol = d + diff;
                    else if (interpol < d - diff) interpol = d - diff;
                    out = interpol < 0 ? 0 : interpol > clip_max ? clip_max : interpol;
                }
            }
            if (bps == 1) ((uint8_t *)dst)[(long)y * dstp + x] = (uint8_t)out;
            else          ((uint16_t *)dst)[(long)y * dstp + x] = (uint16_t)out;
        }
}
