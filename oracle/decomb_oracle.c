/* decomb_oracle.c — CPU restatement of decomb's yadif / blend / cubic line
 * filters (8-bit).  TEST INFRASTRUCTURE ONLY (see oracle.h).
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

/* crop_table[v + 1024] (:23-41): 0 below 0, v inside, 255 above; valid for
 * v in [-1024, 255 + 1024). */
static inline int crop8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

/* :43-48 (C division truncates toward zero) */
static inline int cubic4(int y0, int y1, int y2, int y3)
{
    return crop8((y0 * -3 + y1 * 23 + y2 * 23 + y3 * -3) / 40);
}

/* :50-107 */
static void cubic_line(uint8_t *dst, const uint8_t *cur, int width, int height, int stride, int y)
{
    for (int x = 0; x < width; x++)
    {
        const uint8_t *p = cur + x;
        int a = 0, b = 0, c = 0, d = 0;
        if (y >= 3)                  { a = p[-3 * stride]; b = p[-stride]; }
        else if (y == 2 || y == 1)   { a = b = p[-stride]; }
        else if (y == 0)             { a = b = p[stride]; }
        if (y <= height - 4)                         { c = p[stride]; d = p[3 * stride]; }
        else if (y == height - 3 || y == height - 2) { c = d = p[stride]; }
        else if (y == height - 1)                    { c = d = p[-stride]; }
        dst[x] = cubic4(a, b, c, d);
    }
}

/* :279-361 */
static void blend_line(uint8_t *dst, const uint8_t *cur, int width, int height, int stride, int y)
{
    int u1, u2, d1, d2;
    if (y > 1 && y < height - 2) { u1 = -stride; u2 = -2 * stride; d1 = stride; d2 = 2 * stride; }
    else if (y == 0)             { u1 = u2 = 0; d1 = stride; d2 = 2 * stride; }
    else if (y == 1)             { u1 = u2 = -stride; d1 = stride; d2 = 2 * stride; }
    else if (y == height - 2)    { u1 = -stride; u2 = -2 * stride; d1 = d2 = stride; }
    else                         { u1 = -stride; u2 = -2 * stride; d1 = d2 = 0; }
    for (int x = 0; x < width; x++)
    {
        const uint8_t *p = cur + x;
        const int v = (-p[u2] + 2 * p[u1] + 6 * p[0] + 2 * p[d1] - p[d2]) >> 3;
        dst[x] = crop8(v);
    }
}

/* one spatial candidate of YADIF_CHECK (:530-577): returns 1 when it improved the score */
static int yadif_check(const uint8_t *cur, int sp, int sn, int stride, int j, int cubic_ok,
                       int *score_best, int *pred)
{
    const int score = iabs(cur[sp - 1 + j] - cur[sn - 1 - j]) +
                      iabs(cur[sp + j] - cur[sn - j]) +
                      iabs(cur[sp + 1 + j] - cur[sn + 1 - j]);
    if (score >= *score_best)
        return 0;
    *score_best = score;
    if (cubic_ok)
    {
        switch (j)
        {
            case -1: *pred = cubic4(cur[-3 * stride - 3], cur[-stride - 1], cur[stride + 1], cur[3 * stride + 3]); break;
            case -2: *pred = cubic4((cur[-3 * stride - 4] + cur[-stride - 4]) / 2, cur[-stride - 2],
                                    cur[stride + 2], (cur[3 * stride + 4] + cur[stride + 4]) / 2); break;
            case 1:  *pred = cubic4(cur[-3 * stride + 3], cur[-stride + 1], cur[stride - 1], cur[3 * stride - 3]); break;
            case 2:  *pred = cubic4((cur[-3 * stride + 4] + cur[-stride + 4]) / 2, cur[-stride + 2],
                                    cur[stride - 2], (cur[3 * stride - 4] + cur[stride - 4]) / 2); break;
        }
    }
    else
    {
        *pred = (cur[sp + j] + cur[sn - j]) >> 1;
    }
    return 1;
}

/* :579-712.  `field_parity` is the reference's `parity ^ tff` argument. */
static void yadif_line(uint8_t *dst, const uint8_t *prev, const uint8_t *cur, const uint8_t *next,
                       int stride, const uint8_t *guess, int width, int height,
                       int field_parity, int y, int mode)
{
    const uint8_t *prev2 = field_parity ? prev : cur;
    const uint8_t *next2 = field_parity ? cur : next;
    const int sp = y ? -stride : stride;                 /* mirrored at the first row */
    const int sn = y + 1 < height ? stride : -stride;    /* and at the last           */
    const int vertical_edge = (y < 3) || (y > height - 4);
    const int use_cubic = (mode & ORC_DECOMB_CUBIC) && !vertical_edge;
    const int margin = (mode & ORC_DECOMB_CUBIC) ? 3 : 2;

    for (int x = 0; x < width; x++)
    {
        const uint8_t *pc = cur + x, *pp = prev + x, *pn = next + x, *p2 = prev2 + x, *n2 = next2 + x;
        const int c = pc[sp];
        const int d = (p2[0] + n2[0]) >> 1;
        const int e = pc[sn];
        const int td0 = iabs(p2[0] - n2[0]);
        const int td1 = (iabs(pp[sp] - c) + iabs(pp[sn] - e)) >> 1;
        const int td2 = (iabs(pn[sp] - c) + iabs(pn[sn] - e)) >> 1;
        int diff = imax3(td0 >> 1, td1, td2);
        int pred;

        if (mode & ORC_DECOMB_EEDI2)
        {
            pred = guess[x];
        }
        else
        {
            pred = use_cubic ? cubic4(pc[-3 * stride], pc[-stride], pc[stride], pc[3 * stride]) : (c + e) >> 1;
            if (x > margin && x < width - (margin + 1))
            {
                int best = iabs(pc[sp - 1] - pc[sn - 1]) + iabs(c - e) + iabs(pc[sp + 1] - pc[sn + 1]) - 1;
                /* -1 then, only if it helped, -2; same for +1, +2 */
                if (yadif_check(pc, sp, sn, stride, -1, use_cubic, &best, &pred))
                    yadif_check(pc, sp, sn, stride, -2, use_cubic, &best, &pred);
                if (yadif_check(pc, sp, sn, stride, 1, use_cubic, &best, &pred))
                    yadif_check(pc, sp, sn, stride, 2, use_cubic, &best, &pred);
            }
        }

        if (!vertical_edge)
        {
            const int b = (p2[-2 * stride] + n2[-2 * stride]) >> 1;
            const int f = (p2[2 * stride] + n2[2 * stride]) >> 1;
            const int mx = imax3(d - e, d - c, imin(b - c, f - e));
            const int mn = imin3(d - e, d - c, imax(b - c, f - e));
            diff = imax3(diff, mn, -mx);
        }
        if (pred > d + diff)      pred = d + diff;
        else if (pred < d - diff) pred = d - diff;
        dst[x] = (uint8_t)pred;
    }
}

void orc_decomb_plane(const uint8_t *prev, const uint8_t *cur, const uint8_t *next, int stride,
                      const uint8_t *guess, int guess_stride,
                      uint8_t *dst, int dst_stride, int width, int height,
                      int mode, int parity, int tff)
{
    if (mode == 0)
    {
        /* hb_buffer_copy(dst, ref[1]) (:895-896) */
        for (int y = 0; y < height; y++)
            memcpy(dst + (size_t)y * dst_stride, cur + (size_t)y * stride, width);
        return;
    }
    if ((mode & ORC_DECOMB_EEDI2) && !(mode & ORC_DECOMB_YADIF))
    {
        /* pass the EEDI2 interpolation through (:855-875) */
        for (int y = 0; y < height; y++)
            memcpy(dst + (size_t)y * dst_stride, guess + (size_t)y * guess_stride, width);
        return;
    }
    const int first = parity ? 0 : 1;          /* rows of this parity are rebuilt (:737, :797) */
    for (int y = 0; y < height; y++)
    {
        uint8_t *o = dst + (size_t)y * dst_stride;
        const uint8_t *c = cur + (size_t)y * stride;
        if ((y & 1) != first)
        {
            memcpy(o, c, width);
            continue;
        }
        if (mode == ORC_DECOMB_BLEND)
            blend_line(o, c, width, height, stride, y);
        else if (mode == ORC_DECOMB_CUBIC)
            cubic_line(o, c, width, height, stride, y);
        else if (mode & ORC_DECOMB_YADIF)
            yadif_line(o, prev + (size_t)y * stride, c, next + (size_t)y * stride, stride,
                       guess ? guess + (size_t)y * guess_stride : NULL, width, height, parity ^ tff, y, mode);
        /* any other combination leaves the row untouched, as the reference does */
    }
}
