/* decomb_oracle_px.h — body of decomb_oracle.c, written once for `PIXEL` and included for
 * uint8_t and uint16_t (the reference instantiates its template the same two ways,
 * decomb.c:314-331).  Strides are in samples.  TEST INFRASTRUCTURE ONLY. */
/* :43-48 (C division truncates toward zero) */
static inline int PX(cubic4)(int y0, int y1, int y2, int y3, int maxv)
{
    return cropv((y0 * -3 + y1 * 23 + y2 * 23 + y3 * -3) / 40, maxv);
}

/* :50-107 */
static void PX(cubic_line)(PIXEL *dst, const PIXEL *cur, int width, int height, int stride, int y, int maxv)
{
    for (int x = 0; x < width; x++)
    {
        const PIXEL *p = cur + x;
        int a = 0, b = 0, c = 0, d = 0;
        if (y >= 3)                  { a = p[-3 * stride]; b = p[-stride]; }
        else if (y == 2 || y == 1)   { a = b = p[-stride]; }
        else if (y == 0)             { a = b = p[stride]; }
        if (y <= height - 4)                         { c = p[stride]; d = p[3 * stride]; }
        else if (y == height - 3 || y == height - 2) { c = d = p[stride]; }
        else if (y == height - 1)                    { c = d = p[-stride]; }
        dst[x] = PX(cubic4)(a, b, c, d, maxv);
    }
}

/* :279-361 */
static void PX(blend_line)(PIXEL *dst, const PIXEL *cur, int width, int height, int stride, int y, int maxv)
{
    int u1, u2, d1, d2;
    if (y > 1 && y < height - 2) { u1 = -stride; u2 = -2 * stride; d1 = stride; d2 = 2 * stride; }
    else if (y == 0)             { u1 = u2 = 0; d1 = stride; d2 = 2 * stride; }
    else if (y == 1)             { u1 = u2 = -stride; d1 = stride; d2 = 2 * stride; }
    else if (y == height - 2)    { u1 = -stride; u2 = -2 * stride; d1 = d2 = stride; }
    else                         { u1 = -stride; u2 = -2 * stride; d1 = d2 = 0; }
    for (int x = 0; x < width; x++)
    {
        const PIXEL *p = cur + x;
        const int v = (-p[u2] + 2 * p[u1] + 6 * p[0] + 2 * p[d1] - p[d2]) >> 3;
        dst[x] = cropv(v, maxv);
    }
}

/* one spatial candidate of YADIF_CHECK (:530-577): returns 1 when it improved the score */
static int PX(yadif_check)(const PIXEL *cur, int sp, int sn, int stride, int j, int cubic_ok, int maxv,
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
            case -1: *pred = PX(cubic4)(cur[-3 * stride - 3], cur[-stride - 1], cur[stride + 1], cur[3 * stride + 3], maxv); break;
            case -2: *pred = PX(cubic4)((cur[-3 * stride - 4] + cur[-stride - 4]) / 2, cur[-stride - 2],
                                    cur[stride + 2], (cur[3 * stride + 4] + cur[stride + 4]) / 2, maxv); break;
            case 1:  *pred = PX(cubic4)(cur[-3 * stride + 3], cur[-stride + 1], cur[stride - 1], cur[3 * stride - 3], maxv); break;
            case 2:  *pred = PX(cubic4)((cur[-3 * stride + 4] + cur[-stride + 4]) / 2, cur[-stride + 2],
                                    cur[stride - 2], (cur[3 * stride - 4] + cur[stride - 4]) / 2, maxv); break;
        }
    }
    else
    {
        *pred = (cur[sp + j] + cur[sn - j]) >> 1;
    }
    return 1;
}

/* :579-712.  `field_parity` is the reference's `parity ^ tff` argument. */
static void PX(yadif_line)(PIXEL *dst, const PIXEL *prev, const PIXEL *cur, const PIXEL *next,
                       int stride, const PIXEL *guess, int width, int height,
                       int field_parity, int y, int mode, int maxv)
{
    const PIXEL *prev2 = field_parity ? prev : cur;
    const PIXEL *next2 = field_parity ? cur : next;
    const int sp = y ? -stride : stride;                 /* mirrored at the first row */
    const int sn = y + 1 < height ? stride : -stride;    /* and at the last           */
    const int vertical_edge = (y < 3) || (y > height - 4);
    const int use_cubic = (mode & ORC_DECOMB_CUBIC) && !vertical_edge;
    const int margin = (mode & ORC_DECOMB_CUBIC) ? 3 : 2;

    for (int x = 0; x < width; x++)
    {
        const PIXEL *pc = cur + x, *pp = prev + x, *pn = next + x, *p2 = prev2 + x, *n2 = next2 + x;
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
            pred = use_cubic ? PX(cubic4)(pc[-3 * stride], pc[-stride], pc[stride], pc[3 * stride], maxv) : (c + e) >> 1;
            if (x > margin && x < width - (margin + 1))
            {
                int best = iabs(pc[sp - 1] - pc[sn - 1]) + iabs(c - e) + iabs(pc[sp + 1] - pc[sn + 1]) - 1;
                /* -1 then, only if it helped, -2; same for +1, +2 */
                if (PX(yadif_check)(pc, sp, sn, stride, -1, use_cubic, maxv, &best, &pred))
                    PX(yadif_check)(pc, sp, sn, stride, -2, use_cubic, maxv, &best, &pred);
                if (PX(yadif_check)(pc, sp, sn, stride, 1, use_cubic, maxv, &best, &pred))
                    PX(yadif_check)(pc, sp, sn, stride, 2, use_cubic, maxv, &best, &pred);
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
        dst[x] = (PIXEL)pred;
    }
}

static void PX(decomb_plane)(const PIXEL *prev, const PIXEL *cur, const PIXEL *next, int stride,
                             const PIXEL *guess, int guess_stride,
                             PIXEL *dst, int dst_stride, int width, int height,
                             int mode, int parity, int tff, int maxv)
{
    if (mode == 0)
    {
        /* hb_buffer_copy(dst, ref[1]) (:895-896) */
        for (int y = 0; y < height; y++)
            memcpy(dst + (size_t)y * dst_stride, cur + (size_t)y * stride, sizeof(PIXEL) * width);
        return;
    }
    if ((mode & ORC_DECOMB_EEDI2) && !(mode & ORC_DECOMB_YADIF))
    {
        /* pass the EEDI2 interpolation through (:855-875) */
        for (int y = 0; y < height; y++)
            memcpy(dst + (size_t)y * dst_stride, guess + (size_t)y * guess_stride, sizeof(PIXEL) * width);
        return;
    }
    const int first = parity ? 0 : 1;          /* rows of this parity are rebuilt (:737, :797) */
    for (int y = 0; y < height; y++)
    {
        PIXEL *o = dst + (size_t)y * dst_stride;
        const PIXEL *c = cur + (size_t)y * stride;
        if ((y & 1) != first)
        {
            memcpy(o, c, sizeof(PIXEL) * width);
            continue;
        }
        if (mode == ORC_DECOMB_BLEND)
            PX(blend_line)(o, c, width, height, stride, y, maxv);
        else if (mode == ORC_DECOMB_CUBIC)
            PX(cubic_line)(o, c, width, height, stride, y, maxv);
        else if (mode & ORC_DECOMB_YADIF)
            PX(yadif_line)(o, prev + (size_t)y * stride, c, next + (size_t)y * stride, stride,
                       guess ? guess + (size_t)y * guess_stride : NULL, width, height, parity ^ tff, y, mode, maxv);
        /* any other combination leaves the row untouched, as the reference does */
    }
}
