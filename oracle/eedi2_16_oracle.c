/* eedi2_oracle16.c — CPU restatement of EEDI2 for the 16-bit template instantiation
 * (eedi2_template.c with pixel = uint16_t, decomb.c:324-331; depths 10 and 12).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Groundwork: the HIP EEDI2 passes are 8-bit only so far.
 *
 * Derived from our own 8-bit restatement (eedi2_oracle.c) pass by pass; what the reference does
 * differently above 8 bits is marked "16:" where it happens (thresholds shifted by depth-8 or typed
 * `pixel` = uint16 so that they wrap at 16 bits, PEAK / NEUTRAL from the depth, limlut << (depth-8),
 * sums and squares taken on samples >> (depth-8)).  Pinned plane by plane against the reference's own
 * eedi2_planer_16 (tests/test_oracle_vs_ref.py).  All pitches are in SAMPLES.
 */
#include "oracle.h"

#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define GUARD   4096      /* samples */

/* set by orc_eedi2_16_run_partial from the depth (test infrastructure: not re-entrant) */
static int PEAK, NEUTRAL, SHIFT;
static int LIMLUT[33];        /* eedi2_init_limlut (:23-33): eedi2_limlut << (depth - 8), typed pixel */

static const uint8_t LIMLUT8[33] = { 6, 6, 7, 7, 8, 8, 9, 9, 9, 10, 10, 11, 11, 12, 12, 12, 12, 12, 12, 12,
                                    12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 255, 255 };   /* eedi2.c:21-25 as u8 */

typedef struct
{
    uint16_t *alloc;
    uint16_t *plane[3];
    int stride[3], width[3], height[3];
} frame_t;

struct orc_eedi2_16
{
    orc_eedi2_params_t p;
    int width, height, depth;
    frame_t half[4];   /* SRCPF MSKPF TMPPF DSTPF */
    frame_t full[5];   /* DST2PF TMP2PF2 MSK2PF TMP2PF DST2MPF */
    int *cx2, *cy2, *cxy, *tmpc;   /* decomb.c:398-403: height * stride(luma) ints each, shared by the planes */
};

static inline int iabs(int v) { return v < 0 ? -v : v; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

static void frame_alloc(frame_t *f, int width, int height)
{
    size_t total = 0, off[3];
    for (int c = 0; c < 3; c++)
    {
        f->width[c]  = c ? (width + 1) / 2 : width;
        f->height[c] = c ? (height + 1) / 2 : height;
        f->stride[c] = (f->width[c] * 2 + 63) / 64 * 64 / 2;            /* hb_image_stride in bytes / bps */
        off[c] = total;
        total += (size_t)f->stride[c] * f->height[c];
    }
    f->alloc = calloc(total + 2 * GUARD, sizeof(uint16_t));
    for (int c = 0; c < 3; c++)
        f->plane[c] = f->alloc + GUARD + off[c];
}

orc_eedi2_16_t *orc_eedi2_16_new(int width, int height, int depth, const orc_eedi2_params_t *p)
{
    orc_eedi2_16_t *e = calloc(1, sizeof(*e));
    e->p = *p;
    e->depth = depth;
    e->width = width;
    e->height = height;
    for (int i = 0; i < 4; i++) frame_alloc(&e->half[i], width, height / 2);   /* decomb.c:291-296 */
    for (int i = 0; i < 5; i++) frame_alloc(&e->full[i], width, height);       /* :299-303 */
    if (p->post_processing > 1)
    {
        const size_t n = (size_t)height * e->full[0].stride[0];
        e->cx2 = calloc(n, sizeof(int));
        e->cy2 = calloc(n, sizeof(int));
        e->cxy = calloc(n, sizeof(int));
        e->tmpc = calloc(n, sizeof(int));
    }
    return e;
}

void orc_eedi2_16_free(orc_eedi2_16_t *e)
{
    if (!e) return;
    for (int i = 0; i < 4; i++) free(e->half[i].alloc);
    for (int i = 0; i < 5; i++) free(e->full[i].alloc);
    free(e->cx2); free(e->cy2); free(e->cxy); free(e->tmpc);
    free(e);
}

const uint16_t *orc_eedi2_16_plane(orc_eedi2_16_t *e, int buffer, int plane, int *stride, int *height)
{
    frame_t *f = buffer < 4 ? &e->half[buffer] : &e->full[buffer - 4];
    if (stride) *stride = f->stride[plane];
    if (height) *height = f->height[plane];
    return f->plane[plane];
}

/* eedi2_bit_blit (:46-68) for equal pitches */
static void blit(uint16_t *dst, const uint16_t *src, int pitch, int width, int height)
{
    for (int y = 0; y < height; y++)
        memcpy(dst + (size_t)y * pitch, src + (size_t)y * pitch, sizeof(uint16_t) * width);
}

/* insertion sort of <= 9 values + the reference's midpoint rule (eedi2.c:65-80, e.g. :500-502) */
static int sorted_mid(int *v, int n)
{
    for (int i = 1; i < n; i++)
    {
        const int t = v[i];
        int j = i;
        while (j > 0 && v[j - 1] > t) { v[j] = v[j - 1]; j--; }
        v[j] = t;
    }
    return (n & 1) ? v[n >> 1] : (v[(n - 1) >> 1] + v[n >> 1] + 1) >> 1;
}

/* mean of the values within `lim` of mid, mixed with mid and rounded (:701 etc.);
 * returns the count of values used through *count. */
static int vote(const int *v, int n, int mid, int lim, int *count)
{
    int sum = 0, cnt = 0;
    for (int i = 0; i < n; i++)
        if (iabs(v[i] - mid) <= lim) { cnt++; sum += v[i]; }
    *count = cnt;
    return (int)(((float)(sum + mid) / (float)(cnt + 1)) + 0.5f);
}

/* ---- half-height passes ------------------------------------------------------------ */

/* :122-195 */
static void build_edge_mask(uint16_t *dst, const uint16_t *src, int pitch, int width, int height,
                            int magnitude, int variance, int laplacian)
{
    const int mth = magnitude * 10;
    const int vth = laplacian * 81;       /* sic: the value passed third lands in `vthresh` */
    const int lth = variance;             /* and the second in `lthresh`                    */
    const int ten = (uint16_t)(10 << SHIFT);                           /* 16: `const pixel ten` (:127) */
    memset(dst, 0, sizeof(uint16_t) * (size_t)(height / 2) * pitch);
    for (int y = 1; y < height - 1; y++)
    {
        const uint16_t *p = src + (size_t)(y - 1) * pitch, *c = p + pitch, *n = c + pitch;
        uint16_t *o = dst + (size_t)y * pitch;
        for (int x = 1; x < width - 1; x++)
        {
#define FLATCOL(i) (iabs(p[i] - c[i]) < ten && iabs(c[i] - n[i]) < ten && iabs(p[i] - n[i]) < ten)
            if (FLATCOL(x) || (FLATCOL(x - 1) && FLATCOL(x + 1)))
                continue;
#undef FLATCOL
            int sum = 0, sumsq = 0;
            for (int i = -1; i <= 1; i++)
            {
                sum   += p[x + i] + c[x + i] + n[x + i];
                sumsq += (p[x + i] >> SHIFT) * (p[x + i] >> SHIFT) + (c[x + i] >> SHIFT) * (c[x + i] >> SHIFT) +
                         (n[x + i] >> SHIFT) * (n[x + i] >> SHIFT);                  /* 16: squares of the 8-bit part (:158-166) */
            }
            sum >>= SHIFT;                                                            /* 16: (:154-156) */
            if (9 * sumsq - sum * sum < vth)
                continue;
            const int ix = (c[x + 1] - c[x - 1]) >> SHIFT;
            const int iy = imax(imax(iabs(p[x] - n[x]), iabs(p[x] - c[x])), iabs(c[x] - n[x])) >> SHIFT;
            if (ix * ix + iy * iy >= mth)
            {
                o[x] = PEAK;
                continue;
            }
            const int ixx = (c[x - 1] - 2 * c[x] + c[x + 1]) >> SHIFT;
            const int iyy = (p[x] - 2 * c[x] + n[x]) >> SHIFT;
            if (iabs(ixx) + iabs(iyy) >= lth)
                o[x] = PEAK;
        }
    }
}

static int peaks_around(const uint16_t *p, const uint16_t *c, const uint16_t *n, int x)
{
    return (p[x - 1] == PEAK) + (p[x] == PEAK) + (p[x + 1] == PEAK) + (c[x - 1] == PEAK) +
           (c[x + 1] == PEAK) + (n[x - 1] == PEAK) + (n[x] == PEAK) + (n[x + 1] == PEAK);
}

/* grow != 0: dilate (:207-247), else erode (:259-293) */
static void morph_edge_mask(const uint16_t *msk, uint16_t *dst, int pitch, int width, int height, int thr, int grow)
{
    blit(dst, msk, pitch, width, height);
    for (int y = 1; y < height - 1; y++)
    {
        const uint16_t *p = msk + (size_t)(y - 1) * pitch, *c = p + pitch, *n = c + pitch;
        uint16_t *o = dst + (size_t)y * pitch;
        for (int x = 1; x < width - 1; x++)
        {
            if (grow)
            {
                if (c[x] != 0) continue;
                if (peaks_around(p, c, n, x) >= thr) o[x] = PEAK;
            }
            else
            {
                if (c[x] != PEAK) continue;
                if (peaks_around(p, c, n, x) < thr) o[x] = 0;
            }
        }
    }
}

/* :308-342 */
static void remove_small_gaps(const uint16_t *msk, uint16_t *dst, int pitch, int width, int height)
{
    blit(dst, msk, pitch, width, height);
    for (int y = 1; y < height - 1; y++)
    {
        const uint16_t *m = msk + (size_t)y * pitch;
        uint16_t *o = dst + (size_t)y * pitch;
        for (int x = 3; x < width - 3; x++)
        {
            if (m[x])
            {
                if (m[x - 3] || m[x - 2] || m[x - 1] || m[x + 1] || m[x + 2] || m[x + 3]) continue;
                o[x] = 0;
            }
            else if ((m[x + 1] && (m[x - 1] || m[x - 2] || m[x - 3])) ||
                     (m[x + 2] && (m[x - 1] || m[x - 2])) || (m[x + 3] && m[x - 1]))
                o[x] = PEAK;
        }
    }
}

static inline int sad3(const uint16_t *a, int ai, const uint16_t *b, int bi)
{
    return iabs(a[ai - 1] - b[bi - 1]) + iabs(a[ai] - b[bi]) + iabs(a[ai + 1] - b[bi + 1]);
}

/* :358-525 */
static void calc_directions(int plane, const uint16_t *msk, const uint16_t *src, uint16_t *dst, int pitch,
                            int width, int height, int maxd, int nt)
{
    const int nt13 = (uint16_t)((nt << SHIFT) * 13), nt19 = (uint16_t)((nt << SHIFT) * 19);   /* 16: `pixel` typed, wrap at 16 bits (:364-365) */
    const int maxdt = plane == 0 ? maxd : (maxd >> 1);
    for (size_t i = 0; i < (size_t)pitch * height; i++) dst[i] = (uint16_t)PEAK;             /* (:371-377) */
    for (int y = 1; y < height - 1; y++)
    {
        const uint16_t *mp = msk + (size_t)(y - 1) * pitch, *mc = mp + pitch, *mn = mc + pitch;
        const uint16_t *s2p = src + (ptrdiff_t)(y - 2) * pitch, *sp = s2p + pitch, *sc = sp + pitch,
                      *sn = sc + pitch, *s2n = sn + pitch;
        uint16_t *o = dst + (size_t)y * pitch;
        for (int x = 1; x < width - 1; x++)
        {
            if (mc[x] != PEAK || (mc[x - 1] != PEAK && mc[x + 1] != PEAK))
                continue;
            const int startu = imax(-x + 1, -maxdt), stopu = imin(width - 2 - x, maxdt);
            const int vert = iabs(sc[x] - sn[x]) + iabs(sc[x] - sp[x]);
            int minb = imin(nt13, vert * 6), mina = imin(nt19, vert * 9);
            int minc = mina, mind = minb, mine = minb;
            int dira = -5000, dirb = -5000, dirc = -5000, dird = -5000, dire = -5000;
            for (int u = startu; u <= stopu; u++)
            {
                if (!(y == 1 || mp[x - 1 + u] == PEAK || mp[x + u] == PEAK || mp[x + 1 + u] == PEAK))
                    continue;
                if (!(y == height - 2 || mn[x - 1 - u] == PEAK || mn[x - u] == PEAK || mn[x + 1 - u] == PEAK))
                    continue;
                const int diffsn = sad3(sc, x, sn, x - u);
                const int diffsp = sad3(sc, x, sp, x + u);
                const int diffps = sad3(sp, x, sc, x - u);
                const int diffns = sad3(sn, x, sc, x + u);
                const int diff = diffsn + diffsp + diffps + diffns;
                int diffd = diffsp + diffns, diffe = diffsn + diffps;
                if (diff < minb) { dirb = u; minb = diff; }
                if (y > 1)
                {
                    const int diff2pp = sad3(s2p, x, sp, x - u);
                    const int diffp2p = sad3(sp, x, s2p, x + u);
                    const int diffa = diff + diff2pp + diffp2p;
                    diffd += diffp2p;
                    diffe += diff2pp;
                    if (diffa < mina) { dira = u; mina = diffa; }
                }
                if (y < height - 2)
                {
                    const int diff2nn = sad3(s2n, x, sn, x + u);
                    const int diffn2n = sad3(sn, x, s2n, x - u);
                    const int diffc = diff + diff2nn + diffn2n;
                    diffd += diff2nn;
                    diffe += diffn2n;
                    if (diffc < minc) { dirc = u; minc = diffc; }
                }
                if (diffd < mind) { dird = u; mind = diffd; }
                if (diffe < mine) { dire = u; mine = diffe; }
            }
            int order[5], k = 0;
            if (dira != -5000) order[k++] = dira;
            if (dirb != -5000) order[k++] = dirb;
            if (dirc != -5000) order[k++] = dirc;
            if (dird != -5000) order[k++] = dird;
            if (dire != -5000) order[k++] = dire;
            if (k <= 1)
            {
                o[x] = NEUTRAL;
                continue;
            }
            const int mid = sorted_mid(order, k);
            const int tlim = imax(LIMLUT[iabs(mid)] >> 2, 2);                 /* 16: the SCALED limlut, still >> 2 (:500) */
            int sum = 0, count = 0;
            for (int i = 0; i < k; i++)
                if (iabs(order[i] - mid) <= tlim) { count++; sum += order[i]; }
            if (count > 1)
                o[x] = (uint16_t)(NEUTRAL + ((int)((float)sum / (float)count) << (2 + SHIFT)));   /* shift2 (:509) */
            else
                o[x] = NEUTRAL;
        }
    }
}

static int collect3(int *v, int k, const uint16_t *row, int x, int skip_centre)
{
    if (row[x - 1] != PEAK) v[k++] = row[x - 1];
    if (!skip_centre && row[x] != PEAK) v[k++] = row[x];
    if (row[x + 1] != PEAK) v[k++] = row[x + 1];
    return k;
}

/* filter_dir_map (:649-709) when expand == 0, expand_dir_map (:722-773) when expand != 0 */
static void dir_map_pass(const uint16_t *msk, const uint16_t *dmsk, uint16_t *dst, int pitch,
                         int width, int height, int expand)
{
    blit(dst, dmsk, pitch, width, height);
    for (int y = 1; y < height - 1; y++)
    {
        const uint16_t *dp = dmsk + (size_t)(y - 1) * pitch, *dc = dp + pitch, *dn = dc + pitch;
        const uint16_t *m = msk + (size_t)y * pitch;
        uint16_t *o = dst + (size_t)y * pitch;
        for (int x = 1; x < width - 1; x++)
        {
            if (m[x] != PEAK) continue;
            if (expand && dc[x] != PEAK) continue;
            int order[9], u = 0;
            u = collect3(order, u, dp, x, 0);
            u = collect3(order, u, dc, x, expand);
            u = collect3(order, u, dn, x, 0);
            if (u < (expand ? 5 : 4))
            {
                if (!expand) o[x] = PEAK;
                continue;
            }
            const int mid = sorted_mid(order, u);
            int count;
            const int val = vote(order, u, mid, LIMLUT[iabs(mid - NEUTRAL) >> (2 + SHIFT)], &count);
            if (expand)
            {
                if (count < 5) continue;
            }
            else if (count < 4 || (count < 5 && dc[x] == PEAK))
            {
                o[x] = PEAK;
                continue;
            }
            o[x] = (uint16_t)val;
        }
    }
}

/* does the walk j = from..to along `row`/`other` trip one of the three tests (:565-575)? */
static int trips(const uint16_t *side, const uint16_t *dc, int x, int from, int to, int lim, int side_is_next)
{
    for (int j = from; j <= to; j++)
    {
        const int s = side[x + j], c = dc[x + j], ref = dc[x];
        if ((iabs(s - ref) > lim && s != PEAK) ||
            (side_is_next ? (s == PEAK && c == PEAK) : (c == PEAK && s == PEAK)) ||
            (iabs(c - ref) > lim && c != PEAK))
            return 1;
    }
    return 0;
}

/* :538-635 */
static void filter_map(const uint16_t *msk, const uint16_t *dmsk, uint16_t *dst, int pitch, int width, int height)
{
    blit(dst, dmsk, pitch, width, height);
    for (int y = 1; y < height - 1; y++)
    {
        const uint16_t *dp = dmsk + (size_t)(y - 1) * pitch, *dc = dp + pitch, *dn = dc + pitch;
        const uint16_t *m = msk + (size_t)y * pitch;
        uint16_t *o = dst + (size_t)y * pitch;
        for (int x = 1; x < width - 1; x++)
        {
            if (dc[x] == PEAK || m[x] != PEAK) continue;
            int dir = (dc[x] - NEUTRAL) >> 2;                                    /* 16: still >> 2 here (:563) */
            const int lim = imax(iabs(dir) * 2, 12 << (2 + SHIFT));              /* twelve (:544) */
            dir >>= 2 + SHIFT;                                                   /* (:565) */
            int ict, icb = 0;
            if (dir < 0) ict = trips(dp, dc, x, imax(-x, dir), 0, lim, 0);
            else         ict = trips(dp, dc, x, 0, imin(width - x - 1, dir), lim, 0);
            if (!ict) continue;
            if (dir < 0) icb = trips(dn, dc, x, 0, imin(width - x - 1, iabs(dir)), lim, 1);
            else         icb = trips(dn, dc, x, imax(-x, -dir), 0, lim, 1);
            if (icb) o[x] = PEAK;
        }
    }
}

/* :98-108 */
static void upscale_by_2(const uint16_t *src, uint16_t *dst, int height, int pitch)
{
    for (int y = 0; y < height; y++)
    {
        memcpy(dst + (size_t)(2 * y) * pitch, src + (size_t)y * pitch, sizeof(uint16_t) * pitch);
        memcpy(dst + (size_t)(2 * y + 1) * pitch, src + (size_t)y * pitch, sizeof(uint16_t) * pitch);
    }
}

/* ---- full-height passes ------------------------------------------------------------ */

/* :787-858 */
static void mark_directions_2x(const uint16_t *msk, const uint16_t *dmsk, uint16_t *dst, int pitch,
                               int tff, int width, int height)
{
    for (size_t i = 0; i < (size_t)pitch * height; i++) dst[i] = (uint16_t)PEAK;             /* (:794-804) */
    for (int y = 2 - tff; y < height - 1; y += 2)
    {
        const uint16_t *d0 = dmsk + (size_t)(y - 1) * pitch, *d1 = d0 + 2 * (size_t)pitch;
        const uint16_t *m0 = msk + (size_t)(y - 1) * pitch, *m1 = m0 + 2 * (size_t)pitch;
        uint16_t *o = dst + (size_t)y * pitch;
        for (int x = 1; x < width - 1; x++)
        {
            if (m0[x] != PEAK && m1[x] != PEAK) continue;
            int order[6], v = 0;
            v = collect3(order, v, d0, x, 0);
            v = collect3(order, v, d1, x, 0);
            if (v < 3) continue;
            const int mid = sorted_mid(order, v);
            const int lim = LIMLUT[iabs(mid - NEUTRAL) >> (2 + SHIFT)];
            int u = 0;
            if (iabs(d0[x - 1] - d1[x - 1]) <= lim || d0[x - 1] == PEAK || d1[x - 1] == PEAK) u++;
            if (iabs(d0[x] - d1[x]) <= lim || d0[x] == PEAK || d1[x] == PEAK) u++;
            if (iabs(d0[x + 1] - d1[x - 1]) <= lim || d0[x + 1] == PEAK || d1[x + 1] == PEAK) u++;   /* sic */
            if (u < 2) continue;
            int count;
            const int val = vote(order, v, mid, lim, &count);
            if (count < v - 2 || count < 2) continue;
            o[x] = (uint16_t)val;
        }
    }
}

/* filter_dir_map_2x (:872-939) / expand_dir_map_2x (:953-1011) */
static void dir_map_pass_2x(const uint16_t *msk, const uint16_t *dmsk, uint16_t *dst, int pitch,
                            int field, int width, int height, int expand)
{
    blit(dst, dmsk, pitch, width, height);
    for (int y = 2 - field; y < height - 1; y += 2)
    {
        const uint16_t *dc = dmsk + (size_t)y * pitch;
        const uint16_t *dp = dc - 2 * (ptrdiff_t)pitch, *dn = dc + 2 * (ptrdiff_t)pitch;
        const uint16_t *m0 = msk + (size_t)(y - 1) * pitch, *m1 = m0 + 2 * (size_t)pitch;
        uint16_t *o = dst + (size_t)y * pitch;
        for (int x = 1; x < width - 1; x++)
        {
            if (m0[x] != PEAK && m1[x] != PEAK) continue;
            if (expand && dc[x] != PEAK) continue;
            int order[9], u = 0;
            if (y > 1) u = collect3(order, u, dp, x, 0);
            u = collect3(order, u, dc, x, expand);
            if (y < height - 2) u = collect3(order, u, dn, x, 0);
            if (u < (expand ? 5 : 4))
            {
                if (!expand) o[x] = PEAK;
                continue;
            }
            const int mid = sorted_mid(order, u);
            int count;
            const int val = vote(order, u, mid, LIMLUT[iabs(mid - NEUTRAL) >> (2 + SHIFT)], &count);
            if (expand)
            {
                if (count < 5) continue;
            }
            else if (count < 4 || (count < 5 && dc[x] == PEAK))
            {
                o[x] = PEAK;
                continue;
            }
            o[x] = (uint16_t)val;
        }
    }
}

/* :1025-1132 */
static void fill_gaps_2x(const uint16_t *msk, const uint16_t *dmsk, uint16_t *dst, int pitch,
                         int field, int width, int height)
{
    blit(dst, dmsk, pitch, width, height);
    for (int y = 2 - field; y < height - 1; y += 2)
    {
        const uint16_t *dc = dmsk + (size_t)y * pitch;
        const uint16_t *dp = dc - 2 * (ptrdiff_t)pitch, *dn = dc + 2 * (ptrdiff_t)pitch;
        const uint16_t *mc = msk + (size_t)(y - 1) * pitch;
        const uint16_t *mp = mc - 2 * (ptrdiff_t)pitch, *mn = mc + 2 * (ptrdiff_t)pitch, *mnn = mn + 2 * (ptrdiff_t)pitch;
        uint16_t *o = dst + (size_t)y * pitch;
        for (int x = 1; x < width - 1; x++)
        {
            if (dc[x] != PEAK || (mc[x] != PEAK && mn[x] != PEAK)) continue;
            const int eight = 8 << SHIFT, twenty = 20 << SHIFT, five_hundred = 500 << SHIFT;     /* (:1032-1034) */
            int u = x - 1, back = five_hundred, forward = -five_hundred;
            while (u)
            {
                if (dc[u] != PEAK) { back = dc[u]; break; }
                if (mc[u] != PEAK && mn[u] != PEAK) break;
                u--;
            }
            int v = x + 1;
            while (v < width)
            {
                if (dc[v] != PEAK) { forward = dc[v]; break; }
                if (mc[v] != PEAK && mn[v] != PEAK) break;
                v++;
            }
            int tc = 1, bc = 1, mint = five_hundred, maxt = -twenty, minb = five_hundred, maxb = -twenty;
            for (int j = u; j <= v; j++)
            {
                if (tc)
                {
                    if (y <= 2 || dp[j] == PEAK || (mp[j] != PEAK && mc[j] != PEAK)) { tc = 0; mint = maxt = twenty; }
                    else { if (dp[j] < mint) mint = dp[j]; if (dp[j] > maxt) maxt = dp[j]; }
                }
                if (bc)
                {
                    if (y >= height - 3 || dn[j] == PEAK || (mn[j] != PEAK && mnn[j] != PEAK)) { bc = 0; minb = maxb = twenty; }
                    else { if (dn[j] < minb) minb = dn[j]; if (dn[j] > maxb) maxb = dn[j]; }
                }
            }
            if (maxt == -twenty) maxt = mint = twenty;
            if (maxb == -twenty) maxb = minb = twenty;
            const int far = imax(iabs(forward - NEUTRAL), iabs(back - NEUTRAL));
            const int thresh = imax(imax(far >> 2, eight), imax(iabs(mint - maxt), iabs(minb - maxb)));   /* 16: >> 2 (:1109) */
            const int flim = imin(far >> (2 + SHIFT), 6);                                                  /* 16: >> shift2 (:1112) */
            if (iabs(forward - back) <= thresh && (v - u - 1 <= flim || tc || bc))
            {
                const double step = (double)(forward - back) / (double)(v - u);
                for (int j = 0; j < v - u - 1; j++)
                    o[u + j + 1] = (uint16_t)(back + (int)(j * step + 0.5));
            }
        }
    }
}

/* :1148-1335 — in place: dmsk row y and dst row y are rewritten left to right and the
 * test at x looks at the already rewritten dmsk[x-1]. */
static void interpolate_lattice(int plane, uint16_t *dmsk, uint16_t *dst, const uint16_t *omsk, int pitch,
                                int field, int nt, int width, int height)
{
    const int nt4 = (uint16_t)((nt << SHIFT) * 4), nt7 = (uint16_t)((nt << SHIFT) * 7), nt8 = (uint16_t)((nt << SHIFT) * 8);   /* `pixel` typed (:1158-1160) */
    const int three = 3 << SHIFT, nine = 9 << SHIFT;                                                                              /* (:1156-1157) */
    if (field == 1) memcpy(dst + (size_t)(height - 1) * pitch, dst + (size_t)(height - 2) * pitch, sizeof(uint16_t) * width);
    else            memcpy(dst, dst + pitch, sizeof(uint16_t) * width);
    for (int y = 2 - field; y < height - 1; y += 2)
    {
        uint16_t *top = dst + (size_t)(y - 1) * pitch, *mid = top + pitch, *bot = mid + pitch;
        const uint16_t *ot = omsk + (size_t)(y - 1) * pitch, *ob = ot + 2 * (size_t)pitch;
        uint16_t *dm = dmsk + (size_t)y * pitch;
        for (int x = 0; x < width; x++)
        {
            int dir = dm[x];
            const int lim = LIMLUT[iabs(dir - NEUTRAL) >> (2 + SHIFT)];
            const int avg = (top[x] + bot[x] + 1) >> 1;
            if (dir == PEAK || (iabs(dm[x] - dm[x - 1]) > lim && iabs(dm[x] - dm[x + 1]) > lim))
            {
                mid[x] = (uint16_t)avg;
                if (dir != PEAK) dm[x] = NEUTRAL;
                continue;
            }
            if (lim < nine)
            {
#define SQ8(v) (((v) >> SHIFT) * ((v) >> SHIFT))
                const int sum = (top[x - 1] + top[x] + top[x + 1] + bot[x - 1] + bot[x] + bot[x + 1]) >> SHIFT;
                const int sumsq = SQ8(top[x - 1]) + SQ8(top[x]) + SQ8(top[x + 1]) + SQ8(bot[x - 1]) + SQ8(bot[x]) + SQ8(bot[x + 1]);
#undef SQ8
                if (6 * sumsq - sum * sum < 576)
                {
                    mid[x] = (uint16_t)avg;
                    dm[x] = PEAK;
                    continue;
                }
            }
            if (x > 1 && x < width - 2 &&
                ((top[x] < imax(top[x - 2], top[x - 1]) - three && top[x] < imax(top[x + 2], top[x + 1]) - three &&
                  bot[x] < imax(bot[x - 2], bot[x - 1]) - three && bot[x] < imax(bot[x + 2], bot[x + 1]) - three) ||
                 (top[x] > imin(top[x - 2], top[x - 1]) + three && top[x] > imin(top[x + 2], top[x + 1]) + three &&
                  bot[x] > imin(bot[x - 2], bot[x - 1]) + three && bot[x] > imin(bot[x + 2], bot[x + 1]) + three)))
            {
                mid[x] = (uint16_t)avg;
                dm[x] = NEUTRAL;
                continue;
            }
            dir = (dir - NEUTRAL + (1 << (1 + SHIFT))) >> (2 + SHIFT);           /* (:1233) */
            int val = avg;
            const int startu = (dir - 2 < 0) ? imax(-x + 1, imax(dir - 2, -width + 2 + x))
                                             : imin(x - 1, imin(dir - 2, width - 2 - x));
            const int stopu = (dir + 2 < 0) ? imax(-x + 1, imax(dir + 2, -width + 2 + x))
                                            : imin(x - 1, imin(dir + 2, width - 2 - x));
            int min = nt8;
            const int here = dm[x];
#define NEAR(row, i) ((row)[i] != PEAK && iabs((row)[i] - here) <= lim)
            for (int u = startu; u <= stopu; u++)
            {
                const int diff = sad3(top, x, bot, x - u) + sad3(bot, x, top, x + u);
                if (!(diff < min && (NEAR(ot, x - 1 + u) || NEAR(ot, x + u) || NEAR(ot, x + 1 + u)) &&
                      (NEAR(ob, x - 1 - u) || NEAR(ob, x - u) || NEAR(ob, x + 1 - u))))
                    continue;
                const int h0 = u >> 1, h1 = (u + 1) >> 1;
                const int diff2 = sad3(top, x + h0, bot, x - h0);
                if (!(diff2 < nt4 &&
                      (((iabs(ot[x + h0] - ob[x - h0]) <= lim || iabs(ot[x + h0] - ob[x - h1]) <= lim) && ot[x + h0] != PEAK) ||
                       ((iabs(ot[x + h1] - ob[x - h0]) <= lim || iabs(ot[x + h1] - ob[x - h1]) <= lim) && ot[x + h1] != PEAK))))
                    continue;
                if ((iabs(here - ot[x + h0]) <= lim || iabs(here - ot[x + h1]) <= lim) &&
                    (iabs(here - ob[x - h0]) <= lim || iabs(here - ob[x - h1]) <= lim))
                {
                    val = (top[x + h0] + top[x + h1] + bot[x - h0] + bot[x - h1] + 2) >> 2;
                    min = diff;
                    dir = u;
                }
            }
#undef NEAR
            if (min != nt8)
            {
                mid[x] = (uint16_t)val;
                dm[x] = (uint16_t)(NEUTRAL + (dir << (2 + SHIFT)));
                continue;
            }
            const int lo = imin(top[x], bot[x]), hi = imax(top[x], bot[x]);
            const int d = plane == 0 ? 4 : 2;
            const int su = imax(-x + 1, -d), eu = imin(width - 2 - x, d);
            min = nt7;
            for (int u = su; u <= eu; u++)
            {
                const int h0 = u >> 1, h1 = (u + 1) >> 1;
                const int p1 = top[x + h0] + top[x + h1];
                const int p2 = bot[x - h0] + bot[x - h1];
                const int diff = sad3(top, x, bot, x - u) + sad3(bot, x, top, x + u) + iabs(p1 - p2);
                if (diff < min)
                {
                    const int valt = (p1 + p2 + 2) >> 2;
                    if (valt >= lo && valt <= hi) { val = valt; min = diff; dir = u; }
                }
            }
            mid[x] = (uint16_t)val;
            dm[x] = (min == 7 * nt) ? NEUTRAL : (uint16_t)(NEUTRAL + (dir << (2 + SHIFT)));   /* compares with the unshifted 7*nt (:1324) */
        }
    }
}

/* :1349-1378 */
static void post_process(const uint16_t *nmsk, const uint16_t *omsk, uint16_t *dst, int pitch,
                         int field, int width, int height)
{
    for (int y = 2 - field; y < height - 1; y += 2)
    {
        const uint16_t *nm = nmsk + (size_t)y * pitch, *om = omsk + (size_t)y * pitch;
        uint16_t *d = dst + (size_t)y * pitch;
        for (int x = 0; x < width; x++)
        {
            const int lim = LIMLUT[iabs(nm[x] - NEUTRAL) >> (2 + SHIFT)];
            if (iabs(nm[x] - om[x]) > lim && om[x] != PEAK && om[x] != NEUTRAL)
                d[x] = (uint16_t)((d[x - pitch] + d[x + pitch] + 1) >> 1);
        }
    }
}

/* ---- post-processing 2/3: junctions and corners (eedi2_template.c:1391-1904) ------------ */

/* Both blurs are symmetric FIR filters whose taps, where they would fall outside the row /
 * column, are replaced by their mirror image about the centre (the reference writes this as
 * doubled coefficients on the surviving side: 582 = 2*291 ... :1399-1424, :1549-1594). */
static inline int fold(int centre, int d, int n, int *partner)
{
    int lo = centre - d, hi = centre + d;
    if (lo < 0) lo = hi;
    if (hi >= n) hi = lo;
    *partner = hi;
    return lo;
}

/* eedi2_gaussian_blur1 (:1391-1527): 7 taps, src -> tmp horizontally, tmp -> dst vertically */
static void gaussian_blur1(const uint16_t *src, uint16_t *tmp, uint16_t *dst, int pitch, int width, int height)
{
    static const int W[4] = { 26152, 15862, 3539, 291 };
    for (int y = 0; y < height; y++)
    {
        const uint16_t *s = src + (size_t)y * pitch;
        uint16_t *t = tmp + (size_t)y * pitch;
        for (int x = 0; x < width; x++)
        {
            int acc = s[x] * W[0] + 32768;
            for (int d = 1; d <= 3; d++)
            {
                int hi, lo = fold(x, d, width, &hi);
                acc += (s[lo] + s[hi]) * W[d];
            }
            t[x] = (uint16_t)(acc >> 16);
        }
    }
    for (int y = 0; y < height; y++)
    {
        uint16_t *o = dst + (size_t)y * pitch;
        for (int x = 0; x < width; x++)
        {
            int acc = tmp[(size_t)y * pitch + x] * W[0] + 32768;
            for (int d = 1; d <= 3; d++)
            {
                int hi, lo = fold(y, d, height, &hi);
                acc += (tmp[(size_t)lo * pitch + x] + tmp[(size_t)hi * pitch + x]) * W[d];
            }
            o[x] = (uint16_t)(acc >> 16);
        }
    }
}

/* eedi2_calc_derivatives (:1760-1848): central differences with clamped neighbours
 * (left/right difference one-sided at the row ends, up/down one-sided at the first/last row) */
static void calc_derivatives(const uint16_t *src, int pitch, int width, int height, int *x2, int *y2, int *xy)
{
    for (int y = 0; y < height; y++)
    {
        const uint16_t *s = src + (size_t)y * pitch;
        const uint16_t *up = src + (size_t)imax(y - 1, 0) * pitch;
        const uint16_t *dn = src + (size_t)imin(y + 1, height - 1) * pitch;
        for (int x = 0; x < width; x++)
        {
            const int ix = (s[imin(x + 1, width - 1)] - s[imax(x - 1, 0)]) >> SHIFT;     /* 16: (:1768-1769 ...) */
            const int iy = (up[x] - dn[x]) >> SHIFT;
            x2[(size_t)y * pitch + x] = (ix * ix) >> 1;
            y2[(size_t)y * pitch + x] = (iy * iy) >> 1;
            xy[(size_t)y * pitch + x] = (ix * iy) >> 1;
        }
    }
}

/* eedi2_gaussian_blur_sqrt2 (:1539-1748): 9 taps on an int array, >>16 then >>18 */
static void gaussian_blur_sqrt2(const int *src, int *tmp, int *dst, int pitch, int width, int height)
{
    static const int W[5] = { 18508, 14415, 6809, 1951, 339 };
    for (int y = 0; y < height; y++)
    {
        const int *s = src + (size_t)y * pitch;
        int *t = tmp + (size_t)y * pitch;
        for (int x = 0; x < width; x++)
        {
            int acc = s[x] * W[0] + 32768;
            for (int d = 1; d <= 4; d++)
            {
                int hi, lo = fold(x, d, width, &hi);
                if (d == 3 && x == width - 2) lo = hi = x + 3;       /* :1589 reads x+3, past the row */
                acc += (s[lo] + s[hi]) * W[d];
            }
            t[x] = acc >> 16;
        }
    }
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
        {
            int acc = tmp[(size_t)y * pitch + x] * W[0] + 32768;
            for (int d = 1; d <= 4; d++)
            {
                int hi, lo = fold(y, d, height, &hi);
                acc += (tmp[(size_t)lo * pitch + x] + tmp[(size_t)hi * pitch + x]) * W[d];
            }
            dst[(size_t)y * pitch + x] = acc >> 18;
        }
}

/* eedi2_post_process_corner (:1864-1904): Harris-style response on the blurred derivative
 * products of the two field rows around an interpolated row; > 775 -> vertical average */
static void post_process_corner(const int *x2, const int *y2, const int *xy, int pitch, const uint16_t *msk,
                                uint16_t *dst, int field, int width, int height)
{
    int drow = 3;
    for (int y = 8 - field; y < height - 7; y += 2, drow++)
    {
        const uint16_t *m = msk + (size_t)y * pitch;
        uint16_t *d = dst + (size_t)y * pitch;
        for (int x = 4; x < width - 4; x++)
        {
            if (m[x] == PEAK || m[x] == NEUTRAL) continue;
            int hit = 0;
            for (int k = 0; k < 2; k++)
            {
                const size_t i = (size_t)(drow + k) * pitch + x;
                const int a = x2[i], b = y2[i], c = xy[i];
                const int r = (int)(a * b - c * c - 0.09 * (a + b) * (a + b));
                hit |= r > 775;
            }
            if (hit) d[x] = (uint16_t)((d[x - pitch] + d[x + pitch] + 1) >> 1);
        }
    }
}

/* ---- sequencing (decomb_template.c:366-473) ------------------------------------------ */
static void run_plane(orc_eedi2_16_t *e, int c, int tff, int npasses)
{
    uint16_t *srcp = e->half[0].plane[c], *mskp = e->half[1].plane[c], *tmpp = e->half[2].plane[c], *dstp = e->half[3].plane[c];
    uint16_t *dst2p = e->full[0].plane[c], *tmp2p2 = e->full[1].plane[c], *msk2p = e->full[2].plane[c],
            *tmp2p = e->full[3].plane[c], *dst2mp = e->full[4].plane[c];
    const int pitch = e->full[0].stride[c], height = e->full[0].height[c], width = e->full[0].width[c];
    const int hh = e->half[0].height[c];
    const orc_eedi2_params_t *p = &e->p;
    int n = 0;
#define STEP(call) do { if (n++ >= npasses) return; call; } while (0)
    STEP(build_edge_mask(mskp, srcp, pitch, width, hh, p->magnitude_threshold, p->variance_threshold, p->laplacian_threshold));
    STEP(morph_edge_mask(mskp, tmpp, pitch, width, hh, p->erosion_threshold, 0));
    STEP(morph_edge_mask(tmpp, mskp, pitch, width, hh, p->dilation_threshold, 1));
    STEP(morph_edge_mask(mskp, tmpp, pitch, width, hh, p->erosion_threshold, 0));
    STEP(remove_small_gaps(tmpp, mskp, pitch, width, hh));
    STEP(calc_directions(c, mskp, srcp, tmpp, pitch, width, hh, p->maximum_search_distance, p->noise_threshold));
    STEP(dir_map_pass(mskp, tmpp, dstp, pitch, width, hh, 0));
    STEP(dir_map_pass(mskp, dstp, tmpp, pitch, width, hh, 1));
    STEP(filter_map(mskp, tmpp, dstp, pitch, width, hh));
    STEP(upscale_by_2(srcp, dst2p, hh, pitch));
    STEP(upscale_by_2(dstp, tmp2p2, hh, pitch));
    STEP(upscale_by_2(mskp, msk2p, hh, pitch));
    STEP(mark_directions_2x(msk2p, tmp2p2, tmp2p, pitch, tff, width, height));
    STEP(dir_map_pass_2x(msk2p, tmp2p, dst2mp, pitch, tff, width, height, 0));
    STEP(dir_map_pass_2x(msk2p, dst2mp, tmp2p, pitch, tff, width, height, 1));
    STEP(fill_gaps_2x(msk2p, tmp2p, dst2mp, pitch, tff, width, height));
    STEP(fill_gaps_2x(msk2p, dst2mp, tmp2p, pitch, tff, width, height));
    STEP(interpolate_lattice(c, tmp2p, dst2p, tmp2p2, pitch, tff, p->noise_threshold, width, height));
    if (p->post_processing == 1 || p->post_processing == 3)
    {
        STEP(blit(tmp2p2, tmp2p, pitch, width, height));
        STEP(dir_map_pass_2x(msk2p, tmp2p, dst2mp, pitch, tff, width, height, 0));
        STEP(dir_map_pass_2x(msk2p, dst2mp, tmp2p, pitch, tff, width, height, 1));
        STEP(post_process(tmp2p, tmp2p2, dst2p, pitch, tff, width, height));
    }
    if (p->post_processing == 2 || p->post_processing == 3)
    {
        STEP(gaussian_blur1(srcp, tmpp, srcp, pitch, width, hh));
        STEP(calc_derivatives(srcp, pitch, width, hh, e->cx2, e->cy2, e->cxy));
        STEP(gaussian_blur_sqrt2(e->cx2, e->tmpc, e->cx2, pitch, width, hh));
        STEP(gaussian_blur_sqrt2(e->cy2, e->tmpc, e->cy2, pitch, width, hh));
        STEP(gaussian_blur_sqrt2(e->cxy, e->tmpc, e->cxy, pitch, width, hh));
        STEP(post_process_corner(e->cx2, e->cy2, e->cxy, pitch, tmp2p2, dst2p, tff, width, height));
    }
#undef STEP
}

void orc_eedi2_16_run_partial(orc_eedi2_16_t *e, const uint16_t *const cur[3], const int stride[3], int tff, int npasses)
{
    SHIFT = e->depth - 8;
    PEAK = (1 << e->depth) - 1;
    NEUTRAL = 1 << (e->depth - 1);
    for (int i = 0; i < 33; i++)
        LIMLUT[i] = (uint16_t)((uint16_t)(i < 31 ? LIMLUT8[i] : 0xffff) << SHIFT);      /* (pixel)(-1) << shift, stored as pixel (:30) */
    /* eedi2_fill_half_height_buffer_plane (:77-89): kept-field rows, min(pitch) SAMPLES each; `stride` in samples */
    for (int c = 0; c < 3; c++)
    {
        const int dst_pitch = e->half[0].stride[c];
        const int n = imin(stride[c], dst_pitch);
        const uint16_t *s = cur[c] + (size_t)stride[c] * (!tff);
        uint16_t *d = e->half[0].plane[c];
        for (int y = e->full[0].height[c]; y > 0; y -= 2)
        {
            memcpy(d, s, sizeof(uint16_t) * n);
            d += dst_pitch;
            s += 2 * (size_t)stride[c];
        }
    }
    for (int c = 0; c < 3; c++)
        run_plane(e, c, tff, npasses);
}

void orc_eedi2_16_run(orc_eedi2_16_t *e, const uint16_t *const cur[3], const int stride[3], int tff)
{
    orc_eedi2_16_run_partial(e, cur, stride, tff, 1000);
}
