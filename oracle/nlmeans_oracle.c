/* nlmeans_oracle.c — CPU restatement of libhb's NLMeans (8-bit path).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows /root/reference/libhb/nlmeans.c and templates/nlmeans_template.c;
 * each function cites the lines it restates.  The patch SSD is formed with
 * separable running sums instead of the reference's integral image: both are
 * exact integer box sums of the same squared differences, so the per-pixel
 * `diff` is the same integer (the reference's uint32 integral may wrap, but
 * only 4-corner differences of it are used, nlmeans_template.c:682).
 * Everything after `diff` replays the reference's float/double operations in
 * the reference's order.  Compile with -ffp-contract=off.
 */
#include "oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define EXPSIZE 128

/* nlmeans.c:345-358 */
void orc_nlmeans_tables(double strength, int patch_size,
                        float exptable[128], float *weight_fact_table, int *diff_max)
{
    const float weight_factor       = 1.0 / patch_size / patch_size / (strength * strength);
    const float min_weight_in_table = 0.0005;
    const float stretch             = EXPSIZE / (-log(min_weight_in_table));
    *weight_fact_table = weight_factor * stretch;
    *diff_max          = EXPSIZE / *weight_fact_table;
    for (int i = 0; i < EXPSIZE; i++)
        exptable[i] = exp(-i / stretch);
    exptable[EXPSIZE - 1] = 0;
}

/* nlmeans.c:529 */
int orc_nlmeans_border(int patch_size)
{
    return ((patch_size + 2) / 2 + 15) / 16 * 16;
}

/* nlmeans_template.c:20-43: left/right columns mirror about the edge pixel
 * (edge pixel repeated), then whole bordered rows are mirrored up and down. */
static void mirror_borders(uint8_t *mem, int w, int h, int border)
{
    const int bw = w + 2 * border;
    uint8_t *img = mem + border + (size_t)bw * border;
    for (int y = 0; y < h; y++)
    {
        uint8_t *row = img + (size_t)y * bw;
        for (int i = 0; i < border; i++)
        {
            row[-1 - i] = row[i];
            row[w + i]  = row[w - 1 - i];
        }
    }
    for (int i = 0; i < border; i++)
    {
        memcpy(img - border - (size_t)(i + 1) * bw, img - border + (size_t)i * bw, bw);
        memcpy(img - border + (size_t)(h + i) * bw, img - border + (size_t)(h - 1 - i) * bw, bw);
    }
}

/* nlmeans_template.c:69-101 */
void orc_nlmeans_make_bordered(const uint8_t *src, int w, int h, int src_stride,
                               int border, uint8_t *dst)
{
    const int bw = w + 2 * border;
    uint8_t *img = dst + border + (size_t)bw * border;
    for (int y = 0; y < h; y++)
        memcpy(img + (size_t)y * bw, src + (size_t)y * src_stride, w);
    mirror_borders(dst, w, h, border);
}

/* ---- prefilters (nlmeans_template.c:103-543) -------------------------------- */

#define PF_MEAN3    1
#define PF_MEAN5    2
#define PF_MEDIAN3  4
#define PF_MEDIAN5  8
#define PF_CSM3     16
#define PF_CSM5     32
#define PF_REDUCE25 256
#define PF_REDUCE50 512
#define PF_EDGEBOOST 1024

/* nlmeans_template.c:103-133: uint16 window sum, scaled by a double, truncated */
static void pf_mean(const uint8_t *src, uint8_t *dst, int w, int h, int bw, int size)
{
    const int lo = -((size - 1) / 2), hi = (size + 1) / 2;
    const double scale = 1.0 / (size * size);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            uint16_t sum = 0;
            for (int dx = lo; dx < hi; dx++)
                for (int dy = lo; dy < hi; dy++)
                    sum = sum + src[(ptrdiff_t)bw * (y + dy) + (x + dx)];
            dst[(size_t)bw * y + x] = (uint8_t)(sum * scale);
        }
}

static int cmp_u8(const void *a, const void *b)
{
    return (int)*(const uint8_t *)a - (int)*(const uint8_t *)b;
}

/* nlmeans_template.c:135-230: the Devillard networks return the true median
 * of 9 / 25 values, which is what a full sort yields. */
static void pf_median(const uint8_t *src, uint8_t *dst, int w, int h, int bw, int size)
{
    const int lo = -((size - 1) / 2), hi = (size + 1) / 2;
    uint8_t win[25];
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            int n = 0;
            for (int dx = lo; dx < hi; dx++)
                for (int dy = lo; dy < hi; dy++)
                    win[n++] = src[(ptrdiff_t)bw * (y + dy) + (x + dx)];
            qsort(win, n, 1, cmp_u8);
            dst[(size_t)bw * y + x] = win[n / 2];
        }
}

/* nlmeans_template.c:232-323.  The reference leaves its inner (row) loop with
 * `goto end` both at the first neighbour and at the origin, so: in the first
 * column only the top pixel seeds min/max, and in the centre column only the
 * pixels ABOVE the origin are visited.  Reproduced here on purpose. */
static void pf_csm(const uint8_t *src, uint8_t *dst, int w, int h, int bw, int size)
{
    const int lo = -((size - 1) / 2), hi = (size + 1) / 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            uint8_t vmin = 0, vmax = 0;
            for (int dx = lo; dx < hi; dx++)
            {
                for (int dy = lo; dy < hi; dy++)
                {
                    if (dx == 0 && dy == 0)
                        break;
                    const uint8_t v = src[(ptrdiff_t)bw * (y + dy) + (x + dx)];
                    if (dx == lo && dy == lo)
                    {
                        vmin = vmax = v;
                        break;
                    }
                    if (v < vmin) vmin = v;
                    if (v > vmax) vmax = v;
                }
            }
            const uint8_t mid  = (vmin + vmax) / 2;
            const uint8_t min2 = (vmin + mid) / 2, max2 = (vmax + mid) / 2;
            const uint8_t min3 = (min2 + mid) / 2, max3 = (max2 + mid) / 2;
            const uint8_t v = src[(size_t)bw * y + x];
            uint8_t *o = dst + (size_t)bw * y + x;
            if      (v < vmin) *o = vmin;
            else if (v > vmax) *o = vmax;
            else if (v < min2) *o = min2;
            else if (v > max2) *o = max2;
            else if (v < min3) *o = min3;
            else if (v > max3) *o = max3;
            /* otherwise dst keeps the copy of the source made by the caller */
        }
}

/* nlmeans_template.c:325-426.  Quirks kept: the two gradient sums live in
 * uint16 (negative sums wrap, the `> 0 ? :` is a no-op), the mask byte is the
 * low 8 bits of their scaled sum, and the clean-up pass edits the mask in place
 * in raster order so later pixels see earlier demotions. */
static void pf_edgeboost(const uint8_t *src, uint8_t *dst, int w, int h, int border)
{
    static const int kern[3][3] = { {-31, 0, 31}, {-44, 0, 44}, {-31, 0, 31} };
    const double coef = 1.0 / 126.42;
    const int bw = w + 2 * border, bh = h + 2 * border;
    uint8_t *mask_mem = calloc((size_t)bw * bh, 1);
    uint8_t *mask = mask_mem + border + (size_t)bw * border;

    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            uint16_t g1 = 0, g2 = 0;
            for (int dx = -1; dx <= 1; dx++)
                for (int dy = -1; dy <= 1; dy++)
                {
                    const int v = src[(ptrdiff_t)bw * (y + dy) + (x + dx)];
                    g1 += kern[dy + 1][dx + 1] * v;
                    g2 += kern[dx + 1][dy + 1] * v;
                }
            g1 = (uint16_t)(((double)g1 * coef) + 128);
            g2 = (uint16_t)(((double)g2 * coef) + 128);
            const uint8_t m = (uint8_t)(g1 + g2);
            mask[(size_t)bw * y + x] = m > 160 ? 235 : m > 16 ? 128 : 16;
        }

    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            uint8_t *m = mask + (size_t)bw * y + x;
            if (*m <= 16)
                continue;
            int near = 0;
            for (int dx = -1; dx <= 1; dx++)
                for (int dy = -1; dy <= 1; dy++)
                    if (mask[(ptrdiff_t)bw * (y + dy) + (x + dx)] > 16)
                        near++;
            if (near < 3)
            {
                *m = 16;
                continue;
            }
            const int s = src[(size_t)bw * y + x];
            uint8_t *o = dst + (size_t)bw * y + x;
            if (*m == 235) *o = (3 * s + 1 * *o) / 4;
            else           *o = (2 * s + 3 * *o) / 5;
        }
    free(mask_mem);
}

/* nlmeans_template.c:428-543 */
int orc_nlmeans_prefilter(const uint8_t *bordered, int w, int h, int border,
                          int filter_type, uint8_t *pre)
{
    const int bw = w + 2 * border, bh = h + 2 * border;
    memcpy(pre, bordered, (size_t)bw * bh);
    if (!(filter_type & (PF_MEAN3 | PF_MEAN5 | PF_MEDIAN3 | PF_MEDIAN5 | PF_CSM3 | PF_CSM5)))
        return 0;

    const uint8_t *img = bordered + border + (size_t)bw * border;
    uint8_t *img_pre = pre + border + (size_t)bw * border;

    if      (filter_type & PF_CSM5)    pf_csm(img, img_pre, w, h, bw, 5);
    else if (filter_type & PF_CSM3)    pf_csm(img, img_pre, w, h, bw, 3);
    else if (filter_type & PF_MEDIAN5) pf_median(img, img_pre, w, h, bw, 5);
    else if (filter_type & PF_MEDIAN3) pf_median(img, img_pre, w, h, bw, 3);
    else if (filter_type & PF_MEAN5)   pf_mean(img, img_pre, w, h, bw, 5);
    else if (filter_type & PF_MEAN3)   pf_mean(img, img_pre, w, h, bw, 3);

    if (filter_type & PF_EDGEBOOST)
        pf_edgeboost(img, img_pre, w, h, border);

    int wet = 1, dry = 0;
    if ((filter_type & PF_REDUCE50) && (filter_type & PF_REDUCE25)) { wet = 1; dry = 3; }
    else if (filter_type & PF_REDUCE50)                             { wet = 1; dry = 1; }
    else if (filter_type & PF_REDUCE25)                             { wet = 3; dry = 1; }
    if (dry > 0)
        for (size_t i = 0; i < (size_t)bw * bh; i++)
            pre[i] = (uint8_t)((wet * pre[i] + dry * bordered[i]) / (wet + dry));

    mirror_borders(pre, w, h, border);
    return 1;
}

/* ---- the plane filter (nlmeans_template.c:593-717) -------------------------- */

typedef struct { float weight_sum; float pixel_sum; } acc_t;

/* Box sums of squared differences for one displacement: ssd[y*w+x] =
 * sum over the n x n patch centred on (x,y) of (a - b(+dx,+dy))^2.
 * Equivalent to build_integral (nlmeans_template.c:545-591) + the 4-corner
 * read at :682. */
static void patch_ssd(const uint8_t *a_img, const uint8_t *b_img, int bw,
                      int w, int h, int n, int dx, int dy,
                      uint32_t *colsum, uint32_t *ssd)
{
    const int nh = (n - 1) / 2;
    const int ew = w + n - 1;               /* columns x-nh .. x+nh over the row */
    /* colsum[i] holds, for extended column i (image x = i - nh), the sum over the
     * current n-row window of squared differences. */
    memset(colsum, 0, sizeof(uint32_t) * ew);
    for (int yy = -nh; yy < h + nh; yy++)
    {
        const uint8_t *pa = a_img + (ptrdiff_t)yy * bw - nh;
        const uint8_t *pb = b_img + (ptrdiff_t)(yy + dy) * bw - nh + dx;
        if (yy - n >= -nh)
        {
            const uint8_t *qa = a_img + (ptrdiff_t)(yy - n) * bw - nh;
            const uint8_t *qb = b_img + (ptrdiff_t)(yy - n + dy) * bw - nh + dx;
            for (int i = 0; i < ew; i++)
            {
                const int dn = pa[i] - pb[i], dold = qa[i] - qb[i];
                colsum[i] += (uint32_t)(dn * dn) - (uint32_t)(dold * dold);
            }
        }
        else
        {
            for (int i = 0; i < ew; i++)
            {
                const int dn = pa[i] - pb[i];
                colsum[i] += (uint32_t)(dn * dn);
            }
        }
        const int y = yy - nh;              /* output row whose window just completed */
        if (y < 0)
            continue;
        uint32_t run = 0;
        for (int i = 0; i < n; i++)
            run += colsum[i];
        uint32_t *out = ssd + (size_t)y * w;
        out[0] = run;
        for (int x = 1; x < w; x++)
        {
            run += colsum[x + n - 1] - colsum[x - 1];
            out[x] = run;
        }
    }
}

void orc_nlmeans_plane(const uint8_t *const *frames, const uint8_t *const *frames_pre,
                       const uint8_t *src_pre_plane,
                       int nframes, int w, int h, int border,
                       const orc_nlmeans_params_t *p,
                       uint8_t *dst, int dst_stride)
{
    float exptable[EXPSIZE];
    float wft;
    int diff_max;
    orc_nlmeans_tables(p->strength, p->patch_size, exptable, &wft, &diff_max);

    const int n = p->patch_size;
    const int r_half = (p->range - 1) / 2;
    const int bw = w + 2 * border;
    const size_t origin = border + (size_t)bw * border;
    const double origin_tune = p->origin_tune;

    acc_t *acc = calloc((size_t)w * h, sizeof(acc_t));
    uint32_t *ssd = malloc(sizeof(uint32_t) * (size_t)w * h);
    uint32_t *colsum = malloc(sizeof(uint32_t) * (w + n));

    const uint8_t *src = frames[0] + origin;
    const uint8_t *src_pre = (src_pre_plane ? src_pre_plane : frames_pre[0]) + origin;

    for (int f = 0; f < nframes; f++)
    {
        const uint8_t *cmp = frames[f] + origin;
        const uint8_t *cmp_pre = frames_pre[f] + origin;
        for (int dy = -r_half; dy <= r_half; dy++)
            for (int dx = -r_half; dx <= r_half; dx++)
            {
                if (f == 0 && dx == 0 && dy == 0)
                {
                    /* :644-655 — float += double, evaluated in double */
                    for (int y = 0; y < h; y++)
                        for (int x = 0; x < w; x++)
                        {
                            acc_t *a = &acc[(size_t)y * w + x];
                            a->weight_sum += origin_tune;
                            a->pixel_sum  += origin_tune * src[(size_t)y * bw + x];
                        }
                    continue;
                }
                patch_ssd(src_pre, cmp_pre, bw, w, h, n, dx, dy, colsum, ssd);
                for (int y = 0; y < h; y++)
                    for (int x = 0; x < w; x++)
                    {
                        /* :682-694 */
                        const int diff = (int)ssd[(size_t)y * w + x];
                        if (diff < diff_max)
                        {
                            const int idx = diff * wft;
                            const float weight = exptable[idx];
                            acc_t *a = &acc[(size_t)y * w + x];
                            a->weight_sum += weight;
                            a->pixel_sum  += weight * cmp[(ptrdiff_t)(y + dy) * bw + x + dx];
                        }
                    }
            }
    }

    /* :704-713 */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            const acc_t *a = &acc[(size_t)y * w + x];
            const uint8_t v = (uint8_t)(a->pixel_sum / a->weight_sum);
            dst[(size_t)y * dst_stride + x] = v ? v : src[(size_t)y * bw + x];
        }

    free(colsum);
    free(ssd);
    free(acc);
}


/* ---- 16-bit samples (nlmeans_plane_16 etc.: the same template instantiated with
 * pixel = uint16_t, nlmeans.c:253-262; depths 10 and 12 in 16-bit containers) ----------
 * prefilter = 0 only.  The strength is scaled by (depth-8)^2 before the tables are built
 * (nlmeans.c:343); everything else is the 8-bit algorithm on wider samples. */
static void mirror_borders16(uint16_t *mem, int w, int h, int border)
{
    const int bw = w + 2 * border;
    uint16_t *img = mem + border + (size_t)bw * border;
    for (int y = 0; y < h; y++)
    {
        uint16_t *row = img + (size_t)y * bw;
        for (int i = 0; i < border; i++)
        {
            row[-1 - i] = row[i];
            row[w + i]  = row[w - 1 - i];
        }
    }
    for (int i = 0; i < border; i++)
    {
        memcpy(img - border - (size_t)(i + 1) * bw, img - border + (size_t)i * bw, sizeof(uint16_t) * bw);
        memcpy(img - border + (size_t)(h + i) * bw, img - border + (size_t)(h - 1 - i) * bw, sizeof(uint16_t) * bw);
    }
}

void orc_nlmeans_plane16_pf(const uint16_t *const *planes, int plane_stride, int nframes, int w, int h,
                            int depth, const orc_nlmeans_params_t *p, int src_prefiltered,
                            uint16_t *dst, int dst_stride);
#include "nlmeans_prefilter16.h"

static void patch_ssd16(const uint16_t *a_img, const uint16_t *b_img, int bw,
                        int w, int h, int n, int dx, int dy,
                        uint32_t *colsum, uint32_t *ssd)
{
    const int nh = (n - 1) / 2;
    const int ew = w + n - 1;
    memset(colsum, 0, sizeof(uint32_t) * ew);
    for (int yy = -nh; yy < h + nh; yy++)
    {
        const uint16_t *pa = a_img + (ptrdiff_t)yy * bw - nh;
        const uint16_t *pb = b_img + (ptrdiff_t)(yy + dy) * bw - nh + dx;
        for (int i = 0; i < ew; i++)
        {
            const int dn = pa[i] - pb[i];
            colsum[i] += (uint32_t)dn * (uint32_t)dn;
        }
        if (yy - n >= -nh)
        {
            const uint16_t *qa = a_img + (ptrdiff_t)(yy - n) * bw - nh;
            const uint16_t *qb = b_img + (ptrdiff_t)(yy - n + dy) * bw - nh + dx;
            for (int i = 0; i < ew; i++)
            {
                const int dold = qa[i] - qb[i];
                colsum[i] -= (uint32_t)dold * (uint32_t)dold;
            }
        }
        const int y = yy - nh;
        if (y < 0)
            continue;
        uint32_t run = 0;
        for (int i = 0; i < n; i++)
            run += colsum[i];
        uint32_t *out = ssd + (size_t)y * w;
        out[0] = run;
        for (int x = 1; x < w; x++)
        {
            run += colsum[x + n - 1] - colsum[x - 1];
            out[x] = run;
        }
    }
}

void orc_nlmeans_plane16(const uint16_t *const *planes, int plane_stride, int nframes, int w, int h,
                         int depth, const orc_nlmeans_params_t *p, uint16_t *dst, int dst_stride)
{
    orc_nlmeans_plane16_pf(planes, plane_stride, nframes, w, h, depth, p, 0, dst, dst_stride);
}

/* The same with the prefilters of nlmeans_prefilter_16 (p->prefilter): patches are compared on the
 * prefiltered planes, the weighted sum runs over the raw ones (nlmeans_template.c:614-635).
 * src_prefiltered: whether frame 0's own patches come from its prefiltered plane - the reference
 * latches src_pre before frame 0's prefilter call (:615 vs :631), see oracle_stream.nlmeans_stream. */
void orc_nlmeans_plane16_pf(const uint16_t *const *planes, int plane_stride, int nframes, int w, int h,
                            int depth, const orc_nlmeans_params_t *p, int src_prefiltered,
                            uint16_t *dst, int dst_stride)
{
    float exptable[EXPSIZE];
    float wft;
    int diff_max;
    const double scaled = p->strength * (depth > 8 ? (depth - 8) * (depth - 8) : 1);   /* nlmeans.c:343 */
    orc_nlmeans_tables(scaled, p->patch_size, exptable, &wft, &diff_max);

    const int n = p->patch_size;
    const int r_half = (p->range - 1) / 2;
    const int border = orc_nlmeans_border(n);
    const int bw = w + 2 * border, bh = h + 2 * border;
    const size_t origin = border + (size_t)bw * border;
    const double origin_tune = p->origin_tune;

    uint16_t **fr = malloc(sizeof(uint16_t *) * nframes), **pre = malloc(sizeof(uint16_t *) * nframes);
    for (int f = 0; f < nframes; f++)
    {
        fr[f] = calloc((size_t)bw * bh, sizeof(uint16_t));
        for (int y = 0; y < h; y++)
            memcpy(fr[f] + origin + (size_t)y * bw, planes[f] + (size_t)y * plane_stride, sizeof(uint16_t) * w);
        mirror_borders16(fr[f], w, h, border);
        pre[f] = malloc(sizeof(uint16_t) * (size_t)bw * bh);
        orc_nlmeans_prefilter16(fr[f], w, h, border, p->prefilter, pre[f]);
    }
    const uint16_t *src_pre = (src_prefiltered ? pre[0] : fr[0]) + origin;
    acc_t *acc = calloc((size_t)w * h, sizeof(acc_t));
    uint32_t *ssd = malloc(sizeof(uint32_t) * (size_t)w * h);
    uint32_t *colsum = malloc(sizeof(uint32_t) * (w + n));
    const uint16_t *src = fr[0] + origin;

    for (int f = 0; f < nframes; f++)
    {
        const uint16_t *cmp = fr[f] + origin;
        const uint16_t *cmp_pre = pre[f] + origin;
        for (int dy = -r_half; dy <= r_half; dy++)
            for (int dx = -r_half; dx <= r_half; dx++)
            {
                if (f == 0 && dx == 0 && dy == 0)
                {
                    for (int y = 0; y < h; y++)
                        for (int x = 0; x < w; x++)
                        {
                            acc_t *a = &acc[(size_t)y * w + x];
                            a->weight_sum += origin_tune;
                            a->pixel_sum  += origin_tune * src[(size_t)y * bw + x];
                        }
                    continue;
                }
                patch_ssd16(src_pre, cmp_pre, bw, w, h, n, dx, dy, colsum, ssd);
                for (int y = 0; y < h; y++)
                    for (int x = 0; x < w; x++)
                    {
                        const int diff = (int)ssd[(size_t)y * w + x];
                        if (diff < diff_max)
                        {
                            const int idx = diff * wft;
                            const float weight = exptable[idx];
                            acc_t *a = &acc[(size_t)y * w + x];
                            a->weight_sum += weight;
                            a->pixel_sum  += weight * cmp[(ptrdiff_t)(y + dy) * bw + x + dx];
                        }
                    }
            }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            const acc_t *a = &acc[(size_t)y * w + x];
            const uint16_t v = (uint16_t)(a->pixel_sum / a->weight_sum);
            dst[(size_t)y * dst_stride + x] = v ? v : src[(size_t)y * bw + x];
        }
    free(colsum);
    free(ssd);
    free(acc);
    for (int f = 0; f < nframes; f++) { free(fr[f]); free(pre[f]); }
    free(fr);
    free(pre);
}
