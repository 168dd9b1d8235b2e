/* nlmeans_prefilter16.h — the prefilters of nlmeans_template.c:103-543 for the 16-bit
 * instantiation (pixel = uint16_t, pixel_2 = uint32_t, :10-13).  Included by nlmeans_oracle.c.
 * TEST INFRASTRUCTURE ONLY.  Same structure as the 8-bit restatement above it; what the wider
 * types change is spelled out where it matters:
 *   - the window sum of the mean filter and the two gradient sums of the edge boost live in
 *     uint32 (negative gradients wrap at 32 bits, the `> 0 ? :` is still a no-op), the mask sample
 *     is the low 16 bits of their scaled sum;
 *   - the edge-boost thresholds and mask values (160 / 16, 235 / 128 / 16) are NOT scaled with the
 *     depth (:367-378), so on 10 / 12-bit data nearly every sample classifies as a strong edge;
 *   - min / max / midpoints of CSM are uint16.
 */
static int cmp_u16(const void *a, const void *b)
{
    return (int)*(const uint16_t *)a - (int)*(const uint16_t *)b;
}

static void pf16_mean(const uint16_t *src, uint16_t *dst, int w, int h, int bw, int size)
{
    const int lo = -((size - 1) / 2), hi = (size + 1) / 2;
    const double scale = 1.0 / (size * size);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            uint32_t sum = 0;
            for (int dx = lo; dx < hi; dx++)
                for (int dy = lo; dy < hi; dy++)
                    sum = sum + src[(ptrdiff_t)bw * (y + dy) + (x + dx)];
            dst[(size_t)bw * y + x] = (uint16_t)(sum * scale);
        }
}

static void pf16_median(const uint16_t *src, uint16_t *dst, int w, int h, int bw, int size)
{
    const int lo = -((size - 1) / 2), hi = (size + 1) / 2;
    uint16_t win[25];
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            int n = 0;
            for (int dx = lo; dx < hi; dx++)
                for (int dy = lo; dy < hi; dy++)
                    win[n++] = src[(ptrdiff_t)bw * (y + dy) + (x + dx)];
            qsort(win, n, sizeof(uint16_t), cmp_u16);
            dst[(size_t)bw * y + x] = win[n / 2];
        }
}

static void pf16_csm(const uint16_t *src, uint16_t *dst, int w, int h, int bw, int size)
{
    const int lo = -((size - 1) / 2), hi = (size + 1) / 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            uint16_t vmin = 0, vmax = 0;
            for (int dx = lo; dx < hi; dx++)
            {
                for (int dy = lo; dy < hi; dy++)
                {
                    if (dx == 0 && dy == 0)
                        break;                                   /* the reference's `goto end` (:258-262) */
                    const uint16_t v = src[(ptrdiff_t)bw * (y + dy) + (x + dx)];
                    if (dx == lo && dy == lo)
                    {
                        vmin = vmax = v;
                        break;
                    }
                    if (v < vmin) vmin = v;
                    if (v > vmax) vmax = v;
                }
            }
            const uint16_t mid  = (vmin + vmax) / 2;
            const uint16_t min2 = (vmin + mid) / 2, max2 = (vmax + mid) / 2;
            const uint16_t min3 = (min2 + mid) / 2, max3 = (max2 + mid) / 2;
            const uint16_t v = src[(size_t)bw * y + x];
            uint16_t *o = dst + (size_t)bw * y + x;
            if      (v < vmin) *o = vmin;
            else if (v > vmax) *o = vmax;
            else if (v < min2) *o = min2;
            else if (v > max2) *o = max2;
            else if (v < min3) *o = min3;
            else if (v > max3) *o = max3;
        }
}

static void pf16_edgeboost(const uint16_t *src, uint16_t *dst, int w, int h, int border)
{
    static const int kern[3][3] = { {-31, 0, 31}, {-44, 0, 44}, {-31, 0, 31} };
    const double coef = 1.0 / 126.42;
    const int bw = w + 2 * border, bh = h + 2 * border;
    uint16_t *mask_mem = calloc((size_t)bw * bh, sizeof(uint16_t));
    uint16_t *mask = mask_mem + border + (size_t)bw * border;

    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            uint32_t g1 = 0, g2 = 0;
            for (int dx = -1; dx <= 1; dx++)
                for (int dy = -1; dy <= 1; dy++)
                {
                    const int v = src[(ptrdiff_t)bw * (y + dy) + (x + dx)];
                    g1 += kern[dy + 1][dx + 1] * v;
                    g2 += kern[dx + 1][dy + 1] * v;
                }
            g1 = (uint32_t)(((double)g1 * coef) + 128);
            g2 = (uint32_t)(((double)g2 * coef) + 128);
            const uint16_t m = (uint16_t)(g1 + g2);
            mask[(size_t)bw * y + x] = m > 160 ? 235 : m > 16 ? 128 : 16;
        }

    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            uint16_t *m = mask + (size_t)bw * y + x;
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
            uint16_t *o = dst + (size_t)bw * y + x;
            if (*m == 235) *o = (3 * s + 1 * *o) / 4;
            else           *o = (2 * s + 3 * *o) / 5;
        }
    free(mask_mem);
}

/* nlmeans_template.c:428-543 for uint16 samples: `bordered` and `pre` are (w + 2*border) x
 * (h + 2*border) sample arrays */
int orc_nlmeans_prefilter16(const uint16_t *bordered, int w, int h, int border, int filter_type, uint16_t *pre)
{
    const int bw = w + 2 * border, bh = h + 2 * border;
    memcpy(pre, bordered, sizeof(uint16_t) * (size_t)bw * bh);
    if (!(filter_type & (PF_MEAN3 | PF_MEAN5 | PF_MEDIAN3 | PF_MEDIAN5 | PF_CSM3 | PF_CSM5)))
        return 0;

    const uint16_t *img = bordered + border + (size_t)bw * border;
    uint16_t *img_pre = pre + border + (size_t)bw * border;

    if      (filter_type & PF_CSM5)    pf16_csm(img, img_pre, w, h, bw, 5);
    else if (filter_type & PF_CSM3)    pf16_csm(img, img_pre, w, h, bw, 3);
    else if (filter_type & PF_MEDIAN5) pf16_median(img, img_pre, w, h, bw, 5);
    else if (filter_type & PF_MEDIAN3) pf16_median(img, img_pre, w, h, bw, 3);
    else if (filter_type & PF_MEAN5)   pf16_mean(img, img_pre, w, h, bw, 5);
    else if (filter_type & PF_MEAN3)   pf16_mean(img, img_pre, w, h, bw, 3);

    if (filter_type & PF_EDGEBOOST)
        pf16_edgeboost(img, img_pre, w, h, border);

    int wet = 1, dry = 0;
    if ((filter_type & PF_REDUCE50) && (filter_type & PF_REDUCE25)) { wet = 1; dry = 3; }
    else if (filter_type & PF_REDUCE50)                             { wet = 1; dry = 1; }
    else if (filter_type & PF_REDUCE25)                             { wet = 3; dry = 1; }
    if (dry > 0)
        for (size_t i = 0; i < (size_t)bw * bh; i++)
            pre[i] = (uint16_t)((wet * pre[i] + dry * bordered[i]) / (wet + dry));

    mirror_borders16(pre, w, h, border);
    return 1;
}

/* nlmeans_template.c:69-101 for uint16 samples (src_stride in samples) */
void orc_nlmeans_make_bordered16(const uint16_t *src, int w, int h, int src_stride, int border, uint16_t *dst)
{
    const int bw = w + 2 * border;
    uint16_t *img = dst + border + (size_t)bw * border;
    for (int y = 0; y < h; y++)
        memcpy(img + (size_t)y * bw, src + (size_t)y * src_stride, sizeof(uint16_t) * w);
    mirror_borders16(dst, w, h, border);
}
