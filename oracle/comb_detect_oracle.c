/* comb_detect_oracle.c — CPU restatement of libhb's comb detection (8-bit luma).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows /root/reference/libhb/comb_detect.c and templates/comb_detect_template.c:
 *   template :288-402  gamma detect      template :789-933  integer detect
 *   comb_detect.c :901-966 mask filter   :726-792 erode   :556-622 dilate
 *   :221-276 / :384-454 block scoring    :1029-1049 classification
 *   :1051-1072 pass order                :1074-1081 gamma LUT, :1151-1161 thresholds
 * The reference splits every pass into row segments per CPU; none of the results
 * depends on the split (each pass reads a finished buffer, block rows start at
 * multiples of block_height in every segment), so whole planes are walked.
 *
 * Quirk kept on purpose: the three 3x3 mask passes take their row pointers at
 * column 1 and then index columns 1..width-2 from there (comb_detect.c:939-947),
 * i.e. they produce columns 2..width-1 and read column `width`, which is the
 * stride padding or — when stride == width — the first pixel of the next row.
 */
#include "oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

struct orc_comb
{
    orc_comb_params_t p;
    int width, height, stride;
    uint8_t *mask, *mask_filtered, *mask_temp;
    int depth;
    float *lut;                      /* 1 << depth entries (:1102) */
    float g_mthresh, g_athresh, g_athresh6;
    int athresh_sq, athresh6, c32_min, c32_max;
    int box_x, box_y;                /* pv->mask_box_x / _y: the last block the check recorded (:205-208, 262-265) */
};

orc_comb_t *orc_comb_new(int width, int height, const orc_comb_params_t *p)
{
    return orc_comb_new_depth(width, height, p, 8);
}

/* depth 10 / 12: 16-bit luma samples, thresholds scaled as comb_detect.c:1151-1162 */
orc_comb_t *orc_comb_new_depth(int width, int height, const orc_comb_params_t *p, int depth)
{
    orc_comb_t *c = calloc(1, sizeof(*c));
    c->p = *p;
    c->depth = depth;
    const int max_value = (1 << depth) - 1;
    c->p.motion_threshold  <<= (depth - 8);
    c->p.spatial_threshold <<= (depth - 8);
    if (c->p.block_width > width)   c->p.block_width = width;       /* :1139-1146 */
    if (c->p.block_height > height) c->p.block_height = height;
    c->width = width;
    c->height = height;
    c->stride = (width + 63) / 64 * 64;
    /* +64: the reference's buffers carry AV_INPUT_BUFFER_PADDING_SIZE after the plane */
    c->mask          = calloc((size_t)c->stride * height + 64, 1);
    c->mask_filtered = calloc((size_t)c->stride * height + 64, 1);
    c->mask_temp     = calloc((size_t)c->stride * height + 64, 1);
    c->lut = malloc(sizeof(float) * (max_value + 1));
    for (int i = 0; i <= max_value; i++)
        c->lut[i] = pow(((float)i / (float)max_value), 2.2f);      /* :1074-1081 */
    c->g_mthresh  = (float)c->p.motion_threshold / (float)max_value;     /* :1153-1155 */
    c->g_athresh  = (float)c->p.spatial_threshold / (float)max_value;
    c->g_athresh6 = 6 * c->g_athresh;
    c->athresh_sq = c->p.spatial_threshold * c->p.spatial_threshold;
    c->athresh6   = 6 * c->p.spatial_threshold;
    c->c32_min = 10 << (depth - 8);                                /* :1161-1162 */
    c->c32_max = 15 << (depth - 8);
    return c;
}

void orc_comb_free(orc_comb_t *c)
{
    if (!c) return;
    free(c->mask); free(c->mask_filtered); free(c->mask_temp); free(c->lut);
    free(c);
}

const uint8_t *orc_comb_mask(orc_comb_t *c, int which, int *stride)
{
    if (stride) *stride = c->stride;
    return which == 0 ? c->mask : which == 1 ? c->mask_filtered : c->mask_temp;
}

/* template :288-402 (gamma) and :789-933 (integer) */
/* luma planes: uint8_t, or uint16_t when depth > 8; `stride` in samples */
#define SMP(p, i) (c->depth > 8 ? (int)((const uint16_t *)(p))[i] : (int)((const uint8_t *)(p))[i])
static void detect(orc_comb_t *c, const void *prev, const void *cur, const void *next,
                   int stride, int force)
{
    const int gamma = c->p.mode & 1;
    for (int y = 2; y < c->height - 2; y++)
    {
        const ptrdiff_t row = (ptrdiff_t)y * stride;
        uint8_t *m = c->mask + (size_t)y * c->stride;
        memset(m, 0, c->stride);
        for (int x = 0; x < c->width; x++)
        {
            const int v = SMP(cur, row + x), u1 = SMP(cur, row + x - stride), d1 = SMP(cur, row + x + stride);
            const int u2 = SMP(cur, row + x - 2 * stride), d2 = SMP(cur, row + x + 2 * stride);
            if (gamma)
            {
                const float *L = c->lut;
                const float up = L[v] - L[u1], dn = L[v] - L[d1];
                if (!((up > c->g_athresh && dn > c->g_athresh) || (up < -c->g_athresh && dn < -c->g_athresh)))
                    continue;
                int motion = 0;
                if (c->g_mthresh > 0)
                {
                    if (fabs(L[SMP(prev, row + x)] - L[v]) > c->g_mthresh &&
                        fabs(L[u1] - L[SMP(next, row + x - stride)]) > c->g_mthresh &&
                        fabs(L[d1] - L[SMP(next, row + x + stride)]) > c->g_mthresh)
                        motion++;
                    if (fabs(L[SMP(next, row + x)] - L[v]) > c->g_mthresh &&
                        fabs(L[SMP(prev, row + x - stride)] - L[u1]) > c->g_mthresh &&
                        fabs(L[SMP(prev, row + x + stride)] - L[d1]) > c->g_mthresh)
                        motion++;
                }
                else
                    motion = 1;
                if (motion || force)
                {
                    const float combing = fabs(L[u2] + (4 * L[v]) + L[d2] - (3 * (L[u1] + L[d1])));
                    if (combing > c->g_athresh6)
                        m[x] = 1;
                }
            }
            else
            {
                const int at = c->p.spatial_threshold, mt = c->p.motion_threshold;
                const int up = v - u1, dn = v - d1;
                if (!((up > at && dn > at) || (up < -at && dn < -at)))
                    continue;
                int motion = 0;
                if (mt > 0)
                {
                    if (abs(SMP(prev, row + x) - v) > mt && abs(u1 - SMP(next, row + x - stride)) > mt && abs(d1 - SMP(next, row + x + stride)) > mt)
                        motion++;
                    if (abs(SMP(next, row + x) - v) > mt && abs(SMP(prev, row + x - stride) - u1) > mt && abs(SMP(prev, row + x + stride) - d1) > mt)
                        motion++;
                }
                else
                    motion = 1;
                if (!(motion || force))
                    continue;
                if (c->p.spatial_metric == 0)
                {
                    if (abs(v - d2) < c->c32_min && abs(v - d1) > c->c32_max) m[x] = 1;
                }
                else if (c->p.spatial_metric == 1)
                {
                    if ((u1 - v) * (d1 - v) > c->athresh_sq) m[x] = 1;
                }
                else if (c->p.spatial_metric == 2)
                {
                    if (abs(u2 + 4 * v + d2 - 3 * (u1 + d1)) > c->athresh6) m[x] = 1;
                }
            }
        }
    }
}
#undef SMP

/* The shared frame of the three 3x3 passes: rows 1..height-2, source/destination
 * pointers offset by one column (see the file comment). op: 0 filter, 1 erode, 2 dilate */
static void mask_pass(const orc_comb_t *c, const uint8_t *src, uint8_t *dst, int op)
{
    const int st = c->stride;
    for (int y = 1; y < c->height - 1; y++)
    {
        const uint8_t *p = src + (size_t)(y - 1) * st + 1, *q = src + (size_t)y * st + 1,
                      *n = src + (size_t)(y + 1) * st + 1;
        uint8_t *o = dst + (size_t)y * st + 1;
        for (int x = 1; x < c->width - 1; x++)
        {
            if (op == 0)
            {
                const int hc = q[x - 1] & q[x] & q[x + 1];
                const int vc = p[x] & q[x] & n[x];
                o[x] = c->p.filter_mode == 1 ? hc : (hc & vc);
                continue;
            }
            const int count = p[x - 1] + p[x] + p[x + 1] + q[x - 1] + q[x + 1] + n[x - 1] + n[x] + n[x + 1];
            if (op == 1) o[x] = q[x] == 0 ? 0 : count >= 2;      /* erosion threshold 2, :734 */
            else         o[x] = q[x] ? 1 : count >= 4;           /* dilation threshold 4, :564 */
        }
    }
}

/* :221-276 and :384-454 folded with :1029-1049.  Also keeps mask_box_x / _y the way ONE check thread leaves
 * them (the reference's segment threads race on it, :205-208 - the overlay modes are pinned with cpu_count = 1):
 * every block with score >= threshold / 2 overwrites the position, the scan stops at the first block whose score
 * exceeds the threshold. */
static int score_blocks(orc_comb_t *c, int filtered)
{
    const int bw = c->p.block_width, bh = c->p.block_height, thr = c->p.block_threshold;
    const uint8_t *m = filtered ? c->mask_filtered : c->mask;
    int light = 0;
    for (int y = 0; y + bh <= c->height; y += bh)
        for (int x = 0; x < c->width - bw; x += bw)
        {
            int score = 0;
            for (int by = 0; by < bh; by++)
            {
                const uint8_t *r = m + (size_t)(y + by) * c->stride + x;
                for (int bx = 0; bx < bw; bx++)
                {
                    if (filtered)                         score += r[bx];
                    else if (x + bx == 0)                 score += r[bx] & r[bx + 1];
                    else if (x + bx == c->width - 1)      score += r[bx - 1] & r[bx];
                    else                                  score += r[bx - 1] & r[bx] & r[bx + 1];
                }
            }
            if (score >= thr / 2)  { c->box_x = x; c->box_y = y; }
            if (score > thr)       return 2;
            if (score >= thr / 2)  light = 1;
        }
    return light;
}

int orc_comb_classify(orc_comb_t *c, const uint8_t *prev, const uint8_t *cur, const uint8_t *next,
                      int stride, int force_exhaustive)
{
    /* depth > 8: the pointers address uint16_t samples, `stride` counts samples */
    detect(c, prev, cur, next, stride, force_exhaustive);
    const int filt = (c->p.mode & 2) != 0;
    if (filt)
    {
        if (c->p.filter_mode == 1)
            mask_pass(c, c->mask, c->mask_filtered, 0);
        else
            mask_pass(c, c->mask, c->mask_temp, 0);
        if (c->p.filter_mode == 2)
        {
            mask_pass(c, c->mask_temp, c->mask_filtered, 1);
            mask_pass(c, c->mask_filtered, c->mask_temp, 2);
            mask_pass(c, c->mask_temp, c->mask_filtered, 1);
        }
    }
    return score_blocks(c, filt);
}


/* ---- mask overlay, modes 4 (MODE_MASK) and 8 (MODE_COMPOSITE): comb_detect_template.c:21-136 ----------------
 * process_frame (comb_detect.c:1519-1526) calls this on a COPY of the classified frame when it is combed.
 * draw_mask_box (:21-53) first writes 128 along the outline of the last recorded block INTO THE MASK BUFFER
 * (the filtered one in filter mode) - and nothing clears it: cells no pass rewrites (columns 0-1, rows 0 and
 * height-1 of the filtered mask; rows 0, 1, height-2, height-1 of the detector's mask) keep the 128 for the
 * rest of the stream, where block scoring and the 3x3 passes read it.  Kept as is; the only departure: the
 * bottom line of a box in the last block row of a picture whose height is a multiple of the block height falls
 * on row `height`, one row past the buffer - the reference scribbles into its allocation padding there, here
 * that line is skipped.
 * apply_mask (:72-136): mask-only mode blanks the picture (luma 0, chroma half, whole plane), then luma takes
 * max where the mask is 1 and half where it is 128. */
void orc_comb_overlay(orc_comb_t *c, void *const plane[3], const int stride[3], const int pheight[3])
{
    const int filt = (c->p.mode & 2) != 0;
    uint8_t *m = filt ? c->mask_filtered : c->mask;
    const int st = c->stride, x = c->box_x, y = c->box_y, bw = c->p.block_width, bh = c->p.block_height;
    for (int bx = 0; bx < bw; bx++)
    {
        m[(size_t)y * st + x + bx] = 128;
        if (y + bh < c->height) m[(size_t)(y + bh) * st + x + bx] = 128;
    }
    for (int by = 0; by < bh; by++)
    {
        m[(size_t)(y + by) * st + x] = 128;
        m[(size_t)(y + by) * st + x + bw] = 128;
    }
    const int max = (1 << c->depth) - 1, half = 1 << (c->depth - 1);   /* comb_detect.c:1104-1105 */
    const int bps = c->depth > 8 ? 2 : 1;
    for (int pp = 0; pp < 3; pp++)
    {
        if (!(c->p.mode & 8))
            for (int yy = 0; yy < pheight[pp]; yy++)
                for (int xx = 0; xx < stride[pp] / bps; xx++)
                {
                    if (bps == 1) ((uint8_t *)plane[pp])[(size_t)yy * stride[pp] + xx] = pp ? (uint8_t)half : 0;
                    else ((uint16_t *)((uint8_t *)plane[pp] + (size_t)yy * stride[pp]))[xx] = pp ? (uint16_t)half : 0;
                }
        if (pp != 0) continue;
        for (int yy = 0; yy < c->height; yy++)
            for (int xx = 0; xx < c->width; xx++)
            {
                const int mv = m[(size_t)yy * st + xx];
                if (mv != 1 && mv != 128) continue;
                const int v = mv == 1 ? max : half;
                if (bps == 1) ((uint8_t *)plane[0])[(size_t)yy * stride[0] + xx] = (uint8_t)v;
                else ((uint16_t *)((uint8_t *)plane[0] + (size_t)yy * stride[0]))[xx] = (uint16_t)v;
            }
    }
}
