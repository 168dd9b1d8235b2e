/* motion_metric_oracle.c — CPU restatement of the frame-difference metric vfr.c uses to pick the
 * frame to drop (libhb/motion_metric.c: build_gamma_lut :36-42, approximate_frame_data :44-75,
 * sse_block16 :141-158, motion_metric :164-192, motion_metric_fast :197-219, the >= 1920 x 1080
 * switch to the fast form :245-259).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Pinned: tests/test_oracle_vs_ref.py-style comparison with the reference's own hb_motion_metric
 * object compiled in place (oracle/ref_wrap/wrap_motion_metric.c), tests/test_motion_metric_cpu.py.
 *
 * Luma only.  Samples go through a 2.2-gamma table scaled to 4095, squared differences are summed
 * per 16 x 16 block in 32 bits (which can wrap for a block of extreme differences - kept), blocks in
 * 64 bits, the total is divided by width * height as float.  Pictures >= 1920 wide or >= 1080 high
 * are first reduced 4 x 4 -> 1 with a tree of rounded pair averages.
 * Kept from the reference: above 8 bits the reduced pictures are written with a stride of `width`
 * samples (:207-210) but motion_metric_16 divides the stride it is given by the sample size again
 * (:176-177), so it walks them with HALF that stride - sample (x, y) of the comparison is element
 * y * (width / 2) + x of the reduced picture, i.e. mostly the wrong row.  Deterministic and in
 * bounds, so it is part of the result.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>

static inline unsigned avg2(unsigned a, unsigned b) { return (a + b + 1) >> 1; }
static inline unsigned avg4(unsigned a, unsigned b, unsigned c, unsigned d) { return avg2(avg2(a, b), avg2(c, d)); }

static inline unsigned px(const void *plane, int stride, int x, int y, int bps)
{
    const uint8_t *row = (const uint8_t *)plane + (size_t)y * stride;
    return bps == 1 ? row[x] : ((const uint16_t *)row)[x];
}

/* one sample of the 4x-reduced picture (:44-75): quadrants are averaged (left column pair, right
 * column pair), then the four quadrant values the same way */
static unsigned reduced(const void *plane, int stride, int x, int y, int bps)
{
    unsigned q[4];
    for (int k = 0; k < 4; k++)
    {
        const int sx = 4 * x + 2 * (k & 1), sy = 4 * y + 2 * (k >> 1);
        q[k] = avg4(px(plane, stride, sx, sy, bps), px(plane, stride, sx, sy + 1, bps),
                    px(plane, stride, sx + 1, sy, bps), px(plane, stride, sx + 1, sy + 1, bps));
    }
    return avg4(q[0], q[1], q[2], q[3]);
}

void orc_motion_gamma_lut(unsigned *lut, int depth)
{
    const int max_value = (1 << depth) - 1;
    for (int i = 0; i <= max_value; i++)
        lut[i] = 4095 * pow(((float)i / (float)(max_value - 1)), 2.2f);      /* :40, double pow of float arguments */
}

float orc_motion_metric(const void *a, int stride_a, const void *b, int stride_b, int width, int height, int depth)
{
    const int bps = depth > 8 ? 2 : 1;
    const int fast = width >= 1920 || height >= 1080;
    unsigned *lut = malloc(sizeof(unsigned) << depth);
    orc_motion_gamma_lut(lut, depth);
    const int w = fast ? width / 4 : width, h = fast ? height / 4 : height;
    uint64_t sum = 0;
    for (int by = 0; by < h / 16; by++)
        for (int bx = 0; bx < w / 16; bx++)
        {
            unsigned block = 0;                                               /* 32-bit, as sse_block16 returns */
            for (int y = 16 * by; y < 16 * by + 16; y++)
                for (int x = 16 * bx; x < 16 * bx + 16; x++)
                {
                    int rx = x, ry = y;
                    if (fast && bps == 2)
                    {
                        const int at = y * (w / 2) + x;                       /* the halved stride, see above */
                        rx = at % w;
                        ry = at / w;
                    }
                    const unsigned va = fast ? reduced(a, stride_a, rx, ry, bps) : px(a, stride_a, x, y, bps);
                    const unsigned vb = fast ? reduced(b, stride_b, rx, ry, bps) : px(b, stride_b, x, y, bps);
                    const int diff = (int)(lut[va] - lut[vb]);
                    block += (unsigned)(diff * diff);
                }
            sum += block;
        }
    free(lut);
    return (float)sum / (w * h);
}
