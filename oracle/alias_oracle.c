/* alias_oracle.c — CPU restatement of the libavfilter/zimg-backed alias filters:
 * rotate (transpose/hflip/vflip), grayscale (monochrome), crop+scale (crop, zscale
 * lanczos).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 *                    ***  PARITY UNPINNED  ***
 * The arithmetic of these filters lives in FFmpeg 9.0.1 / zimg snapshot-20250624,
 * which are NOT part of /root/reference (contrib/{ffmpeg,zimg}/module.defs fetch
 * tarballs at build time) and the reference has no test or golden vector for them.
 * This file restates their published algorithms from memory:
 *   - transpose dir: cclock_flip out(x,y)=in(y,x); clock out(x,y)=in(y,h-1-x);
 *     cclock out(x,y)=in(w-1-y,x); clock_flip out(x,y)=in(w-1-y,h-1-x)
 *     [in(col,row)]; hflip / vflip mirror columns / rows.  Pure index permutations.
 *   - monochrome (vf_monochrome.c): per luma sample, with chroma (u,v) centred on 0:
 *       ny = exp(-clip(((b-u)^2 + (r-v)^2) / size, 0, 1)),  b = cb/2, r = cr/2
 *       tt = envelope(y)  (beta 0.6 smoothstep pair), t = tt + (1-tt)*(1-high)
 *       y' = (1-t)*y + t*ny*y ; output lrintf(y'*255) clipped; chroma := 128.
 *     The same formulas appear in the reference's own Metal shader
 *     (libhb/platform/macosx/shaders/grayscale_vt.metal:68-112).
 *   - zscale lanczos (zimg resize): separable 3-lobe Lanczos, support stretched by
 *     1/scale when shrinking, windows centred with round-half-up, taps that fall
 *     outside the picture reflected back in (edge sample repeated), weights
 *     normalised per output sample; left-sited 4:2:0 chroma gets the horizontal
 *     shift 0.25*(1 - src/dst).  zimg itself filters in 16-bit fixed point, horizontal
 *     pass first, each pass stored clamped to the sample range; this restatement keeps
 *     the order and the clamp between the passes but uses double and rounds half up once
 *     at the end, so it can differ from real zimg by 1 LSB (orc_cropscale_plane_fx is
 *     the fixed-point form for 8-bit planes).
 * The HIP path is tested bit-for-bit against THIS file (tables are built by the
 * same formulas with host libm), never against FFmpeg/zimg.
 */
#include "oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* ---- rotate -------------------------------------------------------------------- */
enum { T_NONE, T_CCLOCK_FLIP, T_CLOCK, T_CCLOCK, T_CLOCK_FLIP };

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

void orc_rotate_plane(const uint8_t *src, int sw, int sh, int sstride,
                      uint8_t *dst, int dstride, int angle, int flip)
{
    orc_rotate_plane_d(src, sw, sh, sstride, dst, dstride, angle, flip, 1);
}

/* bps = bytes per sample (1, or 2 for 10 / 12-bit); strides in bytes */
void orc_rotate_plane_d(const void *src, int sw, int sh, int sstride,
                        void *dst, int dstride, int angle, int flip, int bps)
{
    /* rotate.c:190-215: 0 -> hflip if asked; 90 -> clock[_flip]; 180 -> vflip (+hflip
     * unless asked); 270 -> cclock[_flip] */
    int trans = T_NONE, hflip = 0, vflip = 0;
    switch (angle)
    {
        case 0:   hflip = flip; break;
        case 90:  trans = flip ? T_CLOCK_FLIP : T_CLOCK; break;
        case 180: vflip = 1; hflip = !flip; break;
        case 270: trans = flip ? T_CCLOCK_FLIP : T_CCLOCK; break;
    }
    if (trans != T_NONE)
    {
        const int dw = sh, dh = sw;
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++)
            {
                int sx, sy;                     /* source column, row */
                switch (trans)
                {
                    case T_CCLOCK_FLIP: sx = y;          sy = x;          break;
                    case T_CLOCK:       sx = y;          sy = sh - 1 - x; break;
                    case T_CCLOCK:      sx = sw - 1 - y; sy = x;          break;
                    default:            sx = sw - 1 - y; sy = sh - 1 - x; break;
                }
                setpx(dst, dstride, x, y, bps, getpx(src, sstride, sx, sy, bps));
            }
        return;
    }
    for (int y = 0; y < sh; y++)
        for (int x = 0; x < sw; x++)
            setpx(dst, dstride, x, y, bps, getpx(src, sstride, hflip ? sw - 1 - x : x, vflip ? sh - 1 - y : y, bps));
}

/* ---- monochrome ------------------------------------------------------------------ */
static float envelope(const float x)
{
    const float beta = 0.6f;
    if (x < beta)
    {
        const float tmp = fabsf(x / beta - 1.f);
        return 1.f - tmp * tmp;
    }
    const float tmp = (1.f - x) / (1.f - beta);
    return tmp * tmp * (3.f - 2.f * tmp);
}

static float chroma_weight(float b, float r, float u, float v, float size)
{
    float d = ((b - u) * (b - u) + (r - v) * (r - v)) * size;
    d = d < 0.f ? 0.f : d > 1.f ? 1.f : d;
    return expf(-d);
}

void orc_monochrome_luma(const uint8_t *yp, int ystride, const uint8_t *up, const uint8_t *vp, int cstride,
                         uint8_t *dst, int dstride, int w, int h, int subw, int subh,
                         double cb, double cr, double size, double high)
{
    orc_monochrome_luma_d(yp, ystride, up, vp, cstride, dst, dstride, w, h, subw, subh, cb, cr, size, high, 8);
}

/* vf_monochrome.c's 16-bit slice functions are the 8-bit ones with 255 replaced by (1 << depth) - 1 */
void orc_monochrome_luma_d(const void *yp, int ystride, const void *up, const void *vp, int cstride,
                           void *dst, int dstride, int w, int h, int subw, int subh,
                           double cb, double cr, double size, double high, int depth)
{
    const int bps = depth > 8 ? 2 : 1, max = (1 << depth) - 1;
    const float imax = 1.f / max;
    const float ihigh = 1.f - (float)high;
    const float isize = 1.f / (float)size;
    const float b = (float)cb * .5f, r = (float)cr * .5f;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            const int cx = x >> subw, cy = y >> subh;
            const float fy = getpx(yp, ystride, x, y, bps) * imax;
            const float fu = getpx(up, cstride, cx, cy, bps) * imax - .5f;
            const float fv = getpx(vp, cstride, cx, cy, bps) * imax - .5f;
            float ny = chroma_weight(b, r, fu, fv, isize);
            const float tt = envelope(fy);
            const float t = tt + (1.f - tt) * ihigh;
            ny = (1.f - t) * fy + t * ny * fy;
            long q = lrintf(ny * max);
            setpx(dst, dstride, x, y, bps, q < 0 ? 0 : q > max ? max : (unsigned)q);
        }
}

/* ---- lanczos resize ---------------------------------------------------------------- */
static double lanczos3(double x)
{
    const double pi = 3.14159265358979323846;
    x = fabs(x);
    if (x >= 3.0) return 0.0;
    if (x == 0.0) return 1.0;
    const double a = x * pi;
    return (sin(a) / a) * (sin(a / 3.0) / (a / 3.0));
}

int orc_lanczos_table(int src_dim, int dst_dim, double shift, int *idx, double *coef)
{
    const double scale = (double)dst_dim / (double)src_dim;
    const double step = scale < 1.0 ? scale : 1.0;
    const double support = 3.0 / step;
    int taps = (int)ceil(support) * 2;
    if (taps < 1) taps = 1;
    if (taps > 64) taps = 64;
    for (int i = 0; i < dst_dim; i++)
    {
        const double pos = (i + 0.5) / scale + shift;              /* in source sample-edge coordinates */
        const double begin = floor(pos - taps / 2.0 + 0.5);        /* round half up */
        double w[64], total = 0.0;
        for (int k = 0; k < taps; k++)
        {
            w[k] = lanczos3((begin + k + 0.5 - pos) * step);
            total += w[k];
        }
        for (int k = 0; k < taps; k++)
        {
            long j = (long)begin + k;
            if (j < 0) j = -j - 1;                                  /* reflect, edge sample repeated */
            if (j >= src_dim) j = 2L * src_dim - 1 - j;
            if (j < 0) j = 0;
            if (j >= src_dim) j = src_dim - 1;
            idx[(size_t)i * taps + k] = (int)j;
            coef[(size_t)i * taps + k] = w[k] / total;
        }
    }
    return taps;
}

void orc_cropscale_plane(const uint8_t *src, int sstride, int crop_x, int crop_y, int crop_w, int crop_h,
                         uint8_t *dst, int dstride, int dw, int dh, double shift_x, double shift_y)
{
    orc_cropscale_plane_d(src, sstride, crop_x, crop_y, crop_w, crop_h, dst, dstride, dw, dh, shift_x, shift_y, 8);
}

/* the same for `depth`-bit samples (uint16 above 8): only the clip limit changes */
void orc_cropscale_plane_d(const void *src, int sstride, int crop_x, int crop_y, int crop_w, int crop_h,
                           void *dst, int dstride, int dw, int dh, double shift_x, double shift_y, int depth)
{
    const int bps = depth > 8 ? 2 : 1;
    const double vmax = (double)((1 << depth) - 1);
    /* range of the value between the passes: an 8-bit plane is resized as a 16-bit one (65535 = 255.996 * 256) */
    const double hmax = depth == 8 ? 65535.0 / 256.0 : vmax;
    const uint8_t *win = (const uint8_t *)src + (size_t)crop_y * sstride + (size_t)crop_x * bps;
    if (dw == crop_w && dh == crop_h && shift_x == 0.0 && shift_y == 0.0)
    {
        for (int y = 0; y < dh; y++)
            memcpy((uint8_t *)dst + (size_t)y * dstride, win + (size_t)y * sstride, (size_t)dw * bps);
        return;
    }
    int *ix = malloc(sizeof(int) * (size_t)dw * 64), *iy = malloc(sizeof(int) * (size_t)dh * 64);
    double *cx = malloc(sizeof(double) * (size_t)dw * 64), *cy = malloc(sizeof(double) * (size_t)dh * 64);
    const int tx = orc_lanczos_table(crop_w, dw, shift_x, ix, cx);
    const int ty = orc_lanczos_table(crop_h, dh, shift_y, iy, cy);
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++)
        {
            double acc = 0.0;
            for (int j = 0; j < ty; j++)
            {
                const int r = iy[(size_t)y * ty + j];
                double h = 0.0;
                for (int i = 0; i < tx; i++)
                    h += cx[(size_t)x * tx + i] * (double)getpx(win, sstride, ix[(size_t)x * tx + i], r, bps);
                h = h < 0.0 ? 0.0 : h > hmax ? hmax : h;       /* zimg stores the horizontal pass clamped to the sample range */
                acc += cy[(size_t)y * ty + j] * h;
            }
            acc = acc < 0.0 ? 0.0 : acc > vmax ? vmax : acc;
            setpx(dst, dstride, x, y, bps, (unsigned)(int)(acc + 0.5));
        }
    free(ix); free(iy); free(cx); free(cy);
}

/* ---- the same resize in zimg's own arithmetic for 8-bit planes ----------------------------------------
 * zimg has no 8-bit resize kernel: a BYTE plane is first widened to a 16-bit WORD plane (limited-range integer
 * to integer depth conversion = left shift by 8), resized in 16-bit fixed point, and brought back to 8 bits by
 * the depth converter (dither=none: round half up of v / 256).  The WORD resize kernels (resize_impl.cpp,
 * resize_line_h_u16_c / resize_line_v_u16_c, from memory) are
 *     accum = sum_k coeff_i16[k] * (src[k] + INT16_MIN);
 *     dst   = clamp(((accum + (1 << 13)) >> 14) - INT16_MIN, 0, pixel_max)
 * with the filter row quantised to 14 fractional bits such that it still sums to exactly 1 << 14 (filter.cpp:
 * the rounding residue is folded back into the row; here: into its largest tap).  Because a row sums to 1 << 14
 * the bias cancels: dst = clamp((sum_k c[k] * src[k] + 8192) >> 14), floor division.  Horizontal pass first
 * (the cheaper order for an upscale, which is what zimg's cost estimate picks), its result rounded to 16 bits,
 * then the vertical pass.  For an 8-bit source sample v the widened value is 256 v, so the horizontal pass is
 * (sum_k c[k] v[k] + 32) >> 6.
 * PARITY UNPINNED like the double form above (zimg is not in the reference tree); it is the arithmetic the HIP
 * kernel runs, tests/test_alias_cpu.py holds it within 1 LSB of the double form and of Pillow's Lanczos. */
void orc_quantize_taps(const double *coef, int taps, int16_t *q)
{
    int sum = 0, big = 0;
    for (int k = 0; k < taps; k++)
    {
        const long v = lrint(coef[k] * 16384.0);
        q[k] = (int16_t)v;
        sum += (int)v;
        if (fabs(coef[k]) > fabs(coef[big])) big = k;
    }
    q[big] = (int16_t)(q[big] + (16384 - sum));
}

static inline int clamp_u16(int v) { return v < 0 ? 0 : v > 65535 ? 65535 : v; }

void orc_cropscale_plane_fx(const uint8_t *src, int sstride, int crop_x, int crop_y, int crop_w, int crop_h,
                            uint8_t *dst, int dstride, int dw, int dh, double shift_x, double shift_y)
{
    const uint8_t *win = src + (size_t)crop_y * sstride + crop_x;
    if (dw == crop_w && dh == crop_h && shift_x == 0.0 && shift_y == 0.0)
    {
        for (int y = 0; y < dh; y++)
            memcpy(dst + (size_t)y * dstride, win + (size_t)y * sstride, (size_t)dw);
        return;
    }
    int *ix = malloc(sizeof(int) * (size_t)dw * 64), *iy = malloc(sizeof(int) * (size_t)dh * 64);
    double *cx = malloc(sizeof(double) * (size_t)dw * 64), *cy = malloc(sizeof(double) * (size_t)dh * 64);
    const int tx = orc_lanczos_table(crop_w, dw, shift_x, ix, cx);
    const int ty = orc_lanczos_table(crop_h, dh, shift_y, iy, cy);
    int16_t *qx = malloc(sizeof(int16_t) * (size_t)dw * tx), *qy = malloc(sizeof(int16_t) * (size_t)dh * ty);
    for (int x = 0; x < dw; x++) orc_quantize_taps(cx + (size_t)x * tx, tx, qx + (size_t)x * tx);
    for (int y = 0; y < dh; y++) orc_quantize_taps(cy + (size_t)y * ty, ty, qy + (size_t)y * ty);
    uint16_t *hbuf = malloc(sizeof(uint16_t) * (size_t)dw * crop_h);
    for (int r = 0; r < crop_h; r++)
    {
        const uint8_t *row = win + (size_t)r * sstride;
        for (int x = 0; x < dw; x++)
        {
            int s = 0;
            for (int i = 0; i < tx; i++) s += (int)qx[(size_t)x * tx + i] * (int)row[ix[(size_t)x * tx + i]];
            hbuf[(size_t)r * dw + x] = (uint16_t)clamp_u16((s + 32) >> 6);         /* = (256 s + 8192) >> 14 */
        }
    }
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++)
        {
            int acc = 0;
            for (int j = 0; j < ty; j++)
                acc += (int)qy[(size_t)y * ty + j] * ((int)hbuf[(size_t)iy[(size_t)y * ty + j] * dw + x] - 32768);
            const int v16 = clamp_u16(((acc + 8192) >> 14) + 32768);
            const int v8 = (v16 + 128) >> 8;
            dst[(size_t)y * dstride + x] = (uint8_t)(v8 > 255 ? 255 : v8);
        }
    free(ix); free(iy); free(cx); free(cy); free(qx); free(qy); free(hbuf);
}

/* ---- the swscale branch of crop/scale (cropscale.c:159-165) ---------------------------------------------------------
 * When hb_av_can_use_zscale() says no - an odd width or height on either side, or a build without AVX2
 * (hbffmpeg.c:870-915) - crop_scale_init builds `scale=w:h:flags=lanczos+accurate_rnd` instead of zscale: libswscale.
 * PARITY UNPINNED, like the zimg form above and for the same reason (libswscale is not in the reference tree; FFmpeg
 * 9.0.1 per contrib/ffmpeg/module.defs).  What follows restates libswscale's published algorithm for planar 8-bit YUV:
 *   * the filter of one dimension as libswscale/utils.c:initFilter builds it: positions in 16.16 fixed point
 *     (xInc = (src << 16 + dst / 2) / dst; the first tap of output i at (xDstInSrc - (size - 2) << 16) >> 17 with
 *     xDstInSrc = dstPos * xInc >> 7 - srcPos << 16 >> 7, advancing by 2 xInc), Lanczos-3 weights evaluated at the
 *     tap distances as 64-bit integers scaled by 2^54, near-zero taps at either end dropped (SWS_MAX_REDUCE_CUTOFF =
 *     0.002), taps outside the plane folded onto the edge sample, then normalised to `one` with the rounding error carried
 *     from tap to tap;
 *   * horizontal pass (hScale8To15_c): 14-bit coefficients, the sum >> 7 into a 15-bit plane, clamped at 32767;
 *   * vertical pass (yuv2planeX_8_c): 12-bit coefficients, (64 << 12) + sum >> 19, clamped to 8 bits - the C paths'
 *     arithmetic, which `accurate_rnd` makes the SIMD paths reproduce;
 *   * chroma positions as vf_scale hands them over for left-sited 4:2:0 (horizontal position 0, vertical 128): srcPos =
 *     dstPos = 64 horizontally, 128 everywhere else.
 * Held within 2 code values of the zimg form and of Pillow's Lanczos (tests/test_alias_cpu.py); the HIP scaler runs it bit
 * for bit (tests/test_alias_gpu.py). */
#define SWS_FONE_SHIFT 54
static int sws_log2(unsigned v) { int n = 0; while (v >>= 1) n++; return n; }

/* initFilter for SWS_LANCZOS (default parameter 3): pos[dst], coef[dst * size] (caller frees); returns size */
int orc_sws_filter(int src, int dst, int one, int src_pos, int dst_pos, int **out_pos, int16_t **out_coef)
{
    const int64_t fone = 1LL << (SWS_FONE_SHIFT - (sws_log2((unsigned)(src / dst)) < 8 ? sws_log2((unsigned)(src / dst)) : 8));
    const int x_inc = (int)((((int64_t)src << 16) + (dst >> 1)) / dst);
    int *pos = malloc(sizeof(int) * (size_t)dst);
    int size;
    int64_t *f;
    if (llabs((long long)x_inc - 0x10000) < 10 && src_pos == dst_pos)
    {
        size = 1;                                                   /* unscaled: the sample itself */
        f = malloc(sizeof(int64_t) * (size_t)dst);
        for (int i = 0; i < dst; i++) { f[i] = fone; pos[i] = i; }
    }
    else
    {
        const int size_factor = 6;                                  /* lanczos, default parameter */
        size = x_inc <= (1 << 16) ? 1 + size_factor : 1 + (size_factor * src + dst - 1) / dst;
        if (size > src - 2) size = src - 2;
        if (size < 1) size = 1;
        f = malloc(sizeof(int64_t) * (size_t)dst * (size_t)size);
        int64_t x_dst_in_src = (((int64_t)dst_pos * x_inc) >> 7) - (((int64_t)src_pos * 0x10000LL) >> 7);
        for (int i = 0; i < dst; i++)
        {
            int xx = (int)((x_dst_in_src - (int64_t)(size - 2) * (1LL << 16)) / (1 << 17));
            pos[i] = xx;
            for (int j = 0; j < size; j++)
            {
                int64_t d = llabs(((int64_t)xx * (1 << 17)) - x_dst_in_src) << 13;
                if (x_inc > (1 << 16)) d = d * dst / src;
                const double fd = (double)d * (1.0 / (1 << 30));
                int64_t coeff = (int64_t)((d ? sin(fd * M_PI) * sin(fd * M_PI / 3.0) / (fd * fd * M_PI * M_PI / 3.0) : 1.0) * (double)fone);
                if (fd > 3.0) coeff = 0;
                f[(size_t)i * size + j] = coeff;
                xx++;
            }
            x_dst_in_src += 2 * (int64_t)x_inc;
        }
    }
    /* drop near-zero taps: shift each row left past them, find how many taps any row still needs */
    int min_size = 0;
    const double cut = 0.002 * (double)fone;
    for (int i = dst - 1; i >= 0; i--)
    {
        int64_t *row = f + (size_t)i * size;
        int min = size;
        int64_t cut_off = 0;
        for (int j = 0; j < size; j++)
        {
            cut_off += llabs(row[0]);
            if ((double)cut_off > cut) break;
            if (i < dst - 1 && pos[i] >= pos[i + 1]) break;         /* positions stay monotonic */
            for (int k = 1; k < size; k++) row[k - 1] = row[k];
            row[size - 1] = 0;
            pos[i]++;
        }
        cut_off = 0;
        for (int j = size - 1; j > 0; j--)
        {
            cut_off += llabs(row[j]);
            if ((double)cut_off > cut) break;
            min--;
        }
        if (min > min_size) min_size = min;
    }
    const int fsize = min_size;                                     /* (filterAlign only appends zero taps) */
    int64_t *g = malloc(sizeof(int64_t) * (size_t)dst * (size_t)fsize);
    for (int i = 0; i < dst; i++)
        for (int j = 0; j < fsize; j++) g[(size_t)i * fsize + j] = j < size ? f[(size_t)i * size + j] : 0;
    free(f);
    /* taps outside the plane fold onto the edge sample */
    for (int i = 0; i < dst; i++)
    {
        int64_t *row = g + (size_t)i * fsize;
        if (pos[i] < 0)
        {
            for (int j = 1; j < fsize; j++)
            {
                const int left = j + pos[i] > 0 ? j + pos[i] : 0;
                row[left] += row[j];
                row[j] = 0;
            }
            pos[i] = 0;
        }
        if (pos[i] + fsize > src)
        {
            const int shift = pos[i] + (fsize - src < 0 ? fsize - src : 0);
            int64_t acc = 0;
            for (int j = fsize - 1; j >= 0; j--)
                if (pos[i] + j >= src) { acc += row[j]; row[j] = 0; }
            for (int j = fsize - 1; j >= 0; j--)
                row[j] = j < shift ? 0 : row[j - shift];
            pos[i] -= shift;
            row[src - 1 - pos[i]] += acc;
        }
    }
    /* normalise to `one`, the rounding error carried along the row */
    int16_t *coef = malloc(sizeof(int16_t) * (size_t)dst * (size_t)fsize);
    for (int i = 0; i < dst; i++)
    {
        const int64_t *row = g + (size_t)i * fsize;
        int64_t error = 0, sum = 0;
        for (int j = 0; j < fsize; j++) sum += row[j];
        sum = (sum + one / 2) / one;
        if (!sum) sum = 1;
        for (int j = 0; j < fsize; j++)
        {
            const int64_t v = row[j] + error;
            const int64_t iv = v >= 0 ? (v + (sum >> 1)) / sum : -((-v + (sum >> 1)) / sum);   /* ROUNDED_DIV */
            coef[(size_t)i * fsize + j] = (int16_t)iv;
            error = v - iv * sum;
        }
    }
    free(g);
    *out_pos = pos;
    *out_coef = coef;
    return fsize;
}

/* one plane; chroma_h: the plane is a horizontally subsampled, left-sited chroma plane */
void orc_cropscale_plane_sws(const uint8_t *src, int sstride, int crop_x, int crop_y, int crop_w, int crop_h,
                             uint8_t *dst, int dstride, int dw, int dh, int chroma_h)
{
    const uint8_t *win = src + (size_t)crop_y * sstride + crop_x;
    int *px, *py;
    int16_t *qx, *qy;
    const int hpos = chroma_h ? 64 : 128;
    const int tx = orc_sws_filter(crop_w, dw, 1 << 14, hpos, hpos, &px, &qx);
    const int ty = orc_sws_filter(crop_h, dh, 1 << 12, 128, 128, &py, &qy);
    int16_t *hbuf = malloc(sizeof(int16_t) * (size_t)dw * crop_h);
    for (int r = 0; r < crop_h; r++)
    {
        const uint8_t *row = win + (size_t)r * sstride;
        for (int x = 0; x < dw; x++)
        {
            int val = 0;
            for (int j = 0; j < tx; j++) val += (int)row[px[x] + j] * qx[(size_t)x * tx + j];
            val >>= 7;
            hbuf[(size_t)r * dw + x] = (int16_t)(val < (1 << 15) - 1 ? val : (1 << 15) - 1);
        }
    }
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++)
        {
            int val = 64 << 12;
            for (int j = 0; j < ty; j++) val += (int)hbuf[(size_t)(py[y] + j) * dw + x] * qy[(size_t)y * ty + j];
            val >>= 19;
            dst[(size_t)y * dstride + x] = (uint8_t)(val < 0 ? 0 : val > 255 ? 255 : val);
        }
    free(px); free(py); free(qx); free(qy); free(hbuf);
}

/* The swscale branch for 10 / 12-bit planes (libswscale: srcBpc > 8, dstBpc <= 14 -> hScale16To15_c, then the
 * yuv2planeX_10 / _12 template).  Same filter tables; the 16-bit samples come down to the same 15-bit plane between the
 * passes (>> depth - 1 instead of >> 7), the vertical pass rounds with half of its shift (no dither above 8 bits):
 *     h   = min(sum_k q14[k] * src[k] >> (depth - 1), 32767)
 *     out = clip_uintp2(((1 << (26 - depth)) + sum_k q12[k] * h[k]) >> (27 - depth), depth)
 * PARITY UNPINNED like the 8-bit form.  sstride / dstride in bytes. */
void orc_cropscale_plane_sws16(const uint16_t *src, int sstride, int crop_x, int crop_y, int crop_w, int crop_h,
                               uint16_t *dst, int dstride, int dw, int dh, int chroma_h, int depth)
{
    const uint8_t *win = (const uint8_t *)src + (size_t)crop_y * sstride + (size_t)crop_x * 2;
    int *px, *py;
    int16_t *qx, *qy;
    const int hpos = chroma_h ? 64 : 128;
    const int tx = orc_sws_filter(crop_w, dw, 1 << 14, hpos, hpos, &px, &qx);
    const int ty = orc_sws_filter(crop_h, dh, 1 << 12, 128, 128, &py, &qy);
    const int sh = depth - 1, shift = 11 + 16 - depth, vmax = (1 << depth) - 1;
    int16_t *hbuf = malloc(sizeof(int16_t) * (size_t)dw * crop_h);
    for (int r = 0; r < crop_h; r++)
    {
        const uint16_t *row = (const uint16_t *)(win + (size_t)r * sstride);
        for (int x = 0; x < dw; x++)
        {
            int val = 0;
            for (int j = 0; j < tx; j++) val += (int)row[px[x] + j] * qx[(size_t)x * tx + j];
            val >>= sh;
            hbuf[(size_t)r * dw + x] = (int16_t)(val < (1 << 15) - 1 ? val : (1 << 15) - 1);
        }
    }
    for (int y = 0; y < dh; y++)
    {
        uint16_t *out = (uint16_t *)((uint8_t *)dst + (size_t)y * dstride);
        for (int x = 0; x < dw; x++)
        {
            int val = 1 << (shift - 1);
            for (int j = 0; j < ty; j++) val += (int)hbuf[(size_t)(py[y] + j) * dw + x] * qy[(size_t)y * ty + j];
            val >>= shift;
            out[x] = (uint16_t)(val < 0 ? 0 : val > vmax ? vmax : val);
        }
    }
    free(px); free(py); free(qx); free(qy); free(hbuf);
}

/* The same for 10 / 12-bit planes (uint16 samples).  zimg resizes a WORD plane at its own depth: per pass
 *     dst = clamp((sum_k c[k] * src[k] + (1 << 13)) >> 14, 0, (1 << depth) - 1)
 * (it holds the samples biased by -32768 to use signed 16-bit multiplies; a filter row sums to exactly 1 << 14, so
 * the bias is a multiple of 1 << 14 in the sum and comes out unchanged: the plain form above is the same number).
 * Horizontal pass first, the plane between the passes clamped to the depth.  PARITY UNPINNED like the 8-bit form;
 * tests/test_alias_cpu.py holds it within 1 LSB of the double form. */
void orc_cropscale_plane_fx16(const uint16_t *src, int sstride, int crop_x, int crop_y, int crop_w, int crop_h,
                              uint16_t *dst, int dstride, int dw, int dh, double shift_x, double shift_y, int depth)
{
    const int vmax = (1 << depth) - 1;
    const uint8_t *win = (const uint8_t *)src + (size_t)crop_y * sstride + (size_t)crop_x * 2;
    if (dw == crop_w && dh == crop_h && shift_x == 0.0 && shift_y == 0.0)
    {
        for (int y = 0; y < dh; y++)
            memcpy((uint8_t *)dst + (size_t)y * dstride, win + (size_t)y * sstride, (size_t)dw * 2);
        return;
    }
    int *ix = malloc(sizeof(int) * (size_t)dw * 64), *iy = malloc(sizeof(int) * (size_t)dh * 64);
    double *cx = malloc(sizeof(double) * (size_t)dw * 64), *cy = malloc(sizeof(double) * (size_t)dh * 64);
    const int tx = orc_lanczos_table(crop_w, dw, shift_x, ix, cx);
    const int ty = orc_lanczos_table(crop_h, dh, shift_y, iy, cy);
    int16_t *qx = malloc(sizeof(int16_t) * (size_t)dw * tx), *qy = malloc(sizeof(int16_t) * (size_t)dh * ty);
    for (int x = 0; x < dw; x++) orc_quantize_taps(cx + (size_t)x * tx, tx, qx + (size_t)x * tx);
    for (int y = 0; y < dh; y++) orc_quantize_taps(cy + (size_t)y * ty, ty, qy + (size_t)y * ty);
    uint16_t *hbuf = malloc(sizeof(uint16_t) * (size_t)dw * crop_h);
    for (int r = 0; r < crop_h; r++)
    {
        const uint16_t *row = (const uint16_t *)(win + (size_t)r * sstride);
        for (int x = 0; x < dw; x++)
        {
            int s = 0;
            for (int i = 0; i < tx; i++) s += (int)qx[(size_t)x * tx + i] * (int)row[ix[(size_t)x * tx + i]];
            const int v = (s + 8192) >> 14;
            hbuf[(size_t)r * dw + x] = (uint16_t)(v < 0 ? 0 : v > vmax ? vmax : v);
        }
    }
    for (int y = 0; y < dh; y++)
    {
        uint16_t *drow = (uint16_t *)((uint8_t *)dst + (size_t)y * dstride);
        for (int x = 0; x < dw; x++)
        {
            int acc = 0;
            for (int j = 0; j < ty; j++)
                acc += (int)qy[(size_t)y * ty + j] * (int)hbuf[(size_t)iy[(size_t)y * ty + j] * dw + x];
            const int v = (acc + 8192) >> 14;
            drow[x] = (uint16_t)(v < 0 ? 0 : v > vmax ? vmax : v);
        }
    }
    free(ix); free(iy); free(cx); free(cy); free(qx); free(qy); free(hbuf);
}

/* ---- pad (libhb/pad.c:40-148 -> FFmpeg vf_pad.c + drawutils.c; parity unpinned) ----------------
 * vf_pad copies the input picture into a larger one at (x, y) - both rounded down to the chroma
 * subsampling - and fills the rest with one colour.  The colour is given as RGB and converted the
 * way drawutils.c:ff_draw_color does: Y'CbCr through the (Kr, Kb) of the frame's matrix (BT.601 when
 * unspecified), scaled to the frame's range (limited when unspecified), component = (unsigned)(v *
 * ((1 << depth) - 1) + 0.5) with v in [0, 1].
 * The reference's own restatement of the copy-or-fill step, platform/macosx/shaders/pad_vt.metal (a port of
 * FFmpeg's pad.cl): a destination sample at `pos` is the fill colour when pos.x < offset.x, pos.y < offset.y,
 * pos.x >= src_width + offset.x or pos.y >= src_height + offset.y, else the source sample at pos - offset
 * (:54-58, :72-73) - the test orc_pad_plane applies per plane with the offsets shifted by the plane's
 * subsampling; the shader receives the colour already converted (color_y / color_u / color_v, :19-27), which
 * is what orc_pad_color computes. */
void orc_pad_color(int rgb, int matrix, int full_range, int depth, int out[3])
{
    double kr = 0.299, kb = 0.114;                              /* smpte170m, also the fallback */
    switch (matrix)
    {
        case 1: kr = 0.2126; kb = 0.0722; break;
        case 4: kr = 0.30;   kb = 0.11;   break;
        case 7: kr = 0.212;  kb = 0.087;  break;
        case 9: case 10: kr = 0.2627; kb = 0.0593; break;
    }
    const double kg = 1.0 - kr - kb;
    const double r = ((rgb >> 16) & 0xff) / 255., g = ((rgb >> 8) & 0xff) / 255., b = (rgb & 0xff) / 255.;
    double v[3];
    v[0] = kr * r + kg * g + kb * b;
    v[1] = (-kr * r - kg * g + (1.0 - kb) * b) / (2.0 * (1.0 - kb));
    v[2] = ((1.0 - kr) * r - kg * g - kb * b) / (2.0 * (1.0 - kr));
    for (int i = 0; i < 3; i++)
    {
        const int chroma = i > 0;
        if (!full_range)
        {
            v[i] *= (chroma ? 224. : 219.) / 255.;
            v[i] += (chroma ? 128. : 16.) / 255.;
        }
        else if (chroma)
            v[i] += 0.5;
        out[i] = (int)(unsigned)(v[i] * ((1 << depth) - 1) + 0.5);
    }
}

/* one plane: (x, y) and the fill value are the plane's own (already shifted for chroma) */
void orc_pad_plane(const void *src, int sw, int sh, int sstride, void *dst, int dw, int dh, int dstride,
                   int x, int y, int fill, int bps)
{
    for (int yy = 0; yy < dh; yy++)
        for (int xx = 0; xx < dw; xx++)
        {
            const int inside = xx >= x && xx < x + sw && yy >= y && yy < y + sh;
            setpx(dst, dstride, xx, yy, bps, inside ? getpx(src, sstride, xx - x, yy - y, bps) : (unsigned)fill);
        }
}

/* ---- format=pix_fmts=... (libhb/format.c:13-111): depth conversion of planar YUV -------------------------
 * The reference only names the target format; libavfilter's format negotiation inserts a `scale` filter
 * and, sizes being equal, libswscale's unscaled planar copy (swscale_unscaled.c:planarCopyWrapper) does
 * the work.  libswscale is not in the reference tree: PARITY UNPINNED, restated from the published code:
 *   up   (8 -> 10/12, 10 -> 12): limited range, and chroma always, shift only: v << (dd - sd);
 *        full-range luma replicates the top bits: (v << (dd - sd)) | (v >> (2 sd - dd))
 *        (shiftonly = plane == 1 || plane == 2 || (!srcRange && plane == 0))
 *   down (10/12 -> 8, 12 -> 10): ordered dither, tmp = (v + dithers[shift - 1][y & 7][x & 7]) >> shift,
 *        out = tmp - (tmp >> dd)  (the clamp of the one value that can overflow) - the shiftonly form; full-range
 *        luma takes the macro's other arm (DITHER_COPY, !shiftonly), which folds the top code values down BEFORE the
 *        shift so that full scale maps onto full scale: out = (v - (v >> dd) + dither) >> shift.
 *        dithers[1] / dithers[3] (shift 2 / 4) are the 2x2 and 4x4 ordered matrices below.
 * Same chroma subsampling on both sides; anything else (chroma resampling, semi-planar) is out of scope. */
static const uint8_t k_dither2[2][2] = { { 1, 2 }, { 3, 0 } };
static const uint8_t k_dither4[4][4] = { { 4, 8, 7, 11 }, { 12, 0, 15, 3 }, { 6, 10, 5, 9 }, { 14, 2, 13, 1 } };

int orc_format_plane(const void *src, int sstride, int sdepth, void *dst, int dstride, int ddepth,
                     int w, int h, int plane, int full_range)
{
    const int sb = sdepth > 8 ? 2 : 1, db = ddepth > 8 ? 2 : 1;
    const int shiftonly = plane == 1 || plane == 2 || (!full_range && plane == 0);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            const unsigned v = sb == 1 ? ((const uint8_t *)src)[(size_t)y * sstride + x]
                                       : ((const uint16_t *)((const uint8_t *)src + (size_t)y * sstride))[x];
            unsigned o;
            if (ddepth == sdepth)
                o = v;
            else if (ddepth > sdepth)
                o = shiftonly ? v << (ddepth - sdepth) : (v << (ddepth - sdepth)) | (v >> (2 * sdepth - ddepth));
            else
            {
                const int shift = sdepth - ddepth;
                const unsigned d = shift == 2 ? k_dither2[y & 1][x & 1] : k_dither4[y & 3][x & 3];
                if (shift != 2 && shift != 4) return -1;
                if (shiftonly)
                {
                    const unsigned tmp = (v + d) >> shift;
                    o = tmp - (tmp >> ddepth);
                }
                else
                    o = (v - (v >> ddepth) + d) >> shift;
            }
            if (db == 1) ((uint8_t *)dst)[(size_t)y * dstride + x] = (uint8_t)o;
            else         ((uint16_t *)((uint8_t *)dst + (size_t)y * dstride))[x] = (uint16_t)o;
        }
    return 0;
}
