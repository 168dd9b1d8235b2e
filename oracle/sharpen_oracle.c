/* sharpen_oracle.c — CPU restatement of lapsharp, unsharp and chroma-smooth
 * (8-bit).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * lapsharp: /root/reference/libhb/lapsharp.c:37-86 (kernels), :125-182 (loop).
 * unsharp / chroma_smooth: the reference builds a separable binomial blur of
 * order 2*steps out of running pair sums (unsharp.c:128-156,
 * chroma_smooth.c:126-154) in uint32; the same value is the plain double sum of
 * C(2s,i)*C(2s,j)*pixel with edge-clamped coordinates, taken modulo 2^32, which
 * is what is written here.
 */
#include "oracle.h"

#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* ---- lapsharp ------------------------------------------------------------------ */
static const int k_lap[9]     = { 0, -1, 0, -1, 5, -1, 0, -1, 0 };
static const int k_isolap[9]  = { -1, -4, -1, -4, 25, -4, -1, -4, -1 };
static const int k_log[25]    = { 0, 0, -1, 0, 0,  0, -1, -2, -1, 0,  -1, -2, 21, -2, -1,
                                  0, -1, -2, -1, 0,  0, 0, -1, 0, 0 };
static const int k_isolog[25] = { 0, -1, -1, -1, 0,  -1, -3, -4, -3, -1,  -1, -4, 55, -4, -1,
                                  -1, -3, -4, -3, -1,  0, -1, -1, -1, 0 };
static const struct { const int *tap; int size; double coef; } k_tab[4] = {
    { k_lap, 3, 1.0 }, { k_isolap, 3, 1.0 / 5 }, { k_log, 5, 1.0 / 5 }, { k_isolog, 5, 1.0 / 15 } };

void orc_lapsharp_plane(const uint8_t *src, uint8_t *dst, int width, int height,
                        int src_stride, int dst_stride, double strength, int kernel)
{
    const int size = k_tab[kernel].size;
    const int *tap = k_tab[kernel].tap;
    const double coef = k_tab[kernel].coef;
    const int lo = -((size - 1) / 2), hi = (size + 1) / 2;
    const int sb = (src_stride - width) / 2;

    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
        {
            const uint8_t centre = src[(size_t)src_stride * y + x];
            if (y < hi || y > height - hi || x < sb + hi || x > width + sb - hi)
            {
                dst[(size_t)dst_stride * y + x] = centre;
                continue;
            }
            int16_t acc = 0;
            for (int dx = lo; dx < hi; dx++)
                for (int dy = lo; dy < hi; dy++)
                    acc += tap[(dy - lo) * size + dx - lo] * src[(ptrdiff_t)src_stride * (y + dy) + (x + dx)];
            acc = (int16_t)(((acc * coef) - centre) * strength) + centre;
            acc = acc < 0 ? 0 : acc;
            acc = acc > 255 ? 255 : acc;
            dst[(size_t)dst_stride * y + x] = (uint8_t)acc;
        }
}

/* ---- binomial blur shared by unsharp and chroma smooth -------------------------- */
static void binomial_row(uint32_t *coef, int order)
{
    coef[0] = 1;
    for (int n = 1; n <= order; n++)
    {
        coef[n] = 1;
        for (int k = n - 1; k >= 1; k--)
            coef[k] += coef[k - 1];
    }
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

static void blur_mix(const uint8_t *src, uint8_t *dst, int width, int height,
                     int src_stride, int dst_stride, double strength, int size,
                     int sign, int vmin, int vmax)
{
    const int amount = strength * 65536.0;                 /* unsharp.c:258 */
    if (!amount)
    {
        for (int y = 0; y < height; y++)
            memcpy(dst + (size_t)y * dst_stride, src + (size_t)y * src_stride,
                   src_stride < dst_stride ? src_stride : dst_stride);
        return;
    }
    const int steps = size / 2;
    const int scalebits = steps * 4;
    const int32_t halfscale = 1 << (scalebits - 1);
    uint32_t coef[16];
    binomial_row(coef, 2 * steps);

    uint32_t *hrow = malloc(sizeof(uint32_t) * (size_t)width * height);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
        {
            uint32_t s = 0;
            for (int i = 0; i <= 2 * steps; i++)
                s += coef[i] * src[(size_t)y * src_stride + clampi(x - steps + i, 0, width - 1)];
            hrow[(size_t)y * width + x] = s;
        }
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
        {
            uint32_t t = 0;
            for (int j = 0; j <= 2 * steps; j++)
                t += coef[j] * hrow[(size_t)clampi(y - steps + j, 0, height - 1) * width + x];
            const int32_t p = src[(size_t)y * src_stride + x];
            const int32_t d = ((p - (int32_t)((t + halfscale) >> scalebits)) * amount) >> 16;
            const int32_t res = sign > 0 ? p + d : p - d;
            dst[(size_t)y * dst_stride + x] = res > vmax ? vmax : res < vmin ? vmin : (uint8_t)res;
        }
    free(hrow);
}

static int sane_size(int size)
{
    if (size % 2 == 0) size--;
    if (size < 3) size = 3;
    if (size > 15) size = 15;
    return size;
}

void orc_unsharp_plane(const uint8_t *src, uint8_t *dst, int width, int height,
                       int src_stride, int dst_stride, double strength, int size)
{
    blur_mix(src, dst, width, height, src_stride, dst_stride, strength, sane_size(size), +1, 0, 255);
}

void orc_chroma_smooth_plane(const uint8_t *src, uint8_t *dst, int width, int height,
                             int src_stride, int dst_stride, double strength, int size)
{
    blur_mix(src, dst, width, height, src_stride, dst_stride, strength, sane_size(size), -1, 16, 240);
}


/* ---- 16-bit samples (the _16 instantiations: lapsharp.c:184, unsharp.c:171, chroma_smooth.c:170).
 * Strides are in samples; depth 10 or 12. ------------------------------------------------------- */
void orc_lapsharp_plane16(const uint16_t *src, uint16_t *dst, int width, int height,
                          int src_stride, int dst_stride, double strength, int kernel, int depth)
{
    const int size = k_tab[kernel].size;
    const int *tap = k_tab[kernel].tap;
    const double coef = k_tab[kernel].coef;
    const int lo = -((size - 1) / 2), hi = (size + 1) / 2;
    const int sb = (src_stride - width) / 2;
    const int max_value = (1 << depth) - 1;

    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
        {
            const uint16_t centre = src[(size_t)src_stride * y + x];
            if (y < hi || y > height - hi || x < sb + hi || x > width + sb - hi)
            {
                dst[(size_t)dst_stride * y + x] = centre;
                continue;
            }
            int32_t acc = 0;
            for (int dx = lo; dx < hi; dx++)
                for (int dy = lo; dy < hi; dy++)
                    acc += tap[(dy - lo) * size + dx - lo] * src[(ptrdiff_t)src_stride * (y + dy) + (x + dx)];
            acc = (int32_t)(((acc * coef) - centre) * strength) + centre;
            acc = acc < 0 ? 0 : acc;
            acc = acc > max_value ? max_value : acc;
            dst[(size_t)dst_stride * y + x] = (uint16_t)acc;
        }
}

static void blur_mix16(const uint16_t *src, uint16_t *dst, int width, int height,
                       int src_stride, int dst_stride, double strength, int size,
                       int sign, int vmin, int vmax)
{
    const int amount = strength * 65536.0;
    if (!amount)
    {
        for (int y = 0; y < height; y++)
            memcpy(dst + (size_t)y * dst_stride, src + (size_t)y * src_stride,
                   sizeof(uint16_t) * (src_stride < dst_stride ? src_stride : dst_stride));
        return;
    }
    const int steps = size / 2;
    const int scalebits = steps * 4;
    const int32_t halfscale = 1 << (scalebits - 1);
    uint32_t coef[16];
    binomial_row(coef, 2 * steps);

    uint32_t *hrow = malloc(sizeof(uint32_t) * (size_t)width * height);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
        {
            uint32_t s = 0;
            for (int i = 0; i <= 2 * steps; i++)
                s += coef[i] * src[(size_t)y * src_stride + clampi(x - steps + i, 0, width - 1)];
            hrow[(size_t)y * width + x] = s;
        }
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
        {
            uint32_t t = 0;
            for (int j = 0; j <= 2 * steps; j++)
                t += coef[j] * hrow[(size_t)clampi(y - steps + j, 0, height - 1) * width + x];
            const int32_t p = src[(size_t)y * src_stride + x];
            const int32_t d = ((p - (int32_t)((t + halfscale) >> scalebits)) * amount) >> 16;
            const int32_t res = sign > 0 ? p + d : p - d;
            dst[(size_t)y * dst_stride + x] = res > vmax ? vmax : res < vmin ? vmin : (uint16_t)res;
        }
    free(hrow);
}

void orc_unsharp_plane16(const uint16_t *src, uint16_t *dst, int width, int height,
                         int src_stride, int dst_stride, double strength, int size, int depth)
{
    blur_mix16(src, dst, width, height, src_stride, dst_stride, strength, sane_size(size), +1, 0, (1 << depth) - 1);
}

void orc_chroma_smooth_plane16(const uint16_t *src, uint16_t *dst, int width, int height,
                               int src_stride, int dst_stride, double strength, int size, int depth)
{
    const int max = 1 << depth;                            /* chroma_smooth.c:233-235 */
    blur_mix16(src, dst, width, height, src_stride, dst_stride, strength, sane_size(size), -1, max / 16, max - max / 16);
}
