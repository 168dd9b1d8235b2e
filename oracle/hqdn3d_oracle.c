/* hqdn3d_oracle.c — CPU restatement of libhb's hqdn3d denoiser (8-bit).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows /root/reference/libhb/denoise.c: LUT :78-94, the low-pass step :96-100,
 * temporal-only path :102-124, spatial+temporal path :126-165, state seeding
 * :167-201.  The reference fuses three recurrences in one raster scan; they are
 * separated here (horizontal along x, vertical along y on the horizontally
 * filtered value, temporal against the previous output) — same values, same
 * integer types where truncation could matter.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define LUT_BITS 4
#define CENTRE   (256 << LUT_BITS)

void orc_hqdn3d_coef(int16_t ct[8192], double dist25)
{
    const double capped = dist25 > 252.0 ? 252.0 : dist25;
    const double gamma = log(0.25) / log(1.0 - capped / 255.0 - 0.00001);
    for (int i = -CENTRE; i < CENTRE; i++)
    {
        const double f = (i * (1 << (9 - LUT_BITS)) + (1 << (8 - LUT_BITS)) - 1) / 512.0;   /* bin midpoint */
        const double simil = fmax(0, 1.0 - fabs(f) / 255.0);
        ct[CENTRE + i] = lrint(pow(simil, gamma) * 256.0 * f);
    }
    ct[0] = !!dist25;
}

/* sample -> 16-bit fixed point (LOAD, denoise.c:32-33) and back (STORE, :34-35); rows are byte
 * pointers, 16-bit containers for depth > 8.  LUT_BITS is 4 for every depth below 16 (:31). */
static inline uint32_t load_d(const uint8_t *row, int x, int depth)
{
    const uint32_t v = depth == 8 ? row[x] : ((const uint16_t *)row)[x];
    return (v << (16 - depth)) + (((1u << (16 - depth)) - 1) >> 1);
}
static inline void store_d(uint8_t *row, int x, uint32_t val, int depth)
{
    if (depth == 8) row[x] = val >> 8;
    else ((uint16_t *)row)[x] = val >> (16 - depth);
}

/* :96-100 ; `coef` points at the table centre */
static inline uint32_t lowpass(int prev, int cur, const int16_t *coef)
{
    const int d = (prev - cur) >> (8 - LUT_BITS);
    return cur + coef[d];
}

void orc_hqdn3d_plane_d(const uint8_t *src, uint8_t *dst, int w, int h, int sstride, int dstride,
                        uint16_t *frame_ant, int *state_valid,
                        const int16_t spatial_t[8192], const int16_t temporal_t[8192], int depth)
{
    const int16_t *spatial = spatial_t + CENTRE, *temporal = temporal_t + CENTRE;

    if (!*state_valid)
    {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
                frame_ant[(size_t)y * w + x] = load_d(src + (size_t)y * sstride, x, depth);
        *state_valid = 1;
    }

    if (!spatial_t[0])
    {
        /* temporal only (:102-124) */
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
            {
                uint16_t *a = &frame_ant[(size_t)y * w + x];
                const uint32_t t = lowpass(*a, load_d(src + (size_t)y * sstride, x, depth), temporal);
                *a = t;
                store_d(dst + (size_t)y * dstride, x, t, depth);
            }
        return;
    }

    uint32_t *hrow = malloc(sizeof(uint32_t) * w);
    uint16_t *line_ant = malloc(sizeof(uint16_t) * w);
    for (int y = 0; y < h; y++)
    {
        const uint8_t *s = src + (size_t)y * sstride;
        /* horizontal recurrence.  Row 0 starts from lowpass(LOAD(0), LOAD(0)) (:140-146),
         * every other row from LOAD(0) itself (:153-160). */
        uint32_t run = load_d(s, 0, depth);
        if (y == 0)
            run = lowpass(run, load_d(s, 0, depth), spatial);
        hrow[0] = run;
        for (int x = 1; x < w; x++)
        {
            run = lowpass(run, load_d(s, x, depth), spatial);
            hrow[x] = run;
        }
        for (int x = 0; x < w; x++)
        {
            /* vertical on the h-filtered value, through the uint16 line buffer */
            const uint32_t v = y == 0 ? hrow[x] : lowpass(line_ant[x], hrow[x], spatial);
            line_ant[x] = v;
            /* temporal against the previous OUTPUT */
            uint16_t *a = &frame_ant[(size_t)y * w + x];
            const uint32_t t = lowpass(*a, v, temporal);
            *a = t;
            store_d(dst + (size_t)y * dstride, x, t, depth);
        }
    }
    free(line_ant);
    free(hrow);
}

void orc_hqdn3d_plane(const uint8_t *src, uint8_t *dst, int w, int h, int sstride, int dstride,
                      uint16_t *frame_ant, int *state_valid,
                      const int16_t spatial_t[8192], const int16_t temporal_t[8192])
{
    orc_hqdn3d_plane_d(src, dst, w, h, sstride, dstride, frame_ant, state_valid, spatial_t, temporal_t, 8);
}
