/* decomb_oracle.c — CPU restatement of decomb's yadif / blend / cubic line
 * filters (8-bit and 16-bit samples).  TEST INFRASTRUCTURE ONLY (see oracle.h).
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

/* crop_table[v + 1024] (:23-41): 0 below 0, v inside, max_value above. */
static inline int cropv(int v, int maxv) { return v < 0 ? 0 : v > maxv ? maxv : v; }

#define PIXEL uint8_t
#define PX(n) n##_8
#include "decomb_oracle_px.h"
#undef PIXEL
#undef PX
#define PIXEL uint16_t
#define PX(n) n##_16
#include "decomb_oracle_px.h"
#undef PIXEL
#undef PX

void orc_decomb_plane(const uint8_t *prev, const uint8_t *cur, const uint8_t *next, int stride,
                      const uint8_t *guess, int guess_stride,
                      uint8_t *dst, int dst_stride, int width, int height,
                      int mode, int parity, int tff)
{
    decomb_plane_8(prev, cur, next, stride, guess, guess_stride, dst, dst_stride, width, height, mode, parity, tff, 255);
}

void orc_decomb_plane16(const uint16_t *prev, const uint16_t *cur, const uint16_t *next, int stride,
                        const uint16_t *guess, int guess_stride,
                        uint16_t *dst, int dst_stride, int width, int height,
                        int mode, int parity, int tff, int depth)
{
    decomb_plane_16(prev, cur, next, stride, guess, guess_stride, dst, dst_stride, width, height, mode, parity, tff,
                    (1 << depth) - 1);
}
