/* Stand-in for libavutil/bswap.h (only what libhb/blend.c uses). */
#ifndef HBHIP_SHIM_BSWAP_H
#define HBHIP_SHIM_BSWAP_H
#include <stdint.h>
static inline uint16_t av_bswap16(uint16_t x) { return (uint16_t)((x >> 8) | (x << 8)); }
#endif
