/* stand-in for libavutil/intreadwrite.h (denoise.c:25 needs the aligned 16-bit accessors) */
#ifndef HBHIP_SHIM_INTREADWRITE_H
#define HBHIP_SHIM_INTREADWRITE_H
#include <stdint.h>
#define AV_RN16A(p)    (*(const uint16_t *)(p))
#define AV_WN16A(p, v) (*(uint16_t *)(p) = (uint16_t)(v))
#endif
