/* stand-in for libavutil/cpu.h: the declarations live in hbhip_libhb.h */
#include "hbhip_libhb.h"
