/* Compiles the reference's libhb/mt_frame_filter.c in place (found through -I$(REF)/libhb),
 * unmodified, against include/hbhip_libhb.h.  See wrap_common.h. */
#include "wrap_common.h"
#include "mt_frame_filter.c"
