/* Compiles the reference's libhb/nlmeans_x86.c in place (found through -I$(REF)/libhb),
 * unmodified, against include/hbhip_libhb.h.  See wrap_common.h. */
#include "wrap_common.h"
#include "nlmeans_x86.c"
