/* Compiles the reference's libhb/blend.c in place (found through -I$(REF)/libhb),
 * unmodified, against include/hbhip_libhb.h.  See wrap_common.h.
 * `hb_blend` (the reference's compositor object) is exported for the test harness. */
#include "wrap_common.h"
#include "blend.c"
