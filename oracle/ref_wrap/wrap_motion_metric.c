/* Compiles the reference's libhb/motion_metric.c in place (found through -I$(REF)/libhb),
 * unmodified, against include/hbhip_libhb.h.  See wrap_common.h.
 * `hb_motion_metric` (the object vfr.c copies, vfr.c:76-108) is exported for the test harness. */
#include "wrap_common.h"
#include <math.h>
#include "motion_metric.c"
