/* Compiles the reference's libhb/taskset.c in place (found through -I$(REF)/libhb),
 * unmodified, against include/hbhip_libhb.h.  See wrap_common.h. */
#include "wrap_common.h"
#include "taskset.c"
