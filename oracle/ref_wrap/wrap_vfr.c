/* Compiles the reference's libhb/vfr.c in place (found through -I$(REF)/libhb), unmodified, against
 * include/hbhip_libhb.h.  See wrap_common.h.  `hb_filter_vfr` is exported for the test harness, which registers it
 * as HB_FILTER_VFR the way hb_filter_get (common.c:5331-5495) holds it inside libhb.
 *
 * vfr.c picks its frame-difference metric by init->hw_pix_fmt (vfr.c:76-108: hb_motion_metric_vt for VideoToolbox
 * frames, hb_motion_metric otherwise).  A HIP build of libhb adds one case there (INTEGRATION.md §2:
 * `case AV_PIX_FMT_HBHIP: metric = &hb_motion_metric_hip;`).  To keep the file itself untouched the same choice is
 * made from outside: the `default:` branch's `&hb_motion_metric` resolves through the runtime's hw-helper table, which
 * holds what the loaded hw pipeline registered for that hw_pix_fmt - and the reference's own object when nothing did. */
#include "wrap_common.h"
#include <limits.h>
#include <inttypes.h>

static hb_motion_metric_object_t *hbref_metric_for(int hw_pix_fmt)
{
    hb_motion_metric_object_t *m = hbhip_rt_hw_helper(0, hw_pix_fmt);
    return m != NULL ? m : &hb_motion_metric;
}
#define hb_motion_metric (*hbref_metric_for(init->hw_pix_fmt))
#include "vfr.c"
#undef hb_motion_metric
