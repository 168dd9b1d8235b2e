/* motion_metric_hip.c — HIP-backed drop-in for the reference's frame-difference metric object
 * `hb_motion_metric` (libhb/motion_metric.c:306-312; type hb_motion_metric_object_t,
 * handbrake/common.h:1799-1811).
 *
 * vfr.c chooses the object in its own hb_motion_metric_init (:76-108: `hb_motion_metric_vt` for
 * VideoToolbox frames, `hb_motion_metric` otherwise), copies it, calls init once and work(a, b) on
 * consecutive frames while deciding which frame of a run to drop.  `hb_motion_metric_hip` is a third
 * choice with the same entry points.  The gamma table is built here exactly as build_gamma_lut does
 * (:36-42, double pow of float arguments, host libm) and handed to the device.
 */
#include "hbhip_host.h"

#include <math.h>

struct hb_motion_metric_private_s
{
    hbhip_motion_metric *dev;
    hbhip_ctx *ctx;                  /* the context the metric's kernels run on (the job's) */
};

static int motion_metric_hip_init(hb_motion_metric_object_t *metric, hb_filter_init_t *init)
{
    hb_motion_metric_private_t *pv = calloc(1, sizeof(*pv));
    metric->private_data = pv;
    if (pv == NULL)
    {
        hb_error("motion_metric(hip): calloc failed");
        return -1;
    }
    const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(init->pix_fmt);
    const int depth = desc != NULL ? desc->comp[0].depth : 0;
    const int max_value = (1 << depth) - 1;
    unsigned *lut = desc != NULL ? malloc(sizeof(unsigned) * (size_t)(max_value + 1)) : NULL;
    hbhip_ctx *ctx = lut != NULL ? hbhip_host_ctx_for(init) : NULL;
    int rc = ctx == NULL ? HBHIP_ERR_NODEVICE : HBHIP_OK;
    if (rc == HBHIP_OK)
    {
        for (int i = 0; i <= max_value; i++)
            lut[i] = 4095 * pow(((float)i / (float)(max_value - 1)), 2.2f);
        rc = hbhip_motion_metric_create(ctx, init->geometry.width, init->geometry.height, depth, lut, max_value + 1, &pv->dev);
        pv->ctx = ctx;
    }
    free(lut);
    if (rc != HBHIP_OK)
    {
        hb_error("motion_metric(hip): %s", hbhip_strerror(rc));
        free(pv);
        metric->private_data = NULL;
        return -1;
    }
    return 0;
}

static float motion_metric_hip_work(hb_motion_metric_object_t *metric, hb_buffer_t *buf_a, hb_buffer_t *buf_b)
{
    hb_motion_metric_private_t *pv = metric->private_data;
    float value = 0.f;
    int rc;
    hbhip_frame *fa = hbhip_host_frame_of(buf_a), *fb = hbhip_host_frame_of(buf_b);
    if (fa != NULL && fb != NULL)
    {
        hbhip_dev_frame da, db;
        hbhip_frame_describe(fa, &da, NULL, NULL);
        hbhip_frame_describe(fb, &db, NULL, NULL);
        rc = hbhip_frame_use_on(fa, pv->ctx);                    /* frames of another stream of the job (decomb's): behind their producer */
        if (rc == HBHIP_OK) rc = hbhip_frame_use_on(fb, pv->ctx);
        if (rc == HBHIP_OK) rc = hbhip_motion_metric_run_dev(pv->dev, da.plane[0], da.stride[0], db.plane[0], db.stride[0], &value);
    }
    else
        rc = hbhip_motion_metric_run(pv->dev, buf_a->plane[0].data, buf_a->plane[0].stride,
                                     buf_b->plane[0].data, buf_b->plane[0].stride, &value);
    if (rc != HBHIP_OK)
        hb_error("motion_metric(hip): %s", hbhip_strerror(rc));
    return value;
}

static void motion_metric_hip_close(hb_motion_metric_object_t *metric)
{
    hb_motion_metric_private_t *pv = metric->private_data;
    if (pv == NULL) return;
    hbhip_motion_metric_destroy(pv->dev);
    free(pv);
    metric->private_data = NULL;
}

hb_motion_metric_object_t hb_motion_metric_hip =
{
    .name  = "Motion metric (HIP)",
    .init  = motion_metric_hip_init,
    .work  = motion_metric_hip_work,
    .close = motion_metric_hip_close,
};
