/* vfr_standin.c — the stand-in harness's frame-rate shaper (HB_FILTER_VFR).
 *
 * Inside libhb this file does not exist: libhb's own vfr.c is the filter, with the one `case AV_PIX_FMT_HBHIP:` that
 * INTEGRATION.md §2 adds to its hb_motion_metric_init (vfr.c:76-108).  Outside libhb something has to play vfr's part,
 * because every preset-built job carries one between decomb and NLMeans (preset.c:2026-2048; ids common.h:1739-1751)
 * and the PCIe-inclusive pass (handbrake_amd/hostpath.py) must measure the filter list a front-end really produces.
 * Like hb_runtime.c (fifo.c) and hb_harness.c (work.c) it is written from the behaviour of the file it stands in for
 * and is held to it by test: tests/test_vfr_cpu.py drives this object and the reference's vfr.c (compiled unmodified
 * into oracle/_ref) with the same timestamp sequences - gaps, overlaps, short streams, all three modes - and wants
 * equal output (frame identity, start, stop) from both.
 *
 * What the filter does (vfr.c:590-750, 203-353):
 *   stage 1, every mode: frames wait in a queue of three so that the time a dropped frame leaves behind (a gap between
 *            one frame's stop and the next one's start) can be handed out in quarters to the four frames around it;
 *            output times are re-derived so that they stay contiguous;
 *   stage 2, mode 1 (constant) / 2 (peak-limited): frames collect in a window of `depth`; while the window runs ahead
 *            of the target rate the member with the smallest frame-difference metric is dropped; what leaves the
 *            window starts where the previous output stopped; constant mode trims / repeats frames to whole ticks.
 * It never touches a sample: on a device-resident run (init->hw_pix_fmt == AV_PIX_FMT_HBHIP) the hb_buffer_t it queues,
 * drops and duplicates (hb_buffer_shallow_dup shares the device picture) wrap hbhip_frames, and the metric runs on the
 * device (hb_motion_metric_hip).
 */
#ifndef HBHIP_IN_LIBHB
#include "hbhip_host.h"

#include <limits.h>
#include <inttypes.h>

#define VFR_QUEUE 3            /* frames held back by stage 1 */
#define VFR_DEPTH_MAX 10       /* vfr.c:14 */

struct hb_filter_private_s
{
    int           mode;
    hb_rational_t rate_in, rate;

    /* stage 1 */
    hb_buffer_t  *queue[VFR_QUEUE + 1];
    int           queued;
    int64_t       t_start[4], t_stop[4];   /* newest first */
    int64_t       owed[4];
    int64_t       lost_total, gained_total;
    int           n_gaps, n_stretched;

    /* stage 2 */
    hb_buffer_t  *win[VFR_DEPTH_MAX + 1];
    double        score[VFR_DEPTH_MAX + 1];
    int           n_win, depth;
    int64_t       span;
    double        tick, cursor;
    int           n_out, n_drop, n_dup;

    hb_motion_metric_object_t *metric;
};

static void win_remove(hb_filter_private_t *pv, int k)
{
    for (int i = k; i + 1 < pv->n_win; i++)
        pv->win[i] = pv->win[i + 1];
    /* the scores shift over the whole array, whatever the fill (delete_metric, vfr.c:127-133, is called with `count`) */
    for (int i = k; i + 1 < pv->n_win; i++)
        pv->score[i] = pv->score[i + 1];
    pv->n_win--;
}

/* which member of the window to drop, or -1 (vfr.c:135-183) */
static int pick_drop(const hb_filter_private_t *pv, int count)
{
    double target = pv->cursor + pv->tick * (count - 1);
    if (pv->win[count - 1]->s.stop >= (int64_t)target)
        return -1;
    const hb_buffer_t *first = pv->win[0];
    int best = 0, i;
    for (i = 1; i < count; i++)
    {
        if (pv->win[i]->s.stop - first->s.start > pv->span)
            break;
        if (pv->score[i] < pv->score[best])
            best = i;
    }
    target = pv->cursor + pv->tick * (i - 1);
    if (pv->win[i - 1]->s.stop >= (int64_t)target)
        return -1;
    return best;
}

/* stage 2: `in` joins the window (NULL = flushing); what leaves it comes back as a ->next list */
static hb_buffer_t *shape(hb_filter_private_t *pv, hb_buffer_t *in)
{
    if (pv->mode == 0)
    {
        if (in != NULL)
        {
            pv->n_out++;
            pv->cursor = in->s.stop;
        }
        return in;
    }
    if (in != NULL)
    {
        if (pv->cursor == (int64_t)AV_NOPTS_VALUE)
            pv->cursor = in->s.start;
        pv->win[pv->n_win++] = in;
        if (pv->n_win < 2)
            return NULL;
        pv->score[pv->n_win - 1] = pv->metric->work(pv->metric, pv->win[pv->n_win - 2], pv->win[pv->n_win - 1]);
        if (pv->n_win < pv->depth)
            return NULL;
    }
    const int count = pv->n_win;
    const int drop = pick_drop(pv, count);
    if (drop >= 0)
    {
        hb_buffer_t *gone = pv->win[drop];
        win_remove(pv, drop);
        hb_buffer_close(&gone);
        pv->n_drop++;
        return NULL;
    }

    hb_buffer_list_t list;
    hb_buffer_list_clear(&list);
    hb_buffer_t *out = pv->win[0];
    win_remove(pv, 0);
    hb_buffer_list_append(&list, out);
    out->s.start = pv->cursor;
    double edge = pv->cursor + pv->tick;
    pv->n_out++;
    if (pv->mode > 1)
    {
        /* peak-limited: keep the frame's own end unless that would beat the rate */
        if (out->s.stop < edge)
            out->s.stop = pv->cursor = edge;
        else
            pv->cursor = out->s.stop;
    }
    else
    {
        /* constant: one tick per frame; a frame that covers more ticks is repeated */
        double over = (double)out->s.stop - edge;
        out->s.stop = pv->cursor = edge;
        for (; over >= pv->tick; over -= pv->tick)
        {
            hb_buffer_t *again = hb_buffer_shallow_dup(out);
            again->s.new_chap = 0;
            again->s.start = edge;
            edge += pv->tick;
            again->s.stop = pv->cursor = edge;
            hb_buffer_list_append(&list, again);
            pv->n_dup++;
            pv->n_out++;
        }
    }
    return hb_buffer_list_clear(&list);
}

static int vfr_standin_init(hb_filter_object_t *filter, hb_filter_init_t *init)
{
    hb_filter_private_t *pv = calloc(1, sizeof(*pv));
    filter->private_data = pv;
    if (pv == NULL) return -1;
    pv->mode = init->cfr;
    pv->rate_in = pv->rate = init->vrate;
    hb_dict_extract_int(&pv->mode, filter->settings, "mode");
    hb_dict_extract_rational(&pv->rate, filter->settings, "rate");
    if (pv->mode)
    {
        /* inside libhb: the switch on init->hw_pix_fmt in vfr.c:76-108 */
        hb_motion_metric_object_t *proto = hbhip_rt_hw_helper(0, init->hw_pix_fmt);
        if (proto == NULL) proto = hbhip_rt_hw_helper(0, AV_PIX_FMT_HBHIP);   /* host frames: the same object uploads the lumas */
        pv->metric = proto != NULL ? malloc(sizeof(*pv->metric)) : NULL;
        if (pv->metric == NULL) { free(pv); filter->private_data = NULL; return -1; }
        *pv->metric = *proto;
        if (pv->metric->init(pv->metric, init))
        {
            free(pv->metric);
            free(pv);
            filter->private_data = NULL;
            return -1;
        }
    }
    pv->depth = 2;
    const double fps_in = (double)pv->rate_in.num / pv->rate_in.den, fps = (double)pv->rate.num / pv->rate.den;
    if (fps_in > fps)
    {
        /* repeats to expect in a row - or, below 2, fresh frames in a row - plus one to see both ends (vfr.c:389-407) */
        double f = fps_in / fps;
        if (f > 1.0 && f < 2.0) f = 1 / (f - 1);
        pv->depth = ceil(f) + 1;
        if (pv->depth > VFR_DEPTH_MAX) pv->depth = VFR_DEPTH_MAX;
    }
    pv->span = pv->depth * 90000 / fps_in;
    pv->score[0] = INT_MAX;
    if (pv->mode == 2)
    {
        if (fps_in > fps) init->vrate = pv->rate;
    }
    else
        init->vrate = pv->rate;
    pv->tick = (double)pv->rate.den * 90000. / pv->rate.num;
    pv->cursor = (int64_t)AV_NOPTS_VALUE;
    init->cfr = pv->mode;
    return 0;
}

static void vfr_standin_close(hb_filter_object_t *filter)
{
    hb_filter_private_t *pv = filter->private_data;
    if (pv == NULL) return;
    hb_log("vfr: %d frames output, %d dropped and %d duped", pv->n_out, pv->n_drop, pv->n_dup);
    hb_log("vfr: lost time: %" PRId64 " (%d frames), gained: %" PRId64 " (%d frames)", pv->lost_total, pv->n_gaps,
           pv->gained_total, pv->n_stretched);
    for (int i = 0; i < pv->queued; i++) hb_buffer_close(&pv->queue[i]);
    for (int i = 0; i < pv->n_win; i++) hb_buffer_close(&pv->win[i]);
    if (pv->metric != NULL)
    {
        pv->metric->close(pv->metric);
        free(pv->metric);
    }
    free(pv);
    filter->private_data = NULL;
}

static int vfr_standin_work(hb_filter_object_t *filter, hb_buffer_t **buf_in, hb_buffer_t **buf_out)
{
    hb_filter_private_t *pv = filter->private_data;
    hb_buffer_t *in = *buf_in;
    *buf_in = NULL;
    *buf_out = NULL;

    if (in->s.flags & HB_BUF_FLAG_EOF)
    {
        /* what is still queued leaves with the times of slots 2, 1, 0 - in that order whatever the fill (vfr.c:603-620) */
        hb_buffer_list_t list;
        hb_buffer_list_clear(&list);
        int slot = 2;
        for (int i = 0; i < pv->queued; i++)
        {
            hb_buffer_t *b = pv->queue[i];
            b->s.start = pv->t_start[slot];
            b->s.stop = pv->t_stop[slot--];
            hb_buffer_list_append(&list, shape(pv, b));
        }
        pv->queued = 0;
        while (pv->n_win > 0)
            hb_buffer_list_append(&list, shape(pv, NULL));
        hb_buffer_list_append(&list, in);
        *buf_out = hb_buffer_list_clear(&list);
        return HB_FILTER_DONE;
    }

    if (pv->queued > 0 && in->s.start > pv->t_stop[0])
    {
        /* a frame went missing upstream: its time is owed to the four frames around it, the remainder to the oldest */
        const int64_t gap = in->s.start - pv->t_stop[0];
        pv->owed[0] += gap / 4;
        pv->owed[1] += gap / 4;
        pv->owed[2] += gap / 4;
        pv->owed[3] += gap - 3 * (gap / 4);
        pv->lost_total += gap;
        pv->n_gaps++;
    }
    else if (in->s.stop <= pv->t_stop[0])
    {
        pv->n_drop++;                                   /* goes backwards in time: a broken source */
        hb_buffer_close(&in);
        return HB_FILTER_OK;
    }

    for (int i = 3; i >= 1; i--)
    {
        pv->t_start[i] = pv->t_start[i - 1];
        pv->t_stop[i] = pv->t_stop[i - 1];
    }
    if (pv->queued == 0)
    {
        pv->t_start[0] = in->s.start;
        pv->t_stop[0] = in->s.stop;
    }
    else
    {
        pv->t_start[0] = pv->t_stop[1];                  /* contiguous: starts where the previous one stops */
        pv->t_stop[0] = pv->t_start[0] + (in->s.stop - in->s.start);
    }
    pv->queue[pv->queued++] = in;
    if (pv->queued <= VFR_QUEUE)
        return HB_FILTER_OK;

    hb_buffer_t *out = pv->queue[0];
    for (int i = 1; i < pv->queued; i++) pv->queue[i - 1] = pv->queue[i];
    pv->queued--;
    if (pv->owed[3] > 0)
    {
        int64_t shift = 0;
        for (int i = 3; i >= 0; i--)
        {
            pv->t_start[i] += shift;
            pv->t_stop[i] += pv->owed[i] + shift;
            pv->gained_total += pv->owed[i];
            shift += pv->owed[i];
            pv->owed[i] = 0;
            pv->n_stretched++;
        }
    }
    out->s.start = pv->t_start[3];
    out->s.stop = pv->t_stop[3];
    *buf_out = shape(pv, out);
    return HB_FILTER_OK;
}

static hb_filter_info_t *vfr_standin_info(hb_filter_object_t *filter)
{
    hb_filter_private_t *pv = filter->private_data;
    if (pv == NULL) return NULL;
    hb_filter_info_t *info = calloc(1, sizeof(*info));
    if (info == NULL) return NULL;
    info->human_readable_desc = malloc(128);
    const double fps_in = (double)pv->rate_in.num / pv->rate_in.den, fps = (double)pv->rate.num / pv->rate.den;
    info->output.vrate = (pv->mode == 2 && !(fps_in > fps)) ? pv->rate_in : pv->rate;
    info->output.cfr = pv->mode;
    snprintf(info->human_readable_desc, 128, pv->mode == 0 ? "frame rate: same as source (around %.3f fps)"
             : pv->mode == 2 ? "frame rate: %.3f fps -> peak rate limited to %.3f fps" : "frame rate: %.3f fps -> constant %.3f fps",
             pv->mode == 0 ? fps : fps_in, fps);
    return info;
}

hb_filter_object_t hb_filter_vfr_standin =
{
    .id                = HB_FILTER_VFR,
    .enforce_order     = 1,
    .name              = "Framerate Shaper (stand-in)",
    .short_name        = "vfr",
    .settings          = NULL,
    .init              = vfr_standin_init,
    .work              = vfr_standin_work,
    .close             = vfr_standin_close,
    .info              = vfr_standin_info,
    .settings_template = "mode=^([012])$:rate=^" HB_RATIONAL_REG "$",
};
#endif /* !HBHIP_IN_LIBHB */
