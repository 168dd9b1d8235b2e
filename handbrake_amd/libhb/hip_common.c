/* hip_common.c — puts the HIP drop-ins into a job's filter list.
 *
 * The reference has exactly this for its Metal filters: sanitize_filter_list_post() calls
 * hb_vt_setup_hw_filters(job) (work.c:1515-1523), which swaps CPU filter objects for GPU ones IN PLACE with
 * replace_filter(job, CPU_ID, GPU_ID) and adds an adapter (platform/macosx/vt_common.c:486-540).
 * hb_hip_setup_hw_filters() is the same for HIP:
 *   1. every filter of job->list_filter that has a HIP drop-in (hbhip_filter_get, same id) is replaced by a copy of
 *      the drop-in carrying the same settings dict, at the same list position;
 *   2. every run of two or more drop-ins - adjacent, or with only "hw-transparent" filters between them (vfr,
 *      rendersub, rpu: hb_hip_filter_is_hw_transparent) - is bracketed by hb_filter_hip_upload / hb_filter_hip_download,
 *      so frames stay in HBM inside the run (a lone drop-in moves its own frames; adapters would only add two threads).
 * And one thing the VideoToolbox path does not need: a drop-in's init() may refuse settings it has no kernels for.
 * work.c drops a filter whose init fails (:1861-1868) - for a drop-in that would silently lose the filter, so the init
 * loop calls hb_hip_filter_init_failed() first, which puts the CPU filter back (and re-brackets the run around it).
 *
 * The aliased filters (crop/scale, rotate, pad, grayscale, colorspace, format, yadif, bwdif) are .skip = 1 objects in
 * the reference whose work happens in the combined HB_FILTER_AVFILTER (hb_avfilter_combine, hbavfilter.c:510-622);
 * a drop-in is a real filter (.skip = 0, own work()), and because the swap changes the object behind the id,
 * hb_avfilter_combine's switch (hbavfilter.c:520-541) must skip ids whose object is a drop-in: hb_hip_filter_is_hip().
 */
#include "hbhip_host.h"
#include "hip_common.h"

static int is_adapter(const hb_filter_object_t *f)
{
    return f->id == HB_FILTER_HIP_UPLOAD || f->id == HB_FILTER_HIP_DOWNLOAD;
}

int hb_hip_filter_is_hip(const hb_filter_object_t *f)
{
    if (f == NULL) return 0;
    const hb_filter_object_t *p = hbhip_filter_get(f->id);
    return p != NULL && p->init == f->init;
}

static int force_swap(void)
{
    const char *force = getenv("HBHIP_FORCE_SWAP");           /* tests: swap even without a device, so that */
    return force != NULL && atoi(force) != 0;                 /* every init fails and the fallback is exercised */
}

static int hip_enabled(void)
{
    const char *off = getenv("HBHIP_DISABLE");
    if (off != NULL && atoi(off) != 0) return 0;
    if (force_swap()) return 1;
    return hbhip_device_count() > 0;
}

static hb_filter_object_t *new_adapter(int id)
{
    hb_filter_object_t *a = hb_filter_copy(hbhip_filter_get(id));
    if (a != NULL && a->settings == NULL) a->settings = hb_dict_init();
    return a;
}

/* vt_common.c:486-502, by position instead of by id */
static void replace_at(hb_list_t *list, int pos, hb_filter_object_t *proto)
{
    hb_filter_object_t *old = hb_list_item(list, pos);
    if (old->settings == NULL) return;                        /* replace_filter: no settings, no swap (:493-494) */
    hb_filter_object_t *nf = hb_filter_copy(proto);
    if (nf == NULL) return;
    hb_dict_free(&nf->settings);
    nf->settings = hb_value_dup(old->settings);
    hb_list_rem(list, old);
    hb_list_insert(list, pos, nf);
    hb_filter_close(&old);
}

/* Filters that never look at a picture's samples themselves, only at the hb_buffer_t around it: they sit INSIDE a
 * device-resident run without breaking it.  The reference has the same notion for its Metal pipeline -
 * are_filters_supported() (platform/macosx/vt_common.c:424-448) lists HB_FILTER_VFR, RENDER_SUB, FORMAT and RPU
 * beside the filters it has kernels for - and the first two choose their pixel helper by init->hw_pix_fmt:
 * vfr.c:76-108 the motion metric (hb_motion_metric_hip here), rendersub.c:1129-1161 the compositor (hb_blend_hip).
 * FORMAT has a drop-in of its own (format_hip.c); RPU only edits Dolby Vision side data (rpu.c).
 * Every preset-built job has a VFR between decomb (6) and NLMeans (16) (preset.c:2026-2048, ids common.h:1739-1751):
 * without this the run - and the frames - would leave the device there. */
int hb_hip_filter_is_hw_transparent(const hb_filter_object_t *f)
{
    if (f == NULL || hb_hip_filter_is_hip(f)) return 0;
    return f->id == HB_FILTER_VFR || f->id == HB_FILTER_RENDER_SUB || f->id == HB_FILTER_RPU;
}

/* Bracket, from position `from` on, every run [drop-in (drop-in | transparent)* drop-in] that holds at least two
 * drop-ins with hb_filter_hip_upload / hb_filter_hip_download (a lone drop-in moves its own frames; adapters would only
 * add two threads).  Transparent filters at either end of a run stay outside: nothing is gained by uploading for them. */
static void bracket_runs(hb_list_t *list, int from)
{
    for (int i = from; i < hb_list_count(list);)
    {
        if (!hb_hip_filter_is_hip(hb_list_item(list, i)) || is_adapter(hb_list_item(list, i))) { i++; continue; }
        int last = i, n = 1;
        for (int j = i + 1; j < hb_list_count(list); j++)
        {
            hb_filter_object_t *f = hb_list_item(list, j);
            if (is_adapter(f)) break;
            if (hb_hip_filter_is_hip(f)) { last = j; n++; }
            else if (!hb_hip_filter_is_hw_transparent(f)) break;
        }
        if (n >= 2)
        {
            hb_list_insert(list, last + 1, new_adapter(HB_FILTER_HIP_DOWNLOAD));
            hb_list_insert(list, i, new_adapter(HB_FILTER_HIP_UPLOAD));
            last += 2;
        }
        i = last + 1;
    }
}

void hb_hip_setup_hw_filters(hb_job_t *job)
{
    if (job == NULL || job->list_filter == NULL || !hip_enabled()) return;
    if (job->hw_pix_fmt != AV_PIX_FMT_NONE) return;           /* another hw pipeline owns the frames */
    if (!force_swap() && hbhip_host_job_index_is_hip(job) && job->hw_device_index >= hbhip_device_count())
    {
        /* the job names an adapter (common.h:991, hb_json.c "AdapterIndex") this process has no GPU for */
        hb_log("hbhip: job asks for GPU %d, %d present: keeping the CPU filters", job->hw_device_index, hbhip_device_count());
        return;
    }
    hb_list_t *list = job->list_filter;
    for (int i = 0; i < hb_list_count(list); i++)
    {
        hb_filter_object_t *f = hb_list_item(list, i);
        if (is_adapter(f) || hb_hip_filter_is_hip(f)) continue;
        hb_filter_object_t *proto = hbhip_filter_get(f->id);
        if (proto != NULL) replace_at(list, i, proto);
    }
    bracket_runs(list, 0);
}

int hb_hip_filter_init_failed(hb_job_t *job, int index, hb_filter_init_t *init)
{
    if (job == NULL || job->list_filter == NULL) return 0;
    hb_list_t *list = job->list_filter;
    hb_filter_object_t *f = hb_list_item(list, index);
    if (f == NULL || is_adapter(f) || !hb_hip_filter_is_hip(f)) return 0;
    hb_filter_object_t *cpu = hb_filter_init(f->id);
    if (cpu == NULL) return 0;                                /* no CPU filter of that id: work.c drops it */
    hb_dict_free(&cpu->settings);
    cpu->settings = f->settings ? hb_value_dup(f->settings) : hb_dict_init();
    if (cpu->sub_filter != NULL)                              /* mt_frame wrapper: hb_add_filter_dict copies them down */
    {
        hb_dict_free(&cpu->sub_filter->settings);
        cpu->sub_filter->settings = hb_value_dup(cpu->settings);
    }
    hb_log("hbhip: '%s' keeps its CPU filter (the HIP drop-in declined these settings)", cpu->name);
    hb_list_rem(list, f);
    hb_filter_close(&f);

    /* everything from `index` on has not been initialised yet: take its adapters out, put the CPU filter in, and
     * bracket what is left of the runs afresh.  What lies before `index` has seen its init and stays as it is. */
    for (int i = index; i < hb_list_count(list);)
    {
        hb_filter_object_t *a = hb_list_item(list, i);
        if (!is_adapter(a)) { i++; continue; }
        hb_list_rem(list, a);
        hb_filter_close(&a);
    }
    int pos = index, ret = 1;
    if (hbhip_host_dev_io(init))
    {
        /* inside a device-resident run: the CPU filter needs host frames */
        hb_filter_object_t *prev = pos > 0 ? hb_list_item(list, pos - 1) : NULL;
        if (prev != NULL && prev->id == HB_FILTER_HIP_UPLOAD)
        {
            /* the run's own upload sits right in front: undo it instead of downloading straight again */
            if (prev->close != NULL) prev->close(prev);
            hb_list_rem(list, prev);
            hb_filter_close(&prev);
            init->hw_pix_fmt = AV_PIX_FMT_NONE;
            pos--;
            ret = 2;                                          /* the CPU filter now sits one slot earlier */
        }
        else
            hb_list_insert(list, pos++, new_adapter(HB_FILTER_HIP_DOWNLOAD));
    }
    hb_list_insert(list, pos, cpu);
    bracket_runs(list, pos + 1);
    return ret;
}
