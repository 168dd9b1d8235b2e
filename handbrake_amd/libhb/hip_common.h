/* hip_common.h — the two calls libhb's work.c makes to put the HIP drop-ins into a job (hip_common.c).
 * Counterpart of handbrake/platform/macosx/vt_common.h:hb_vt_setup_hw_filters. */
#ifndef HBHIP_HIP_COMMON_H
#define HBHIP_HIP_COMMON_H

/* sanitize_filter_list_post (work.c:1515-1523): swap CPU filters for HIP drop-ins in place, bracket runs of
 * drop-ins with the upload / download adapters. */
void hb_hip_setup_hw_filters(hb_job_t *job);
/* the filter init loop (work.c:1855-1868), when init() of job->list_filter[index] failed: if that filter is a HIP
 * drop-in, put the CPU filter of the same id and settings back in its place (fixing the adapters around it) and
 * return > 0 - the loop then CONTINUES AT index - (return value - 1) instead of dropping the filter; 0 = not ours. */
int  hb_hip_filter_init_failed(hb_job_t *job, int index, hb_filter_init_t *init);
/* do_job's clean-up, once the job's filters have been closed (work.c:2311-2321): the HIP stream the job had leased goes
 * back to the pool (libhb/hbhip_registry.c: concurrent jobs on one GPU run on streams of their own) */
void hb_hip_job_close(hb_job_t *job);
/* hb_avfilter_combine (hbavfilter.c:520-541): an aliased id whose object is a drop-in is a real filter */
int  hb_hip_filter_is_hip(const hb_filter_object_t *filter);
/* a filter that only handles the hb_buffer_t around a picture (vfr, rendersub, rpu): a member of a device-resident
 * run like the ones are_filters_supported() lists (platform/macosx/vt_common.c:424-448) */
int  hb_hip_filter_is_hw_transparent(const hb_filter_object_t *filter);

#endif
