/* hb_harness.h — C interface of the filter-chain test/bench driver
 * (hb_harness.c).  Plain C types only so Python can bind it with ctypes. */
#ifndef HBHIP_HARNESS_H
#define HBHIP_HARNESS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hbh_chain_s hbh_chain_t;

typedef struct hbh_frame_info_s
{
    int     is_eof;
    int64_t start;
    int64_t stop;
    int     flags;
    int     combed;
    int     width;
    int     height;
    int     fmt;
    int     nplanes;
    int     plane_width[4];
    int     plane_height[4];
    int     plane_stride[4];
} hbh_frame_info_t;

/* protos[i] = address of a registered hb_filter_object_t (e.g. &hb_filter_nlmeans);
 * settings[i] = "key=value:key=value" or NULL. */
hbh_chain_t *hbh_chain_open(int nstages, void *const *protos, const char *const *settings,
                            int pix_fmt, int width, int height,
                            int vrate_num, int vrate_den);
/* The same chain built the way do_job() builds it: filters by ID from the registered (CPU) objects, the HIP swap
 * (hb_hip_setup_hw_filters) when use_hip, init with drop / CPU-fallback on failure (work.c:1820-1870). */
hbh_chain_t *hbh_job_open(int nfilters, const int *ids, const char *const *settings, int pix_fmt, int width, int height,
                          int vrate_num, int vrate_den, int use_hip);
/* job->hw_device_index (common.h:991) of jobs opened from now on: which GPU the job's drop-ins run on; -1 = not set */
void hbh_set_job_device(int index);
/* jobs opened from now on carry one subtitle track of this source (enum subsource: 0 VOBSUB, 6 PGSSUB, ...) marked for
 * burn-in, which is what makes rendersub.c's init find its track (:1199-1209); -1 = none */
void hbh_set_job_subtitle(int source);
/* "|"-separated names of the stages as initialised; returns their count */
int hbh_chain_describe(hbh_chain_t *c, char *buf, int len);
/* Chains opened from now on run every stage on its own thread with a fifo in front (filter_loop, work.c:2527-2600);
 * output is collected after hbh_chain_push_eof(), which joins the stages. */
void hbh_set_threaded(int on);
/* Threaded chains opened from now on drop the frames their last stage makes (a consumer that keeps up, e.g. an
 * encoder) and only count them: hbh_chain_produced() = frames made so far, callable while the stages run. */
void hbh_set_discard_output(int on);
int  hbh_chain_produced(hbh_chain_t *c);
/* threaded chains: milliseconds stage's thread has spent inside its filter's work() so far (which stage a pipeline waits for) */
double hbh_chain_stage_busy_ms(hbh_chain_t *c, int stage);
/* Colour description of the source for chains opened from now on (init->color_*; AVCOL_* numbers,
 * range 1 = tv, 2 = pc).  Default bt709 / tv. */
void hbh_set_source_color(int prim, int transfer, int matrix, int range);
int  hbh_chain_push(hbh_chain_t *c, const uint8_t *const plane[3], const int stride[3],
                    int64_t start, int64_t stop, int flags, int combed);
/* `count` frames from `n_unique` prepared pictures (plane[3 * k + p], one stride per plane), numbered first, first + 1, ...:
 * filled by `nthreads` threads (the decoder's part in libhb), pushed in order; returns when the last one is in */
int  hbh_chain_feed(hbh_chain_t *c, const uint8_t *const *plane, const int stride[3], int n_unique, int first, int count,
                    int64_t duration, int flags, int nthreads);
int  hbh_chain_push_eof(hbh_chain_t *c);
int  hbh_chain_pending(hbh_chain_t *c);
int  hbh_chain_peek(hbh_chain_t *c, hbh_frame_info_t *info);
int  hbh_chain_pop(hbh_chain_t *c, uint8_t *const plane[3], const int stride[3]);
void hbh_chain_output_geometry(hbh_chain_t *c, int *width, int *height, int *vrate_num, int *vrate_den);
void hbh_chain_close(hbh_chain_t *c);

/* Run a frame-difference metric object (address of an hb_motion_metric_object_t) on two lumas. */
int hbh_motion_metric_run(const void *proto, int pix_fmt, int width, int height,
                          const uint8_t *luma_a, int stride_a, const uint8_t *luma_b, int stride_b, float *out);

/* One rendered subtitle bitmap: 4 planes (Y, Cb, Cr, alpha; 8-bit), placed at (x, y) of the frame. */
typedef struct
{
    const uint8_t *plane[4];
    int            stride[4];
    int            x, y, width, height;
} hbh_overlay_t;
/* Run a compositor object (address of an hb_blend_object_t) on one frame, in place. */
int hbh_blend_run(const void *proto, int pix_fmt, int width, int height, int chroma_location, int overlay_fmt,
                  uint8_t *const plane[3], const int stride[3], int n_overlays, const hbh_overlay_t *ov, int passes);

/* a decoded bitmap subtitle for the job's burn-in track (see hbh_set_job_subtitle): shown from start to stop (90 kHz, stop < 0:
 * until the next one) on a window_w x window_h canvas */
int hbh_chain_push_subtitle(hbh_chain_t *c, const hbh_overlay_t *ov, int64_t start, int64_t stop, int window_w, int window_h);

void hbhip_set_log_level(int level);

#ifdef __cplusplus
}
#endif
#endif
