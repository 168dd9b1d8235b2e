/* hbhip.h — C ABI of libhbhip.so: libhb's per-pixel video-filter hot path as
 * hand-written HIP kernels for AMD Instinct MI355X (gfx950 / CDNA4).
 *
 * This is the drop-in boundary.  libhb stays C: each HIP-backed
 * hb_filter_object_t (handbrake_amd/libhb/<filter>_hip.c) keeps the reference's
 * init/work/close surface (libhb/handbrake/common.h:1670-1711) and only
 * translates hb_buffer_t <-> plane pointers before calling the functions below.
 * Plain pointers and sizes only; no C++ or torch types cross this boundary; no
 * exception crosses it either.  Every entry point names the reference
 * interface it replaces (paths relative to /root/reference/libhb).
 *
 * Conventions
 *   - return 0 (HBHIP_OK) on success, HBHIP_AGAIN (1) when a pull has no frame
 *     ready yet, negative HBHIP_ERR_* on failure (hbhip_strerror()).
 *   - the caller owns host memory; the library owns device memory and, unless
 *     one is adopted (hbhip_ctx_create_on_stream), one hipStream_t per context.
 *   - one caller thread per filter instance (that is what filter_loop gives,
 *     work.c:2527-2600); distinct instances / contexts are independent.
 *   - a missing GPU is an error (HBHIP_ERR_NODEVICE): there is NO CPU fallback
 *     inside this library.  libhb falls back by re-inserting its CPU filter
 *     when a HIP filter's init() fails (work.c:1861-1868 semantics).
 */
#ifndef HBHIP_H
#define HBHIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HBHIP_OK               0
#define HBHIP_AGAIN            1
#define HBHIP_ERR_NODEVICE    (-1)
#define HBHIP_ERR_HIP         (-2)
#define HBHIP_ERR_ARG         (-3)
#define HBHIP_ERR_NOMEM       (-4)
#define HBHIP_ERR_UNSUPPORTED (-5)
#define HBHIP_ERR_STATE       (-6)

#define HBHIP_ABI_VERSION 1

typedef struct hbhip_ctx    hbhip_ctx;
typedef struct hbhip_filter hbhip_filter;

/* A planar picture living in HOST memory: what hb_buffer_t.plane[] describes
 * (handbrake/internal.h:137-144).  stride in bytes. */
typedef struct hbhip_host_frame
{
    uint8_t *plane[3];
    int      stride[3];
} hbhip_host_frame;

/* A planar picture living in DEVICE memory (HBM).  Used by the device-resident
 * hand-off between adjacent HIP filters (the reference's
 * precedent is hb_buffer_t.storage_type = COREMEDIA, internal.h:152-153) and
 * by bench.py, whose inputs are resident in HBM before the timed region. */
typedef struct hbhip_dev_frame
{
    void *plane[3];
    int   stride[3];
} hbhip_dev_frame;

/* ---- library / context ------------------------------------------------------ */
int         hbhip_abi_version(void);
int         hbhip_device_count(void);                 /* hb_get_cpu_count() analogue, ports.c:292 */
const char *hbhip_strerror(int code);
int         hbhip_ctx_create(int device, hbhip_ctx **out);
/* Adopt an existing hipStream_t (passed as void*) instead of creating one. */
int         hbhip_ctx_create_on_stream(int device, void *hip_stream, hbhip_ctx **out);
void        hbhip_ctx_destroy(hbhip_ctx *ctx);
int         hbhip_ctx_sync(hbhip_ctx *ctx);           /* hipStreamSynchronize */
const char *hbhip_ctx_last_error(hbhip_ctx *ctx);     /* text of the last HIP failure */
int         hbhip_ctx_device_name(hbhip_ctx *ctx, char *buf, int len);
int         hbhip_ctx_device_index(hbhip_ctx *ctx);   /* the `device` it was created on; < 0 on a NULL context */
/* The on-box HBM ceiling: a float4 copy kernel over two buffers of `bytes` each (take them well past the 256 MB Infinity
 * Cache), best of `iters` timed passes with HIP events; *gbps = (read + write) bytes / time.  bench.py reports roofline
 * fractions against this as well as against the nominal 8 TB/s. */
int         hbhip_ctx_copy_bandwidth(hbhip_ctx *ctx, size_t bytes, int iters, double *gbps);

/* Per-kernel timing with HIP events on the context's stream (off by default).
 * When enabled every kernel launch is bracketed by two events; stats are read
 * back per kernel name.  bench.py derives roofline.achieved from these. */
int  hbhip_ctx_profile_enable(hbhip_ctx *ctx, int on);
int  hbhip_ctx_profile_reset(hbhip_ctx *ctx);
int  hbhip_ctx_profile_count(hbhip_ctx *ctx);         /* distinct kernel names seen (syncs) */
int  hbhip_ctx_profile_get(hbhip_ctx *ctx, int idx, char *name, int name_len,
                           int64_t *launches, double *total_ms);
/* Two user events on the stream, for whole-region timing. */
int  hbhip_ctx_mark(hbhip_ctx *ctx, int slot);        /* slot 0..7: record event */
int  hbhip_ctx_elapsed_ms(hbhip_ctx *ctx, int slot_a, int slot_b, double *ms); /* syncs slot_b */

/* Device memory helpers (thin hipMalloc/hipFree/hipMemcpy wrappers so a C host
 * never needs the HIP headers). */
/* Page-locked host memory (hipHostMalloc): frame buffers allocated from it make the H2D / D2H
 * copies of hbhip_filter_push / pull true DMA transfers instead of staged pageable copies.
 * What fifo.c's buffer pools would allocate from in a HIP build (hb_buffer_init, fifo.c:358-457). */
int  hbhip_host_alloc(size_t bytes, void **out);
void hbhip_host_free(void *p);
int  hbhip_dev_alloc(hbhip_ctx *ctx, size_t bytes, void **out);
int  hbhip_dev_free(hbhip_ctx *ctx, void *p);
int  hbhip_dev_upload(hbhip_ctx *ctx, void *dst, const void *src, size_t bytes);
int  hbhip_dev_download(hbhip_ctx *ctx, void *dst, const void *src, size_t bytes);

/* ---- device-resident frames (hand-off between adjacent HIP filters) -----------------
 * The reference's GPU precedent keeps frames on the device between its Metal filters
 * by giving hb_buffer_t a storage_type (COREMEDIA, internal.h:152-153) and bracketing the
 * run of GPU filters with an adapter (HB_FILTER_ADAPTER_VT, platform/macosx/adapter_vt.c).
 * hbhip_frame is the HIP equivalent of the CVPixelBuffer behind such a buffer: a
 * reference-counted picture in HBM, recycled through a per-context pool. */
typedef struct hbhip_frame hbhip_frame;
int  hbhip_frame_alloc(hbhip_ctx *ctx, int width, int height, int depth,
                       int log2_chroma_w, int log2_chroma_h, hbhip_frame **out);
void hbhip_frame_retain(hbhip_frame *fr);
void hbhip_frame_release(hbhip_frame *fr);            /* back to the pool at refcount 0 */
/* A filter on ANOTHER context of the same GPU is about to queue work that reads the frame: its stream is ordered behind
 * the frame's producer (the ready mark only) and the frame goes idle behind that stream (a job with more than one HIP
 * stream, libhb/hbhip_registry.c).  No-op for the owner's context while nobody else has read the frame. */
int  hbhip_frame_use_on(hbhip_frame *fr, hbhip_ctx *ctx);
int  hbhip_frame_refs(hbhip_frame *fr);               /* holders right now; 1 = the caller is the only one (it may write in place) */
int  hbhip_frame_describe(hbhip_frame *fr, hbhip_dev_frame *out, int *width, int *height);
hbhip_ctx *hbhip_frame_context(hbhip_frame *fr);      /* the context (device, stream) whose pool the frame belongs to */
int  hbhip_frame_copy(hbhip_frame *dst, hbhip_frame *src);                  /* same geometry; stream-ordered D2D */
int  hbhip_frame_upload(hbhip_frame *fr, const hbhip_host_frame *src);      /* H2D, returns when src is consumed */
int  hbhip_frame_download(hbhip_frame *fr, const hbhip_host_frame *dst);    /* D2H, synchronous */
/* The pipelined H2D: the copy is queued on the context's upload stream and the call returns; `src` must stay valid until
 * hbhip_ctx_upload_done(ctx, token, block) has answered HBHIP_OK (HBHIP_AGAIN: not yet; block != 0 waits).  The frame's
 * ready mark is the copy itself: readers (hbhip_frame_use_on, hbhip_frame_copy, a download) wait for it and nothing else. */
int  hbhip_frame_upload_async(hbhip_frame *fr, const hbhip_host_frame *src, void **token);
int  hbhip_ctx_upload_done(hbhip_ctx *ctx, void *token, int block);
/* The producer of a frame marks the point of the context's stream behind which its contents are complete; a
 * download then waits for that point only (not for what other filter threads have queued since).  The pipelined
 * D2H: queue the copy on the download stream and return; `dst` and the frame must stay valid until
 * hbhip_frame_download_wait(fr, token) has returned.  A few in flight keep the bus busy (the download adapter). */
int  hbhip_frame_mark_ready(hbhip_frame *fr);
int  hbhip_frame_download_async(hbhip_frame *fr, const hbhip_host_frame *dst, void **token);
int  hbhip_frame_download_wait(hbhip_frame *fr, void *token);

/* ---- frames in, frames out: what a drop-in inside a device-resident run uses instead of push_dev / pull_dev -----------
 * hbhip_filter_use_frames(f) (once, before the first push): the filter's pictures are frames of its context's pool.
 * hbhip_filter_push_frame: the frame becomes the filter's input picture without a copy (the filter holds a reference of its
 * own; a frame with other holders, or of another geometry, is copied as hbhip_filter_push_dev would).
 * hbhip_filter_pull_frame: the next output AS a frame (its one reference passes to the caller); HBHIP_AGAIN when none. */
int hbhip_filter_use_frames(hbhip_filter *f);
int hbhip_filter_push_frame(hbhip_filter *f, hbhip_frame *fr, int64_t tag);
int hbhip_filter_pull_frame(hbhip_filter *f, hbhip_frame **out, int64_t *tag);

/* ---- generic streaming surface of a filter instance ---------------------------
 * Mirrors hb_filter_object_t.work (common.h:1682-1685): push one input frame,
 * pull zero or more output frames, flush at EOF, destroy in close().
 * `tag` travels with the frame (the host filter keeps hb_buffer_t props by tag). */
int  hbhip_filter_push(hbhip_filter *f, const hbhip_host_frame *in, int64_t tag);
int  hbhip_filter_push_dev(hbhip_filter *f, const hbhip_dev_frame *in, int64_t tag);
int  hbhip_filter_pull(hbhip_filter *f, const hbhip_host_frame *out, int64_t *tag);
int  hbhip_filter_pull_dev(hbhip_filter *f, const hbhip_dev_frame *out, int64_t *tag);
/* Batch form of push_dev/pull_dev: push n_in frames (tags tag0, tag0+1, ...) and
 * pull every frame that becomes ready into out[0..out_cap); *n_out = frames
 * written.  One ABI crossing per batch instead of two per frame. */
int  hbhip_filter_process_dev(hbhip_filter *f, const hbhip_dev_frame *in, int n_in, int64_t tag0,
                              const hbhip_dev_frame *out, int out_cap, int *n_out);
/* Pipelined form of push + pull for filters that make one frame from one frame: queue the upload of `in`, the
 * filter, and the download into `out`, and return at once.  `in` and `out` must stay valid until hbhip_filter_wait()
 * has returned this submission (they finish in submission order; `tag` comes back with it).  Two or three submissions
 * in flight overlap H2D, kernels and D2H - the role `threads` frames in flight play in the reference
 * (nlmeans.c:464-597, mt_frame_filter.c:45-237).  HBHIP_ERR_UNSUPPORTED: not such a filter, use push / pull. */
int  hbhip_filter_submit_async(hbhip_filter *f, const hbhip_host_frame *in, const hbhip_host_frame *out, int64_t tag);
int  hbhip_filter_wait(hbhip_filter *f, int64_t *tag);      /* oldest submission done; HBHIP_AGAIN when none is in flight */
int  hbhip_filter_inflight(hbhip_filter *f);
int  hbhip_filter_flush(hbhip_filter *f);             /* input ended (HB_BUF_FLAG_EOF) */
int  hbhip_filter_pending(hbhip_filter *f);           /* frames a pull would return now */
/* Batching across work() calls: with defer on, a filter that gathers frames for a common launch (decomb / EEDI2:
 * the fields of several frames per launch, NLMeans) launches nothing until hbhip_filter_kick() - the frames pushed
 * meanwhile are pending but NOT complete until the kick has been given.  What libhb's filters do with `threads`
 * frames in flight (nlmeans.c:464-597): answer HB_FILTER_DELAY for a few frames, then emit a burst. */
int  hbhip_filter_defer(hbhip_filter *f, int on);
int  hbhip_filter_kick(hbhip_filter *f);
void hbhip_filter_destroy(hbhip_filter *f);
/* Output geometry (cropscale / rotate change it; init->geometry, cropscale.c:170-178). */
int  hbhip_filter_out_geometry(hbhip_filter *f, int *width, int *height);
hbhip_ctx *hbhip_filter_context(hbhip_filter *f);     /* the context the instance was created on */

/* ---- a run of adjacent HIP filters fused into one object ------------------------------------
 * The reference merges runs of libavfilter-backed filters into one filter object whose work()
 * pushes a frame through the whole graph (hb_avfilter_combine, hbavfilter.c:510-622); this is the
 * same for a run of HIP filters: one caller thread, pictures handed from stage to stage in HBM by
 * pointer (no copies between stages), a batch of frames walked stage by stage so that batching
 * stages (NLMeans) cover the batch in one launch.  The chain BORROWS the filters: consecutive
 * geometries must match, and after hbhip_chain_destroy the caller destroys them LAST STAGE FIRST
 * (a stage may still hold pictures of the stage before it).
 * Stages created on `ctx` all run on its stream.  Stages created on contexts of their own (same
 * device) run on THEIR streams - libhb's one thread per filter (work.c:2527-2600) as one stream per
 * filter: the chain orders each hand-over with an event, so the first stages start on the next batch
 * while the last ones finish this one.  Input frames must be complete in `ctx`'s stream order when
 * hbhip_chain_process_dev is called; output frames are complete after hbhip_chain_sync().
 * pic_flags / combed: per input frame, what hbhip_decomb_push carries (NULL = 0); out_tags: the
 * tag of each output frame (decomb stages shift tags as documented at hbhip_decomb_push). */
typedef struct hbhip_chain hbhip_chain;
int  hbhip_chain_create(hbhip_ctx *ctx, hbhip_filter *const *stages, int n_stages, hbhip_chain **out);
int  hbhip_chain_process_dev(hbhip_chain *c, const hbhip_dev_frame *in, const int *pic_flags, const int *combed,
                             int n_in, int64_t tag0, const hbhip_dev_frame *out, int64_t *out_tags, int out_cap,
                             int *n_out);
int  hbhip_chain_flush_dev(hbhip_chain *c, const hbhip_dev_frame *out, int64_t *out_tags, int out_cap, int *n_out);
int  hbhip_chain_pending(hbhip_chain *c);             /* finished frames waiting for room in `out` */
int  hbhip_chain_sync(hbhip_chain *c);                /* wait for every stage's stream */
void hbhip_chain_destroy(hbhip_chain *c);

/* ---- NLMeans  (replaces nlmeans.c:223-419 init tables + nlmeans_template.c:545-717) */
#define HBHIP_NLMEANS_FRAMES_MAX 32                   /* NLMEANS_FRAMES_MAX, nlmeans.c:87 */
typedef struct hbhip_nlmeans_params
{
    /* per plane Y,Cb,Cr, already cascaded/sanitised exactly as nlmeans.c:306-343 */
    double strength[3];
    double origin_tune[3];
    int    patch_size[3];
    int    range[3];
    int    nframes[3];
    int    prefilter[3];
    /* tables built on the host with libm, as nlmeans.c:345-358 does */
    float  exptable[3][128];
    float  weight_fact_table[3];
    int    diff_max[3];
} hbhip_nlmeans_params;

int hbhip_nlmeans_create(hbhip_ctx *ctx, const hbhip_nlmeans_params *p,
                         int width, int height, int depth,
                         int log2_chroma_w, int log2_chroma_h,
                         hbhip_filter **out);
/* Max frames processed per kernel launch when several are queued (default 8). */
int hbhip_nlmeans_set_batch(hbhip_filter *f, int frames);

/* ---- Lapsharp  (replaces lapsharp_8, lapsharp.c:125-182) ------------------------ */
typedef struct hbhip_lapsharp_params
{
    double strength[3];   /* sanitised 0..1.5 (lapsharp.c:304-305)                         */
    int    kernel[3];     /* 0 lap, 1 isolap, 2 log, 3 isolog (lapsharp.c:80-86)           */
} hbhip_lapsharp_params;
int hbhip_lapsharp_create(hbhip_ctx *ctx, const hbhip_lapsharp_params *p, int width, int height,
                          int depth, int log2_chroma_w, int log2_chroma_h, hbhip_filter **out);

/* ---- Unsharp / chroma smooth (replace unsharp_8 unsharp.c:89-173 and
 *      chroma_smooth_8 chroma_smooth.c:87-172) ------------------------------------- */
typedef struct hbhip_blur_params
{
    int amount[3];        /* (int)(strength * 65536.0); 0 = plane is copied (unsharp.c:258) */
    int size[3];          /* odd 3..15 (unsharp.c:251-254)                                 */
} hbhip_blur_params;
int hbhip_unsharp_create(hbhip_ctx *ctx, const hbhip_blur_params *p, int width, int height,
                         int depth, int log2_chroma_w, int log2_chroma_h, hbhip_filter **out);
int hbhip_chroma_smooth_create(hbhip_ctx *ctx, const hbhip_blur_params *p, int width, int height,
                               int depth, int log2_chroma_w, int log2_chroma_h, hbhip_filter **out);

/* ---- hqdn3d (replaces hqdn3d_denoise_spatial/temporal/depth, denoise.c:102-201) --- */
typedef struct hbhip_hqdn3d_params
{
    /* coef[2*c] spatial, coef[2*c+1] temporal table of plane c, each built on the host
     * exactly as hqdn3d_precalc_coef builds it (denoise.c:78-94; entry 0 = strength != 0) */
    int16_t coef[6][8192];
} hbhip_hqdn3d_params;
int hbhip_hqdn3d_create(hbhip_ctx *ctx, const hbhip_hqdn3d_params *p, int width, int height,
                        int depth, int log2_chroma_w, int log2_chroma_h, hbhip_filter **out);

/* ---- Decomb (replaces decomb.c:495-612 frame logic + decomb_template.c:579-898
 *      line filters + eedi2_template.c passes) -------------------------------------- */
typedef struct hbhip_decomb_params
{
    int mode;                       /* MODE_DECOMB_* bits, decomb.h:13-18                  */
    int parity;                     /* -1 = from picture flags (decomb.c:519-528)          */
    /* EEDI2 (decomb.c:234-243) */
    int magnitude_threshold, variance_threshold, laplacian_threshold;
    int dilation_threshold, erosion_threshold, noise_threshold;
    int maximum_search_distance, post_processing;
} hbhip_decomb_params;
int hbhip_decomb_create(hbhip_ctx *ctx, const hbhip_decomb_params *p, int width, int height,
                        int depth, int log2_chroma_w, int log2_chroma_h, hbhip_filter **out);
/* push with the per-buffer state decomb looks at: s.flags (PIC_FLAG_*) and
 * s.combed (HB_COMB_*, set upstream by comb detect).  Pulled tags are
 * (input tag << 1) | field_index, field_index = 1 for the second frame of a bob pair. */
int hbhip_decomb_push(hbhip_filter *f, const hbhip_host_frame *in, int64_t tag, int pic_flags, int combed);
int hbhip_decomb_push_dev(hbhip_filter *f, const hbhip_dev_frame *in, int64_t tag, int pic_flags, int combed);
int hbhip_decomb_push_frame(hbhip_filter *f, hbhip_frame *fr, int64_t tag, int pic_flags, int combed);   /* as hbhip_filter_push_frame */
/* The reference's "Deinterlace" filter = FFmpeg yadif as deinterlace_init configures it
 * (deinterlace.c:72-143): spatial_check 0 = send_*_nospatial, bob = send_field (two frames per
 * input), selective = deint=interlaced (only frames whose s.combed is set), parity -1 / 0 (tff) /
 * 1 (bff).  Frames go in with hbhip_decomb_push[_dev] (flags + combed) and come out like decomb's.
 * Arithmetic of vf_yadif.c, parity unpinned (oracle/decomb_oracle.c:orc_yadif_ff_plane). */
int hbhip_yadif_create(hbhip_ctx *ctx, int spatial_check, int bob, int selective, int parity,
                       int width, int height, int depth, int log2_chroma_w, int log2_chroma_h,
                       hbhip_filter **out);
/* The reference's "Bwdif" filter = FFmpeg bwdif as deinterlace_init configures it (deinterlace.c:46, 72-143; the
 * spatial bit is yadif-only, :98-122): bob = send_field, selective = deint=interlaced, parity as above.  Same
 * push / pull surface as hbhip_yadif_create.  Arithmetic of vf_bwdif.c, parity unpinned
 * (oracle/decomb_oracle.c:orc_bwdif_plane, which follows platform/macosx/shaders/bwdif_vt.metal where the two agree). */
int hbhip_bwdif_create(hbhip_ctx *ctx, int bob, int selective, int parity,
                       int width, int height, int depth, int log2_chroma_w, int log2_chroma_h,
                       hbhip_filter **out);
/* Test hook: copy one plane of an EEDI2 scratch frame to the host (buffer 0..3 = eedi_half[],
 * 4..8 = eedi_full[], decomb.c:64-74); dst == NULL only queries stride/height. */
int hbhip_decomb_debug_eedi_plane(hbhip_filter *f, int buffer, int plane, uint8_t *dst, int dst_stride,
                                  int *stride, int *height);

/* ---- Comb detect (replaces comb_detect.c:1051-1072 comb_segmenter and the passes it
 *      runs: comb_detect_template.c:288-402/789-933, comb_detect.c:221-276, 384-454,
 *      556-622, 726-792, 901-966, 1029-1049) ------------------------------------------ */
typedef struct hbhip_comb_detect_params
{
    int mode, spatial_metric, motion_threshold, spatial_threshold;
    int filter_mode, block_threshold, block_width, block_height;
    float gamma_lut[256];           /* built on the host as comb_detect.c:1074-1081 does   */
} hbhip_comb_detect_params;
int hbhip_comb_detect_create(hbhip_ctx *ctx, const hbhip_comb_detect_params *p, int width, int height,
                             int depth, hbhip_filter **out);
/* store_ref (comb_detect.c:1007-1018): the luma plane becomes the newest of the
 * prev/cur/next ring.  luma == NULL repeats the newest plane (first frame / EOF). */
/* depth > 8: the gamma table has 1 << depth entries (comb_detect.c:1102, 1074-1081), more than
 * the params struct holds; hand it over before the first classify.  (depth 8 uses p->gamma_lut.) */
int hbhip_comb_detect_set_gamma_lut(hbhip_filter *f, const float *lut, int entries);
int hbhip_comb_detect_store(hbhip_filter *f, const uint8_t *luma, int stride);
int hbhip_comb_detect_store_dev(hbhip_filter *f, const void *luma, int stride);
/* comb_segmenter on the ring: *combed = HB_COMB_NONE/LIGHT/HEAVY for the middle plane. */
int hbhip_comb_detect_classify(hbhip_filter *f, int force_exhaustive, int *combed);
/* The same verdicts for n_frames (<= 16) frames at once, for callers that hold the lumas in HBM: frame i is judged
 * from lumas[i], lumas[i + 1], lumas[i + 2] (prev / cur / next as the ring would hold them - n_frames + 2 device
 * pointers, `stride` bytes a row, 4-byte aligned), bit i of force_bits = force_exhaustive for frame i.  Three
 * launches and one read-back whatever n_frames is; does not touch the ring.  8-bit, modes 0-3, filter-mode 0 or 2. */
int hbhip_comb_detect_classify_many_dev(hbhip_filter *f, const void *const *lumas, int stride, int n_frames,
                                        unsigned force_bits, int *combed);

/* Mask overlay, modes 4 (MODE_MASK) / 8 (MODE_COMPOSITE): draw_mask_box + apply_mask (comb_detect_template.c:21-136)
 * on `frame`, which holds a COPY of the frame the last classify judged combed (process_frame, comb_detect.c:1519-1526).
 * plane_w / plane_h: samples per row and rows of each plane.  The box position is the one a single check thread
 * leaves (the reference's segment threads race on it, comb_detect.c:205-208); its outline stays in the mask, as in
 * the reference. */
int hbhip_comb_detect_overlay(hbhip_filter *f, const hbhip_host_frame *frame, const int plane_w[3], const int plane_h[3]);
int hbhip_comb_detect_overlay_dev(hbhip_filter *f, const hbhip_dev_frame *frame, const int plane_w[3], const int plane_h[3]);

/* ---- Alias family (libavfilter/zimg-backed in the reference; arithmetic external,
 *      parity pinned to oracle/alias_oracle.c only — see DESIGN.md) ------------------- */
/* rotate_init's transpose/hflip/vflip composition (rotate.c:169-256). angle 0/90/180/270. */
int hbhip_rotate_create(hbhip_ctx *ctx, int angle, int hflip, int width, int height, int depth,
                        int log2_chroma_w, int log2_chroma_h, hbhip_filter **out);
/* FFmpeg `monochrome=cb:cr:size:high` as grayscale_init sets it up (grayscale.c:43-61). */
int hbhip_grayscale_create(hbhip_ctx *ctx, double cb, double cr, double size, double high,
                           int width, int height, int depth, int log2_chroma_w, int log2_chroma_h,
                           hbhip_filter **out);
/* `crop` + `zscale=filter=lanczos` as crop_scale_init sets them up (cropscale.c:63-165). */
typedef struct hbhip_cropscale_params
{
    int width, height;                                   /* output size                   */
    int crop_top, crop_bottom, crop_left, crop_right;    /* cropscale.c:68-73             */
} hbhip_cropscale_params;
int hbhip_cropscale_create(hbhip_ctx *ctx, const hbhip_cropscale_params *p, int width, int height,
                           int depth, int log2_chroma_w, int log2_chroma_h, hbhip_filter **out);
/* The other branch of crop_scale_init (cropscale.c:159-165): `crop` + `scale=flags=lanczos+accurate_rnd`, which the
 * reference builds when hb_av_can_use_zscale() says no (an odd width or height, hbffmpeg.c:870-915) - libswscale's
 * arithmetic instead of zimg's (8-bit planes: hScale8To15 + yuv2planeX_8; 10 / 12-bit: hScale16To15 + yuv2planeX_10 / _12). */
int hbhip_cropscale_sws_create(hbhip_ctx *ctx, const hbhip_cropscale_params *p, int width, int height,
                               int depth, int log2_chroma_w, int log2_chroma_h, hbhip_filter **out);
/* FFmpeg `pad=width:height:x:y:color` as pad_init sets it up (pad.c:40-148): the picture at (x, y)
 * of a width x height one, the rest filled with fill[] (Y, Cb, Cr sample values, already converted
 * from the RGB colour the way drawutils.c:ff_draw_color does).  x, y multiples of the chroma subsampling. */
typedef struct hbhip_pad_params
{
    int width, height, x, y;
    int fill[3];
} hbhip_pad_params;
int hbhip_pad_create(hbhip_ctx *ctx, const hbhip_pad_params *p, int width, int height, int depth,
                     int log2_chroma_w, int log2_chroma_h, hbhip_filter **out);
/* `format=pix_fmts=<fmt>` as format_init sets it up (format.c:13-111): libavfilter then converts with a same-size
 * `scale`, i.e. libswscale's unscaled planar copy.  Built: planar YUV depth changes 8 / 10 / 12 -> 8 / 10 / 12 with the
 * chroma subsampling unchanged (up: shift, full-range luma replicates the top bits; down: ordered dither - the
 * shift-only form for chroma and limited-range luma, (v - (v >> dst_depth) + d) >> shift for full-range luma, both arms of
 * swscale_unscaled.c's DITHER_COPY).  Arithmetic pinned to
 * oracle/alias_oracle.c:orc_format_plane only (parity unpinned). */
int hbhip_format_create(hbhip_ctx *ctx, int width, int height, int src_depth, int dst_depth,
                        int log2_chroma_w, int log2_chroma_h, int full_range, hbhip_filter **out);
/* The zscale [-> format=gbrpf32le -> tonemap] -> zscale -> format graph colorspace_init builds
 * (colorspace.c:126-193): matrix / range / transfer / primaries conversion, with tone mapping
 * when the source transfer is SMPTE 2084 or ARIB STD-B67 and the transfer changes.  Colour ids
 * are the AVCOL_* = HB_COLR_* numbers (common.h), range 1 = tv, 2 = pc; tonemap ids are
 * vf_tonemap's.  Arithmetic pinned to oracle/colorspace_oracle.c only (parity unpinned).
 * 8/10/12-bit, 4:2:0 / 4:2:2 / 4:4:4.  HBHIP_ERR_UNSUPPORTED for conversions outside its tables. */
enum { HBHIP_TONEMAP_NONE = 0, HBHIP_TONEMAP_LINEAR = 1, HBHIP_TONEMAP_GAMMA = 2, HBHIP_TONEMAP_CLIP = 3,
       HBHIP_TONEMAP_REINHARD = 4, HBHIP_TONEMAP_HABLE = 5, HBHIP_TONEMAP_MOBIUS = 6 };
typedef struct hbhip_colorspace_params
{
    int in_prim, in_transfer, in_matrix, in_range;       /* init->color_* (colorspace.c:96-99)      */
    int out_prim, out_transfer, out_matrix, out_range;   /* after the settings are applied (:101-120) */
    int tonemap;                                         /* HBHIP_TONEMAP_*; "hable" by default (:151) */
    double param;                                        /* NAN = the operator's default (:154-157)  */
    double desat;                                        /* carried; without effect, as in FFmpeg on GBR frames */
    double npl;                                          /* nominal peak luminance, 100 (:74)        */
    double peak;                                         /* determine_signal_peak (:37-49)           */
} hbhip_colorspace_params;
int hbhip_colorspace_create(hbhip_ctx *ctx, const hbhip_colorspace_params *p, int width, int height,
                            int depth, int log2_chroma_w, int log2_chroma_h, hbhip_filter **out);

/* ---- Frame-difference metric (replaces hb_motion_metric, motion_metric.c: the object vfr.c
 *      calls on consecutive frames to choose the one to drop, vfr.c:76-108, 380) ---------------
 * gamma_lut: the 1 << depth entries build_gamma_lut produces (:36-42), built by the caller. */
typedef struct hbhip_motion_metric hbhip_motion_metric;
int  hbhip_motion_metric_create(hbhip_ctx *ctx, int width, int height, int depth,
                                const unsigned *gamma_lut, int entries, hbhip_motion_metric **out);
/* hb_motion_metric_work (:268-279): luma planes of two frames -> the metric.  Synchronous. */
int  hbhip_motion_metric_run(hbhip_motion_metric *m, const uint8_t *luma_a, int stride_a,
                             const uint8_t *luma_b, int stride_b, float *out);             /* host planes */
int  hbhip_motion_metric_run_dev(hbhip_motion_metric *m, const void *luma_a, int stride_a,
                                 const void *luma_b, int stride_b, float *out);            /* planes in HBM */
void hbhip_motion_metric_destroy(hbhip_motion_metric *m);

/* ---- Subtitle compositor (replaces hb_blend, blend.c: the object rendersub.c hands every frame
 *      and the list of rendered overlays, rendersub.c:467, 1129-1161) -------------------------
 * Planar 8/10/12-bit frames; overlays are 8-bit Y/Cb/Cr/alpha bitmaps either in the frame's chroma
 * subsampling (blend8on8 / blend8on1x, blend.c:425-604) or 4:4:4 on a subsampled frame
 * (blend_subsample_8on8 / _8on1x, :48-328, weights from the chroma location, common.c:7054-7091). */
typedef struct hbhip_blend hbhip_blend;
typedef struct hbhip_overlay
{
    const uint8_t *plane[4];          /* Y, Cb, Cr, alpha (host memory) */
    int            stride[4];
    int            x, y, width, height;   /* hb_buffer_t.f.x / .y / .width / .height of the overlay */
} hbhip_overlay;
/* hb_blend_init (:788-846); chroma_location is the AVCHROMA_LOC_* number */
int  hbhip_blend_create(hbhip_ctx *ctx, int width, int height, int depth, int log2_chroma_w, int log2_chroma_h,
                        int chroma_location, int overlay_log2_chroma_w, int overlay_log2_chroma_h, hbhip_blend **out);
/* Upload the current overlay list (call again only when it changes: rendersub's `changed`).
 * The bitmaps are consumed when this returns. */
int  hbhip_blend_set_overlays(hbhip_blend *b, const hbhip_overlay *ov, int n);
/* hb_blend_work (:848-873): composite the overlays, in order, onto the frame in place. */
int  hbhip_blend_apply(hbhip_blend *b, const hbhip_host_frame *frame);       /* H2D, blend, D2H; synchronous */
int  hbhip_blend_apply_dev(hbhip_blend *b, const hbhip_dev_frame *frame);    /* frame already in HBM */
void hbhip_blend_destroy(hbhip_blend *b);                                    /* hb_blend_close (:875-885) */

/* ---- test hook ------------------------------------------------------------------------------
 * The EEDI2 mask passes of a batch run as ONE launch whose tiles wait for the previous field's tiles (csrc/eedi2_engine.h:
 * MaskChain).  A wait that runs out never aborts: the launch ends, and a repair pass behind it recomputes the batch's
 * masks field by field.  spin_limit > 0: polls a wait makes before it gives up, for every mask launch from now on
 * (1 = at once, i.e. every launch takes the repair path); 0: back to the default (about a second).  Returns how many
 * mask launches have been repaired in this process so far. */
unsigned hbhip_debug_mask_chain(int spin_limit);

#ifdef __cplusplus
}
#endif
#endif /* HBHIP_H */
