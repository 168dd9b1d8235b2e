/* hbhip_registry.c — registry of the HIP drop-in filter objects and the shared
 * device context they run on.
 *
 * hbhip_filter_get() is the counterpart of hb_filter_get() (common.c:5331-5495)
 * for the HIP-backed objects: same ids, so a job's filter list does not change.
 * Inside libhb the hook is the one the reference already has for its own GPU
 * filters: sanitize_filter_list_post() swaps CPU objects for GPU ones in place
 * (work.c:1515-1523, platform/macosx/vt_common.c:486-540); INTEGRATION.md shows
 * the equivalent hb_hip_setup_hw_filters().
 */
#include "hbhip_host.h"

#include <pthread.h>

static pthread_mutex_t g_ctx_lock = PTHREAD_MUTEX_INITIALIZER;
static hbhip_ctx      *g_ctx = NULL;
static int             g_ctx_refs = 0;

hbhip_ctx *hbhip_host_ctx(void)
{
    pthread_mutex_lock(&g_ctx_lock);
    if (g_ctx == NULL)
    {
        int device = 0;
        const char *env = getenv("HBHIP_DEVICE");
        if (env != NULL) device = atoi(env);
        int rc = hbhip_ctx_create(device, &g_ctx);
        if (rc != HBHIP_OK)
        {
            hb_error("hbhip: cannot create device context on GPU %d: %s", device, hbhip_strerror(rc));
            g_ctx = NULL;
        }
        else
        {
            char name[256];
            hbhip_ctx_device_name(g_ctx, name, sizeof(name));
            hb_log("hbhip: using GPU %d: %s", device, name);
        }
    }
    if (g_ctx != NULL) g_ctx_refs++;
    hbhip_ctx *c = g_ctx;
    pthread_mutex_unlock(&g_ctx_lock);
    return c;
}

void hbhip_host_ctx_release(void)
{
    pthread_mutex_lock(&g_ctx_lock);
    if (g_ctx != NULL && --g_ctx_refs <= 0)
    {
        /* keep the context alive for the life of the process: filters of later
         * jobs reuse it and HIP tears it down at exit */
        g_ctx_refs = 0;
    }
    pthread_mutex_unlock(&g_ctx_lock);
}

hb_filter_object_t *hbhip_filter_get(int filter_id)
{
    switch (filter_id)
    {
        case HB_FILTER_NLMEANS:       return &hb_filter_nlmeans_hip;
        case HB_FILTER_LAPSHARP:      return &hb_filter_lapsharp_hip;
        case HB_FILTER_UNSHARP:       return &hb_filter_unsharp_hip;
        case HB_FILTER_CHROMA_SMOOTH: return &hb_filter_chroma_smooth_hip;
        case HB_FILTER_DENOISE:       return &hb_filter_denoise_hip;
        case HB_FILTER_CROP_SCALE:    return &hb_filter_crop_scale_hip;
        case HB_FILTER_GRAYSCALE:     return &hb_filter_grayscale_hip;
        case HB_FILTER_ROTATE:        return &hb_filter_rotate_hip;
        case HB_FILTER_DECOMB:        return &hb_filter_decomb_hip;
        case HB_FILTER_COMB_DETECT:   return &hb_filter_comb_detect_hip;
        default:                return NULL;
    }
}

/* Address of the shared context, for the bench/test bindings. */
void *hbhip_host_ctx_ptr(void)
{
    return hbhip_host_ctx();
}

int hbhip_host_simple_work(hbhip_filter *dev, const hb_filter_init_t *output, const char *who,
                           hb_buffer_t **buf_in, hb_buffer_t **buf_out)
{
    hb_buffer_t *in = *buf_in;
    if (in->s.flags & HB_BUF_FLAG_EOF)
    {
        *buf_out = in;
        *buf_in = NULL;
        return HB_FILTER_DONE;
    }
    int ow = in->f.width, oh = in->f.height;
    hbhip_filter_out_geometry(dev, &ow, &oh);
    hb_buffer_t *out = hbhip_host_alloc_out(output, ow, oh);
    if (out == NULL)
        return HB_FILTER_FAILED;

    hbhip_host_frame fin, fout;
    hbhip_host_frame_from_buf(&fin, in);
    hbhip_host_frame_from_buf(&fout, out);
    int64_t tag;
    int rc = hbhip_filter_push(dev, &fin, 0);
    if (rc == HBHIP_OK)
        rc = hbhip_filter_pull(dev, &fout, &tag);
    if (rc != HBHIP_OK)
    {
        hb_error("%s(hip): %s", who, hbhip_strerror(rc));
        hb_buffer_close(&out);
        return HB_FILTER_FAILED;
    }
    hb_buffer_copy_props(out, in);
    *buf_out = out;
    return HB_FILTER_OK;
}
