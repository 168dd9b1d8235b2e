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
        case HB_FILTER_NLMEANS: return &hb_filter_nlmeans_hip;
        default:                return NULL;
    }
}

/* Address of the shared context, for the bench/test bindings. */
void *hbhip_host_ctx_ptr(void)
{
    return hbhip_host_ctx();
}
