/* host_path_bench.c — frames/s of the hb_filter_object_t path (host hb_buffer_t in, host
 * hb_buffer_t out, i.e. PCIe included) driven from C through the chain harness, without the
 * Python test plumbing.  DESIGN.md quotes this next to bench.py's HBM-resident `value`.
 *
 *   cc -O2 -Iinclude -Ihandbrake_amd/libhb tools/host_path_bench.c -o tools/host_path_bench \
 *      -Lhandbrake_amd -lhbhip_filters -lhbhip -lhbrt -Wl,-rpath,$PWD/handbrake_amd
 *   tools/host_path_bench [nframes] [width] [height]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "hbhip_libhb.h"
#include "hb_harness.h"

extern hb_filter_object_t hb_filter_nlmeans_hip, hb_filter_lapsharp_hip, hb_filter_decomb_hip;

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void run(const char *name, void *proto, const char *settings, int n, int w, int h)
{
    void *protos[1] = { proto };
    const char *sets[1] = { settings };
    hbh_chain_t *c = hbh_chain_open(1, protos, sets, AV_PIX_FMT_YUV420P, w, h, 30000, 1001);
    if (!c) { printf("%s: init failed\n", name); return; }
    const int cw = (w + 1) / 2, ch = (h + 1) / 2;
    uint8_t *src[8][3];
    uint32_t s = 12345;
    for (int f = 0; f < 8; f++)
        for (int p = 0; p < 3; p++)
        {
            const size_t sz = (size_t)(p ? cw : w) * (p ? ch : h);
            src[f][p] = malloc(sz);
            for (size_t i = 0; i < sz; i++) { s = s * 1664525u + 1013904223u; src[f][p][i] = (uint8_t)(96 + ((s >> 24) & 31)); }
        }
    uint8_t *dst[3] = { malloc((size_t)w * h), malloc((size_t)cw * ch), malloc((size_t)cw * ch) };
    const int sstride[3] = { w, cw, cw };
    int out = 0;
    double t0 = 0, t_push = 0, t_pop = 0;
    for (int i = 0; i < n + 8; i++)
    {
        if (i == 8) { t0 = now(); out = 0; }                   /* first 8 frames: warm-up */
        const uint8_t *pl[3] = { src[i & 7][0], src[i & 7][1], src[i & 7][2] };
        double a = now();
        if (hbh_chain_push(c, pl, sstride, (int64_t)i * 3003, (int64_t)(i + 1) * 3003, 0x10, 0) != 0) break;
        double b = now();
        while (hbh_chain_pending(c) > 0)
        {
            hbh_frame_info_t info;
            if (hbh_chain_peek(c, &info) != 0) break;
            hbh_chain_pop(c, info.is_eof ? NULL : dst, info.is_eof ? NULL : sstride);
            if (!info.is_eof) out++;
        }
        if (i >= 8) { t_push += b - a; t_pop += now() - b; }
    }
    const double dt = now() - t0;
    printf("%-28s %dx%d: %8.1f frames/s through hb_filter_object_t (%d frames, %.2f ms each; work() %.2f ms, harness copy-out %.2f ms)\n",
           name, w, h, out / dt, out, 1e3 * dt / (out ? out : 1), 1e3 * t_push / (out ? out : 1), 1e3 * t_pop / (out ? out : 1));
    hbh_chain_close(c);
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 200;
    const int w = argc > 2 ? atoi(argv[2]) : 1920, h = argc > 3 ? atoi(argv[3]) : 1080;
    const char *only = argc > 4 ? argv[4] : "";
    if (!*only || !strcmp(only, "nlmeans"))
    run("nlmeans medium", &hb_filter_nlmeans_hip,
        "y-strength=6:y-origin-tune=1:y-patch-size=7:y-range=3:y-frame-count=2:y-prefilter=0:"
        "cb-strength=6:cb-origin-tune=1:cb-patch-size=7:cb-range=3:cb-frame-count=2:cb-prefilter=0", n, w, h);
    if (!*only || !strcmp(only, "lapsharp"))
    run("lapsharp", &hb_filter_lapsharp_hip, "y-strength=0.2:y-kernel=isolap:cb-strength=0.2:cb-kernel=isolap", n, w, h);
    if (!*only || !strcmp(only, "decomb"))
    run("decomb mode 7", &hb_filter_decomb_hip, "mode=7", n, w, h);
    if (!strcmp(only, "decomb4"))
    run("decomb mode 4", &hb_filter_decomb_hip, "mode=4", n, w, h);
    return 0;
}
