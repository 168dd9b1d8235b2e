"""The golden-vector cases: (synthetic stream, reference filter chain).
make_golden.py runs them through the reference; tests replay them through the
oracle restatement (CPU) and the HIP drop-ins (GPU)."""

NLM_MEDIUM = ("y-strength=6:y-origin-tune=1:y-patch-size=7:y-range=3:y-frame-count=2:y-prefilter=0:"
              "cb-strength=6:cb-origin-tune=1:cb-patch-size=7:cb-range=3:cb-frame-count=2:cb-prefilter=0")
NLM_TAPE = ("y-strength=3:y-origin-tune=0.8:y-patch-size=3:y-range=5:y-frame-count=2:y-prefilter=0:"
            "cb-strength=6:cb-origin-tune=0.8:cb-patch-size=5:cb-range=5:cb-frame-count=2:cb-prefilter=0")
NLM_ANIM_LIGHT = ("y-strength=3:y-origin-tune=0.15:y-patch-size=5:y-range=7:y-frame-count=3:y-prefilter=0:"
                  "cb-strength=2.25:cb-origin-tune=0.15:cb-patch-size=5:cb-range=7:cb-frame-count=3:cb-prefilter=0")

NLM_PRE_A = ("y-strength=6:y-origin-tune=1:y-patch-size=7:y-range=3:y-frame-count=2:y-prefilter=272:"
             "cb-strength=6:cb-origin-tune=1:cb-patch-size=5:cb-range=3:cb-frame-count=2:cb-prefilter=8")
NLM_PRE_B = ("y-strength=5:y-origin-tune=0.5:y-patch-size=5:y-range=5:y-frame-count=3:y-prefilter=1537:"
             "cb-strength=4:cb-origin-tune=1:cb-patch-size=3:cb-range=3:cb-frame-count=1:cb-prefilter=2")
NLM_PRE_C = ("y-strength=6:y-origin-tune=1:y-patch-size=7:y-range=3:y-frame-count=2:y-prefilter=2049:"
             "cb-strength=6:cb-origin-tune=0.8:cb-patch-size=7:cb-range=3:cb-frame-count=2:cb-prefilter=1028:"
             "cr-strength=6:cr-origin-tune=0.8:cr-patch-size=7:cr-range=3:cr-frame-count=2:cr-prefilter=800")


def nlm(strength=6, origin_tune=1.0, patch=7, rng=3, nframes=2, prefilter=0, depth=8):
    return dict(strength=strength, origin_tune=origin_tune, patch=patch, range=rng,
                nframes=nframes, prefilter=prefilter, depth=depth)


def lap(strength=0.2, kernel="isolap", depth=8):
    return dict(strength=strength, kernel=kernel, depth=depth)


def blur(strength=0.25, size=7, depth=8):
    return dict(strength=strength, size=size, depth=depth)


COMB_DEFAULT = ("mode=3:spatial-metric=2:motion-thresh=1:spatial-thresh=1:filter-mode=2:"
                "block-thresh=40:block-width=16:block-height=16")          # param.c:204-207
COMB_DEFAULT_PAR = dict(mode=3, spatial_metric=2, motion_thresh=1, spatial_thresh=1, filter_mode=2,
                        block_thresh=40, block_width=16, block_height=16)

# name -> case.  `chain` = [(reference filter object symbol, settings)], `hip` = the drop-in symbols.
CASES = {
    "nlmeans_medium_96x64": dict(model="progressive", w=96, h=64, n=4,
                                 chain=[("hb_filter_nlmeans", NLM_MEDIUM + ":threads=2")],
                                 hip=[("hb_filter_nlmeans_hip", NLM_MEDIUM)],
                                 orc=[("nlmeans", [nlm(), nlm(), nlm()])]),
    "nlmeans_tape_70x50": dict(model="progressive", w=70, h=50, n=4,
                               chain=[("hb_filter_nlmeans", NLM_TAPE + ":threads=1")],
                               hip=[("hb_filter_nlmeans_hip", NLM_TAPE)],
                               orc=[("nlmeans", [nlm(3, 0.8, 3, 5, 2), nlm(6, 0.8, 5, 5, 2), nlm(6, 0.8, 5, 5, 2)])]),
    "nlmeans_animation_light_80x48": dict(model="random", w=80, h=48, n=5,
                                          chain=[("hb_filter_nlmeans", NLM_ANIM_LIGHT + ":threads=3")],
                                          hip=[("hb_filter_nlmeans_hip", NLM_ANIM_LIGHT)],
                                          orc=[("nlmeans", [nlm(3, 0.15, 5, 7, 3), nlm(2.25, 0.15, 5, 7, 3),
                                                            nlm(2.25, 0.15, 5, 7, 3)])]),
    # NLMeans prefilters (nlmeans_template.c:103-543).  threads=1: with more threads the reference
    # races on which frames are already prefiltered when src_pre is latched.
    "nlmeans_pre_csm3_median5_96x64": dict(model="progressive", w=96, h=64, n=4,
                                           chain=[("hb_filter_nlmeans", NLM_PRE_A + ":threads=1")],
                                           hip=[("hb_filter_nlmeans_hip", NLM_PRE_A)],
                                           orc=[("nlmeans", [nlm(6, 1.0, 7, 3, 2, 16 + 256), nlm(6, 1.0, 5, 3, 2, 8),
                                                             nlm(6, 1.0, 5, 3, 2, 8)])]),
    "nlmeans_pre_edgeboost_80x48": dict(model="random", w=80, h=48, n=4,
                                        chain=[("hb_filter_nlmeans", NLM_PRE_B + ":threads=1")],
                                        hip=[("hb_filter_nlmeans_hip", NLM_PRE_B)],
                                        orc=[("nlmeans", [nlm(5, 0.5, 5, 5, 3, 1 + 1024 + 512), nlm(4, 1.0, 3, 3, 1, 2),
                                                          nlm(4, 1.0, 3, 3, 1, 2)])]),
    "nlmeans_pre_passthru_70x50": dict(model="progressive", w=70, h=50, n=3,
                                       chain=[("hb_filter_nlmeans", NLM_PRE_C + ":threads=1")],
                                       hip=[("hb_filter_nlmeans_hip", NLM_PRE_C)],
                                       orc=[("nlmeans", [nlm(6, 1.0, 7, 3, 2, 2049), nlm(6, 0.8, 7, 3, 2, 4 + 1024),
                                                         nlm(6, 0.8, 7, 3, 2, 32 + 256 + 512)])]),
    # 16-bit samples (the _16 template instantiations, nlmeans.c:253-262): YUV420P10 / P12
    "nlmeans_medium_10bit_96x64": dict(model="progressive", w=96, h=64, n=4, depth=10,
                                       chain=[("hb_filter_nlmeans", NLM_MEDIUM + ":threads=2")],
                                       hip=[("hb_filter_nlmeans_hip", NLM_MEDIUM)],
                                       orc=[("nlmeans", [nlm(depth=10), nlm(depth=10), nlm(depth=10)])]),
    "nlmeans_tape_12bit_70x50": dict(model="random", w=70, h=50, n=3, depth=12,
                                     chain=[("hb_filter_nlmeans", NLM_TAPE + ":threads=1")],
                                     hip=[("hb_filter_nlmeans_hip", NLM_TAPE)],
                                     orc=[("nlmeans", [nlm(3, 0.8, 3, 5, 2, depth=12), nlm(6, 0.8, 5, 5, 2, depth=12),
                                                       nlm(6, 0.8, 5, 5, 2, depth=12)])]),
    "lapsharp_isolog_10bit_134x70": dict(model="progressive", w=134, h=70, n=2, depth=10,
                                         chain=[("hb_filter_lapsharp", "y-strength=1.1:y-kernel=isolog:cb-strength=0.4:cb-kernel=lap")],
                                         hip=[("hb_filter_lapsharp_hip", "y-strength=1.1:y-kernel=isolog:cb-strength=0.4:cb-kernel=lap")],
                                         orc=[("lapsharp", [lap(1.1, "isolog", 10), lap(0.4, "lap", 10), lap(0.4, "lap", 10)])]),
    "unsharp_12bit_96x64": dict(model="random", w=96, h=64, n=2, depth=12,
                                chain=[("hb_filter_unsharp", "y-strength=0.9:y-size=15:cb-strength=0.5:cb-size=5")],
                                hip=[("hb_filter_unsharp_hip", "y-strength=0.9:y-size=15:cb-strength=0.5:cb-size=5")],
                                orc=[("unsharp", [blur(0.9, 15, 12), blur(0.5, 5, 12), blur(0.5, 5, 12)])]),
    "chroma_smooth_10bit_134x70": dict(model="random", w=134, h=70, n=2, depth=10,
                                       chain=[("hb_filter_chroma_smooth", "cb-strength=0.8:cb-size=9")],
                                       hip=[("hb_filter_chroma_smooth_hip", "cb-strength=0.8:cb-size=9")],
                                       orc=[("chroma_smooth", [blur(0.8, 9, 10), blur(0.8, 9, 10)])]),
    "lapsharp_medium_134x70": dict(model="progressive", w=134, h=70, n=2,
                                   chain=[("hb_filter_lapsharp", "y-strength=0.2:y-kernel=isolap:cb-strength=0.2:cb-kernel=isolap")],
                                   hip=[("hb_filter_lapsharp_hip", "y-strength=0.2:y-kernel=isolap:cb-strength=0.2:cb-kernel=isolap")],
                                   orc=[("lapsharp", [lap(), lap(), lap()])]),
    "lapsharp_isolog_strong_128x64": dict(model="random", w=128, h=64, n=2,
                                          chain=[("hb_filter_lapsharp", "y-strength=1.3:y-kernel=isolog:cb-strength=0.6:cb-kernel=log")],
                                          hip=[("hb_filter_lapsharp_hip", "y-strength=1.3:y-kernel=isolog:cb-strength=0.6:cb-kernel=log")],
                                          orc=[("lapsharp", [lap(1.3, "isolog"), lap(0.6, "log"), lap(0.6, "log")])]),
    "unsharp_medium_134x70": dict(model="progressive", w=134, h=70, n=2,
                                  chain=[("hb_filter_unsharp", "y-strength=0.25:y-size=7:cb-strength=0.25:cb-size=7")],
                                  hip=[("hb_filter_unsharp_hip", "y-strength=0.25:y-size=7:cb-strength=0.25:cb-size=7")],
                                  orc=[("unsharp", [blur(), blur(), blur()])]),
    "unsharp_size15_96x64": dict(model="random", w=96, h=64, n=1,
                                 chain=[("hb_filter_unsharp", "y-strength=1.5:y-size=15:cb-strength=0.8:cb-size=3")],
                                 hip=[("hb_filter_unsharp_hip", "y-strength=1.5:y-size=15:cb-strength=0.8:cb-size=3")],
                                 orc=[("unsharp", [blur(1.5, 15), blur(0.8, 3), blur(0.8, 3)])]),
    "unsharp_sizes_5_9_190x96": dict(model="random", w=190, h=96, n=2,
                                     chain=[("hb_filter_unsharp", "y-strength=0.75:y-size=5:cb-strength=0.5:cb-size=9")],
                                     hip=[("hb_filter_unsharp_hip", "y-strength=0.75:y-size=5:cb-strength=0.5:cb-size=9")],
                                     orc=[("unsharp", [blur(0.75, 5), blur(0.5, 9), blur(0.5, 9)])]),
    "chroma_smooth_medium_134x70": dict(model="random", w=134, h=70, n=2,
                                        chain=[("hb_filter_chroma_smooth", "cb-strength=0.6:cb-size=7")],
                                        hip=[("hb_filter_chroma_smooth_hip", "cb-strength=0.6:cb-size=7")],
                                        orc=[("chroma_smooth", [blur(0.6, 7), blur(0.6, 7)])]),
    "decomb_default_134x70": dict(model="interlaced", w=134, h=70, n=4,
                                  chain=[("hb_filter_decomb", "mode=7")],
                                  hip=[("hb_filter_decomb_hip", "mode=7")],
                                  orc=[("decomb", dict(mode=7))]),
    "decomb_bob_128x64": dict(model="interlaced", w=128, h=64, n=4,
                              chain=[("hb_filter_decomb", "mode=23")],
                              hip=[("hb_filter_decomb_hip", "mode=23")],
                              orc=[("decomb", dict(mode=23))]),
    "combdetect_decomb_selective_192x96": dict(model="interlaced", w=192, h=96, n=5,
                                               chain=[("hb_filter_comb_detect", COMB_DEFAULT), ("hb_filter_decomb", "mode=39")],
                                               hip=[("hb_filter_comb_detect_hip", COMB_DEFAULT), ("hb_filter_decomb_hip", "mode=39")],
                                               orc=[("comb_detect", COMB_DEFAULT_PAR), ("decomb", dict(mode=39))]),
    "decomb_eedi2_bob_128x64": dict(model="interlaced", w=128, h=64, n=3,
                                    chain=[("hb_filter_decomb", "mode=31")],
                                    hip=[("hb_filter_decomb_hip", "mode=31")],
                                    orc=[("decomb", dict(mode=31))]),
    # 10 / 12-bit NLMeans with prefilters; threads=1 for the same reason as the 8-bit prefilter cases
    "nlmeans_prefilter_10bit_134x70": dict(model="progressive", w=134, h=70, n=4, depth=10,
                                           chain=[("hb_filter_nlmeans", NLM_PRE_A + ":threads=1")],
                                           hip=[("hb_filter_nlmeans_hip", NLM_PRE_A)],
                                           orc=[("nlmeans", [nlm(prefilter=272, depth=10), nlm(patch=5, prefilter=8, depth=10),
                                                             nlm(patch=5, prefilter=8, depth=10)])]),
    "nlmeans_prefilter_12bit_96x64": dict(model="random", w=96, h=64, n=4, depth=12,
                                          chain=[("hb_filter_nlmeans", NLM_PRE_C + ":threads=1")],
                                          hip=[("hb_filter_nlmeans_hip", NLM_PRE_C)],
                                          orc=[("nlmeans", [nlm(prefilter=2049, depth=12),
                                                            nlm(origin_tune=0.8, prefilter=1028, depth=12),
                                                            nlm(origin_tune=0.8, prefilter=800, depth=12)])]),
    # 10 / 12-bit EEDI2 (oracle eedi2_16_oracle.c, kernels csrc/eedi2_16.hip)
    "decomb_eedi2_bob_10bit_128x64": dict(model="interlaced", w=128, h=64, n=3, depth=10,
                                          chain=[("hb_filter_decomb", "mode=31")],
                                          hip=[("hb_filter_decomb_hip", "mode=31")],
                                          orc=[("decomb", dict(mode=31, depth=10))]),
    "decomb_eedi2_cubic_12bit_190x96": dict(model="interlaced", w=190, h=96, n=3, depth=12,
                                            chain=[("hb_filter_decomb", "mode=15:noise-thresh=30")],
                                            hip=[("hb_filter_decomb_hip", "mode=15:noise-thresh=30")],
                                            orc=[("decomb", dict(mode=15, noise=30, depth=12))]),
    "decomb_eedi2_only_190x96": dict(model="interlaced", w=190, h=96, n=3,
                                     chain=[("hb_filter_decomb", "mode=8")],
                                     hip=[("hb_filter_decomb_hip", "mode=8")],
                                     orc=[("decomb", dict(mode=8))]),
    "combdetect_decomb_selective_10bit_192x96": dict(model="interlaced", w=192, h=96, n=5, depth=10,
                                                     chain=[("hb_filter_comb_detect", COMB_DEFAULT), ("hb_filter_decomb", "mode=39")],
                                                     hip=[("hb_filter_comb_detect_hip", COMB_DEFAULT), ("hb_filter_decomb_hip", "mode=39")],
                                                     orc=[("comb_detect", dict(COMB_DEFAULT_PAR, depth=10)), ("decomb", dict(mode=39, depth=10))]),
    "combdetect_nogamma_12bit_128x64": dict(model="progressive", w=128, h=64, n=4, depth=12,
                                            chain=[("hb_filter_comb_detect", "mode=2:spatial-metric=0:motion-thresh=2:spatial-thresh=2:filter-mode=1:block-thresh=20:block-width=16:block-height=16"),
                                                   ("hb_filter_decomb", "mode=39")],
                                            hip=[("hb_filter_comb_detect_hip", "mode=2:spatial-metric=0:motion-thresh=2:spatial-thresh=2:filter-mode=1:block-thresh=20:block-width=16:block-height=16"),
                                                 ("hb_filter_decomb_hip", "mode=39")],
                                            orc=[("comb_detect", dict(mode=2, spatial_metric=0, motion_thresh=2, spatial_thresh=2, filter_mode=1,
                                                                      block_thresh=20, block_width=16, block_height=16, depth=12)),
                                                 ("decomb", dict(mode=39, depth=12))]),
    "decomb_default_10bit_134x70": dict(model="interlaced", w=134, h=70, n=4, depth=10,
                                        chain=[("hb_filter_decomb", "mode=7")],
                                        hip=[("hb_filter_decomb_hip", "mode=7")],
                                        orc=[("decomb", dict(mode=7, depth=10))]),
    "decomb_bob_12bit_128x64": dict(model="interlaced", w=128, h=64, n=4, depth=12,
                                    chain=[("hb_filter_decomb", "mode=23")],
                                    hip=[("hb_filter_decomb_hip", "mode=23")],
                                    orc=[("decomb", dict(mode=23, depth=12))]),
    "decomb_cubic_10bit_96x64": dict(model="interlaced", w=96, h=64, n=3, depth=10,
                                     chain=[("hb_filter_decomb", "mode=4")],
                                     hip=[("hb_filter_decomb_hip", "mode=4")],
                                     orc=[("decomb", dict(mode=4, depth=10))]),
    "hqdn3d_10bit_134x70": dict(model="progressive", w=134, h=70, n=4, depth=10,
                                chain=[("hb_filter_denoise", "y-spatial=3:cb-spatial=2:cr-spatial=0:y-temporal=2:cb-temporal=3:cr-temporal=4")],
                                hip=[("hb_filter_denoise_hip", "y-spatial=3:cb-spatial=2:cr-spatial=0:y-temporal=2:cb-temporal=3:cr-temporal=4")],
                                orc=[("hqdn3d", dict(y_spatial=3, cb_spatial=2, cr_spatial=0, y_temporal=2, cb_temporal=3, cr_temporal=4, depth=10))]),
    "hqdn3d_12bit_96x64": dict(model="random", w=96, h=64, n=3, depth=12,
                               chain=[("hb_filter_denoise", "y-spatial=4")],
                               hip=[("hb_filter_denoise_hip", "y-spatial=4")],
                               orc=[("hqdn3d", dict(y_spatial=4, depth=12))]),
    "hqdn3d_medium_134x70": dict(model="progressive", w=134, h=70, n=4,
                                 chain=[("hb_filter_denoise", "y-spatial=3:cb-spatial=2:cr-spatial=2:y-temporal=2:cb-temporal=3:cr-temporal=3")],
                                 hip=[("hb_filter_denoise_hip", "y-spatial=3:cb-spatial=2:cr-spatial=2:y-temporal=2:cb-temporal=3:cr-temporal=3")],
                                 orc=[("hqdn3d", dict(y_spatial=3, cb_spatial=2, cr_spatial=2, y_temporal=2, cb_temporal=3, cr_temporal=3))]),
}
