"""handbrake_amd — MI355X-native (HIP / gfx950) drop-ins for libhb's per-pixel
video-filter hot path (NLMeans, decomb/EEDI2, comb-detect, crop/scale,
lapsharp/unsharp, chroma-smooth, hqdn3d, grayscale, rotate).

The product is native code:

* ``libhbhip.so``          - HIP kernels + the C ABI of ``include/hbhip.h``
* ``libhbhip_filters.so``  - C ``hb_filter_object_t`` drop-ins (``libhb/*_hip.c``)
* ``libhbrt.so``           - stand-in for libhb's runtime + filter-chain driver,
                             used only when the filters run outside libhb

This Python package is plumbing for tests and ``bench.py``: ctypes bindings
(``hip``, ``hbrt``) and the synthetic stream generator (``synth``).  Nothing in
here computes pixels, and nothing in here falls back to a CPU path: if the HIP
library or a GPU is missing the bindings raise.
"""
__all__ = ["hip", "hbrt", "synth"]
