// metal_wrap.h — runs the reference's own Metal compute shaders on the CPU (TEST INFRASTRUCTURE, never shipped).
//
// The alias family's arithmetic lives in FFmpeg / zimg, which are not in the image: its oracles are restatements of
// published algorithms and stay "parity unpinned" (SURVEY 8c).  The reference tree does hold a second implementation of
// four of those filters, though: the compute shaders of its VideoToolbox pipeline (libhb/platform/macosx/shaders/
// grayscale_vt.metal, yadif_vt.metal, bwdif_vt.metal, pad_vt.metal - ports of vf_monochrome / vf_yadif / vf_bwdif /
// vf_pad).  The Metal Shading Language is C++14 with vector types, textures and a few attributes, so those files compile
// as host C++ against a small stand-in library (oracle/shim/metal/: metal_stdlib, metal_texture) WHERE THEY LIE, unmodified,
// and a loop over the grid plays the GPU.  What that gives is an independent implementation from the reference tree to
// hold the restatements against - structure (which rows, which frames, which neighbours, the edge rules) and values
// within the shaders' own arithmetic: they compute in float on samples normalised to [0, 1] where the FFmpeg filters
// compute in integers, so `(a + b) / 2` is not truncated and the result is rounded once at the texture write.  Agreement
// is therefore expected within one code value, not bit for bit; tests/test_metal_cpu.py says what is compared and how
// close it is.  It does NOT pin the restatements to FFmpeg.
#pragma once
#include <metal_stdlib>
#include <metal_integer>
#include <metal_texture>

#define HBMTL_EXPORT extern "C" __attribute__((visibility("default")))

static inline metal::texture_plane hbmtl_plane(const void *data, int pitch, int w, int h)
{
    metal::texture_plane p = { (uint8_t *)data, pitch, w, h, 1 };
    return p;
}
