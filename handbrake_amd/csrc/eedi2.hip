// eedi2.hip — EEDI2 (edge-directed interpolation of a field) for gfx950, 8-bit.
//
// The reference's passes (libhb/templates/eedi2_template.c), sequenced as
// eedi2_interpolate_plane does (libhb/templates/decomb_template.c:366-441), each
// launch covering the three planes of EVERY field of a batch (blockIdx.z = 3 * field +
// plane; Eedi2Engine at the end of the file).  A field's scratch frames live in HBM with
// the byte layout hb_frame_buffer_init gives them (fifo.c:820-881) between zeroed
// guards, because the reference's passes index a flat buffer and read a few bytes
// outside rows and planes (e.g. eedi2_template.c:395-447, 1194-1195); with the same
// layout the same bytes are read and the result is bit-identical.  The edge mask keeps
// state from field to field exactly as the reference's does (:132).
//
//   k_mask_fused4      eedi2_fill_half_height_buffer_plane :77-89, eedi2_build_edge_mask :122-195,
//                      erode / dilate_edge_mask :207-293, remove_small_gaps :308-342 — the field extraction and
//                      five passes, one LDS-tiled launch
//   k_calc_dir_rows    eedi2_calc_directions                 :358-525   (the time sink; k_calc_dir_mark / work
//                      is the fallback for search distances beyond the LDS halo)
//   k_dir_map4 / k_dir_map_c   eedi2_filter_dir_map / expand_dir_map :649-773 and the _2x forms :872-1011
//   k_filter_map       eedi2_filter_map                      :538-635
//   k_mark_2x4         eedi2_upscale_by_2 (x3) :98-108 + eedi2_mark_directions_2x :787-858
//   k_fill_gaps_b      eedi2_fill_gaps_2x                    :1025-1132
//   k_lattice_cand_q / k_lattice_resolve   eedi2_interpolate_lattice   :1148-1335
//   k_post             eedi2_post_process :1349-1378 (normally folded into the last expand_dir_map_2x; the
//                      eedi2_bit_blit before it, :46-68, into the dir-map filter that follows it)
//   k_blur1 / k_derivatives / k_blur_sqrt2 / k_post_corner   post-processing 2/3: eedi2_gaussian_blur1
//                      :1391-1527, eedi2_calc_derivatives :1760-1848, eedi2_gaussian_blur_sqrt2 :1539-1748,
//                      eedi2_post_process_corner :1864-1904
//
// interpolate_lattice rewrites its direction row in place and tests the value it
// just wrote at x-1 (:1194), a left-to-right dependency.  Each row is given to one
// workgroup: lanes evaluate both possible outcomes of their pixel in parallel and
// the chain is resolved with a prefix composition of 2-state maps, the carry running
// from chunk to chunk.
#include "eedi2_vote.h"
#include "eedi2_engine.h"

#include <atomic>
#include <algorithm>

namespace {

constexpr int PEAK = 255, NEUTRAL = 128;
constexpr int EEDI_MAX_FIELDS = EEDI_MAX_BATCH;   // fields per launch (P3::tffbits is one word)
constexpr size_t GUARD = 32768;   // >= 2 rows + halo of the widest plane the LDS staging may touch

__constant__ uint8_t c_limlut[33] = { 6, 6, 7, 7, 8, 8, 9, 9, 9, 10, 10, 11, 11, 12, 12, 12, 12, 12, 12, 12,
                                      12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 255, 255 };   // eedi2.c:21-25 stored as u8

// The table in LDS.  From constant memory a look-up is a vector memory load with a per-lane address - a round trip
// through the L1 in the middle of a pixel's dependency chain (behind its sort, ahead of its vote); from LDS it is one
// ds_read_u8.  Kernels with a barrier ahead of their first look-up fill one copy per workgroup (lim_fill + their
// barrier: lattice candidates 276 -> 272, expand_dir_map_2x 78 -> 75 us per launch); the 4-pixel dir-map kernels have no
// barrier and keep the constant table (a copy per wave, written and read by the wave itself, cost more than it saved).
constexpr int LIM_N = 33, LIM_PAD = 40;
__device__ __forceinline__ void lim_fill(uint8_t *tab, int tid)
{
    if (tid < LIM_N) tab[tid] = c_limlut[tid];
}

struct P3
{
    uint8_t *a[3];       // pass specific roles, see each kernel
    uint8_t *b[3];
    uint8_t *c[3];
    uint8_t *d[3];
    uint8_t *e[3];
    uint8_t *f[3];
    uint8_t *g[3];
    int pitch[3], width[3], height[3];   // height = rows of the buffers this pass walks
    size_t   fstride;    // field batching: the scratch buffers of consecutive fields of a launch lie this many bytes apart
    uint32_t tffbits;    // bit f: pv->tff of field f of the launch (the rebuilt rows start at y0 = 2 - tff)
    const uint32_t *pflags;   // [field][plane] of the launch: == pepoch when the plane's edge mask has a pixel set (MaskChain::pflags)
    uint32_t pepoch;
};

// A launch covers the three planes of every field of a batch: blockIdx.z = 3 * field + plane.  All scratch frames of
// a field sit in one slab (EediEngineBase::init_slots), so one offset moves every pointer of the pass to the block's field.
// The block's pointers are picked once into Q (scalar loads from the kernel arguments; P itself is never written -
// a dynamically indexed store would push the whole struct into scratch memory).
// `maskless`: the plane's edge mask is empty (no mask tile left a pixel set: MaskChain::pflags) - chroma without edges, a
// flat or letterboxed picture.  Every pass behind the mask only works on or next to mask pixels and copies (or fills) the
// rest, so for such a plane a pass is a copy, and its kernel takes the shortest way to it: the planes of the synthetic
// stream's chroma, a third of a field's pixels, cost 18 % of the decomb workload before (measured by leaving them out).
struct PL { uint8_t *a, *b, *c, *d, *e, *f, *g; };
__device__ __forceinline__ PL plane_ptrs(const P3 &P, int pl, size_t off)
{
    // only d is ever tested for "not bound" (the optional copy of the dir-map passes); the others are either bound or not
    // looked at by the pass, so they move without the test (a compare and two selects on the scalar unit per pointer and
    // wave - these passes run near the scalar issue rate, tools/salu_rate.hip)
    PL q = { P.a[pl] + off, P.b[pl] + off, P.c[pl] + off, P.d[pl], P.e[pl] + off, P.f[pl] + off, P.g[pl] + off };
    if (q.d) q.d += off;
    return q;
}
#define FIELD_PLANE(P)                                                      \
    const int fld = (int)blockIdx.z / 3, pl = (int)blockIdx.z - 3 * fld;    \
    const int tff = (int)(((P).tffbits >> fld) & 1u);                       \
    const PL Q = plane_ptrs((P), pl, (size_t)fld * (P).fstride);            \
    /* pflags: bound by every enqueue_passes path (never null); a mask launch's number is never 0 (enqueue_mask),  */ \
    /* so a flag that was only ever zeroed reads as "empty mask" - which is what no mask launch having set it means */ \
    const bool maskless = (P).pflags[blockIdx.z] != (P).pepoch;             \
    (void)tff; (void)maskless

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int hbhip_align_up_dev(int v, int a) { return (v + a - 1) / a * a; }

__device__ __forceinline__ int sad3(const uint8_t *a, int ai, const uint8_t *b, int bi)
{
    return iabs((int)a[ai - 1] - (int)b[bi - 1]) + iabs((int)a[ai] - (int)b[bi]) + iabs((int)a[ai + 1] - (int)b[bi + 1]);
}

// insertion sort + midpoint rule (eedi2.c:65-80)
__device__ __forceinline__ int sorted_mid(int *v, int n)
{
    for (int i = 1; i < n; i++)
    {
        const int t = v[i];
        int j = i;
        while (j > 0 && v[j - 1] > t) { v[j] = v[j - 1]; j--; }
        v[j] = t;
    }
    return (n & 1) ? v[n >> 1] : (v[(n - 1) >> 1] + v[n >> 1] + 1) >> 1;
}


// Register-only variant of the two helpers above for the dir-map kernels: the candidates sit in
// fixed slots (an absent one holds ABSENT, larger than any value and farther from any midpoint
// than any vote limit), a sorting network orders them, and the n present values are then the
// first n -- no data-dependent loop, no indexed register file.
constexpr int ABSENT = 1000;

__device__ __forceinline__ void cswap(int &a, int &b)
{
    const int lo = min(a, b), hi = max(a, b);
    a = lo; b = hi;
}

// midpoint of the n present values among 9 slots (n >= 4); the slots end up sorted
__device__ __forceinline__ int mid9(int &v0, int &v1, int &v2, int &v3, int &v4, int &v5, int &v6, int &v7, int &v8, int n)
{
    cswap(v0, v3); cswap(v1, v7); cswap(v2, v5); cswap(v4, v8);
    cswap(v0, v7); cswap(v2, v4); cswap(v3, v8); cswap(v5, v6);
    cswap(v0, v2); cswap(v1, v3); cswap(v4, v5); cswap(v7, v8);
    cswap(v1, v4); cswap(v3, v6); cswap(v5, v7);
    cswap(v0, v1); cswap(v2, v4); cswap(v3, v5); cswap(v6, v8);
    cswap(v2, v3); cswap(v4, v5); cswap(v6, v7);
    cswap(v1, v2); cswap(v3, v4); cswap(v5, v6);
    // n = 4..9: lower middle index (n-1)>>1 = 1,2,2,3,3,4 ; upper n>>1 = 2,2,3,3,4,4.  The lower one only counts for even n,
    // where it is the upper one's left neighbour: both come off the same two comparisons
    // (tests/test_eedi2_identities_cpu.py::test_midpoint_selection_and_unsorted_vote)
    const bool n5 = n <= 5, n7 = n <= 7;
    const int hi = n5 ? v2 : (n7 ? v3 : v4);
    const int lo = n5 ? v1 : (n7 ? v2 : v3);
    return (n & 1) ? hi : (lo + hi + 1) >> 1;
}

// midpoint of the n present values among 6 slots (n >= 3)
__device__ __forceinline__ int mid6(int &v0, int &v1, int &v2, int &v3, int &v4, int &v5, int n)
{
    cswap(v0, v5); cswap(v1, v3); cswap(v2, v4);
    cswap(v1, v2); cswap(v3, v4);
    cswap(v0, v3); cswap(v2, v5);
    cswap(v0, v1); cswap(v2, v3); cswap(v4, v5);
    cswap(v1, v2); cswap(v3, v4);
    // n = 3..6: lower middle index 1,1,2,2 ; upper 1,2,2,3
    const int lo = n <= 4 ? v1 : v2;
    const int hi = n <= 3 ? v1 : (n <= 5 ? v2 : v3);
    return (n & 1) ? hi : (lo + hi + 1) >> 1;
}

__device__ __forceinline__ void vote1(int v, int mid, int lim, int &sum, int &cnt)
{
    // |v - mid| as one v_sad_u16 (both are below 2^16 with empty upper halves); never within lim for ABSENT (lim <= 255)
    const bool in = (int)__builtin_amdgcn_sad_u16((uint32_t)v, (uint32_t)mid, 0u) <= lim;
    cnt += in;
    sum += in ? v : 0;
}

#define XY_PLANE(P)                                                         \
    FIELD_PLANE(P);                                                         \
    const int x = blockIdx.x * blockDim.x + threadIdx.x;                    \
    const int y = blockIdx.y * blockDim.y + threadIdx.y;                    \
    const int pitch = (P).pitch[pl], width = (P).width[pl], height = (P).height[pl]; \
    (void)width; (void)height; (void)pitch

// ------------------------------------------------------------------------------------------
// Four pixels per thread.  One byte per thread makes these passes latency bound: a wave lives
// for one memory round trip whatever it computes, so the time is (#waves / resident waves) x
// latency.  The *4 kernels below give a thread one aligned dword of its row and the dwords either
// side of it (bytes x-4 .. x+7) from each row it needs: 4x fewer waves, same bytes, same flat
// addressing (reads left of column 0 / right of the pitch land in the neighbouring row exactly
// as the reference's pointer arithmetic does).
struct Win12 { uint32_t w0, w1, w2; };

__device__ __forceinline__ Win12 ldwin(const uint8_t *row_at_x)
{
    const uint32_t *p = reinterpret_cast<const uint32_t *>(row_at_x);
    return Win12{p[-1], p[0], p[1]};
}

// byte at column x+i, i in [-4, 7] (compile-time after unrolling)
__device__ __forceinline__ int wb(const Win12 &w, int i)
{
    const int k = i + 4;
    const uint32_t d = k < 4 ? w.w0 : (k < 8 ? w.w1 : w.w2);
    return (int)((d >> (8 * (k & 3))) & 0xffu);
}

// store the 4 result bytes of columns x..x+3, of which only those < limit exist in the reference's loop
__device__ __forceinline__ void st4(uint8_t *dst_at_x, const int (&out)[4], int x, int limit)
{
    if (x + 3 < limit)
        *reinterpret_cast<uint32_t *>(dst_at_x) = (uint32_t)out[0] | ((uint32_t)out[1] << 8) | ((uint32_t)out[2] << 16) | ((uint32_t)out[3] << 24);
    else
        for (int k = 0; k < 4 && x + k < limit; k++) dst_at_x[k] = (uint8_t)out[k];
}

#define XY4_PLANE(P)                                                        \
    FIELD_PLANE(P);                                                         \
    const int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x);              \
    const int y = blockIdx.y * blockDim.y + threadIdx.y;                    \
    const int pitch = (P).pitch[pl], width = (P).width[pl], height = (P).height[pl]; \
    (void)width; (void)height; (void)pitch

// ------------------------------------------------------------------------------------------
// The five mask passes in one launch: build_edge_mask -> erode -> dilate -> erode ->
// remove_small_gaps (decomb_template.c:390-397) each only look one pixel (three along x for the
// last) around themselves, so a workgroup can carry a 128 x 16 tile of the final mask through all
// of them in LDS with a shrinking halo.  a = srcp, b = the PREVIOUS field's final mask (the rows of
// the lower half that build_edge_mask does not touch keep it, :146-150), c = the new mask.  The old
// and the new mask are different buffers (the engine alternates them) because neighbouring tiles
// read each other's halos.  Pixels a pass does not process keep their input, as in the reference.
// The tile's shape is a build parameter (tools/variant.sh; -DMF_TILE_W / MF_TILE_H / MF_STRIP_ROWS / MF_THREADS).  What runs
// since round 6 is 64 x 16 pixels on 256 threads; until then 128 x 16 on 512.  Measured on one box, the chain / decomb bob
// / the kernel alone (output fps / fps / us per 16 fields, profiles/r6Z_mask_tile_shapes.log): 128 x 16 x 512 8 617 / 12 725 /
// 117; 64 x 16 x 256 8 888 / 13 201 / 112-115; 128 x 8 x 256 8 824 / 13 373 / 112; 128 x 16 x 256 with four rows per
// thread 8 754 / 12 715 / 164 - a slower kernel and a faster chain: what the chain gains is not the kernel's own time but
// the 256-thread workgroup, which shares the CUs with the kernels of the other streams (the next part's mask beside this
// part's passes, decomb beside the other stages) where four 512-thread workgroups of a CU kept them out; 128 x 32, 128 x
// 24, 128 x 12, 128 x 8 on 512 threads, 256 x 8 and 64 x 8 lose (138-160 us: a taller tile is a longer link of the chain,
// a flatter one stages 2 x its pixels).
#ifndef MF_TILE_H
#define MF_TILE_H 16
#endif
#ifndef MF_STRIP_ROWS
#define MF_STRIP_ROWS 2
#endif
#ifndef MF_TILE_W
#define MF_TILE_W 64
#endif
constexpr int MF_W = MF_TILE_W, MF_H = MF_TILE_H, MF_OX = 8, MF_OY = 4;      // tile and the LDS frame's origin offset
constexpr int MF_LP = MF_W + 2 * MF_OX, MF_LR = MF_H + 2 * MF_OY; // the LDS frame: 80 x 24


// The passes work on dwords.  Every value of the mask is 0 or 255, so inside the kernel a mask
// pixel is one byte holding 0 / 1 and four of them are handled by one 32-bit operation: the 8-neighbour
// count of erode / dilate is a sum of byte-shifted dwords (at most 8 per byte, no carries), the
// threshold test one add (bit 7 of count + 0x80 - thr), remove_small_gaps a handful of ANDs / ORs of
// shifted dwords.  A thread owns one dword column of the LDS frame and a strip of 4 rows; it loads the
// 6 rows x 3 dwords around the strip once and keeps the per-row partial sums in registers.  Each pass
// computes the whole frame minus one more row top and bottom; the cells next to the frame's left / right
// edge come out wrong by design (they read the unwritten pad column), one byte further in per pass,
// which the 8-byte column halo absorbs (the tile needs x0 - 3 .. x0 + MF_W + 2 from the last erode).
constexpr int MF_DW = MF_LP / 4;                 // 20 dwords per LDS row
constexpr int MF_DP = MF_DW + 2;                 // + one pad dword either side
constexpr int MF_SR = MF_STRIP_ROWS;             // rows per thread and pass
// The chain link's round trips (build knob, bits: 1 = the look at the flags of the field before goes out in front of the
// tile's own loads, 2 = the edge tests are computed while the old mask is on its way, 4 = no look at the plane flag at the
// tile's end - a word per tile, folded into the plane flags by the pass behind the launch).  Measured, us per 16 fields
// (profiles/r6Z_mask_link.log): none 133-135, 1: 135-136, 2: 142-147, 1 + 2: 144, 4: 127 - the launch is not the pure chain
// of latencies the first two assume (its vector instructions alone are 93 us of issue); only the last one is on.
#ifndef MF_OPT
#define MF_OPT 4
#endif
#ifndef MF_THREADS
#define MF_THREADS 256
#endif
constexpr int MF_T = MF_THREADS;                // threads: 11 strips of MF_SR rows x 20 dword columns = 220 of them work in a pass
// (round 2, on the 128-pixel tile as a launch per field: 4 rows per thread and 256 threads 15.0 us per launch, 2 rows and
// 512 threads 12.6 us, 1 row and 1024 threads 12.6 us - with the all-fields launch of the upper part at 55 / 57 / 77 us
// per 16 fields)

// 0xff in byte k when lo <= X + k < hi
__device__ __forceinline__ uint32_t mf_bytes_in(int X, int lo, int hi)
{
    uint32_t m = 0xffffffffu;
    const int a = lo - X, b = hi - X;
    if (a > 0) m = a >= 4 ? 0u : (m << (8 * a));
    if (b < 4) m = b <= 0 ? 0u : (m & (0xffffffffu >> (8 * (4 - b))));
    return m;
}

// the four bytes at columns X .. X + 3 with those at or beyond `width` (the row's padding) replaced by padv's: the reference's
// memset(dstp, 255, pitch * height) writes the padding, its bit_blit of `width` columns does not
__device__ __forceinline__ uint32_t pad_bytes(uint32_t v, int X, int width, uint32_t padv)
{
    const uint32_t in = mf_bytes_in(X, 0, width);
    return (v & in) | (padv & ~in);
}

// erode (GROW = false) / dilate (GROW = true) of LDS rows ra .. rb
template <bool GROW>
__device__ __forceinline__ void mf_morph4(const uint32_t (*src)[MF_DP], uint32_t (*dst)[MF_DP], int c4, int strip,
                                          int ra, int rb, int thr, uint32_t px1, int fy, int height)
{
    const int r0 = ra + strip * MF_SR;
    if (r0 <= rb)
    {
        const uint32_t K = (uint32_t)(0x80 - min(max(thr, 0), 9)) * 0x01010101u;
        uint32_t S2[MF_SR + 2], S3[MF_SR + 2], C[MF_SR + 2];
#pragma unroll
        for (int i = 0; i < MF_SR + 2; i++)
        {
            const int r = min(r0 - 1 + i, MF_LR - 1);
            const uint32_t l = src[r][c4], c = src[r][c4 + 1], rr = src[r][c4 + 2];
            const uint32_t lb = __builtin_amdgcn_alignbyte(c, l, 3), rbv = __builtin_amdgcn_alignbyte(rr, c, 1);
            C[i] = c;
            S2[i] = lb + rbv;
            S3[i] = S2[i] + c;
        }
#pragma unroll
        for (int i = 0; i < MF_SR; i++)
        {
            const int r = r0 + i;
            if (r > rb) break;
            const int y = fy + r;
            const uint32_t count = S3[i] + S2[i + 1] + S3[i + 2];
            const uint32_t ge = ((count + K) >> 7) & 0x01010101u;          // count >= thr, per byte
            const uint32_t pm = (y >= 1 && y < height - 1) ? px1 : 0u;
            const uint32_t c = C[i + 1];
            dst[r][c4 + 1] = GROW ? (c | (ge & pm)) : (c & ~((ge ^ 0x01010101u) & pm));
        }
    }
    __syncthreads();
}

// The field extraction (eedi2_fill_half, decomb_template.c:455-473) rides along: the source rows are read from the
// frame itself (S.frame[field] = its planes, row start_line + 2y at pitch S.spitch[pl], start_line = !tff) and the
// tile's own part of SRCPF (P.a) is written for the passes that follow; bytes at x >= width read as 0, as fill_half
// writes them (device pictures have no row padding to drag along).
//
// Fields.  The previous field's mask only enters through the rows of the lower half, so the tiles whose LDS frame
// stays above height / 2 (`part` 1) are independent of it and go out in ONE launch for all fields of a batch; the
// rest (`part` 2) is a chain: one launch per field, each reading the mask the launch before it completed.  Field 0
// of a launch reads P.b (the last field of the previous batch), field f > 0 the new mask of field f - 1.
struct MaskSrc { const uint8_t *frame[EEDI_MAX_FIELDS][3]; int spitch[3]; };

// CHAIN: the tile is one of k_mask_chain's (MaskChain, eedi2_engine.h)
template <bool CHAIN>
__device__ __forceinline__ void mask_tile(const P3 &P, const MaskSrc &S, const MaskChain &C, int fld, int pl, int bx, int by,
                                          int mth, int vth, int lth, int erode_thr, int dilate_thr,
                                          uint32_t (*s_src)[MF_DP], uint32_t (*s_a)[MF_DP], uint32_t (*s_b)[MF_DP])
{
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    const int x0 = bx * MF_W, y0 = by * MF_H;
    const bool upper = y0 + MF_H + MF_OY <= height / 2;            // no row of the LDS frame reaches the kept half
    const size_t foff = (size_t)fld * P.fstride;
    const uint8_t *oldm = fld == 0 ? P.b[pl] : P.c[pl] + foff - P.fstride;
    const uint8_t *frame = S.frame[fld][pl];
    const int start_line = (int)(((P.tffbits >> fld) & 1u) ^ 1u);
    uint8_t *srcp = P.a[pl] + foff, *newm = P.c[pl] + foff;
    const int t = threadIdx.x, fx = x0 - MF_OX, fy = y0 - MF_OY;

    // The LDS frame is MF_LR x MF_DW = 480 dwords for 256 threads: two per thread (MF_LD).  All loads of a thread go out before
    // anything is done with the first (as a loop, the store of SRCPF between them made the second wait for the first:
    // two round trips in a row, and again for the old mask - on the chain's critical path from tile to tile).
    constexpr int MF_LD = (MF_LR * MF_DW + MF_T - 1) / MF_T;     // frame dwords per thread (two for the 16-row tile)
    static_assert(((MF_LR - 2 + MF_SR - 1) / MF_SR) * MF_DW <= MF_T, "a thread per strip and dword column");
    int fr[MF_LD], fc[MF_LD], fyy[MF_LD], fxx[MF_LD];
    bool fin[MF_LD];
    uint32_t sv[MF_LD];
    // a link of the chain: the look at the flags of the field before in front of the tile's own loads (a flag that is up
    // - the rule: that tile ran a field's worth of workgroups ago - then costs no round trip of its own)
    const uint32_t *cflag = nullptr;
    uint32_t cseen = 0;
    if (CHAIN && fld > 0)
    {
        cflag = eedi_chain_flag(C, fld, pl, bx, by);
        cseen = C.epoch + 1u;
        if ((MF_OPT & 1) && cflag) cseen = __hip_atomic_load(cflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int k = 0; k < MF_LD; k++)
    {
        const int i = t + MF_T * k;
        fin[k] = i < MF_LR * MF_DW;
        fr[k] = i / MF_DW; fc[k] = i - fr[k] * MF_DW;
        fyy[k] = fy + fr[k]; fxx[k] = fx + 4 * fc[k];
        sv[k] = 0;
        if (fin[k] && fyy[k] >= 0 && fyy[k] < height && fxx[k] >= 0 && fxx[k] < width)
            sv[k] = *reinterpret_cast<const uint32_t *>(frame + (size_t)(start_line + 2 * fyy[k]) * S.spitch[pl] + fxx[k]);
    }
#pragma unroll
    for (int k = 0; k < MF_LD; k++)
    {
        if (!fin[k]) continue;
        const int r = fr[k], c4 = fc[k], y = fyy[k], x = fxx[k];
        if (x + 3 >= width && x < width) sv[k] &= 0xffffffffu >> (8 * (x + 4 - width));
        // the tile's own cells go out as SRCPF (every cell of the plane belongs to exactly one tile)
        if (y >= 0 && y < height && x >= 0 && x < pitch &&
            r >= MF_OY && r < MF_OY + MF_H && c4 >= MF_OX / 4 && c4 < (MF_OX + MF_W) / 4)
            *reinterpret_cast<uint32_t *>(srcp + (size_t)y * pitch + x) = sv[k];
        s_src[r][c4 + 1] = sv[k];
    }
    if (CHAIN && fld > 0) eedi_chain_wait(C, cflag, cseen);        // (the source rows above are already on their way)
    else if ((MF_OPT & 2) || upper) __syncthreads();              // (s_src)
    uint32_t mv[MF_LD];
#pragma unroll
    for (int k = 0; k < MF_LD; k++)
    {
        mv[k] = 0;
        // (only the rows of the kept half are used, and those were written by lower tiles)
        if (fin[k] && !upper && fyy[k] >= 0 && fyy[k] < height && fxx[k] >= 0 && fxx[k] < pitch)
        {
            const uint32_t *m = reinterpret_cast<const uint32_t *>(oldm + (size_t)fyy[k] * pitch + fxx[k]);
            mv[k] = CHAIN ? __hip_atomic_load(m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *m;
        }
    }
    // an upper tile keeps nothing of the old mask, and with MF_OPT & 2 the edge tests (which never needed it) are computed
    // while it is on its way: s_src is complete here (the barrier above), the old mask goes to LDS behind them
    const bool edge_first = (MF_OPT & 2) || upper;
    if (!edge_first)
    {
#pragma unroll
        for (int k = 0; k < MF_LD; k++)
            if (fin[k]) s_a[fr[k]][fc[k] + 1] = mv[k] & 0x01010101u;
        __syncthreads();
    }

    const int c4 = t % MF_DW, strip = t / MF_DW;               // strips past the frame have no rows in any pass
    const int X = fx + 4 * c4;
    const uint32_t px1 = mf_bytes_in(X, 1, width - 1) & 0x01010101u;

    // build_edge_mask (:122-195), in place on the old mask; LDS rows 1 .. 22
    uint32_t edges[MF_SR];
#pragma unroll
    for (int i = 0; i < MF_SR; i++) edges[i] = 0;
    {
        const int r0 = 1 + strip * MF_SR;
        if (r0 <= MF_LR - 2)
        {
            int b[MF_SR + 2][6], q[MF_SR + 2][6];
#pragma unroll
            for (int i = 0; i < MF_SR + 2; i++)
            {
                const int r = min(r0 - 1 + i, MF_LR - 1);
                const uint32_t l = s_src[r][c4], c = s_src[r][c4 + 1], rr = s_src[r][c4 + 2];
                b[i][0] = (int)(l >> 24);
                b[i][1] = (int)(c & 0xffu); b[i][2] = (int)((c >> 8) & 0xffu); b[i][3] = (int)((c >> 16) & 0xffu); b[i][4] = (int)(c >> 24);
                b[i][5] = (int)(rr & 0xffu);
#pragma unroll
                for (int j = 0; j < 6; j++) q[i][j] = b[i][j] * b[i][j];
            }
#pragma unroll
            for (int i = 0; i < MF_SR; i++)
            {
                const int r = r0 + i;
                if (r > MF_LR - 2) break;
                const int y = fy + r;
                const int (&Pr)[6] = b[i], (&Cr)[6] = b[i + 1], (&Nr)[6] = b[i + 2];
                // the three samples of a column: all pairwise differences below 10 (:157-160) is max - min < 10, and the largest
                // pairwise difference (Iy, :172-173) is that same max - min: one v_max3 / v_min3 pair per column serves both
                // (tests/test_eedi2_identities_cpu.py::test_edge_mask_flatness_and_iy_are_one_range, ::test_edge_mask_laplacian_as_two_unsigned_sads)
                int cs[6], cq[6], rng[6];
                bool fl[6];
#pragma unroll
                for (int j = 0; j < 6; j++)
                {
                    cs[j] = Pr[j] + Cr[j] + Nr[j];
                    cq[j] = q[i][j] + q[i + 1][j] + q[i + 2][j];
                    rng[j] = max(max(Pr[j], Cr[j]), Nr[j]) - min(min(Pr[j], Cr[j]), Nr[j]);
                    fl[j] = rng[j] < 10;
                }
                uint32_t edge = 0;
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    // (no short-circuit: as `&&` / `||` the tests became 118 exec-mask branches per tile pair of the kernel)
                    const int sum = cs[k] + cs[k + 1] + cs[k + 2], sumsq = cq[k] + cq[k + 1] + cq[k + 2];
                    const int C0 = Cr[k], C1 = Cr[k + 1], C2 = Cr[k + 2];
                    const int ix = C2 - C0;
                    const int iy = rng[k + 1];
                    // |Ixx| + |Iyy| = |(C0 + C2) - 2 C1| + |(P1 + N1) - 2 C1|, both sides non-negative: two v_sad_u32
                    const uint32_t c2 = 2u * (uint32_t)C1;
                    uint32_t lap;
                    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(lap) : "v"((uint32_t)(cs[k + 1] - C1)), "v"(c2));
                    asm("v_sad_u32 %0, %1, %2, %0" : "+v"(lap) : "v"((uint32_t)(C0 + C2)), "v"(c2));
                    const bool notflat = !(fl[k + 1] | (fl[k] & fl[k + 2]));
                    const bool var = 9 * sumsq - sum * sum >= vth;
                    const bool mag = ix * ix + iy * iy >= mth;
                    const bool e = notflat & var & (mag | ((int)lap >= lth));
                    edge |= (e ? 1u : 0u) << (8 * k);
                }
                const uint32_t pm = (y >= 1 && y < height - 1) ? px1 : 0u;
                edges[i] = edge & pm;
                if (!edge_first)
                {
                    const uint32_t keep = (y < height / 2) ? 0u : s_a[r][c4 + 1];
                    s_a[r][c4 + 1] = keep | edges[i];
                }
            }
        }
    }
    if (edge_first)
    {
        // the old mask to LDS - every cell of the frame, also rows 0 and 23, which no edge test writes and the first erode
        // reads - then the edges on top of what is kept of it
        // (an upper tile: zeros in those two rows, the edges everywhere else - no cell is written twice, no barrier)
#pragma unroll
        for (int k = 0; k < MF_LD; k++)
            if (fin[k] && (!upper || fr[k] == 0 || fr[k] == MF_LR - 1)) s_a[fr[k]][fc[k] + 1] = upper ? 0u : mv[k] & 0x01010101u;
        if (!upper) __syncthreads();
        const int r0 = 1 + strip * MF_SR;
#pragma unroll
        for (int i = 0; i < MF_SR; i++)
        {
            const int r = r0 + i;
            if (r > MF_LR - 2) break;
            if (upper) { s_a[r][c4 + 1] = edges[i]; continue; }
            const uint32_t keep = (fy + r < height / 2) ? 0u : s_a[r][c4 + 1];
            s_a[r][c4 + 1] = keep | edges[i];
        }
    }
    __syncthreads();

    mf_morph4<false>(s_a, s_b, c4, strip, 2, MF_LR - 3, erode_thr, px1, fy, height);
    mf_morph4<true>(s_b, s_a, c4, strip, 3, MF_LR - 4, dilate_thr, px1, fy, height);
    mf_morph4<false>(s_a, s_b, c4, strip, 4, MF_LR - 5, erode_thr, px1, fy, height);

    // remove_small_gaps (:308-342) on the tile's 16 rows x 32 dwords, straight to the new mask
    uint32_t anyset = 0;
    for (int i = t; i < MF_H * (MF_W / 4); i += MF_T)
    {
        const int r = MF_OY + i / (MF_W / 4), g4 = MF_OX / 4 + (i & (MF_W / 4 - 1));
        const int y = fy + r, x = fx + 4 * g4;
        if (y >= height || x >= width) continue;
        const uint32_t l = s_b[r][g4], c = s_b[r][g4 + 1], rr = s_b[r][g4 + 2];
        const uint32_t a1 = __builtin_amdgcn_alignbyte(c, l, 3), a2 = __builtin_amdgcn_alignbyte(c, l, 2), a3 = __builtin_amdgcn_alignbyte(c, l, 1);
        const uint32_t b1 = __builtin_amdgcn_alignbyte(rr, c, 1), b2 = __builtin_amdgcn_alignbyte(rr, c, 2), b3 = __builtin_amdgcn_alignbyte(rr, c, 3);
        const uint32_t a12 = a1 | a2, a123 = a12 | a3;
        const uint32_t set = c & (a123 | b1 | b2 | b3);                               // a set pixel survives with any neighbour set
        const uint32_t fill = ((b1 & a123) | (b2 & a12) | (b3 & a1)) & (c ^ 0x01010101u);
        const uint32_t pm = (y >= 1 && y < height - 1) ? (mf_bytes_in(x, 3, width - 3) & 0x01010101u) : 0u;
        const uint32_t res = (((set | fill) & pm) | (c & ~pm)) * 255u;
        anyset |= x + 3 < width ? res : res & (0xffffffffu >> (8 * (x + 4 - width)));
        uint8_t *d = newm + (size_t)y * pitch + x;
        if (CHAIN)
        {
            if (x + 3 < width) __hip_atomic_store(reinterpret_cast<uint32_t *>(d), res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else for (int k = 0; k < 4 && x + k < width; k++) __hip_atomic_store(d + k, (uint8_t)(res >> (8 * k)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        else if (x + 3 < width) *reinterpret_cast<uint32_t *>(d) = res;
        else for (int k = 0; k < 4 && x + k < width; k++) d[k] = (uint8_t)(res >> (8 * k));
    }
    // a plane with a mask pixel somewhere: say so (P3::pflags - the passes behind the mask take a shortcut for the others).
    // One thread of the tile, and only while the flag is not up yet: thousands of tiles storing to the same few words
    // serialise in the L2 (the mask launch went from 120 us to 2 ms with a store per wave).
    // (behind the tile's own flag: the next field's tiles wait for that one)
    const bool has = CHAIN ? eedi_chain_signal(C, fld, pl, bx, by, anyset != 0u) : (bool)__syncthreads_or(anyset != 0u);
    eedi_chain_note_has(C, fld, pl, bx, by, has);
}

__global__ __launch_bounds__(MF_T) void k_mask_fused4(P3 P, MaskSrc S, int f0, int part, int mth, int vth, int lth, int erode_thr, int dilate_thr,
                                                      uint32_t *pflags, uint32_t epoch)
{
    __shared__ uint32_t s_src[MF_LR][MF_DP];
    __shared__ uint32_t s_a[MF_LR][MF_DP];
    __shared__ uint32_t s_b[MF_LR][MF_DP];
    const int zf = (int)blockIdx.z / 3, pl = (int)blockIdx.z - 3 * zf, fld = f0 + zf;   // f0: first field of this launch
    const int x0 = blockIdx.x * MF_W, y0 = blockIdx.y * MF_H;
    if (x0 >= P.width[pl] || y0 >= P.height[pl]) return;
    const bool upper = y0 + MF_H + MF_OY <= P.height[pl] / 2;
    if (part != 0 && upper != (part == 1)) return;
    MaskChain none;
    none.pflags = pflags; none.epoch = epoch; none.has = nullptr;
    mask_tile<false>(P, S, none, fld, pl, (int)blockIdx.x, (int)blockIdx.y, mth, vth, lth, erode_thr, dilate_thr, s_src, s_a, s_b);
}

// blockIdx.x = field * C.ntiles + tile: field-major, see MaskChain
__global__ __launch_bounds__(MF_T) void k_mask_chain(P3 P, MaskSrc S, MaskChain C, int mth, int vth, int lth, int erode_thr, int dilate_thr)
{
    __shared__ uint32_t s_src[MF_LR][MF_DP];
    __shared__ uint32_t s_a[MF_LR][MF_DP];
    __shared__ uint32_t s_b[MF_LR][MF_DP];
    int fld, pl, bx, by;
    if (eedi_chain_tile(C, fld, pl, bx, by))                      // block-uniform
        mask_tile<true>(P, S, C, fld, pl, bx, by, mth, vth, lth, erode_thr, dilate_thr, s_src, s_a, s_b);
    else
        mask_tile<false>(P, S, C, fld, pl, bx, by, mth, vth, lth, erode_thr, dilate_thr, s_src, s_a, s_b);
}

// Queued behind every k_mask_chain launch, a workgroup per field and plane.
//  * Each folds the words its plane's tiles left in C.has into the plane flag (P3::pflags: the passes behind the mask take
//    a shortcut for a plane without a mask pixel) - the tiles themselves no longer look at the flag, which cost every
//    workgroup of the chain a round trip at its end.
//  * Workgroup 0 then has nothing to do unless a wait of the chain ran out (C.err).  Then the launch's lower tiles are
//    recomputed field after field, tile after tile, by this one workgroup - program order is the dependency order, every
//    flag already carries the epoch (each tile publishes itself, timed out or not), so the waits inside mask_tile<true>
//    pass at once - from the same sources with the same arithmetic: the new masks end up as the per-field launches would
//    have left them.  (The upper tiles and SRCPF never depended on another field.)  A plane flag that went up for a mask the
//    repair empties only costs the shortcut.
__global__ __launch_bounds__(MF_T) void k_mask_chain_repair(P3 P, MaskSrc S, MaskChain C, int nfields, int mth, int vth, int lth,
                                                            int erode_thr, int dilate_thr)
{
    __shared__ uint32_t s_src[MF_LR][MF_DP];
    __shared__ uint32_t s_a[MF_LR][MF_DP];
    __shared__ uint32_t s_b[MF_LR][MF_DP];
    eedi_chain_fold_has(C, MF_T);
    if (blockIdx.x != 0 || __hip_atomic_load(C.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;      // block-uniform
    MaskChain R = C;
    R.has = nullptr;                                               // the repaired tiles raise the plane flags themselves
    for (int fld = 0; fld < nfields; fld++)
        for (int tile = 0; tile < C.ntiles; tile++)
        {
            int pl, bx, by;
            eedi_chain_lower_tile(C, tile, pl, bx, by);
            mask_tile<true>(P, S, R, fld, pl, bx, by, mth, vth, lth, erode_thr, dilate_thr, s_src, s_a, s_b);
            __syncthreads();
        }
    if (threadIdx.x == 0)
    {
        __hip_atomic_store(C.err, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(C.fallbacks, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// calc_directions in two launches so that no lane idles while its neighbour walks the
// +-maxd search: k_calc_dir_mark fills the plane with 255 (the reference's memset) and
// appends every pixel that passes the edge test (:392-393) to a work list; k_calc_dir_work
// gives each listed pixel its own lane.  a = mskp, b = srcp, c = out (tmpp).
__global__ void k_calc_dir_mark(P3 P, uint32_t *__restrict__ list, int *__restrict__ count)
{
    XY_PLANE(P);
    if (x >= pitch || y >= height) return;
    Q.c[(size_t)y * pitch + x] = 255;                      // memset(dstp, 255, pitch*height)
    if (x >= 1 && x < width - 1 && y >= 1 && y < height - 1)
    {
        const uint8_t *mc = Q.a + (size_t)y * pitch;
        if (mc[x] == PEAK && (mc[x - 1] == PEAK || mc[x + 1] == PEAK))
            list[atomicAdd(count, 1)] = (uint32_t)x | ((uint32_t)y << 14) | ((uint32_t)pl << 28);
    }
}

__global__ __launch_bounds__(256) void k_calc_dir_work(P3 P, const uint32_t *__restrict__ list,
                                                       const int *__restrict__ count, int maxd, int nt13, int nt19)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= *count) return;
    const uint32_t e = list[gid];
    const int x = e & 0x3fff, y = (e >> 14) & 0x3fff, pl = e >> 28;
    const PL Q = plane_ptrs(P, pl, 0);                           // one field per launch (see Eedi2Engine::enqueue_passes)
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    const uint8_t *mc = Q.a + (size_t)y * pitch;
    const uint8_t *mp = mc - pitch, *mn = mc + pitch;
    const uint8_t *sc = Q.b + (size_t)y * pitch;
    const uint8_t *sp = sc - pitch, *sn = sc + pitch, *s2p = sp - pitch, *s2n = sn + pitch;
    const int maxdt = pl == 0 ? maxd : (maxd >> 1);
    const int startu = max(-x + 1, -maxdt), stopu = min(width - 2 - x, maxdt);
    const int vert = iabs((int)sc[x] - (int)sn[x]) + iabs((int)sc[x] - (int)sp[x]);
    int minb = min(nt13, vert * 6), mina = min(nt19, vert * 9);
    int minc = mina, mind = minb, mine = minb;
    int dira = -5000, dirb = -5000, dirc = -5000, dird = -5000, dire = -5000;
    for (int u = startu; u <= stopu; u++)
    {
        if (!(y == 1 || mp[x - 1 + u] == PEAK || mp[x + u] == PEAK || mp[x + 1 + u] == PEAK)) continue;
        if (!(y == height - 2 || mn[x - 1 - u] == PEAK || mn[x - u] == PEAK || mn[x + 1 - u] == PEAK)) continue;
        const int diffsn = sad3(sc, x, sn, x - u);
        const int diffsp = sad3(sc, x, sp, x + u);
        const int diffps = sad3(sp, x, sc, x - u);
        const int diffns = sad3(sn, x, sc, x + u);
        const int diff = diffsn + diffsp + diffps + diffns;
        int diffd = diffsp + diffns, diffe = diffsn + diffps;
        if (diff < minb) { dirb = u; minb = diff; }
        if (y > 1)
        {
            const int diff2pp = sad3(s2p, x, sp, x - u);
            const int diffp2p = sad3(sp, x, s2p, x + u);
            const int diffa = diff + diff2pp + diffp2p;
            diffd += diffp2p;
            diffe += diff2pp;
            if (diffa < mina) { dira = u; mina = diffa; }
        }
        if (y < height - 2)
        {
            const int diff2nn = sad3(s2n, x, sn, x + u);
            const int diffn2n = sad3(sn, x, s2n, x - u);
            const int diffc = diff + diff2nn + diffn2n;
            diffd += diff2nn;
            diffe += diffn2n;
            if (diffc < minc) { dirc = u; minc = diffc; }
        }
        if (diffd < mind) { dird = u; mind = diffd; }
        if (diffe < mine) { dire = u; mine = diffe; }
    }
    int order[5], k = 0;
    if (dira != -5000) order[k++] = dira;
    if (dirb != -5000) order[k++] = dirb;
    if (dirc != -5000) order[k++] = dirc;
    if (dird != -5000) order[k++] = dird;
    if (dire != -5000) order[k++] = dire;
    int out = NEUTRAL;
    if (k > 1)
    {
        const int mid = sorted_mid(order, k);
        const int tlim = max((int)c_limlut[iabs(mid)] >> 2, 2);
        int sum = 0, cnt = 0;
        for (int i = 0; i < k; i++)
            if (iabs(order[i] - mid) <= tlim) { cnt++; sum += order[i]; }
        if (cnt > 1) out = (NEUTRAL + ((int)((float)sum / (float)cnt) * 4)) & 0xff;
    }
    Q.c[(size_t)y * pitch + x] = (uint8_t)out;
}

// calc_directions, fast path (search distance <= 30).  The source and mask rows the search touches are staged in LDS
// (with the same flat addressing, so out-of-row offsets pick up the same bytes), the pixels that pass the edge test are
// compacted inside the block so that busy lanes are contiguous, and each listed pixel walks its +-maxd window out of LDS.
#ifdef HBHIP_DEV_STATS
// development builds with -DHBHIP_DEV_STATS: what the search schedule really runs (tools/exp_knobs.sh prints it with HBHIP_DEV_STATS=1).
// 0 workgroups past the early exit, 1 workgroups with listed pixels, 2 listed pixels, 3 wave trips of the list loop,
// 4 steps of all lanes (useful), 5 wave-steps (the longest lane of each wave trip), 6 lanes in wave trips,
// 7 workgroups in the dense form, 8 their waves with listed pixels, 9 those of them without a step to leave out
__device__ unsigned long long g_cd_stats[24];
#define CD_STAT(i, v) atomicAdd(&g_cd_stats[i], (unsigned long long)(v))
#else
#define CD_STAT(i, v) ((void)0)
#endif
constexpr int CD_W = 256, CD_HALO = 32, CD_LW = CD_W + 2 * CD_HALO;

// What the search loop does per step is stripped to what has to happen per step:
//  * the 3-byte groups every step needs ("triples": bytes i..i+2 of a staged row, i = cc-1+-u) are formed ONCE per block
//    into LDS tables, one dword per column and source row - a step reads eight dwords at per-lane columns instead of
//    twenty and realigns nothing;
//  * the steps that can count at all (a mask peak among the three pixels above at +u AND below at -u, :395-399) are a
//    64-bit set per pixel cut from ballot bitmaps of the mask rows, and the loop walks its set bits only;
//  * each running minimum and its offset are ONE integer key: `if (sum < min) { min = sum; dir = u; }` with u ascending
//    is exactly min() on (sum << 16) | tag (ties keep the earlier = smaller u, the initial threshold is (thr << 16) | 0
//    so only strictly smaller sums get in, and tag 0 means "never set" = the reference's -5000).
// Values are the reference's (:358-525): same sums, same order of comparisons.

// A block takes 256 columns x R rows.
//  * Rows share their staging: R + 4 source and R + 2 mask rows (and the triple tables / peak bitmaps made from them)
//    serve R rows of pixels instead of 5 + 3 for one, and the masked pixels of R rows fill the waves of the search (one row
//    of this content leaves the last wave of a block's list three quarters empty).
//  * The list keeps column order: lanes of a wave read neighbouring table columns, so the eight table reads of a step stay
//    nearly free of LDS bank conflicts.  Ordering the list by the number of steps (counting sort, or a stable partition
//    into coarse bins) evens out the trip counts of a wave but scatters its columns: SQ_LDS_BANK_CONFLICT 1.6 M -> 4.2 M
//    cycles per launch and a slower kernel (57 / 55 against 52 us), with no fewer VALU instructions to show for it (the
//    search is a quarter of them) - measured, not kept.
//  * A running minimum and its offset are one integer built by v_sad_hi_u8 chains (see calc_dir_search), a step forms its
//    two table addresses with one instruction each: 31 VALU instructions per step against 45.
//  * The output rows are assembled in LDS and stored as dwords.
// maxd <= 30.  Per 1080i field of the bench's content: SQ_INSTS_VALU 16.7 M, SQ_INSTS_SALU 2.9 M, SQ_INSTS_LDS 2.1 M
// (one row per block without the key chains: 26.5 M / 11.3 M / 3.2 M).

// The vote over the five offsets (:486-517).  ta..te: 0 = never set (the reference's -5000), else u + 32.
__device__ __forceinline__ int calc_dir_vote(int ta, int tb, int tc, int td, int te)
{
    // the offsets that were set, sorted (unset ones as a large sentinel at the end): 9-exchange network on 5 values
    constexpr int BIG = 1 << 20;
    int v0 = ta ? ta - 32 : BIG, v1 = tb ? tb - 32 : BIG, v2 = tc ? tc - 32 : BIG, v3 = td ? td - 32 : BIG, v4 = te ? te - 32 : BIG;
    const int k = (ta != 0) + (tb != 0) + (tc != 0) + (td != 0) + (te != 0);
#define CD_CX(a, b) { const int lo_ = min(a, b), hi_ = max(a, b); a = lo_; b = hi_; }
    CD_CX(v0, v1) CD_CX(v3, v4) CD_CX(v2, v4) CD_CX(v2, v3) CD_CX(v0, v3) CD_CX(v0, v2) CD_CX(v1, v4) CD_CX(v1, v3) CD_CX(v1, v2)
#undef CD_CX
    int out = NEUTRAL;
    if (k > 1)
    {
        // sorted_mid's midpoint rule (eedi2.c:65-80): odd k -> v[k/2], even k -> (v[(k-1)/2] + v[k/2] + 1) >> 1; one formula
        // serves both ((2v + 1) >> 1 == v)
        const int lo = k == 2 ? v0 : (k == 5 ? v2 : v1);        // v[(k-1)>>1]: 0 1 1 2 for k = 2 3 4 5
        const int hi = k >= 4 ? v2 : v1;                         // v[k>>1]:     1 1 2 2
        const int mid = (lo + hi + 1) >> 1;
        // max(limlut[|mid|] >> 2, 2) without the table (a dependent load per pixel): |mid| <= 30 here, and the table
        // (eedi2.c:21-25) is 12 from entry 13 on and below 12 before it - 3 and 2 after the shift and the max
        // (tests/test_eedi2_identities_cpu.py::test_calc_directions_vote_limit_closed_form)
        static_assert(CD_HALO - 2 <= 30, "the closed form of limlut covers entries 0..30");
        const int tlim = iabs(mid) >= 13 ? 3 : 2;
        int sum = 0, cnt = 0;
        if (iabs(v0 - mid) <= tlim) { cnt++; sum += v0; }        // the sentinels fail the test by themselves
        if (iabs(v1 - mid) <= tlim) { cnt++; sum += v1; }
        if (iabs(v2 - mid) <= tlim) { cnt++; sum += v2; }
        if (iabs(v3 - mid) <= tlim) { cnt++; sum += v3; }
        if (iabs(v4 - mid) <= tlim) { cnt++; sum += v4; }
        if (cnt > 1) out = (NEUTRAL + ((int)((float)sum / (float)cnt) * 4)) & 0xff;
    }
    return out;
}

// PASS = uint64_t, or uint32_t when the window fits (chroma: 2 * 12 + 1 steps): walking the set bits of one word costs
// four vector instructions a step instead of nine.
template <bool EDGE, typename PASS>
__device__ __forceinline__ int calc_dir_search(const uint32_t *tr, PASS pass, int maxdt, bool first, bool last, int nt13, int nt19)
{
    // tr = &s_tri[j][b]: row k of the table is k * CD_LW further (k = 0..4: rows y-2 .. y+2)
    const uint32_t F2p = tr[0], Fp = tr[CD_LW], Fc = tr[2 * CD_LW], Fn = tr[3 * CD_LW], F2n = tr[4 * CD_LW];
    const int ctr = (int)((Fc >> 8) & 0xff);
    const int vert = iabs(ctr - (int)((Fn >> 8) & 0xff)) + iabs(ctr - (int)((Fp >> 8) & 0xff));
    // Keys: (running minimum << 16) | tag, tag = u + 32 (twice that for key a), tag 0 = unset.  `if (sum < min) { min = sum;
    // dir = u; }` with u ascending is min() on that key: the initial key is (threshold << 16) | 0, so only strictly smaller
    // sums get in, and among equal sums the smaller tag = the earlier u stays.  v_sad_hi_u8 adds its sum already shifted
    // by 16, so a key is a chain of them started from the tag; the tags are placed so that every key collects exactly one
    // (d1, s2pp carry it; key a = d1 + e1 + s2pp + sp2p gets two).  Sums stay below 2^16 (at most 20 bytes).
    uint32_t kb = (uint32_t)min(nt13, vert * 6) << 16, ka = (uint32_t)min(nt19, vert * 9) << 16;
    uint32_t kc = ka, kd = kb, ke = kb;
    // LDS byte addresses of columns b+u / b-u at jj = u + maxdt = 0, kept opaque so that a step forms its two addresses
    // and its tag with one instruction each
    typedef __attribute__((address_space(3))) const uint32_t *lds_u32;
    uint32_t ap0 = (uint32_t)(uintptr_t)(lds_u32)(tr - maxdt), am0 = (uint32_t)(uintptr_t)(lds_u32)(tr + maxdt);
    uint32_t tag0 = (uint32_t)(32 - maxdt);
    asm volatile("" : "+v"(ap0), "+v"(am0), "+s"(tag0));
#define SADH(a, b, acc) __builtin_amdgcn_sad_hi_u8((a), (b), (acc))
    while (pass)
    {
        const int jj = sizeof(PASS) == 8 ? __ffsll((unsigned long long)pass) - 1 : __ffs((unsigned int)pass) - 1;
        pass &= pass - (PASS)1;
        const uint32_t tag = (uint32_t)jj + tag0;
        const lds_u32 tp = (lds_u32)(uintptr_t)(ap0 + 4u * (uint32_t)jj), tm = (lds_u32)(uintptr_t)(am0 - 4u * (uint32_t)jj);
        const uint32_t e1 = SADH(Fp, tm[2 * CD_LW], SADH(Fc, tm[3 * CD_LW], 0u));     // diffsn + diffps
        const uint32_t d1 = SADH(Fn, tp[2 * CD_LW], SADH(Fc, tp[1 * CD_LW], tag));    // diffsp + diffns (+ tag)
        const uint32_t diff = e1 + d1;
        uint32_t diffd = d1, diffe = e1;
        kb = min(kb, diff);
        if (!EDGE || !first)
        {
            const uint32_t diff2pp = SADH(F2p, tm[1 * CD_LW], tag);
            const uint32_t diffp2p = SADH(Fp, tp[0 * CD_LW], 0u);
            diffd += diffp2p;
            diffe += diff2pp;
            ka = min(ka, diff + diff2pp + diffp2p);
        }
        else
            diffe += tag;
        if (!EDGE || !last)
        {
            const uint32_t diff2nn = SADH(F2n, tp[3 * CD_LW], 0u);
            const uint32_t diffn2n = SADH(Fn, tm[4 * CD_LW], 0u);
            diffd += diff2nn;
            diffe += diffn2n;
            kc = min(kc, diff + diff2nn + diffn2n);
        }
        kd = min(kd, diffd);
        ke = min(ke, diffe);
    }
#undef SADH
    return calc_dir_vote((int)((ka & 0xffffu) >> 1), (int)(kb & 0xffffu), (int)(kc & 0xffffu), (int)(kd & 0xffffu), (int)(ke & 0xffffu));
}

// The dense form of the search.  The reference clears only the upper half of the edge mask before it marks a field's
// edges (eedi2_template.c:132, `height / 2`), so the lower half accumulates: after a few fields of moving content
// nearly every pixel there is listed and takes every one of its 2 * maxd + 1 steps (measured on the bench's stream:
// 65 % of a luma field listed, 39 of 49 steps per listed pixel, 97 % of all steps in the lower half).  Walking bit
// sets buys nothing there, and the eight 3-byte SADs of a step are shared: with
//     E_r(x, u) = sum_k |row_r[x + k] - row_{r+1}[x + k - u]|,   k = -1..1
// the reference's sums (:402-470) are
//     diffsn = E_y(x, u)      diffns = E_y(x + u, u)      diffps = E_{y-1}(x, u)    diffsp = E_{y-1}(x + u, u)
//     diff2pp = E_{y-2}(x, u) diffp2p = E_{y-2}(x + u, u) diffn2n = E_{y+1}(x, u)   diff2nn = E_{y+1}(x + u, u)
// so a thread that owns column x of R rows needs P_r = E_r(x, u) and Q_r = E_r(x + u, u) for the R + 3 row pairs around
// its rows: 2 (R + 3) SADs serve R pixels (2.75 a pixel at R = 8 instead of 8), its own triples stay in registers, the
// other side comes from the table at column x -+ u - which u and -u read alike, so a trip of the loop takes the pair and
// the ten keys of a pixel take two candidates with one v_min3_u32 each.  With S = P + Q:
//     diff (b) = S_{y-1} + S_y     diffa = diff + S_{y-2}     diffc = diff + S_{y+1}
//     diffe = P_{y-2} + .. + P_{y+1}     diffd = Q_{y-2} + .. + Q_{y+1}
// Keys as in calc_dir_search, the tag riding in every P and Q (v_sad_hi_u8's addend): b, d and e collect four tags,
// a and c six.  A step a pixel does not take (PRED: its bit in `up` / `dn` - the complement of the step set, bit d for
// u = +d / -d) gets 2^30 added to its five sums and cannot win; a wave whose pixels all take every step skips that.
// At R = 8 that is 199 vector instructions a trip (two steps of 8 x 64 pixels) = 41 vector-ALU cycles per 64 pixel-steps
// by the instruction classes of tools/valu_rate.hip, against 103 in the list form; at R = 4 (what runs: registers, see
// the launch) 128 a trip = 53.
template <int R, bool PRED>
__device__ __forceinline__ void calc_dir_dense(const uint32_t *tr, int maxdt, const uint32_t (&up)[R], const uint32_t (&dn)[R],
                                               int nt13, int nt19, uint32_t (&ka)[R], uint32_t (&kb)[R], uint32_t (&kc)[R],
                                               uint32_t (&kd)[R], uint32_t (&ke)[R])
{
    constexpr int NS = R + 4, NE = R + 3;
#define SADH(a, b, acc) __builtin_amdgcn_sad_hi_u8((a), (b), (acc))
    uint32_t T[NS];
#pragma unroll
    for (int r = 0; r < NS; r++) T[r] = tr[r * CD_LW];
#pragma unroll
    for (int j = 0; j < R; j++)
    {
        const int ctr = (int)((T[j + 2] >> 8) & 0xff);
        const int vert = iabs(ctr - (int)((T[j + 3] >> 8) & 0xff)) + iabs(ctr - (int)((T[j + 1] >> 8) & 0xff));
        kb[j] = (uint32_t)min(nt13, vert * 6) << 16; ka[j] = (uint32_t)min(nt19, vert * 9) << 16;
        kc[j] = ka[j]; kd[j] = kb[j]; ke[j] = kb[j];
    }
    // the five sums of every row from the P / Q of one step; X: the poison of the rows that do not take it
    auto sums = [&](const uint32_t (&Pv)[NE], const uint32_t (&Qv)[NE], const uint32_t (&X)[R], uint32_t (&ca)[R], uint32_t (&cb)[R],
                    uint32_t (&cc)[R], uint32_t (&cd)[R], uint32_t (&ce)[R]) {
        uint32_t S[NE], P2[NE - 1], Q2[NE - 1];
#pragma unroll
        for (int r = 0; r < NE; r++) S[r] = Pv[r] + Qv[r];
#pragma unroll
        for (int r = 0; r < NE - 1; r++) { P2[r] = Pv[r] + Pv[r + 1]; Q2[r] = Qv[r] + Qv[r + 1]; }
        if (PRED)
        {
#pragma unroll
            for (int j = 0; j < R; j++)
            {
                cb[j] = S[j + 1] + S[j + 2] + X[j]; ca[j] = cb[j] + S[j]; cc[j] = cb[j] + S[j + 3];
                ce[j] = P2[j] + P2[j + 2] + X[j];   cd[j] = Q2[j] + Q2[j + 2] + X[j];
            }
        }
        else
        {
            // diffc of a row is diffa of the next one
            uint32_t S2[NE - 1], S3[NE - 2];
#pragma unroll
            for (int r = 0; r < NE - 1; r++) S2[r] = S[r] + S[r + 1];
#pragma unroll
            for (int r = 0; r < NE - 2; r++) S3[r] = S2[r] + S[r + 2];
#pragma unroll
            for (int j = 0; j < R; j++)
            {
                cb[j] = S2[j + 1]; ca[j] = S3[j]; cc[j] = S3[j + 1];
                ce[j] = P2[j] + P2[j + 2]; cd[j] = Q2[j] + Q2[j + 2];
            }
        }
    };
    uint32_t X1[R], X2[R];
#pragma unroll
    for (int j = 0; j < R; j++) { X1[j] = 0; X2[j] = 0; }
    {
        // u = 0: both sides are the thread's own column
        uint32_t Pv[NE], ca[R], cb[R], cc[R], cd[R], ce[R];
#pragma unroll
        for (int r = 0; r < NE; r++) Pv[r] = SADH(T[r], T[r + 1], 32u);
        if (PRED)
        {
#pragma unroll
            for (int j = 0; j < R; j++) X1[j] = (up[j] << 30) & 0x40000000u;
        }
        sums(Pv, Pv, X1, ca, cb, cc, cd, ce);
#pragma unroll
        for (int j = 0; j < R; j++)
        {
            ka[j] = min(ka[j], ca[j]); kb[j] = min(kb[j], cb[j]); kc[j] = min(kc[j], cc[j]);
            kd[j] = min(kd[j], cd[j]); ke[j] = min(ke[j], ce[j]);
        }
    }
    for (int d = 1; d <= maxdt; d++)
    {
        const uint32_t *tp = tr + d, *tm = tr - d;
        uint32_t Pl[NS], Mi[NS];
#pragma unroll
        for (int r = 0; r < NS; r++) { Pl[r] = tp[r * CD_LW]; Mi[r] = tm[r * CD_LW]; }
        const uint32_t t1 = 32u + (uint32_t)d, t2 = 32u - (uint32_t)d;
        uint32_t P1[NE], Q1[NE], Pn[NE], Qn[NE];
#pragma unroll
        for (int r = 0; r < NE; r++)
        {
            P1[r] = SADH(T[r], Mi[r + 1], t1); Q1[r] = SADH(Pl[r], T[r + 1], t1);      // u = +d: x - u left, x + u right
            Pn[r] = SADH(T[r], Pl[r + 1], t2); Qn[r] = SADH(Mi[r], T[r + 1], t2);      // u = -d: the sides change places
        }
        if (PRED)
        {
            const uint32_t sh = 30u - (uint32_t)d;
#pragma unroll
            for (int j = 0; j < R; j++) { X1[j] = (up[j] << sh) & 0x40000000u; X2[j] = (dn[j] << sh) & 0x40000000u; }
        }
#ifdef CD_SEQ
        // development form: u = +d and u = -d one after the other, so that only one direction's SADs
        // and candidates are live at a time (5 R more v_min, ~5 R + 2 (R + 3) fewer live registers); measured in
        // profiles/r6Z_calcdir_seq.log
        (void)P1; (void)Q1; (void)Pn; (void)Qn;
        {
            uint32_t Pa[NE], Qa[NE], ca[R], cb[R], cc[R], cd[R], ce[R];
#pragma unroll
            for (int r = 0; r < NE; r++) { Pa[r] = SADH(T[r], Mi[r + 1], t1); Qa[r] = SADH(Pl[r], T[r + 1], t1); }
            if (PRED)
            {
                const uint32_t sh = 30u - (uint32_t)d;
#pragma unroll
                for (int j = 0; j < R; j++) X1[j] = (up[j] << sh) & 0x40000000u;
            }
            sums(Pa, Qa, X1, ca, cb, cc, cd, ce);
#pragma unroll
            for (int j = 0; j < R; j++)
            {
                ka[j] = min(ka[j], ca[j]); kb[j] = min(kb[j], cb[j]); kc[j] = min(kc[j], cc[j]);
                kd[j] = min(kd[j], cd[j]); ke[j] = min(ke[j], ce[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < R; j++) asm volatile("" : "+v"(ka[j]), "+v"(kb[j]), "+v"(kc[j]), "+v"(kd[j]), "+v"(ke[j]));
        __builtin_amdgcn_sched_barrier(0);
        {
            uint32_t Pa[NE], Qa[NE], ca[R], cb[R], cc[R], cd[R], ce[R];
#pragma unroll
            for (int r = 0; r < NE; r++) { Pa[r] = SADH(T[r], Pl[r + 1], t2); Qa[r] = SADH(Mi[r], T[r + 1], t2); }
            if (PRED)
            {
                const uint32_t sh = 30u - (uint32_t)d;
#pragma unroll
                for (int j = 0; j < R; j++) X2[j] = (dn[j] << sh) & 0x40000000u;
            }
            sums(Pa, Qa, X2, ca, cb, cc, cd, ce);
#pragma unroll
            for (int j = 0; j < R; j++)
            {
                ka[j] = min(ka[j], ca[j]); kb[j] = min(kb[j], cb[j]); kc[j] = min(kc[j], cc[j]);
                kd[j] = min(kd[j], cd[j]); ke[j] = min(ke[j], ce[j]);
            }
        }
        continue;
#endif
        uint32_t ca[R], cb[R], cc[R], cd[R], ce[R], na[R], nb[R], nc[R], nd[R], ne[R];
        sums(P1, Q1, X1, ca, cb, cc, cd, ce);
        sums(Pn, Qn, X2, na, nb, nc, nd, ne);
#pragma unroll
        for (int j = 0; j < R; j++)
        {
            ka[j] = min(ka[j], min(ca[j], na[j])); kb[j] = min(kb[j], min(cb[j], nb[j])); kc[j] = min(kc[j], min(cc[j], nc[j]));
            kd[j] = min(kd[j], min(cd[j], nd[j])); ke[j] = min(ke[j], min(ce[j], ne[j]));
        }
    }
#undef SADH
}

// bit t of a mask row's window: a peak among columns start + t .. + 2 (len + 2 <= 63 bits of the row's bitmap)
__device__ __forceinline__ uint64_t calc_dir_window(const uint64_t *bits, int start, uint64_t lenmask)
{
    const int wq = start >> 6, sh = start & 63;
    const uint64_t lo = bits[wq], hi = bits[wq + 1];
    const uint64_t w = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
    return (w | (w >> 1) | (w >> 2)) & lenmask;
}

// dense_min: a block whose list holds at least this many pixels (and that touches neither the first nor the last row)
// takes the dense form of the search (R >= 4)
#ifndef CD_WAVES_ATTR
#define CD_WAVES_ATTR
#endif
template <int R>
__global__ __launch_bounds__(CD_W) CD_WAVES_ATTR void k_calc_dir_rows(P3 P, int maxd, int nt13, int nt19, int dense_min, uint32_t padv)
{
    constexpr int NS = R + 4, NM = R + 2, RW = CD_LW / 4, RQ = CD_LW / 16;
    __shared__ __attribute__((aligned(16))) uint8_t s_band[NS + NM][CD_LW];   // staged rows: 0..NS-1 source y0-2.., NS.. mask y0-1..
    __shared__ __attribute__((aligned(16))) uint32_t s_tri[NS][CD_LW];         // [r][i] = bytes i..i+2 of source row r
    __shared__ uint64_t s_bits[NM][8];                                         // bit i: mask row m is 255 at column i
    __shared__ uint16_t s_list[R * CD_W];                                      // the listed pixels, (row << 8) | column
    __shared__ __attribute__((aligned(16))) uint8_t s_out[R][CD_W];
    __shared__ int s_count;
    FIELD_PLANE(P);
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    const int x0 = blockIdx.x * CD_W, y0 = blockIdx.y * R;
    if (y0 >= height || x0 >= pitch) return;
    const int tid = threadIdx.x, lane = tid & 63;
    if (maskless)
    {
        // no mask pixel in the plane: the reference's memset(dstp, 255, ...) is all there is (:371)
#pragma unroll
        for (int jr = 0; jr < R; jr += CD_W / 64)
        {
            const int j = jr + (tid >> 6), y = y0 + j, xb = x0 + 4 * lane;
            if (j < R && y < height && xb < pitch) *reinterpret_cast<uint32_t *>(Q.c + (size_t)y * pitch + xb) = pad_bytes(0xffffffffu, xb, width, padv);
        }
        return;
    }
    if (tid == 0) s_count = 0;
    if (tid < NM * 3) s_bits[tid / 3][5 + tid % 3] = 0;           // words past the staged columns (the window reads one of them)
    uint32_t *band = reinterpret_cast<uint32_t *>(&s_band[0][0]);
    {
        // flat addressing as in the one-row forms (out-of-row columns pick up the neighbouring rows' bytes, as the
        // reference's pointer arithmetic does); rows past height + 1 serve no pixel and are not touched.  16 bytes per
        // load (the scratch planes, their pitches and x0 - CD_HALO are multiples of 16).
        const uint8_t *sb = Q.b + x0 - CD_HALO, *mb = Q.a + x0 - CD_HALO;
        for (int i = tid; i < (NS + NM) * RQ; i += CD_W)
        {
            const int r = i / RQ, c16 = i - r * RQ;
            const uint8_t *src = r < NS ? sb + (ptrdiff_t)min(y0 - 2 + r, height + 1) * pitch
                                        : mb + (ptrdiff_t)min(y0 - 1 + (r - NS), height) * pitch;
            reinterpret_cast<uint4 *>(band)[i] = reinterpret_cast<const uint4 *>(src)[c16];
        }
    }
    __syncthreads();
    // the pixels that pass the edge test (:392-393), four per thread from dwords of the mask row; the rest of the
    // output is 255 (memset(dstp, 255, pitch*height))
#pragma unroll
    for (int jr = 0; jr < R; jr += CD_W / 64)
    {
        const int j = jr + (tid >> 6), y = y0 + j, xb = x0 + 4 * lane;
        if (R % (CD_W / 64) != 0 && j >= R) break;                // wave-uniform (R smaller than the block's waves)
        const uint32_t *mrow = band + (NS + j + 1) * RW + CD_HALO / 4 + lane;
        const uint32_t prev = mrow[-1], cur = mrow[0], next = mrow[1];
        auto eq255 = [](uint32_t v) { return (((v & 0x7f7f7f7fu) + 0x01010101u) & v) & 0x80808080u; };
        uint32_t act = eq255(cur) & (eq255(__builtin_amdgcn_alignbyte(cur, prev, 3)) | eq255(__builtin_amdgcn_alignbyte(next, cur, 1)));
        if (y < 1 || y >= height - 1) act = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (xb + k < 1 || xb + k >= width - 1) act &= ~(0x80u << (8 * k));
        reinterpret_cast<uint32_t *>(&s_out[j][0])[lane] = 0xffffffffu;
        // listed in column order (see above)
        int total = 0, before = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const uint64_t bal = __ballot((act >> (8 * k + 7)) & 1u);
            before += (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            total += __popcll(bal);
        }
        int base = 0;
        if (lane == 0 && total) base = atomicAdd(&s_count, total);
        base = __builtin_amdgcn_readfirstlane(base) + before;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((act >> (8 * k + 7)) & 1u) s_list[base++] = (uint16_t)((j << 8) | (4 * lane + k));
    }
    __syncthreads();
    const int count = s_count;
    const int maxdt = pl == 0 ? maxd : (maxd >> 1);
    const int len = 2 * maxdt + 1;
    const uint64_t lenmask = (1ull << len) - 1ull;
    const bool edge = y0 <= 1 || y0 + R - 1 >= height - 2;
    const bool dense = R >= 4 && !edge && count >= dense_min;          // block-uniform
    if (count)                                                         // block-uniform
    {
        // (behind the list: a block without a listed pixel - off the mask, like whole chroma planes without edges - makes
        // neither tables nor bitmaps)
        // tables: the four columns of a staged dword at a time (two dwords realigned three ways, one 16-byte write); columns
        // 0 .. CD_LW-4 are looked at (b+-u, |u| <= CD_HALO-2), the last dword's are made but never read
        for (int i = tid; i < NS * RW; i += CD_W)
        {
            const uint32_t d0 = band[i], d1 = band[i + 1];
            reinterpret_cast<uint4 *>(&s_tri[0][0])[i] =
                make_uint4(d0 & 0x00ffffffu, __builtin_amdgcn_alignbyte(d1, d0, 1) & 0x00ffffffu,
                           __builtin_amdgcn_alignbyte(d1, d0, 2) & 0x00ffffffu, __builtin_amdgcn_alignbyte(d1, d0, 3) & 0x00ffffffu);
        }
        // peak bitmaps of the mask rows: a first round for all, the 64 columns left over for the first wave
        for (int k = 0; k < 2; k++)
        {
            if (k == 1 && tid >= 64) break;                          // wave-uniform
            const int i = tid + k * CD_W;
    #pragma unroll
            for (int m = 0; m < NM; m++)
            {
                const uint64_t w = __ballot(s_band[NS + m][i] == PEAK);   // the wave's 64 consecutive columns = one word
                if (lane == 0) s_bits[m][i >> 6] = w;
            }
        }
        __syncthreads();
    }
#ifdef HBHIP_DEV_STATS
    if (tid == 0) { CD_STAT(0, 1); CD_STAT(1, count != 0); CD_STAT(2, count); CD_STAT(7, dense); }
#endif
    if (R >= 4 && dense)
    {
        // a thread owns column tid of the block's R rows (all of them interior rows here)
        constexpr int RD = R >= 4 ? R : 1;                             // (the two-row instantiation never gets here)
        const int lx = tid, px = x0 + lx, b = lx + CD_HALO - 1;
        const int startu = max(-px + 1, -maxdt), stopu = min(width - 2 - px, maxdt);
        uint64_t range = 0;
        if (stopu >= startu)
        {
            const int nb = stopu - startu + 1;
            range = (nb >= 64 ? ~0ull : ((1ull << nb) - 1ull)) << (startu + maxdt);
        }
        // the windows of the block's R + 2 mask rows at this column, each serving the row below it (as it stands) and the
        // row above it (reversed)
        uint64_t win[RD + 2];
#pragma unroll
        for (int m = 0; m < RD + 2; m++) win[m] = calc_dir_window(s_bits[m], b - maxdt, lenmask);
        uint32_t up[RD], dn[RD], act = 0, inactive = 0;
#pragma unroll
        for (int j = 0; j < RD; j++)
        {
            const uint8_t *mr = &s_band[NS + j + 1][CD_HALO + lx];
            const bool a = px >= 1 && px < width - 1 && mr[0] == PEAK && (mr[-1] == PEAK || mr[1] == PEAK);    // :392-393
            const uint64_t pass = a ? range & win[j] & (__brevll(win[j + 2]) >> (64 - len)) : 0ull;
            const uint64_t npass = ~pass;
            up[j] = (uint32_t)(npass >> maxdt);
            dn[j] = __brev((uint32_t)npass << (31 - maxdt));
            act |= (uint32_t)a << j;
            inactive |= (up[j] | dn[j]) & ((2u << maxdt) - 1u);
        }
        if (__any(act != 0))                                           // wave-uniform
        {
            uint32_t ka[RD], kb[RD], kc[RD], kd[RD], ke[RD];
            const uint32_t *tr = &s_tri[0][b];
            if (__all(inactive == 0)) calc_dir_dense<RD, false>(tr, maxdt, up, dn, nt13, nt19, ka, kb, kc, kd, ke);
            else                      calc_dir_dense<RD, true>(tr, maxdt, up, dn, nt13, nt19, ka, kb, kc, kd, ke);
#pragma unroll
            for (int j = 0; j < RD; j++)
                if ((act >> j) & 1u)
                    // b, d and e hold four tags, a and c six (calc_dir_dense)
                    s_out[j][lx] = (uint8_t)calc_dir_vote((int)((ka[j] & 0xffffu) / 6u), (int)((kb[j] & 0xffffu) >> 2), (int)((kc[j] & 0xffffu) / 6u),
                                                          (int)((kd[j] & 0xffffu) >> 2), (int)((ke[j] & 0xffffu) >> 2));
#ifdef HBHIP_DEV_STATS
            if (lane == 0) { CD_STAT(8, 1); CD_STAT(9, __all(inactive == 0)); }
#endif
        }
        __syncthreads();
    }
    else if (count)                                   // block-uniform
    {
        for (int p = tid; p < count; p += CD_W)
        {
            const uint32_t id = s_list[p];
            const int j = (int)(id >> 8), lx = (int)(id & 255u), y = y0 + j, px = x0 + lx, b = lx + CD_HALO - 1;
            // The steps this pixel takes, as a bit set (bit jj: u = jj - maxdt): inside its range, and - unless on the first /
            // last row - with a mask peak above at +u and below at -u (:395-399).  Above: bits b-maxdt .. b+maxdt of the row's
            // bitmap in that order; below: the same bits of the other row's in reverse.
            const int startu = max(-px + 1, -maxdt), stopu = min(width - 2 - px, maxdt);
            uint64_t pass = 0;
            if (stopu >= startu)
            {
                const int nb = stopu - startu + 1;
                pass = (nb >= 64 ? ~0ull : ((1ull << nb) - 1ull)) << (startu + maxdt);
                if (y != 1)          pass &= calc_dir_window(s_bits[j], b - maxdt, lenmask);
                if (y != height - 2) pass &= __brevll(calc_dir_window(s_bits[j + 2], b - maxdt, lenmask)) >> (64 - len);
            }
#ifdef HBHIP_DEV_STATS
            {
                int n = __popcll(pass), mx = n, sm = n, ln = 1;
                for (int o = 32; o; o >>= 1) { mx = max(mx, __shfl_xor(mx, o)); sm += __shfl_xor(sm, o); ln += __shfl_xor(ln, o); }
                const uint64_t act = __ballot(1);
                if (lane == __ffsll((unsigned long long)act) - 1) { CD_STAT(3, 1); CD_STAT(4, sm); CD_STAT(5, mx); CD_STAT(6, ln); }
            }
#endif
            const uint32_t *tr = &s_tri[j][b];
            int out;
            if (len <= 32)                                                    // block-uniform (the plane's search distance)
                out = edge ? calc_dir_search<true, uint32_t>(tr, (uint32_t)pass, maxdt, y == 1, y == height - 2, nt13, nt19)
                           : calc_dir_search<false, uint32_t>(tr, (uint32_t)pass, maxdt, false, false, nt13, nt19);
            else
                out = edge ? calc_dir_search<true, uint64_t>(tr, pass, maxdt, y == 1, y == height - 2, nt13, nt19)
                           : calc_dir_search<false, uint64_t>(tr, pass, maxdt, false, false, nt13, nt19);
            s_out[j][lx] = (uint8_t)out;
        }
        __syncthreads();
    }
#pragma unroll
    for (int jr = 0; jr < R; jr += CD_W / 64)
    {
        const int j = jr + (tid >> 6), y = y0 + j, xb = x0 + 4 * lane;
        if (j < R && y < height && xb < pitch)
            *reinterpret_cast<uint32_t *>(Q.c + (size_t)y * pitch + xb) = pad_bytes(reinterpret_cast<const uint32_t *>(&s_out[j][0])[lane], xb, width, padv);
    }
}

// filter_dir_map / expand_dir_map and their _2x forms.
// a = edge mask, b = direction map in, c = out.  step = 1 (half height) or 2.
// step 1: rows 1..height-2, neighbours y+-1, mask row y.
// step 2: rows y0, y0+2, ... < height-1, neighbours y+-2 (guarded by y>1 / y<height-2), mask rows y-1 and y+1.

// Four pixels per thread.  One byte per thread makes these passes latency bound: a wave lives for two dependent memory
// round trips whatever it computes, so the time is (#waves / resident waves) x that.  Here a thread owns one aligned dword of
// its row: three 12-byte windows (rows y-step, y, y+step of the direction map) + one or two mask dwords, a quarter of
// the waves and of the load instructions, one dword store.  In the _2x forms (step 2) every other row is only copied:
// that is decided per row (a wave = one row), before anything but the row's own dword is loaded.
__device__ __forceinline__ int dir_map_px(int u0, int u1, int u2, int c0, int c1, int c2, int n0, int n1, int n2,
                                          bool up_ok, bool dn_ok, int expand, const uint8_t *limlut)
{
    const bool h0 = up_ok && u0 != PEAK, h1 = up_ok && u1 != PEAK, h2 = up_ok && u2 != PEAK;
    const bool h3 = c0 != PEAK, h4 = !expand && c1 != PEAK, h5 = c2 != PEAK;
    const bool h6 = dn_ok && n0 != PEAK, h7 = dn_ok && n1 != PEAK, h8 = dn_ok && n2 != PEAK;
    const int u = h0 + h1 + h2 + h3 + h4 + h5 + h6 + h7 + h8;
    if (u < (expand ? 5 : 4)) return expand ? c1 : PEAK;
    int v0 = h0 ? u0 : ABSENT, v1 = h1 ? u1 : ABSENT, v2 = h2 ? u2 : ABSENT;
    int v3 = h3 ? c0 : ABSENT, v4 = h4 ? c1 : ABSENT, v5 = h5 ? c2 : ABSENT;
    int v6 = h6 ? n0 : ABSENT, v7 = h7 ? n1 : ABSENT, v8 = h8 ? n2 : ABSENT;
    // the midpoint from a sorted COPY: the vote below is a sum and a count, whatever the order - it takes the slots as they
    // are, and of the network only the exchanges that reach the middle entries are left (39 of its 50 min / max)
    int s0 = v0, s1 = v1, s2 = v2, s3 = v3, s4 = v4, s5 = v5, s6 = v6, s7 = v7, s8 = v8;
    const int mid = mid9(s0, s1, s2, s3, s4, s5, s6, s7, s8, u);
    const int lim = limlut[iabs(mid - NEUTRAL) >> 2];
    int sum = 0, count = 0;
    vote1(v0, mid, lim, sum, count); vote1(v1, mid, lim, sum, count); vote1(v2, mid, lim, sum, count);
    vote1(v3, mid, lim, sum, count); vote1(v4, mid, lim, sum, count); vote1(v5, mid, lim, sum, count);
    vote1(v6, mid, lim, sum, count); vote1(v7, mid, lim, sum, count); vote1(v8, mid, lim, sum, count);
    const int val = vote_avg(sum + mid, count + 1);              // (int)((float)(sum + mid) / (float)(count + 1) + 0.5f)
    if (expand) return count >= 5 ? (val & 0xff) : c1;
    if (count < 4 || (count < 5 && c1 == PEAK)) return PEAK;
    return val & 0xff;
}

__device__ __forceinline__ uint32_t ff_bytes(uint32_t v)          // 0x80 in every byte that is 0xff
{
    return (((v & 0x7f7f7f7fu) + 0x01010101u) & v) & 0x80808080u;
}

// ------------------------------------------------------------------------------------------
// The dir-map vote on PAIRS of pixels.  With a dense edge mask (the lower half of a field's mask saturates,
// eedi2_template.c:132) nearly every pixel of the dir-map passes reaches its sort and its vote, and both are 16-bit work:
// two horizontally adjacent pixels ride in the halves of a dword through v_pk_min_u16 / v_pk_max_u16 (the 9-input
// network once for both), v_pk_sub / add / mad_u16 and v_bfi_b32 selects.  A slot that holds no value (the reference
// leaves peaks out of order[], :659-668) carries PK_ABSENT, above every value and farther from every midpoint than any
// limit (also the 255 of limlut's last entries, eedi2.c:24).
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
typedef int16_t i16x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t PK_ABSENT_HI = 0x7f00u;                            // 0x00ff + 0x7f00 = 0x7fff
__device__ __forceinline__ u16x2 pk(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t un(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ u16x2 pk1(uint32_t both) { return pk(both * 0x00010001u); }
// [a < b] per half as 0 / 1 for halves below 2^15: the borrow of a - b (two packed instructions; written as a comparison
// or as min(saturated difference, 1) the compiler unpacks it into a compare and a select per half)
__device__ __forceinline__ u16x2 pk_lt(u16x2 a, u16x2 b) { return (u16x2)((u16x2)(a - b) >> 15); }
__device__ __forceinline__ void cswap2(u16x2 &a, u16x2 &b)
{
    const u16x2 lo = __builtin_elementwise_min(a, b), hi = __builtin_elementwise_max(a, b);
    a = lo; b = hi;
}
// limlut[i] (eedi2.c:21-25 as 8-bit pixels: 6 6 7 7 8 8 9 9 9 10 10 11 11 12 ... 12, then 255 255 for i = 31, 32) in
// closed form, both halves: min(12, 6 + ((i - [i >= 8]) >> 1)), 255 from 31 on
// (tests/test_eedi2_identities_cpu.py::test_limlut_closed_form)
__device__ __forceinline__ u16x2 limlut2(u16x2 i)
{
    const u16x2 g = pk_lt(pk1(7), i);
    const u16x2 l = __builtin_elementwise_min((u16x2)(((i - g) >> 1) + pk1(6)), pk1(12));
    const u16x2 big = pk_lt(pk1(30), i) * pk1(255);
    return __builtin_elementwise_max(l, big);
}
// two bytes of the 8-byte window {hi, lo} as the halves of a dword (v_perm_b32; selector bytes: 0-3 = lo, 4-7 = hi, 12 = 0)
#define PK_BYTES(hi, lo, a, b) __builtin_amdgcn_perm((hi), (lo), 0x0c000c00u | ((uint32_t)(b) << 16) | (uint32_t)(a))

// The two pixels at columns k and k + 1 of a thread's dword (k = 0 or 2; windows wu / wc / wd = rows above, own, below as
// the bytes x-4 .. x+7, rows that do not count - first / last rows of the _2x forms - already all 0xff).
// Returns the pass's values for both (low byte of each half): filter_dir_map :649-707 (expand == 0) or expand_dir_map
// :722-773 (expand != 0; the caller only takes them for pixels the pass works on).
template <int K>
__device__ __forceinline__ uint32_t dir_map_pair(const Win12 &wu, const Win12 &wc, const Win12 &wd, int expand)
{
    // slot s of pixel k is byte k - 1 + s % 3 of its row: the pairs (k-1, k), (k, k+1), (k+1, k+2); window byte index = column + 4
    u16x2 v[9];
    {
        const Win12 *rows[3] = { &wu, &wc, &wd };
#pragma unroll
        for (int r = 0; r < 3; r++)
        {
            const uint32_t w0 = rows[r]->w0, w1 = rows[r]->w1, w2 = rows[r]->w2;
            if (K == 0)
            {
                v[3 * r + 0] = pk(PK_BYTES(w1, w0, 3, 4));            // columns -1, 0
                v[3 * r + 1] = pk(PK_BYTES(w1, w0, 4, 5));            // 0, 1
                v[3 * r + 2] = pk(PK_BYTES(w1, w0, 5, 6));            // 1, 2
            }
            else
            {
                v[3 * r + 0] = pk(PK_BYTES(w1, w0, 5, 6));            // 1, 2
                v[3 * r + 1] = pk(PK_BYTES(w1, w0, 6, 7));            // 2, 3
                v[3 * r + 2] = pk(PK_BYTES(w2, w1, 3, 4));            // 3, 4
            }
        }
    }
    const u16x2 c1 = v[4];
    // absent slots: 1 per half that holds a peak (a pixel expand works on has a peak in its centre: left out as :739-746
    // leave it out), the value pushed up to PK_ABSENT
    u16x2 a[9];
    uint32_t absent = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
    {
        a[i] = (u16x2)((v[i] + pk1(1)) >> 8);
        absent += un(a[i]);
        v[i] = v[i] + a[i] * pk1(PK_ABSENT_HI);
    }
    // midpoint of the n = 9 - absent present values (mid9's selection with masks): n <= 5 <=> absent >= 4, n <= 7 <=> absent >= 2
    u16x2 s0 = v[0], s1 = v[1], s2 = v[2], s3 = v[3], s4 = v[4], s5 = v[5], s6 = v[6], s7 = v[7], s8 = v[8];
    cswap2(s0, s3); cswap2(s1, s7); cswap2(s2, s5); cswap2(s4, s8);
    cswap2(s0, s7); cswap2(s2, s4); cswap2(s3, s8); cswap2(s5, s6);
    cswap2(s0, s2); cswap2(s1, s3); cswap2(s4, s5); cswap2(s7, s8);
    cswap2(s1, s4); cswap2(s3, s6); cswap2(s5, s7);
    cswap2(s0, s1); cswap2(s2, s4); cswap2(s3, s5); cswap2(s6, s8);
    cswap2(s2, s3); cswap2(s4, s5); cswap2(s6, s7);
    cswap2(s1, s2); cswap2(s3, s4); cswap2(s5, s6);
    const u16x2 ab = pk(absent), one = pk1(1), zero = pk1(0);
    const uint32_t m5 = un(zero - pk_lt(pk1(3), ab));
    const uint32_t m7 = un(zero - pk_lt(one, ab));
    const uint32_t modd = un((ab & one) - one);                        // n odd <=> absent even
#define PK_SEL(m, x, y) (((m) & (x)) | (~(m) & (y)))                    /* v_bfi_b32 */
    const uint32_t hi = PK_SEL(m5, un(s2), PK_SEL(m7, un(s3), un(s4)));
    const uint32_t lo = PK_SEL(m5, un(s1), PK_SEL(m7, un(s2), un(s3)));
    const u16x2 mid = pk(PK_SEL(modd, hi, un((u16x2)((pk(lo) + pk(hi) + one) >> 1))));
    // the vote (:685-697): values within limlut[|mid - neutral| >> 2] of the midpoint
    const i16x2 t = __builtin_bit_cast(i16x2, (u16x2)(mid - pk1(NEUTRAL)));
    const u16x2 lim1 = limlut2(__builtin_bit_cast(u16x2, __builtin_elementwise_max(t, (i16x2)(-t))) >> 2) + one;
    u16x2 sum = zero, cnt = zero;
#pragma unroll
    for (int i = 0; i < 9; i++)
    {
        const u16x2 d = __builtin_elementwise_max(v[i], mid) - __builtin_elementwise_min(v[i], mid);
        const u16x2 in = pk_lt(d, lim1);
        cnt += in;
        sum += in * v[i];
    }
    const uint32_t sm = un((u16x2)(sum + mid)), ct = un(cnt);
    uint32_t out = 0;
#pragma unroll
    for (int h = 0; h < 2; h++)
    {
        // too few values (:669-673 / :747): the midpoint above may be a PK_ABSENT then, so the count says nothing
        const int n = 9 - (int)((absent >> (16 * h)) & 0xffffu);
        const int count = n >= 4 ? (int)((ct >> (16 * h)) & 0xffffu) : 0;
        const int val = vote_avg((int)((sm >> (16 * h)) & 0xffffu), count + 1) & 0xff;
        const int c = (int)((un(c1) >> (16 * h)) & 0xffu);
        int res;
        if (expand) res = count >= 5 ? val : c;
        else        res = (count < 4 || (count < 5 && c == PEAK)) ? PEAK : val;
        out |= (uint32_t)res << (16 * h);
    }
#undef PK_SEL
    return out;
}

// In the _2x forms a thread takes the rows 2r and 2r + 1 of its dword column: the one with the parity of the rebuilt rows
// is worked on, the other only copied (a wave per copied row spent more on finding its plane and field - scalar
// instructions - than on its dword).
__global__ __launch_bounds__(256) void k_dir_map4(P3 P, int step, int expand)
{
    FIELD_PLANE(P);
    const int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    const int r = blockIdx.y * blockDim.y + threadIdx.y;
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    const int y0 = step == 1 ? 1 : 2 - tff;
    const int y = step == 1 ? r : 2 * r + (y0 & 1);
    if (x >= width) return;
    if (maskless)
    {
        // No mask pixel in the plane: the pass is its bit_blit (:656 / :729) - of a map that calc_directions filled with
        // peaks for want of a mask pixel (:371) and that every pass since has only copied: the output is peaks too, and a
        // thread stores its dwords without loading anything (as copies the chroma planes of 16 fields were 66 MB and 14 us
        // per full-height launch).
        const int ya = step == 1 ? r : 2 * r, nrows = step == 1 ? 1 : 2;
        const int out[4] = { PEAK, PEAK, PEAK, PEAK };
        for (int i = 0; i < nrows && ya + i < height; i++)
        {
            const size_t at = (size_t)(ya + i) * pitch + x;
            st4(Q.c + at, out, x, width);
            if (Q.d) st4(Q.d + at, out, x, width);
        }
        return;
    }
    // the row of the pair that is never rebuilt (bit_blit only): fetched now, stored after the other row's work, so that
    // its load is one of the thread's loads in flight and not a round trip of its own
    const int yc = 2 * r + 1 - (y0 & 1);
    const bool copy = step != 1 && yc < height;
    uint32_t vcopy = 0;
    if (copy) vcopy = *reinterpret_cast<const uint32_t *>(Q.b + (size_t)yc * pitch + x);
    auto put_copy = [&]() {
        if (!copy) return;
        int out[4] = { (int)(vcopy & 0xff), (int)((vcopy >> 8) & 0xff), (int)((vcopy >> 16) & 0xff), (int)(vcopy >> 24) };
        st4(Q.c + (size_t)yc * pitch + x, out, x, width);
        if (Q.d) st4(Q.d + (size_t)yc * pitch + x, out, x, width);
    };
    if (y >= height) { put_copy(); return; }
    const uint8_t *dc = Q.b + (size_t)y * pitch + x;
    uint8_t *o = Q.c + (size_t)y * pitch + x;
    const bool row_ok = step == 1 ? (y >= 1 && y < height - 1) : (y >= y0 && y < height - 1);
    if (!row_ok)
    {
        // bit_blit only (and the optional copy of the input, the eedi2_bit_blit before post-processing)
        const uint32_t v = *reinterpret_cast<const uint32_t *>(dc);
        int out[4] = { (int)(v & 0xff), (int)((v >> 8) & 0xff), (int)((v >> 16) & 0xff), (int)(v >> 24) };
        st4(o, out, x, width);
        if (Q.d) st4(Q.d + (size_t)y * pitch + x, out, x, width);
        put_copy();
        return;
    }
    // The row's own dword and the mask first: a thread without a pixel to work on (off the mask - whole planes of them
    // where the chroma has no edges: a third of a field's pixels) copies its dword and never fetches the 32 bytes around it.
    const uint32_t own = *reinterpret_cast<const uint32_t *>(dc);
    const uint8_t *mk = Q.a + (size_t)y * pitch + x;
    const uint32_t m0 = *reinterpret_cast<const uint32_t *>(step == 1 ? mk : mk - (ptrdiff_t)pitch);
    const uint32_t m1 = step == 1 ? 0u : *reinterpret_cast<const uint32_t *>(mk + pitch);
    const bool up_ok = step == 1 || y > 1, dn_ok = step == 1 || y < height - 2;
    // the pixels the pass works on, one flag byte each (:658 / :738): inside the row, on the mask, and for expand a peak
    uint32_t work = ((ff_bytes(m0) | ff_bytes(m1)) >> 7) & mf_bytes_in(x, 1, width - 1);
    if (expand) work &= ff_bytes(own) >> 7;
    uint32_t res = own;
    if (work)
    {
        asm volatile("" ::: "memory");                         // a real branch: waves without such a pixel skip the loads and the votes
        const Win12 wc = ldwin(dc), wu = ldwin(dc - (ptrdiff_t)step * pitch), wd = ldwin(dc + (ptrdiff_t)step * pitch);
        const Win12 none = { 0xffffffffu, 0xffffffffu, 0xffffffffu };
        const Win12 eu = up_ok ? wu : none, ed = dn_ok ? wd : none;
        const uint32_t p01 = dir_map_pair<0>(eu, wc, ed, expand), p23 = dir_map_pair<2>(eu, wc, ed, expand);
        const uint32_t votes = __builtin_amdgcn_perm(p23, p01, 0x06040200u);     // the low bytes of the four halves
        const uint32_t sel = work * 255u;
        res = (res & ~sel) | (votes & sel);
    }
    int out[4] = { (int)(res & 0xff), (int)((res >> 8) & 0xff), (int)((res >> 16) & 0xff), (int)(res >> 24) };
    st4(o, out, x, width);
    if (Q.d)
    {
        int in[4] = { (int)(own & 0xff), (int)((own >> 8) & 0xff), (int)((own >> 16) & 0xff), (int)(own >> 24) };
        st4(Q.d + (size_t)y * pitch + x, in, x, width);
    }
    put_copy();
}

// k_dir_map4 in two phases.  Which pixels reach the sort is decided by byte arithmetic on whole dwords
// (phase 1: peak tests, the 3x3 count of usable neighbours and the thresholds on it, four pixels per
// 32-bit operation); on real pictures that is a small minority, spread so that nearly every wave
// holds a few - and a wave pays for the sort if one lane needs it.  So phase 1 only queues them (LDS
// list, per workgroup of 4 rows x 256 pixels) and phase 2 gives each queued pixel a lane of its own:
// the sort runs on full waves, on as few of them as the queue fills.
__device__ __forceinline__ uint32_t live3(const Win12 &w, uint32_t &centre)   // per byte: usable (non-peak) values among x-1, x, x+1
{
    const uint32_t n0 = (~ff_bytes(w.w0) >> 7) & 0x01010101u, n1 = (~ff_bytes(w.w1) >> 7) & 0x01010101u, n2 = (~ff_bytes(w.w2) >> 7) & 0x01010101u;
    centre = n1;
    return __builtin_amdgcn_alignbyte(n1, n0, 3) + n1 + __builtin_amdgcn_alignbyte(n2, n1, 1);
}

// post != 0 (the last expand_dir_map_2x of a field): eedi2_post_process (:1349-1378, k_post) rides along - it is
// pointwise in the map this pass has just made (e = the map before the post filters, f = dst2p, rebuilt rows only).
// DC_R thread rows per thread: the pass is two phases around a queue with little arithmetic in the first, and a wave that
// takes one dword row of it spends more scalar instructions on finding its plane, field and rows than vector ones on the
// row (91 scalar instructions per wave with one row each; expand_dir_map_2x 73 -> 66 us per launch at two rows, 63 at four).
#ifndef DC_THREAD_ROWS
#define DC_THREAD_ROWS 4
#endif
constexpr int DC_R = DC_THREAD_ROWS, DC_ROWS = 4 * DC_R;
__global__ __launch_bounds__(256) void k_dir_map_c(P3 P, int step, int expand, int post)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_out[DC_ROWS][256];
    __shared__ uint16_t s_list[DC_ROWS * 256];
    __shared__ int s_count;
    __shared__ uint8_t s_lim[LIM_PAD];
    FIELD_PLANE(P);
    const int y0 = step == 1 ? 1 : 2 - tff;
    // step 2: a thread row takes the PAIR of rows 2r, 2r + 1 - the one with the rebuilt rows' parity goes through the
    // pass, the other is only copied (k_dir_map4)
    const int rb = blockIdx.y * DC_ROWS;
    const int bx0 = 4 * (blockIdx.x * 64), x = bx0 + 4 * threadIdx.x;
    const int yb = step == 1 ? rb : 2 * rb + (y0 & 1);                              // row of thread row 0
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    if (bx0 >= width || (step == 1 ? rb : 2 * rb) >= height) return;                 // whole workgroup outside
    if (maskless)
    {
        // no mask pixel in the plane: the pass is its bit_blit of a map of peaks (k_dir_map4), and the post-processing that
        // may ride along finds every direction a peak and leaves the picture alone (:1364)
        if (x < width)
        {
            const int out[4] = { PEAK, PEAK, PEAK, PEAK };
#pragma unroll
            for (int h = 0; h < DC_R; h++)
            {
                const int r = rb + (int)threadIdx.y + 4 * h, ya = step == 1 ? r : 2 * r, nrows = step == 1 ? 1 : 2;
                for (int i = 0; i < nrows && ya + i < height; i++)
                {
                    const size_t at = (size_t)(ya + i) * pitch + x;
                    st4(Q.c + at, out, x, width);
                    if (Q.d) st4(Q.d + at, out, x, width);
                }
            }
        }
        return;
    }
    const int tid = threadIdx.y * 64 + threadIdx.x;
    if (tid == 0) s_count = 0;
    lim_fill(s_lim, tid);
    __syncthreads();
    // Three sweeps over the thread's DC_R rows, so that the loads of all of them are in flight together: as one loop (load the
    // row's dword and masks, wait, load its windows, wait, count) a thread went through 2 DC_R memory round trips one after
    // the other, and the pass - mostly a copy - ran at half of what its bytes allow.
    uint32_t vcopy[DC_R], own[DC_R], m0[DC_R], m1[DC_R], cand[DC_R];
#pragma unroll
    for (int h = 0; h < DC_R; h++)
    {
        const int lr = (int)threadIdx.y + 4 * h, r = rb + lr, y = step == 1 ? r : 2 * r + (y0 & 1);
        // the copied row of the pair: fetched now, stored when the workgroup is done (see k_dir_map4)
        const int yc = 2 * r + 1 - (y0 & 1);
        vcopy[h] = 0; own[h] = 0; m0[h] = 0; m1[h] = 0;
        if (step != 1 && x < width && yc < height) vcopy[h] = *reinterpret_cast<const uint32_t *>(Q.b + (size_t)yc * pitch + x);
        const bool inside = x < width && y < height;
        const bool row_ok = step == 1 ? (y >= 1 && y < height - 1) : (y >= y0 && y < height - 1);
        if (inside)
        {
            own[h] = *reinterpret_cast<const uint32_t *>(Q.b + (size_t)y * pitch + x);
            if (row_ok)
            {
                const uint8_t *mk = Q.a + (size_t)y * pitch + x;
                m0[h] = *reinterpret_cast<const uint32_t *>(step == 1 ? mk : mk - (ptrdiff_t)pitch);
                if (step != 1) m1[h] = *reinterpret_cast<const uint32_t *>(mk + pitch);
            }
        }
    }
    Win12 wc[DC_R], wu[DC_R], wd[DC_R];
#pragma unroll
    for (int h = 0; h < DC_R; h++)
    {
        const int lr = (int)threadIdx.y + 4 * h, r = rb + lr, y = step == 1 ? r : 2 * r + (y0 & 1);
        const bool inside = x < width && y < height;
        const bool row_ok = step == 1 ? (y >= 1 && y < height - 1) : (y >= y0 && y < height - 1);
        // the row's own dword and the mask first: without a candidate among its four pixels a thread copies its dword and
        // fetches nothing else (k_dir_map4)
        cand[h] = 0;
        wc[h] = wu[h] = wd[h] = Win12{ 0u, 0u, 0u };
        if (inside && row_ok)
        {
            cand[h] = ((ff_bytes(m0[h]) | ff_bytes(m1[h])) >> 7) & mf_bytes_in(x, 1, width - 1);
            if (expand) cand[h] &= ff_bytes(own[h]) >> 7;                                  // expand only fills peak pixels
            if (cand[h])
            {
                const uint8_t *dc = Q.b + (size_t)y * pitch + x;
                wc[h] = ldwin(dc); wu[h] = ldwin(dc - (ptrdiff_t)step * pitch); wd[h] = ldwin(dc + (ptrdiff_t)step * pitch);
            }
        }
    }
#pragma unroll
    for (int h = 0; h < DC_R; h++)
    {
        const int lr = (int)threadIdx.y + 4 * h, r = rb + lr, y = step == 1 ? r : 2 * r + (y0 & 1);
        const bool inside = x < width && y < height;
        if (inside)
        {
            uint32_t out = own[h];
            if (cand[h])
            {
                const bool up_ok = step == 1 || y > 1, dn_ok = step == 1 || y < height - 2;
                uint32_t nc, nu, nd;
                const uint32_t s3c = live3(wc[h], nc), s3u = live3(wu[h], nu), s3d = live3(wd[h], nd);
                uint32_t u = expand ? s3c - nc : s3c;                                  // expand leaves the centre out (:671)
                if (up_ok) u += s3u;
                if (dn_ok) u += s3d;
                const uint32_t enough = ((u + (uint32_t)(0x80 - (expand ? 5 : 4)) * 0x01010101u) >> 7) & 0x01010101u;
                if (!expand) out |= (cand[h] & ~enough) * 255u;                        // too few neighbours: peak
                const uint32_t sortpx = cand[h] & enough;
                if (sortpx)
                {
                    const int n = __popc(sortpx);
                    int at = atomicAdd(&s_count, n);
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if ((sortpx >> (8 * k)) & 1u) s_list[at++] = (uint16_t)((lr << 8) | (4 * threadIdx.x + k));
                }
                (void)nu; (void)nd;
            }
            *reinterpret_cast<uint32_t *>(&s_out[lr][4 * threadIdx.x]) = out;
            if (Q.d)                                                                    // optional copy of the input (the eedi2_bit_blit before post-processing)
            {
                int in[4] = { (int)(own[h] & 0xff), (int)((own[h] >> 8) & 0xff), (int)((own[h] >> 16) & 0xff), (int)(own[h] >> 24) };
                st4(Q.d + (size_t)y * pitch + x, in, x, width);
            }
        }
    }
    __syncthreads();
    const int count = s_count;
    for (int i = tid; i < count; i += 256)
    {
        const int e = s_list[i], ly = e >> 8, lx = e & 255;
        const int yy = yb + step * ly;
        const uint8_t *c = Q.b + (size_t)yy * pitch + bx0 + lx;
        const uint8_t *up = c - (ptrdiff_t)step * pitch, *dn = c + (ptrdiff_t)step * pitch;
        const bool up_ok = step == 1 || yy > 1, dn_ok = step == 1 || yy < height - 2;
        s_out[ly][lx] = (uint8_t)dir_map_px(up[-1], up[0], up[1], c[-1], c[0], c[1], dn[-1], dn[0], dn[1], up_ok, dn_ok, expand, s_lim);
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < DC_R; h++)
    {
        const int lr = (int)threadIdx.y + 4 * h, r = rb + lr, y = step == 1 ? r : 2 * r + (y0 & 1);
        const int yc = 2 * r + 1 - (y0 & 1);
        const bool row_ok = step == 1 ? (y >= 1 && y < height - 1) : (y >= y0 && y < height - 1);
        if (x < width && y < height)
        {
            const uint32_t v = *reinterpret_cast<const uint32_t *>(&s_out[lr][4 * threadIdx.x]);
            uint8_t *o = Q.c + (size_t)y * pitch + x;
            if (x + 3 < width) *reinterpret_cast<uint32_t *>(o) = v;
            else for (int k = 0; k < 4 && x + k < width; k++) o[k] = (uint8_t)(v >> (8 * k));
            if (post && row_ok)
            {
                const size_t at = (size_t)y * pitch + x;
                const uint32_t om4 = *reinterpret_cast<const uint32_t *>(Q.e + at);
                uint8_t *d = Q.f + at;
                const uint32_t up4 = *reinterpret_cast<const uint32_t *>(d - pitch), dn4 = *reinterpret_cast<const uint32_t *>(d + pitch);
                const uint32_t cur4 = *reinterpret_cast<const uint32_t *>(d);
                int out[4];
                bool any = false;
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const int nm = (v >> (8 * k)) & 0xffu, om = (om4 >> (8 * k)) & 0xffu;
                    const int lim = s_lim[iabs(nm - NEUTRAL) >> 2];
                    const bool fix = iabs(nm - om) > lim && om != PEAK && om != NEUTRAL;
                    out[k] = fix ? (int)((((up4 >> (8 * k)) & 0xffu) + ((dn4 >> (8 * k)) & 0xffu) + 1) >> 1) : (int)((cur4 >> (8 * k)) & 0xffu);
                    any |= fix;
                }
                if (any) st4(d, out, x, width);
            }
        }
        if (step != 1 && x < width && yc < height)
        {
            int in[4] = { (int)(vcopy[h] & 0xff), (int)((vcopy[h] >> 8) & 0xff), (int)((vcopy[h] >> 16) & 0xff), (int)(vcopy[h] >> 24) };
            st4(Q.c + (size_t)yc * pitch + x, in, x, width);
            if (Q.d) st4(Q.d + (size_t)yc * pitch + x, in, x, width);
        }
    }
}

// eedi2_filter_dir_map and the eedi2_expand_dir_map behind it (half height, step 1) in one launch.  A workgroup makes the
// filtered map of its FE_R rows x 256 columns AND of the one-pixel ring around them (a dword column either side, a row
// above and below) in LDS, expands out of that, and stores the expanded map only: the filtered map never reaches memory.
// a = mskp, b = map in, c = out (which must not be b: the ring of a tile is its neighbours' input).
#ifndef FE_ROWS
#define FE_ROWS 14
#endif
constexpr int FE_R = FE_ROWS, FE_LW = 68;

// STEP 2 is the pair of _2x passes (:872-1011) on the lattice of the rows they rebuild: lattice row r is plane row
// y = 2 r + (y0 & 1), its neighbours the plane rows y -+ 2 (where they exist: up_ok / dn_ok), its mask rows y - 1 and y + 1;
// the plane rows between them are the passes' bit_blit and leave as they came.
//
// eedi2_expand_dir_map (:719-773) out of the filtered tile s_f (k_dir_map_c): fe_expand_row decides a dword of output
// row lr - copied, or its pixels with enough usable neighbours queued for the vote -, fe_expand_finish votes and stores.
// cand: the row's pixels on the mask and inside the row (one flag byte each).
__device__ __forceinline__ void fe_expand_row(const uint32_t (*s_f)[FE_LW], int lr, int tx, uint32_t cand, bool up_ok, bool dn_ok,
                                              uint8_t (*s_out)[256], uint16_t *s_list, int *s_count)
{
    const uint32_t own = s_f[lr + 1][tx + 1];
    cand &= ff_bytes(own) >> 7;                                                      // expand only fills peak pixels
    if (cand)
    {
        const Win12 none = { 0xffffffffu, 0xffffffffu, 0xffffffffu };
        const Win12 wc = { s_f[lr + 1][tx], own, s_f[lr + 1][tx + 2] };
        const Win12 wu = up_ok ? Win12{ s_f[lr][tx], s_f[lr][tx + 1], s_f[lr][tx + 2] } : none;
        const Win12 wd = dn_ok ? Win12{ s_f[lr + 2][tx], s_f[lr + 2][tx + 1], s_f[lr + 2][tx + 2] } : none;
        uint32_t nc, nu, nd;
        const uint32_t s3c = live3(wc, nc), s3u = live3(wu, nu), s3d = live3(wd, nd);
        const uint32_t u = s3c - nc + s3u + s3d;                                     // the centre is left out (:671)
        const uint32_t enough = ((u + (uint32_t)(0x80 - 5) * 0x01010101u) >> 7) & 0x01010101u;
        const uint32_t sortpx = cand & enough;
        if (sortpx)
        {
            int at = atomicAdd(s_count, __popc(sortpx));
#pragma unroll
            for (int k = 0; k < 4; k++)
                if ((sortpx >> (8 * k)) & 1u) s_list[at++] = (uint16_t)((lr << 8) | (4 * tx + k));
        }
        (void)nu; (void)nd;
    }
    *reinterpret_cast<uint32_t *>(&s_out[lr][4 * tx]) = own;
}

// POST (the pair in front of eedi2_post_process, :1349-1378, which rides along as in k_dir_map_c): d = where the filtered
// map goes as well (the reference leaves it in dst2mp), f = the picture; the old map is the pass's own input, so the
// eedi2_bit_blit that keeps a copy of it is not needed - the caller has the input in the plane the copy would go to.
template <int STEP, bool POST>
__global__ __launch_bounds__(256) void k_dir_map_fe(P3 P, uint32_t padv)
{
    __shared__ uint32_t s_f[FE_R + 2][FE_LW];                       // lattice rows rb - 1 .. rb + FE_R, dword columns -1 .. 64
    __shared__ __attribute__((aligned(16))) uint8_t s_out[FE_R][256];
    __shared__ uint16_t s_list[FE_R * 256];
    __shared__ int s_count;
    __shared__ uint8_t s_lim[LIM_PAD];
    FIELD_PLANE(P);
    const int y0 = STEP == 1 ? 1 : 2 - tff, par = STEP == 1 ? 0 : (y0 & 1);
    const int rb = blockIdx.y * FE_R;
    const int bx0 = 256 * blockIdx.x, x = bx0 + 4 * threadIdx.x;
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    if (bx0 >= pitch || STEP * rb >= height) return;
    auto row_ok = [&](int y) { return STEP == 1 ? (y >= 1 && y < height - 1) : (y >= y0 && y < height - 1); };
    if (maskless)
    {
        // no mask pixel in the plane: both passes are their bit_blit of a map of peaks (k_dir_map4)
        if (x < pitch)
            for (int lr = threadIdx.y; lr < FE_R; lr += 4)
#pragma unroll
                for (int i = 0; i < STEP; i++)
                    if (STEP * (rb + lr) + i < height)
                    {
                        const size_t at = (size_t)(STEP * (rb + lr) + i) * pitch + x;
                        *reinterpret_cast<uint32_t *>(Q.c + at) = pad_bytes(0xffffffffu, x, width, padv);
                        if (POST && x < width) { const int out[4] = { PEAK, PEAK, PEAK, PEAK }; st4(Q.d + at, out, x, width); }
                    }
        return;
    }
    const int tid = threadIdx.y * 64 + threadIdx.x;
    if (tid == 0) s_count = 0;
    lim_fill(s_lim, tid);
    // the rows between the lattice's: fetched now, stored at the end
    constexpr int NCOPY = STEP == 1 ? 1 : (FE_R + 3) / 4;
    uint32_t vcopy[NCOPY];
    if (STEP != 1)
#pragma unroll
        for (int h = 0; h < NCOPY; h++)
        {
            const int lr = threadIdx.y + 4 * h, yc = 2 * (rb + lr) + 1 - par;
            vcopy[h] = (lr < FE_R && yc < height && x < width) ? *reinterpret_cast<const uint32_t *>(Q.b + (size_t)yc * pitch + x) : 0u;
        }
    // SIDE 0: the whole dword; -1 / +1: the ring's dword left / right of the tile, of which only the pixel next to the tile
    // is ever read (half the vote)
    auto filtered = [&](int r, int xx, auto side) -> uint32_t {
        constexpr int SIDE = decltype(side)::value;
        const int y = STEP * r + par;
        if (r < 0 || y >= height || xx < 0 || xx >= width) return 0u;
        const uint8_t *dc = Q.b + (size_t)y * pitch + xx;
        const uint32_t own = *reinterpret_cast<const uint32_t *>(dc);
        if (!row_ok(y)) return own;
        const uint8_t *mk = Q.a + (size_t)y * pitch + xx;
        const uint32_t m0 = *reinterpret_cast<const uint32_t *>(STEP == 1 ? mk : mk - (ptrdiff_t)pitch);
        const uint32_t m1 = STEP == 1 ? 0u : *reinterpret_cast<const uint32_t *>(mk + pitch);
        uint32_t work = ((ff_bytes(m0) | ff_bytes(m1)) >> 7) & mf_bytes_in(xx, 1, width - 1);
        if (SIDE < 0) work &= 0xff000000u;
        if (SIDE > 0) work &= 0x000000ffu;
        uint32_t res = own;
        if (work)
        {
            asm volatile("" ::: "memory");
            const bool up_ok = STEP == 1 || y > 1, dn_ok = STEP == 1 || y < height - 2;
            const Win12 none = { 0xffffffffu, 0xffffffffu, 0xffffffffu };
            const Win12 wc = ldwin(dc), wu = up_ok ? ldwin(dc - (ptrdiff_t)STEP * pitch) : none, wd = dn_ok ? ldwin(dc + (ptrdiff_t)STEP * pitch) : none;
            const uint32_t p01 = SIDE < 0 ? 0u : dir_map_pair<0>(wu, wc, wd, 0), p23 = SIDE > 0 ? 0u : dir_map_pair<2>(wu, wc, wd, 0);
            const uint32_t votes = __builtin_amdgcn_perm(p23, p01, 0x06040200u);
            const uint32_t sel = work * 255u;
            res = (res & ~sel) | (votes & sel);
        }
        return res;
    };
    for (int lr = threadIdx.y; lr < FE_R + 2; lr += 4) s_f[lr][threadIdx.x + 1] = filtered(rb - 1 + lr, x, std::integral_constant<int, 0>());
    static_assert(FE_R + 2 <= 64, "a lane per ring row");
    if (threadIdx.y == 0 && (int)threadIdx.x < FE_R + 2) s_f[threadIdx.x][0] = filtered(rb - 1 + (int)threadIdx.x, bx0 - 4, std::integral_constant<int, -1>());
    if (threadIdx.y == 1 && (int)threadIdx.x < FE_R + 2) s_f[threadIdx.x][65] = filtered(rb - 1 + (int)threadIdx.x, bx0 + 256, std::integral_constant<int, 1>());
    __syncthreads();
    for (int lr = threadIdx.y; lr < FE_R; lr += 4)
    {
        const int y = STEP * (rb + lr) + par;
        if (x >= width || y >= height) continue;
        uint32_t cand = 0;
        if (row_ok(y))
        {
            const uint8_t *mk = Q.a + (size_t)y * pitch + x;
            const uint32_t m0 = *reinterpret_cast<const uint32_t *>(STEP == 1 ? mk : mk - (ptrdiff_t)pitch);
            const uint32_t m1 = STEP == 1 ? 0u : *reinterpret_cast<const uint32_t *>(mk + pitch);
            cand = ((ff_bytes(m0) | ff_bytes(m1)) >> 7) & mf_bytes_in(x, 1, width - 1);
        }
        fe_expand_row(s_f, lr, threadIdx.x, cand, STEP == 1 || y > 1, STEP == 1 || y < height - 2, s_out, s_list, &s_count);
    }
    __syncthreads();
    const int count = s_count;
    for (int i = tid; i < count; i += 256)
    {
        const int e = s_list[i], ly = e >> 8, lx = e & 255;
        const int y = STEP * (rb + ly) + par;
        const uint8_t *c = reinterpret_cast<const uint8_t *>(&s_f[ly + 1][1]) + lx;
        const uint8_t *up = c - 4 * FE_LW, *dn = c + 4 * FE_LW;
        s_out[ly][lx] = (uint8_t)dir_map_px(up[-1], up[0], up[1], c[-1], c[0], c[1], dn[-1], dn[0], dn[1],
                                            STEP == 1 || y > 1, STEP == 1 || y < height - 2, 1, s_lim);
    }
    __syncthreads();
    // the padding of the rows (padv): 255 where the reference expands into a plane that the fill in front of the passes
    // (calc_directions', mark_directions_2x's memset) has covered, 0 where into one nothing has
#pragma unroll
    for (int h = 0; h < (FE_R + 3) / 4; h++)
    {
        const int lr = threadIdx.y + 4 * h, y = STEP * (rb + lr) + par;
        if (lr >= FE_R || x >= pitch) continue;
        auto put_d = [&](int yy, uint32_t v) {                     // the filtered map: the row's pixels only, as the pass's bit_blit
            if (x >= width) return;
            uint8_t *o = Q.d + (size_t)yy * pitch + x;
            if (x + 3 < width) *reinterpret_cast<uint32_t *>(o) = v;
            else for (int k = 0; k < 4 && x + k < width; k++) o[k] = (uint8_t)(v >> (8 * k));
        };
        if (y < height)
        {
            const uint32_t v = x < width ? *reinterpret_cast<const uint32_t *>(&s_out[lr][4 * threadIdx.x]) : 0u;
            *reinterpret_cast<uint32_t *>(Q.c + (size_t)y * pitch + x) = pad_bytes(v, x, width, padv);
            if (POST) put_d(y, s_f[lr + 1][threadIdx.x + 1]);
            if (POST && x < width && row_ok(y))
            {
                const size_t at = (size_t)y * pitch + x;
                const uint32_t om4 = *reinterpret_cast<const uint32_t *>(Q.b + at);
                uint8_t *d = Q.f + at;
                const uint32_t up4 = *reinterpret_cast<const uint32_t *>(d - pitch), dn4 = *reinterpret_cast<const uint32_t *>(d + pitch);
                const uint32_t cur4 = *reinterpret_cast<const uint32_t *>(d);
                int out[4];
                bool any = false;
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const int nm = (v >> (8 * k)) & 0xffu, om = (om4 >> (8 * k)) & 0xffu;
                    const int lim = s_lim[iabs(nm - NEUTRAL) >> 2];
                    const bool fix = iabs(nm - om) > lim && om != PEAK && om != NEUTRAL;
                    out[k] = fix ? (int)((((up4 >> (8 * k)) & 0xffu) + ((dn4 >> (8 * k)) & 0xffu) + 1) >> 1) : (int)((cur4 >> (8 * k)) & 0xffu);
                    any |= fix;
                }
                if (any) st4(d, out, x, width);
            }
        }
        if (STEP != 1)
        {
            const int yc = 2 * (rb + lr) + 1 - par;
            if (yc < height)
            {
                *reinterpret_cast<uint32_t *>(Q.c + (size_t)yc * pitch + x) = pad_bytes(vcopy[h < NCOPY ? h : 0], x, width, padv);
                if (POST) put_d(yc, vcopy[h < NCOPY ? h : 0]);
            }
        }
    }
}

// (Measured and dropped, DESIGN.md 4.2.3: the filter pass's votes on a queue - the tile staged in LDS, the aligned pixel
// pairs with a mask pixel listed, a lane per listed pair - 131 us per launch against the 103 of this form: a wave here
// votes for all 64 of its dwords when one holds a mask pixel, and half of those votes are for nothing on the bench's
// pictures, but the list costs what it saves - 48 M vector instructions per launch against 42 + 6 M - and adds four barriers.)

// a = mskp, b = dmsk in, c = out.  eedi2_filter_map (:538-635): a pixel that carries a direction loses it (-> peak) when
// the map breaks along that direction on the row above AND on the row below - two short walks (at most 9 pixels: the
// direction / 16) with an early exit.  One pixel per thread made this pass 70 scalar and 38 vector instructions per wave
// around four dependent byte loads from memory (mask, map, then a pair per step of each walk): it ran at the scalar unit's
// rate and the memory latency, not at its bytes.  Here a workgroup stages its 4 + 2 rows of the map (256 columns + 8 either
// side) in LDS with dword loads, a thread owns an aligned dword of its row (mask and output as dwords), and the walks of
// its four pixels read bytes from LDS.
constexpr int FM_W = 256, FM_R = 4, FM_HALO = 8, FM_LW = FM_W + 2 * FM_HALO;

// The test of a step (:569-571 and its three siblings) without a branch per clause: every clause is the SIGN of an
// integer - lim - |s - ref| is negative when s is off by more than lim, s - 255 is negative unless s is the peak,
// 509 - c - s is negative only when both are the peak - and the step trips when the sign of
// (off(s) & notpeak(s)) | (off(c) & notpeak(c)) | bothpeak is set.  v_sad_u16 of two bytes is their absolute difference.
// (Every s, c, ref and limit: tests/test_eedi2_identities_cpu.py, which also holds the unified ranges of the loop below.)
__device__ __forceinline__ int fm_step(uint32_t s, uint32_t c, uint32_t ref, int lim)
{
    const int off_s = lim - (int)__builtin_amdgcn_sad_u16(s, ref, 0u), off_c = lim - (int)__builtin_amdgcn_sad_u16(c, ref, 0u);
    return (off_s & (int)(s - 255u)) | (off_c & (int)(c - 255u)) | (509 - (int)c - (int)s);
}

__global__ __launch_bounds__(256) void k_filter_map(P3 P)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_d[FM_R + 2][FM_LW];
    FIELD_PLANE(P);
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    const int bx0 = blockIdx.x * FM_W, by0 = blockIdx.y * FM_R;
    if (bx0 >= width || by0 >= height) return;                   // whole workgroup outside
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int x = bx0 + 4 * threadIdx.x, y = by0 + threadIdx.y;
    if (maskless)
    {
        // no mask pixel in the plane: the pass is its bit_blit (:547) of a map of peaks (k_dir_map4)
        if (x < width && y < height)
        {
            uint8_t *o = Q.c + (size_t)y * pitch + x;
            if (x + 3 < width) *reinterpret_cast<uint32_t *>(o) = 0xffffffffu;
            else for (int k = 0; k < 4 && x + k < width; k++) o[k] = (uint8_t)PEAK;
        }
        return;
    }
    const bool inside = x < width && y < height;
    // The thread's own dword and its mask first: a workgroup without a candidate (mask peak, map not peak, inside the frame
    // of :557-560) - off the mask, like whole chroma planes without edges - copies its rows and stages nothing.
    uint32_t m4 = 0, own = 0;
    if (inside)
    {
        own = *reinterpret_cast<const uint32_t *>(Q.b + (size_t)y * pitch + x);
        if (y >= 1 && y < height - 1) m4 = *reinterpret_cast<const uint32_t *>(Q.a + (size_t)y * pitch + x);   // (m4 stays 0 on the first / last row)
    }
    uint32_t cand = ((ff_bytes(m4) & ~ff_bytes(own)) >> 7) & mf_bytes_in(x, 1, width - 1);
    if (!__syncthreads_or(cand != 0u))                           // block-uniform
    {
        if (inside)
        {
            uint8_t *o = Q.c + (size_t)y * pitch + x;
            if (x + 3 < width) *reinterpret_cast<uint32_t *>(o) = own;
            else for (int k = 0; k < 4 && x + k < width; k++) o[k] = (uint8_t)(own >> (8 * k));
        }
        return;
    }
    {
        // 6 x 68 dwords for 256 threads: both loads of a thread in flight before the first LDS store
        static_assert((FM_R + 2) * (FM_LW / 4) <= 2 * 256, "two staged dwords per thread");
        uint32_t v[2] = { 0u, 0u };                              // outside the plane: never looked at (the walks are clipped to the row)
#pragma unroll
        for (int k = 0; k < 2; k++)
        {
            const int i = tid + 256 * k, r = i / (FM_LW / 4), c4 = i - r * (FM_LW / 4);
            const int yy = by0 - 1 + r, col = bx0 - FM_HALO + 4 * c4;
            if (i < (FM_R + 2) * (FM_LW / 4) && yy >= 0 && yy < height && col >= 0 && col < pitch)
                v[k] = *reinterpret_cast<const uint32_t *>(Q.b + (size_t)yy * pitch + col);
        }
#pragma unroll
        for (int k = 0; k < 2; k++)
            if (tid + 256 * k < (FM_R + 2) * (FM_LW / 4)) reinterpret_cast<uint32_t *>(&s_d[0][0])[tid + 256 * k] = v[k];
    }
    __syncthreads();
    if (!inside) return;
    const uint8_t *rc = &s_d[threadIdx.y + 1][FM_HALO + 4 * threadIdx.x];
    uint32_t out4 = own;
    // Both walks of a pixel in one loop, neither with an early exit: the pixel turns peak when the walk above trips
    // somewhere AND the walk below does (:565-627 take the second only after the first has tripped - the same verdict).
    // With neg = min(dir, 0), pos = max(dir, 0) the four ranges of :565-620 are [max(-x, neg), min(w - x - 1, pos)] above
    // and [max(-x, -pos), min(w - x - 1, -neg)] below; the shorter walk repeats its last step.
#pragma unroll 1
    for (int k = 0; k < 4; k++)
    {
        if (!((cand >> (8 * k)) & 1u)) continue;
        const int xx = x + k;
        const uint8_t *dc = rc + k, *dp = dc - FM_LW, *dn = dc + FM_LW;
        const uint32_t ref = dc[0];
        int dir = ((int)ref - NEUTRAL) >> 2;
        const int lim = max(iabs(dir) * 2, 12 << 2);
        dir >>= 2;
        const int neg = min(dir, 0), pos = max(dir, 0);
        const int tf = max(-xx, neg), tt = min(width - xx - 1, pos), bf = max(-xx, -pos), bt = min(width - xx - 1, -neg);
        const int n = max(tt - tf, bt - bf);
        int any_t = 0, any_b = 0;
        for (int i = 0; i <= n; i++)
        {
            const int jt = min(tf + i, tt), jb = min(bf + i, bt);
            any_t |= fm_step(dp[jt], dc[jt], ref, lim);
            any_b |= fm_step(dn[jb], dc[jb], ref, lim);
        }
        if ((any_t & any_b) < 0) out4 |= 0xffu << (8 * k);
    }
    uint8_t *o = Q.c + (size_t)y * pitch + x;
    if (x + 3 < width) *reinterpret_cast<uint32_t *>(o) = out4;
    else for (int k = 0; k < 4 && x + k < width; k++) o[k] = (uint8_t)(out4 >> (8 * k));
}

// eedi2_mark_directions_2x's vote (:800-868) for the two pixels at columns K and K + 1 of a thread's dword (K = 0 or 2), the
// way dir_map_pair votes: wa / wb = the half-height direction rows above and below the rebuilt row (bytes x - 4 .. x + 7), six
// slots, an absent one (a peak) lifted to PK_ABSENT; returns the pass's value (or PEAK) in the low byte of each half.
template <int K>
__device__ __forceinline__ uint32_t mark_pair(const Win12 &wa, const Win12 &wb)
{
    u16x2 v[6];
    {
        const Win12 *rows[2] = { &wa, &wb };
#pragma unroll
        for (int r = 0; r < 2; r++)
        {
            const uint32_t w0 = rows[r]->w0, w1 = rows[r]->w1, w2 = rows[r]->w2;
            if (K == 0)
            {
                v[3 * r + 0] = pk(PK_BYTES(w1, w0, 3, 4));            // columns -1, 0
                v[3 * r + 1] = pk(PK_BYTES(w1, w0, 4, 5));            // 0, 1
                v[3 * r + 2] = pk(PK_BYTES(w1, w0, 5, 6));            // 1, 2
            }
            else
            {
                v[3 * r + 0] = pk(PK_BYTES(w1, w0, 5, 6));            // 1, 2
                v[3 * r + 1] = pk(PK_BYTES(w1, w0, 6, 7));            // 2, 3
                v[3 * r + 2] = pk(PK_BYTES(w2, w1, 3, 4));            // 3, 4
            }
        }
    }
    const u16x2 one = pk1(1), zero = pk1(0);
    u16x2 a[6], raw[6];
    uint32_t absent = 0;
#pragma unroll
    for (int i = 0; i < 6; i++)
    {
        raw[i] = v[i];
        a[i] = (u16x2)((v[i] + one) >> 8);                              // 1 per half that holds a peak
        absent += un(a[i]);
        v[i] = v[i] + a[i] * pk1(PK_ABSENT_HI);
    }
    // mid6's network and selection with masks: n = 6 - absent present values; n <= 4 <=> absent >= 2, n <= 3 <=> absent >= 3,
    // n <= 5 <=> absent >= 1, n odd <=> absent odd
    u16x2 s0 = v[0], s1 = v[1], s2 = v[2], s3 = v[3], s4 = v[4], s5 = v[5];
    cswap2(s0, s5); cswap2(s1, s3); cswap2(s2, s4);
    cswap2(s1, s2); cswap2(s3, s4);
    cswap2(s0, s3); cswap2(s2, s5);
    cswap2(s0, s1); cswap2(s2, s3); cswap2(s4, s5);
    cswap2(s1, s2); cswap2(s3, s4);
    const u16x2 ab = pk(absent);
    const uint32_t m4 = un(zero - pk_lt(one, ab));                     // absent >= 2
    const uint32_t m3 = un(zero - pk_lt(pk1(2), ab));                  // absent >= 3
    const uint32_t m5 = un(zero - pk_lt(zero, ab));                    // absent >= 1
    const uint32_t modd = un(zero - (ab & one));                       // n odd
#define PK_SEL(m, x, y) (((m) & (x)) | (~(m) & (y)))                    /* v_bfi_b32 */
    const uint32_t lo = PK_SEL(m4, un(s1), un(s2));
    const uint32_t hi = PK_SEL(m3, un(s1), PK_SEL(m5, un(s2), un(s3)));
    const u16x2 mid = pk(PK_SEL(modd, hi, un((u16x2)((pk(lo) + pk(hi) + one) >> 1))));
#undef PK_SEL
    const i16x2 t = __builtin_bit_cast(i16x2, (u16x2)(mid - pk1(NEUTRAL)));
    const u16x2 lim1 = limlut2(__builtin_bit_cast(u16x2, __builtin_elementwise_max(t, (i16x2)(-t))) >> 2) + one;
    // the three pairs across the rebuilt row (:833-835; the third one compares a2 with b0 and excuses b2 - sic)
    auto close = [&](int i, int j, int k) -> u16x2 {
        const u16x2 d = __builtin_elementwise_max(raw[i], raw[j]) - __builtin_elementwise_min(raw[i], raw[j]);
        return pk_lt(d, lim1) | a[i] | a[k];
    };
    const u16x2 u = close(0, 3, 3) + close(1, 4, 4) + close(2, 3, 5);
    u16x2 sum = zero, cnt = zero;
#pragma unroll
    for (int i = 0; i < 6; i++)
    {
        const u16x2 d = __builtin_elementwise_max(v[i], mid) - __builtin_elementwise_min(v[i], mid);
        const u16x2 in = pk_lt(d, lim1);
        cnt += in;
        sum += in * v[i];
    }
    const uint32_t sm = un((u16x2)(sum + mid)), ct = un(cnt), uu = un(u);
    uint32_t out = 0;
#pragma unroll
    for (int h = 0; h < 2; h++)
    {
        const int n = 6 - (int)((absent >> (16 * h)) & 0xffffu);
        const int count = (int)((ct >> (16 * h)) & 0xffffu);
        int res = PEAK;
        // (with fewer than three values the midpoint above may be a PK_ABSENT: nothing of it is used)
        if (n >= 3 && (int)((uu >> (16 * h)) & 0xffffu) >= 2 && !(count < n - 2 || count < 2))
            res = vote_avg((int)((sm >> (16 * h)) & 0xffffu), count + 1) & 0xff;
        out |= (uint32_t)res << (16 * h);
    }
    return out;
}

// a = msk2p, b = dmsk (tmp2p2), c = out (tmp2p)
// Also performs the three eedi2_upscale_by_2 line doublings (:98-108, decomb_template.c:408-410):
// g (srcp) -> d (dst2p), b (dstp, the half-height direction map) -> e (tmp2p2), a (mskp) -> f (msk2p).
// The rows y-1 / y+1 of the doubled maps that mark_directions reads are rows (y-1)>>1 / (y+1)>>1 of
// the half-height ones, so it reads those directly and no separate upscale launch is needed.
// a = mskp, b = dstp, c = out (tmp2p), g = srcp; `height` = full height.

// Four pixels per thread (see k_dir_map4): the three line doublings are dword copies, and only the
// rows mark_directions_2x rebuilds (every other one) load the two neighbouring half-height rows.
// A thread takes the rows 2r and 2r + 1 of its dword column: both are doubled from half-height row r (three loads for six
// stores), one of them at most is a row mark_directions_2x rebuilds, the other is the memset's 255 - a wave of its own
// per row spent more scalar instructions on finding its plane and field than vector ones on its dword.
__global__ __launch_bounds__(256) void k_mark_2x4(P3 P, uint32_t padv)
{
    FIELD_PLANE(P);
    const int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    const int r = blockIdx.y * blockDim.y + threadIdx.y;
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    const int y0 = 2 - tff;
    if (x >= pitch || 2 * r >= height) return;
    const bool two = 2 * r + 1 < height;
    // the row of the pair with the parity of the rebuilt rows, and the other one
    const int y = 2 * r + (y0 & 1), yc = 2 * r + 1 - (y0 & 1);
    const bool rebuilt = y < height && !maskless && y >= y0 && y < height - 1;
    const size_t hs = (size_t)r * pitch + x, fs = (size_t)(2 * r) * pitch + x;
    // every load the thread may need goes out ahead of its first store: the memory counter is in order, and a load behind
    // a store is only known to be there once the store is (the mask rows (y - 1) >> 1 and (y + 1) >> 1 are row r and the one
    // above or below it)
    const uint32_t vg = *reinterpret_cast<const uint32_t *>(Q.g + hs), vb = *reinterpret_cast<const uint32_t *>(Q.b + hs),
                   va = *reinterpret_cast<const uint32_t *>(Q.a + hs);
    uint32_t k0w = 0, k1w = 0;
    if (rebuilt)
    {
        const int ra = (y - 1) >> 1, rb = (y + 1) >> 1;
        k0w = ra == r ? va : *reinterpret_cast<const uint32_t *>(Q.a + (ptrdiff_t)ra * pitch + x);
        k1w = rb == r ? va : *reinterpret_cast<const uint32_t *>(Q.a + (ptrdiff_t)rb * pitch + x);
    }
    // the mask first: a thread none of whose four pixels sits under or above a mask pixel writes peaks and never fetches the
    // direction rows (k_dir_map4)
    const bool vote = rebuilt && ((ff_bytes(k0w) | ff_bytes(k1w)) & mf_bytes_in(x, 1, width - 1)) != 0u;
    Win12 wa = { 0u, 0u, 0u }, wbn = { 0u, 0u, 0u };
    if (vote)
    {
        wa = ldwin(Q.b + (ptrdiff_t)((y - 1) >> 1) * pitch + x);
        wbn = ldwin(Q.b + (ptrdiff_t)((y + 1) >> 1) * pitch + x);
    }
    {
        *reinterpret_cast<uint32_t *>(Q.d + fs) = vg;
        *reinterpret_cast<uint32_t *>(Q.e + fs) = vb;
        *reinterpret_cast<uint32_t *>(Q.f + fs) = va;
        if (two)
        {
            *reinterpret_cast<uint32_t *>(Q.d + fs + pitch) = vg;
            *reinterpret_cast<uint32_t *>(Q.e + fs + pitch) = vb;
            *reinterpret_cast<uint32_t *>(Q.f + fs + pitch) = va;
        }
    }
    // memset(dstp, 255, pitch*height); padv: what the rows' padding gets (0 when the map goes to the plane the fused dir-map
    // pass reads, whose padding the reference never writes)
    if (yc < height) *reinterpret_cast<uint32_t *>(Q.c + (size_t)yc * pitch + x) = pad_bytes(0xffffffffu, x, width, padv);
    if (y >= height) return;
    uint32_t *o = reinterpret_cast<uint32_t *>(Q.c + (size_t)y * pitch + x);
    uint32_t packed = 0xffffffffu;
    if (!vote)                                                 // (no mask pixel in the plane or near the thread: nothing but the memset, :800)
    {
        *o = pad_bytes(packed, x, width, padv);
        return;
    }
    {
        // the pixels under or above a mask pixel, one flag byte each; the vote on pixel pairs (mark_pair)
        const uint32_t work = ((ff_bytes(k0w) | ff_bytes(k1w)) >> 7) & mf_bytes_in(x, 1, width - 1);
        asm volatile("" ::: "memory");                            // keep the branch above (see k_dir_map4)
        const uint32_t p01 = mark_pair<0>(wa, wbn), p23 = mark_pair<2>(wa, wbn);
        const uint32_t votes = __builtin_amdgcn_perm(p23, p01, 0x06040200u);
        const uint32_t sel = work * 255u;
        packed = (packed & ~sel) | (votes & sel);
    }
    *o = pad_bytes(packed, x, width, padv);
}

// fill_gaps_2x.  a = msk2p, b = dmsk in, c = out.  Every pixel of a fillable gap computes the same
// (u, v, back, forward, verdict) as its neighbours in the gap (:1053-1120), so each thread only writes its own pixel.
// Two things cost time in a pixel-per-thread form: most waves exist only to copy 64 bytes, and the few pixels that
// really are gaps (direction unknown inside the mask) walk left and right on dependent byte loads (:1053-1073) while
// the rest of their wave idles.  Here a workgroup owns FG_W consecutive pixels of one rebuilt row:
//   1. the seven rows involved (dc = y, mask y-1 / y+1 / y-3 / y+3, direction y-2 / y+2) are staged in LDS with
//      FG_HALO pixels either side (dwords, same flat addressing);
//   2. every thread takes four pixels of the span: copies them to the output row (in LDS) and appends the gap pixels
//      among them to a list - the gap pixels of the span end up in consecutive lanes;
//   3. the list is worked off one gap pixel per thread; a walk reads LDS while it stays inside the staged span and
//      falls back to memory beyond it (rare);
//   4. the output row leaves as dwords.
// Rows the pass does not rebuild are only copied (the reference's bit_blit), decided per workgroup.
// (Tried and dropped: four pixels per thread without compaction - nearly every wave then pays four serial walks, 45 us
// against 29; the walk on LDS without compaction, 38 us.)
#ifndef FG_WIDTH
#define FG_WIDTH 1024
#endif
constexpr int FG_W = FG_WIDTH, FG_T = FG_W / 4, FG_HALO = 64, FG_LW = FG_W + 2 * FG_HALO;


// The walks are done on bitmaps: walked byte by byte, every pixel of a gap goes the gap's whole length on dependent
// reads (1.9 M VALU instructions in a 24.7 us kernel: its time was the longest chain).  The staged rows are first
// turned into four bit rows (ballots), and a pixel finds its gap's ends, and whether the rows above / below break
// the gap's support, with a few 64-bit operations; only the min / max over a supported gap still walks bytes
// (independent reads).  Walks that leave the staged span take the byte path (rare).
// FG_R rebuilt rows (and the FG_R copied rows between them) per workgroup: the rows a rebuilt row looks at (direction
// rows y - 2 / y / y + 2, mask rows y - 3 / y - 1 / y + 1 / y + 3) are its neighbours' too - 2 FG_R + 5 staged rows serve
// FG_R of them instead of 7 serving one - and a quarter of the waves find their plane and field (the one-row form ran
// at the scalar unit's rate: 104 scalar instructions per wave, half of the waves only there to copy a row or to return).
#ifndef FG_ROWS
#define FG_ROWS 4
#endif
constexpr int FG_R = FG_ROWS, FG_ND = FG_R + 2, FG_NM = FG_R + 3, FG_WORDS = FG_LW / 64 + 1;
__global__ __launch_bounds__(FG_T) void k_fill_gaps_b(P3 P)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_d[FG_ND][FG_LW];      // direction rows yb - 2, yb, .. , yb + 2 FG_R
    __shared__ __attribute__((aligned(16))) uint8_t s_m[FG_NM][FG_LW];      // mask rows yb - 3, yb - 1, .. , yb + 2 FG_R + 1
    __shared__ __attribute__((aligned(16))) uint8_t s_out[FG_R][FG_W];
    __shared__ uint16_t s_list[FG_R * FG_W];
    __shared__ int s_count;
    __shared__ uint64_t s_stop[FG_R][FG_WORDS], s_np[FG_R][FG_WORDS], s_bt[FG_R][FG_WORDS], s_bb[FG_R][FG_WORDS];   // one bit per staged column
    FIELD_PLANE(P);
    const int y0 = 2 - tff;
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    const int x0 = blockIdx.x * FG_W, tid = threadIdx.x;
    const int ya = (int)blockIdx.y * 2 * FG_R;                     // the workgroup's 2 FG_R plane rows: ya .. ya + 2 FG_R - 1
    if (ya >= height || x0 >= width) return;
    const int yb = ya + (y0 & 1);                                  // its rows of the rebuilt parity: yb, yb + 2, ..
    const int x = x0 + 4 * tid;
    if (maskless)
    {
        // no mask pixel in the plane, no gap to fill (:1048-1050): the pass is its bit_blit of a map of peaks (k_dir_map4)
        if (x < width)
        {
#pragma unroll
            for (int i = 0; i < 2 * FG_R; i++)
            {
                const int y = ya + i;
                if (y >= height) break;
                uint8_t *o = Q.c + (size_t)y * pitch + x;
                if (x + 3 < width) *reinterpret_cast<uint32_t *>(o) = 0xffffffffu;
                else for (int k = 0; k < 4 && x + k < width; k++) o[k] = (uint8_t)PEAK;
            }
        }
        return;
    }
    auto rebuilt = [&](int y) { return y >= y0 && y < height - 1; };
    // the rows that are only copied (the reference's bit_blit): the other parity, and what lies outside y0 .. height - 2
    uint32_t vcopy[2 * FG_R];
#pragma unroll
    for (int i = 0; i < 2 * FG_R; i++)
    {
        const int y = ya + i;
        vcopy[i] = 0;
        if (x < width && y < height && !((((y - y0) & 1) == 0) && rebuilt(y)))
            vcopy[i] = *reinterpret_cast<const uint32_t *>(Q.b + (size_t)y * pitch + x);
    }
    if (tid == 0) s_count = 0;
    const int lo = x0 - FG_HALO;                                   // column of staged byte 0 (a multiple of 4)
    const int ndw = (min(FG_W, hbhip_align_up_dev(width - x0, 4)) + 2 * FG_HALO) / 4;
    {
        // all loads of a thread in flight before its first LDS store (ndw <= 288: two dwords per row and thread); rows
        // outside the plane are rows no pixel's tests reach (:1076, :1090): they are read from the nearest row inside
        const bool h0 = tid < ndw, h1 = tid + FG_T < ndw;
        uint32_t vd[FG_ND][2] = {}, vm[FG_NM][2] = {};
        if (h0)
        {
#pragma unroll
            for (int r = 0; r < FG_ND; r++)
                vd[r][0] = reinterpret_cast<const uint32_t *>(Q.b + (size_t)min(max(yb - 2 + 2 * r, 0), height - 1) * pitch + lo)[tid];
#pragma unroll
            for (int r = 0; r < FG_NM; r++)
                vm[r][0] = reinterpret_cast<const uint32_t *>(Q.a + (size_t)min(max(yb - 3 + 2 * r, 0), height - 1) * pitch + lo)[tid];
        }
        if (h1)
        {
#pragma unroll
            for (int r = 0; r < FG_ND; r++)
                vd[r][1] = reinterpret_cast<const uint32_t *>(Q.b + (size_t)min(max(yb - 2 + 2 * r, 0), height - 1) * pitch + lo)[tid + FG_T];
#pragma unroll
            for (int r = 0; r < FG_NM; r++)
                vm[r][1] = reinterpret_cast<const uint32_t *>(Q.a + (size_t)min(max(yb - 3 + 2 * r, 0), height - 1) * pitch + lo)[tid + FG_T];
        }
        if (h0)
        {
#pragma unroll
            for (int r = 0; r < FG_ND; r++) reinterpret_cast<uint32_t *>(s_d[r])[tid] = vd[r][0];
#pragma unroll
            for (int r = 0; r < FG_NM; r++) reinterpret_cast<uint32_t *>(s_m[r])[tid] = vm[r][0];
        }
        if (h1)
        {
#pragma unroll
            for (int r = 0; r < FG_ND; r++) reinterpret_cast<uint32_t *>(s_d[r])[tid + FG_T] = vd[r][1];
#pragma unroll
            for (int r = 0; r < FG_NM; r++) reinterpret_cast<uint32_t *>(s_m[r])[tid + FG_T] = vm[r][1];
        }
    }
    __syncthreads();
    // rebuilt row k = yb + 2k: direction rows s_d[k] (y - 2), s_d[k + 1] (y), s_d[k + 2] (y + 2); mask rows s_m[k] (y - 3),
    // s_m[k + 1] (y - 1), s_m[k + 2] (y + 1), s_m[k + 3] (y + 3)
    const unsigned staged = 4u * (unsigned)ndw;
#pragma unroll
    for (int r = 0; r < FG_R; r++)
    {
        const int y = yb + 2 * r;
        if (x < width && y < height && rebuilt(y))                              // (rows that are copied have their dword in vcopy)
        {
            const int c = 4 * tid + FG_HALO;
            const uint32_t cw = *reinterpret_cast<const uint32_t *>(&s_d[r + 1][c]);
            const uint32_t mcw = *reinterpret_cast<const uint32_t *>(&s_m[r + 1][c]), mnw = *reinterpret_cast<const uint32_t *>(&s_m[r + 2][c]);
            *reinterpret_cast<uint32_t *>(&s_out[r][4 * tid]) = cw;
            // the gap pixels among the four (direction unknown, inside the mask: :1046-1050) as byte arithmetic, one atomic
            // for all of them
            const uint32_t gap = ((ff_bytes(cw) & (ff_bytes(mcw) | ff_bytes(mnw))) >> 7) & mf_bytes_in(x, 1, width - 1);
            if (gap)
            {
                int at = atomicAdd(&s_count, __popc(gap));
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if ((gap >> (8 * k)) & 1u) s_list[at++] = (uint16_t)((r << 12) | (4 * tid + k));
            }
        }
    }
    __syncthreads();
    const int count = s_count;
#ifdef HBHIP_DEV_STATS
    if (tid == 0) CD_STAT(12, 1);
#endif
    if (count)                                                                     // block-uniform: no gap pixel (off the mask,
    {                                                                              // whole chroma planes without edges), no bitmaps
        // Per staged column and rebuilt row: does a walk stop here (a known direction, or outside the mask: :1055-1058), is the
        // direction known, does the row above / below end the "top / bottom continues" state (:1078-1093).  The walks then
        // are bit scans instead of chains of dependent byte reads.  First the rows' own bits - direction known, mask set -
        // once per staged row, then the four combinations per rebuilt row as word arithmetic.
        for (int k = 0; k < (FG_LW + FG_T - 1) / FG_T; k++)
        {
            // (no branch: the bytes are read whatever they hold - from the last staged column for the lanes past it, whose
            // bits are then cleared)
            const unsigned col = tid + FG_T * k, cc = min(col, staged - 1u);
            const bool in = col < staged;
            uint64_t nd[FG_ND], pm[FG_NM];
    #pragma unroll
            for (int r = 0; r < FG_ND; r++) nd[r] = __ballot(in & (s_d[r][cc] != PEAK));       // direction known
    #pragma unroll
            for (int r = 0; r < FG_NM; r++) pm[r] = __ballot(in & (s_m[r][cc] == PEAK));       // on the mask
            const uint64_t inw = __ballot(in);
            if ((tid & 63) == 0 && (col >> 6) < (unsigned)FG_WORDS)
            {
    #pragma unroll
                for (int r = 0; r < FG_R; r++)
                {
                    const uint64_t mc = pm[r + 1], mn = pm[r + 2];
                    s_np[r][col >> 6] = nd[r + 1];
                    s_stop[r][col >> 6] = nd[r + 1] | (inw & ~mc & ~mn);
                    s_bt[r][col >> 6] = inw & (~nd[r] | (~pm[r] & ~mc));
                    s_bb[r][col >> 6] = inw & (~nd[r + 2] | (~mn & ~pm[r + 3]));
                }
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < count; i += FG_T)
    {
        const int e = s_list[i], r = e >> 12, lx = e & 0xfff, px = x0 + lx, y = yb + 2 * r;
        const uint8_t *DC = s_d[r + 1], *DP = s_d[r], *DN = s_d[r + 2];
        const uint8_t *MC = s_m[r + 1], *MN = s_m[r + 2], *MP = s_m[r], *MNN = s_m[r + 3];
        // (columns outside the staged span: from memory; the rows these are read from exist for every pixel that gets here)
        const uint8_t *gd = Q.b + (size_t)y * pitch, *gma = Q.a + (ptrdiff_t)(y - 1) * pitch;
        auto rd = [&](const uint8_t *srow, const uint8_t *grow, int col) -> int {
            const unsigned k = (unsigned)(col - lo);
            return k < staged ? (int)srow[k] : (int)grow[col];
        };
        int u = px - 1, back = 500, forward = -500;
        int v = px + 1;
        int tc = 1, bc = 1, mint = 500, maxt = -20, minb = 500, maxb = -20;
        // the two walks as bit scans over the staged columns [first, last) = plane columns [max(lo, 1), min(lo + staged, width))
        const int c = px - lo;
        const int first = max(1 - lo, 0), last = min((int)staged, width - lo);
        int ul = -1, vl = -1;
        for (int wi = (c - 1) >> 6; wi >= (first >> 6) && c - 1 >= first; wi--)      // highest stop bit in [first, c - 1]
        {
            uint64_t w = s_stop[r][wi];
            if (wi == ((c - 1) >> 6) && ((c - 1) & 63) != 63) w &= (2ull << ((c - 1) & 63)) - 1ull;
            if (wi == (first >> 6)) w &= ~0ull << (first & 63);
            if (w) { ul = 64 * wi + 63 - __clzll((long long)w); break; }
        }
        for (int wi = (c + 1) >> 6; wi <= ((last - 1) >> 6) && c + 1 < last; wi++)   // lowest stop bit in [c + 1, last - 1]
        {
            uint64_t w = s_stop[r][wi];
            if (wi == ((c + 1) >> 6)) w &= ~0ull << ((c + 1) & 63);
            if (wi == ((last - 1) >> 6) && ((last - 1) & 63) != 63) w &= (2ull << ((last - 1) & 63)) - 1ull;
            if (w) { vl = 64 * wi + __ffsll((long long)w) - 1; break; }
        }
        // inside the staged span when each walk found its stop there or ran into the row end inside it
        const bool fast = (ul >= 0 || lo <= 1) && (vl >= 0 || lo + (int)staged > width);     // column `width` itself must be staged too
        if (fast)
        {
            if (ul >= 0) { u = lo + ul; if ((s_np[r][ul >> 6] >> (ul & 63)) & 1ull) back = DC[ul]; }
            else u = 0;
            if (vl >= 0) { v = lo + vl; if ((s_np[r][vl >> 6] >> (vl & 63)) & 1ull) forward = DC[vl]; }
            else v = width;
            // columns u .. v (v = width included, as the reference's loop includes it) are staged
            const int a0 = u - lo, a1 = v - lo;
            auto any_in = [&](const uint64_t *bits) {
                for (int wi = a0 >> 6; wi <= (a1 >> 6); wi++)
                {
                    uint64_t w = bits[wi];
                    if (wi == (a0 >> 6)) w &= ~0ull << (a0 & 63);
                    if (wi == (a1 >> 6) && (a1 & 63) != 63) w &= (2ull << (a1 & 63)) - 1ull;
                    if (w) return true;
                }
                return false;
            };
            if (y <= 2 || any_in(s_bt[r])) { tc = 0; mint = maxt = 20; }
            else for (int j = a0; j <= a1; j++) { const int t = DP[j]; mint = min(mint, t); maxt = max(maxt, t); }
            if (y >= height - 3 || any_in(s_bb[r])) { bc = 0; minb = maxb = 20; }
            else for (int j = a0; j <= a1; j++) { const int t = DN[j]; minb = min(minb, t); maxb = max(maxb, t); }
        }
        else
        {
            const uint8_t *gmn = gma + 2 * (ptrdiff_t)pitch;
            while (u)
            {
                const int d = rd(DC, gd, u);
                if (d != PEAK) { back = d; break; }
                if (rd(MC, gma, u) != PEAK && rd(MN, gmn, u) != PEAK) break;
                u--;
            }
            while (v < width)
            {
                const int d = rd(DC, gd, v);
                if (d != PEAK) { forward = d; break; }
                if (rd(MC, gma, v) != PEAK && rd(MN, gmn, v) != PEAK) break;
                v++;
            }
            // (rows y - 2 / y - 3 and y + 2 / y + 3 are only looked at where the reference looks at them: :1076, :1090)
            const uint8_t *gdp = gd - 2 * (ptrdiff_t)pitch, *gdn = gd + 2 * (ptrdiff_t)pitch;
            const uint8_t *gmp = gma - 2 * (ptrdiff_t)pitch, *gmnn = gmn + 2 * (ptrdiff_t)pitch;
            for (int j = u; j <= v; j++)
            {
                if (tc)
                {
                    int t;
                    if (y <= 2 || (t = rd(DP, gdp, j)) == PEAK || (rd(MP, gmp, j) != PEAK && rd(MC, gma, j) != PEAK)) { tc = 0; mint = maxt = 20; }
                    else { mint = min(mint, t); maxt = max(maxt, t); }
                }
                if (bc)
                {
                    int t;
                    if (y >= height - 3 || (t = rd(DN, gdn, j)) == PEAK || (rd(MN, gmn, j) != PEAK && rd(MNN, gmnn, j) != PEAK)) { bc = 0; minb = maxb = 20; }
                    else { minb = min(minb, t); maxb = max(maxb, t); }
                }
            }
        }
#ifdef HBHIP_DEV_STATS
        CD_STAT(13, 1); CD_STAT(14, v - u); CD_STAT(15, fast); CD_STAT(16, tc + bc); CD_STAT(17, (tc + bc) * (v - u + 1));
#endif
        if (maxt == -20) maxt = mint = 20;
        if (maxb == -20) maxb = minb = 20;
        const int far = max(iabs(forward - NEUTRAL), iabs(back - NEUTRAL));
        const int thresh = max(max(far >> 2, 8), max(iabs(mint - maxt), iabs(minb - maxb)));
        const int flim = min(far >> 2, 6);
        if (iabs(forward - back) <= thresh && (v - u - 1 <= flim || tc || bc))
        {
            const double stepd = (double)(forward - back) / (double)(v - u);
            const int j = px - u - 1;
            s_out[r][lx] = (uint8_t)((back + (int)(j * stepd + 0.5)) & 0xff);
        }
    }
    __syncthreads();
    if (x < width)
    {
#pragma unroll
        for (int i = 0; i < 2 * FG_R; i++)
        {
            const int y = ya + i;
            if (y >= height) break;
            const bool work = (((y - y0) & 1) == 0) && rebuilt(y);
            const uint32_t v = work ? *reinterpret_cast<const uint32_t *>(&s_out[(i - (y0 & 1)) >> 1][4 * tid]) : vcopy[i];
            uint8_t *o = Q.c + (size_t)y * pitch + x;
            if (x + 3 < width) *reinterpret_cast<uint32_t *>(o) = v;
            else for (int k = 0; k < 4 && x + k < width; k++) o[k] = (uint8_t)(v >> (8 * k));
        }
    }
}

// lattice_px cut at its two decision points, for k_lattice_cand_q: each stage either finishes the word or hands
// the pixel to the next one.  `base` carries the bits every outcome shares (valA and the right-hand test).
constexpr uint32_t LAT_MORE = 0x80000000u;

// the variance and the edge test on the fixed 2 x 5 neighbourhood (:1213-1240) for one of the four pixels of a thread, out
// of registers: {thi, tlo} / {bhi, blo} are eight staged bytes of the rows above / below with the pixel's column - 2 at
// byte O.  The six-pixel sums are a v_sad_u8 against zero and a v_dot4_u32_u8 per row; the edge test rides in 16-bit
// halves - (T0, T4) against (T1, T3): one packed max gives (left, right) maxima, one packed min the minima, and a borrow
// bit per half says on which side of them T2 -+ 3 lies.  Returns the pixel's word, LAT_MORE set when it has to search.
template <int O>
__device__ __forceinline__ uint32_t lattice_stage_a4(uint32_t thi, uint32_t tlo, uint32_t bhi, uint32_t blo, int lim, bool inner, uint32_t base)
{
    const uint32_t avg = base & 0xffu;
    constexpr uint32_t SEL3 = (uint32_t)(O + 1) | ((uint32_t)(O + 2) << 8) | ((uint32_t)(O + 3) << 16) | 0x0c000000u;
    const uint32_t mt = __builtin_amdgcn_perm(thi, tlo, SEL3), mb = __builtin_amdgcn_perm(bhi, blo, SEL3);   // columns x-1 .. x+1
    const int sum = (int)__builtin_amdgcn_sad_u8(mt, 0u, __builtin_amdgcn_sad_u8(mb, 0u, 0u));
    const int sumsq = (int)__builtin_amdgcn_udot4(mt, mt, __builtin_amdgcn_udot4(mb, mb, 0u, false), false);
    if (lim < 9 && 6 * sumsq - sum * sum < 576) return base | (avg << 8) | ((uint32_t)PEAK << 16);
    const u16x2 three = pk1(3);
    auto sides = [&](uint32_t hi, uint32_t lo, uint32_t &below, uint32_t &above) {
        const u16x2 outer = pk(PK_BYTES(hi, lo, O, O + 4)), near = pk(PK_BYTES(hi, lo, O + 1, O + 3));
        const u16x2 mid = pk(PK_BYTES(hi, lo, O + 2, O + 2));
        const u16x2 mx = __builtin_elementwise_max(outer, near), mn = __builtin_elementwise_min(outer, near);
        below = un((u16x2)(mid + three - mx));        // bit 15 of a half: mid < max - 3 on that side
        above = un((u16x2)(mn + three - mid));        //                   mid > min + 3
    };
    uint32_t tbelow, tabove, bbelow, babove;
    sides(thi, tlo, tbelow, tabove);
    sides(bhi, blo, bbelow, babove);
    const bool edge = ((tbelow & bbelow & 0x80008000u) == 0x80008000u) || ((tabove & babove & 0x80008000u) == 0x80008000u);
    if (inner && edge) return base | (avg << 8) | ((uint32_t)NEUTRAL << 16);
    return base | LAT_MORE;
}

// (Tried and dropped: the three bytes around a column as ONE unaligned ds_read_b32 - gfx950 returns the right dword at every
// byte offset, tools/lds_unaligned.hip - with v_sad_u8 on the masked dwords: a third of the LDS instructions and of the
// arithmetic of the two searches below, and the candidates kernel went from 272 to 314 us per launch.  An LDS dword that
// straddles two banks is not one access.)
// the search around the pixel's direction (:1242-1290)
__device__ __forceinline__ uint32_t lattice_stage_b(const uint8_t *top, const uint8_t *bot, const uint8_t *ot, const uint8_t *ob,
                                                    const uint8_t *dm, int x, int width, int nt4, int nt8, uint32_t base, const uint8_t *limlut)
{
    const int d = dm[x];
    const int lim = limlut[iabs(d - NEUTRAL) >> 2];
    int dir = (d - NEUTRAL + 2) >> 2;
    int val = (int)(base & 0xffu);
    const int startu = (dir - 2 < 0) ? max(-x + 1, max(dir - 2, -width + 2 + x)) : min(x - 1, min(dir - 2, width - 2 - x));
    const int stopu = (dir + 2 < 0) ? max(-x + 1, max(dir + 2, -width + 2 + x)) : min(x - 1, min(dir + 2, width - 2 - x));
    int mn = nt8;
#define NEAR(row, i) ((row)[i] != PEAK && iabs((int)(row)[i] - d) <= lim)
    for (int u = startu; u <= stopu; u++)
    {
        const int diff = sad3(top, x, bot, x - u) + sad3(bot, x, top, x + u);
        if (!(diff < mn && (NEAR(ot, x - 1 + u) || NEAR(ot, x + u) || NEAR(ot, x + 1 + u)) &&
              (NEAR(ob, x - 1 - u) || NEAR(ob, x - u) || NEAR(ob, x + 1 - u))))
            continue;
        const int h0 = u >> 1, h1 = (u + 1) >> 1;
        const int diff2 = sad3(top, x + h0, bot, x - h0);
        const int o0 = ot[x + h0], o1 = ot[x + h1], q0 = ob[x - h0], q1 = ob[x - h1];
        if (!(diff2 < nt4 && (((iabs(o0 - q0) <= lim || iabs(o0 - q1) <= lim) && o0 != PEAK) ||
                              ((iabs(o1 - q0) <= lim || iabs(o1 - q1) <= lim) && o1 != PEAK))))
            continue;
        if ((iabs(d - o0) <= lim || iabs(d - o1) <= lim) && (iabs(d - q0) <= lim || iabs(d - q1) <= lim))
        {
            val = ((int)top[x + h0] + (int)top[x + h1] + (int)bot[x - h0] + (int)bot[x - h1] + 2) >> 2;
            mn = diff;
            dir = u;
        }
    }
#undef NEAR
    if (mn != nt8) return base | ((uint32_t)val << 8) | ((uint32_t)((NEUTRAL + dir * 4) & 0xff) << 16);
    return base | LAT_MORE;
}

// The same search out of four 8-byte windows.  Its five steps u = dir - 2 .. dir + 2 look at the rows below / above at
// x - u -+ 1 and x + u -+ 1: seven consecutive bytes around x - dir and x + dir of each of the four rows, which three
// aligned ds_read_b32 and two v_alignbyte bring into a register pair (as byte reads the five steps took 60 to 90 of
// them per pixel, and the kernel was bound by the LDS pipe: 154 of its 209 us per 16 fields).  A step's two SADs are a
// v_perm (three window bytes and a zero) and a v_sad_u8 each, the six "near the pixel's direction" tests two bit tests
// on masks made once per window, two bytes per packed operation.  A pixel whose steps the row ends clamp
// (:1244-1247 - the range then lies outside dir -+ 2) takes the plain walk above.
struct LatRows { const uint32_t *top, *bot, *ot, *ob; int org; };      // the staged rows as dwords; org: local index of column 0

// row[c] .. row[c + 7] (any alignment) as two dwords
__device__ __forceinline__ void lat_window8(const uint32_t *row, int li, uint32_t &lo, uint32_t &hi)
{
    const uint32_t *w = row + (li >> 2);
    const uint32_t r0 = w[0], r1 = w[1], r2 = w[2];
    const uint32_t sh = (uint32_t)li & 3u;
    lo = __builtin_amdgcn_alignbyte(r1, r0, sh);
    hi = __builtin_amdgcn_alignbyte(r2, r1, sh);
}
// row[c - 1], row[c], row[c + 1] and a zero
__device__ __forceinline__ uint32_t lat_three(const uint32_t *row, int li)
{
    const uint32_t *w = row + ((li - 1) >> 2);
    return __builtin_amdgcn_alignbyte(w[1], w[0], (uint32_t)(li - 1) & 3u) & 0x00ffffffu;
}
// which of a window's eight bytes are a direction (not the peak) within lim of d: byte 2i at bit 2i, byte 2i + 1 at bit 16 + 2i
__device__ __forceinline__ uint32_t lat_near8(uint32_t lo, uint32_t hi, u16x2 d2, u16x2 lim1)
{
    uint32_t m = 0;
    const u16x2 peak = pk1(PEAK);
#define LAT_NEAR2(i)                                                                                   \
    {                                                                                                  \
        const u16x2 h = pk(PK_BYTES(hi, lo, 2 * i, 2 * i + 1));                                        \
        const i16x2 t = __builtin_bit_cast(i16x2, (u16x2)(h - d2));                                    \
        const u16x2 a = __builtin_bit_cast(u16x2, __builtin_elementwise_max(t, (i16x2)(-t)));          \
        m |= un(pk_lt(a, lim1) & pk_lt(h, peak)) << (2 * i);                                           \
    }
    LAT_NEAR2(0) LAT_NEAR2(1) LAT_NEAR2(2) LAT_NEAR2(3)
#undef LAT_NEAR2
    return m;
}
constexpr uint32_t lat_bit(int b) { return (b & 1) ? 1u << (16 + b - 1) : 1u << b; }
constexpr uint32_t lat_bits3(int g) { return lat_bit(g) | lat_bit(g + 1) | lat_bit(g + 2); }
constexpr uint32_t lat_sel3(int g) { return (uint32_t)g | ((uint32_t)(g + 1) << 8) | ((uint32_t)(g + 2) << 16) | 0x0c000000u; }

template <int K> __device__ __forceinline__ uint32_t lat_byte(uint32_t lo, uint32_t hi)
{
    return K < 4 ? (lo >> (8 * (K & 3))) & 0xffu : (hi >> (8 * (K & 3))) & 0xffu;
}
// row[c] .. row[c + 3]
__device__ __forceinline__ uint32_t lat_window4(const uint32_t *row, int li)
{
    const uint32_t *w = row + (li >> 2);
    return __builtin_amdgcn_alignbyte(w[1], w[0], (uint32_t)li & 3u);
}

// The loop over u (:1249-1290) keeps the step with the smallest diff among those that pass its tests (the first of
// them on a tie: `diff < min`), and no test looks at what an earlier step left behind - so the steps are evaluated
// side by side and the winner is the minimum of the keys (diff << 3) | step.  Counted from the even number at or below
// dir - 2 (six steps, the first or the last of them outside dir -+ 2) the half steps u >> 1 and (u + 1) >> 1 are
// constants too, and the second test's pixels (:1262-1279) come out of four more windows; only the winner's four
// pixels are read per byte.  (As a loop with the second test under a branch: 2.0 of a pixel's 5 steps took it, which
// for a wave meant all five, each with 14 byte reads on a third of its lanes.)  The claims on the CPU:
// tests/test_eedi2_identities_cpu.py::test_lattice_search_steps_side_by_side, ::test_lattice_half_steps_are_constants_of_the_step,
// ::test_within_limit_as_one_unsigned_compare.
__device__ __forceinline__ uint32_t lattice_stage_b_win(const LatRows R, const uint8_t *top, const uint8_t *bot, const uint8_t *ot,
                                                        const uint8_t *ob, const uint8_t *dm, int x, int width, int nt4, int nt8,
                                                        uint32_t base, const uint8_t *limlut)
{
    const int d = dm[x];
    const int lim = limlut[iabs(d - NEUTRAL) >> 2];
    const int dir = (d - NEUTRAL + 2) >> 2;
    const int startu = (dir - 2 < 0) ? max(-x + 1, max(dir - 2, -width + 2 + x)) : min(x - 1, min(dir - 2, width - 2 - x));
    const int stopu = (dir + 2 < 0) ? max(-x + 1, max(dir + 2, -width + 2 + x)) : min(x - 1, min(dir + 2, width - 2 - x));
    if (startu != dir - 2 || stopu != dir + 2)
    {
        CD_STAT(22, 1);
        return lattice_stage_b(top, bot, ot, ob, dm, x, width, nt4, nt8, base, limlut);
    }
    const int de = dir & ~1, par = dir & 1, hb = (de >> 1) - 1;     // step i: u = de - 2 + i, u >> 1 = hb + (i >> 1), (u + 1) >> 1 = hb + ((i + 1) >> 1)
    const int lx = R.org + x;
    uint32_t blo, bhi, tlo, thi, olo, ohi, qlo, qhi, t2lo, t2hi, b2lo, b2hi;
    lat_window8(R.bot, lx - de - 4, blo, bhi);               // x - u - 1 at byte 5 - i
    lat_window8(R.ob, lx - de - 4, qlo, qhi);
    lat_window8(R.top, lx + de - 3, tlo, thi);               // x + u - 1 at byte i
    lat_window8(R.ot, lx + de - 3, olo, ohi);
    lat_window8(R.top, lx + hb - 1, t2lo, t2hi);             // x + (u >> 1) - 1 at byte i >> 1
    lat_window8(R.bot, lx - hb - 4, b2lo, b2hi);             // x - (u >> 1) - 1 at byte 3 - (i >> 1)
    const uint32_t o2 = lat_window4(R.ot, lx + hb);          // x + (u >> 1) at byte i >> 1
    const uint32_t q2 = lat_window4(R.ob, lx - hb - 3);      // x - (u >> 1) at byte 3 - (i >> 1)
    const uint32_t tc = lat_three(R.top, lx), bc = lat_three(R.bot, lx);
    const u16x2 d2 = pk1((uint32_t)d), lim1 = pk1((uint32_t)lim + 1u);
    const uint32_t near_t = lat_near8(olo, ohi, d2, lim1), near_b = lat_near8(qlo, qhi, d2, lim1);
    const uint32_t ulim = (uint32_t)lim, ulim2 = 2u * ulim;
    auto within = [&](uint32_t a, uint32_t b) { return a - b + ulim <= ulim2; };              // |a - b| <= lim
    const uint32_t o[4] = { o2 & 0xffu, (o2 >> 8) & 0xffu, (o2 >> 16) & 0xffu, o2 >> 24 };
    const uint32_t q[4] = { q2 & 0xffu, (q2 >> 8) & 0xffu, (q2 >> 16) & 0xffu, q2 >> 24 };
    const bool od[4] = { within(o[0], (uint32_t)d), within(o[1], (uint32_t)d), within(o[2], (uint32_t)d), within(o[3], (uint32_t)d) };
    const bool qd[4] = { within(q[0], (uint32_t)d), within(q[1], (uint32_t)d), within(q[2], (uint32_t)d), within(q[3], (uint32_t)d) };
    // the three distinct second SADs (they only know u >> 1)
    const int diff2[3] = {
        (int)__builtin_amdgcn_sad_u8(__builtin_amdgcn_perm(t2hi, t2lo, lat_sel3(0)), __builtin_amdgcn_perm(b2hi, b2lo, lat_sel3(3)), 0u),
        (int)__builtin_amdgcn_sad_u8(__builtin_amdgcn_perm(t2hi, t2lo, lat_sel3(1)), __builtin_amdgcn_perm(b2hi, b2lo, lat_sel3(2)), 0u),
        (int)__builtin_amdgcn_sad_u8(__builtin_amdgcn_perm(t2hi, t2lo, lat_sel3(2)), __builtin_amdgcn_perm(b2hi, b2lo, lat_sel3(1)), 0u) };
    uint32_t best = 0x7fffffffu;
#define LAT_STEP(i)                                                                                                   \
    {                                                                                                                 \
        constexpr int A = (i) >> 1, B = ((i) + 1) >> 1;                                                               \
        const uint32_t bg = __builtin_amdgcn_perm(bhi, blo, lat_sel3(5 - (i))), tg = __builtin_amdgcn_perm(thi, tlo, lat_sel3(i)); \
        const int diff = (int)__builtin_amdgcn_sad_u8(tc, bg, __builtin_amdgcn_sad_u8(bc, tg, 0u));                   \
        const bool oq = A == B ? (within(o[A], q[3 - A]) && o[A] != PEAK)                                             \
                               : (((within(o[A], q[3 - A]) || within(o[A], q[3 - B])) && o[A] != PEAK) ||             \
                                  ((within(o[B], q[3 - A]) || within(o[B], q[3 - B])) && o[B] != PEAK));              \
        const bool ok = ((i) == 0 ? par == 0 : (i) == 5 ? par != 0 : true) && diff < nt8 &&                           \
                        (near_t & lat_bits3(i)) && (near_b & lat_bits3(5 - (i))) && diff2[A] < nt4 && oq &&           \
                        (od[A] || od[B]) && (qd[3 - A] || qd[3 - B]);                                                 \
        best = min(best, ok ? ((uint32_t)diff << 3) | (uint32_t)(i) : 0x7fffffffu);                                   \
    }
    LAT_STEP(0) LAT_STEP(1) LAT_STEP(2) LAT_STEP(3) LAT_STEP(4) LAT_STEP(5)
#undef LAT_STEP
    if (best == 0x7fffffffu) return base | LAT_MORE;
    const int u = de - 2 + (int)(best & 7u);
    const int h0 = u >> 1, h1 = (u + 1) >> 1;
    const int val = ((int)top[x + h0] + (int)top[x + h1] + (int)bot[x - h0] + (int)bot[x - h1] + 2) >> 2;
    return base | ((uint32_t)val << 8) | ((uint32_t)((NEUTRAL + u * 4) & 0xff) << 16);
}

// the short search for pixels the first one left without a match (:1292-1318)
__device__ __forceinline__ uint32_t lattice_stage_c(const uint8_t *top, const uint8_t *bot, const uint8_t *dm, int x, int width, int pl,
                                                    int nt7, int nt, uint32_t base)
{
    int dir = ((int)dm[x] - NEUTRAL + 2) >> 2;
    int val = (int)(base & 0xffu);
    const int lo = min((int)top[x], (int)bot[x]), hi = max((int)top[x], (int)bot[x]);
    const int dd = pl == 0 ? 4 : 2;
    const int su = max(-x + 1, -dd), eu = min(width - 2 - x, dd);
    int mn = nt7;
    for (int u = su; u <= eu; u++)
    {
        const int h0 = u >> 1, h1 = (u + 1) >> 1;
        const int p1 = (int)top[x + h0] + (int)top[x + h1];
        const int p2 = (int)bot[x - h0] + (int)bot[x - h1];
        const int diff = sad3(top, x, bot, x - u) + sad3(bot, x, top, x + u) + iabs(p1 - p2);
        if (diff < mn)
        {
            const int valt = (p1 + p2 + 2) >> 2;
            if (valt >= lo && valt <= hi) { val = valt; mn = diff; dir = u; }
        }
    }
    const int newB = (mn == 7 * nt) ? NEUTRAL : ((NEUTRAL + dir * 4) & 0xff);
    return base | ((uint32_t)val << 8) | ((uint32_t)newB << 16);
}

// interpolate_lattice in two launches.
// k_lattice_cand_q: everything about a pixel of the rows being rebuilt that does not depend on its left
// neighbour's NEW direction value, packed into 32 bits:
//   [7:0] valA  = vertical average (outcome A)      [15:8]  valB = outcome-B pixel value
//   [23:16] newB = outcome-B direction value        bit 24 = "always A" (dir == peak)
//   bit 25 = right-hand test |d[x]-d[x+1]| > lim    (newA is NEUTRAL, or PEAK when always A)
// k_lattice_resolve (one wavefront per row): resolves which outcome each pixel takes —
// that depends on the value just written at x-1 (:1194) — with a 64-lane prefix composition
// of 2-state maps, carrying the last written value from chunk to chunk, then writes the row.
// a = dmsk (tmp2p, in/out), b = dst (dst2p, in/out), c = omsk (tmp2p2).
constexpr int LC_HALO = 40;   // |u| <= 34, +-1 for the triples, rounded to dwords


// The searching pixels are queued.  Only pixels that carry a direction (d != peak) go
// through the variance / edge tests and the two searches - on real pictures roughly one in ten, but
// nearly every wave holds some, and pays for all of it.  A workgroup takes 1024 pixels of a row:
// every thread packs the word of its four pixels as if they were "always A" (vertical average, the
// right-hand test) and queues those that are not; the queue then gets one lane per pixel.
constexpr int LQ_W = 1024, LQ_LW = LQ_W + 2 * LC_HALO;

__global__ __launch_bounds__(256) void k_lattice_cand_q(P3 P, uint32_t *__restrict__ cand, int cand_pitch,
                                                        int cand_plane_stride, int nt4, int nt7, int nt8, int nt)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_rows[5][LQ_LW];
    __shared__ __attribute__((aligned(16))) uint32_t s_cand[LQ_W];
    __shared__ uint16_t s_list[3][LQ_W];                          // [1], [2]: the queues of the two searches
    __shared__ int s_count[3];
    __shared__ uint8_t s_lim[LIM_PAD];
    FIELD_PLANE(P);
    const int field = tff;
    cand += (size_t)fld * (P.fstride / sizeof(uint32_t));          // the candidates live in the field's slab too
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    const int x0 = blockIdx.x * LQ_W, t = threadIdx.x;
    const int ri = blockIdx.y;
    const int nrows = (height - (2 - field)) / 2;
    if (x0 >= width || ri >= nrows) return;
    if (maskless) return;                                          // no direction anywhere: k_lattice_resolve averages by itself
    const int y = (2 - field) + 2 * ri;
    if (t < 3) s_count[t] = 0;
    lim_fill(s_lim, t);
    {
        const uint8_t *g[5] = { Q.b + (size_t)(y - 1) * pitch, Q.b + (size_t)(y + 1) * pitch,
                                Q.c + (size_t)(y - 1) * pitch, Q.c + (size_t)(y + 1) * pitch,
                                Q.a + (size_t)y * pitch };
        // only as far right as the row's pixels (+ halo) reach
        const int need4 = (min(LQ_W, hbhip_align_up_dev(width - x0, 4)) + 2 * LC_HALO) / 4;
        // all loads of a thread in flight before its first LDS store (need4 <= 276: two dwords per row and thread) - as a
        // loop of load -> store pairs the ten round trips of a thread followed one another
        static_assert(LQ_LW / 4 <= 2 * 256, "two staged dwords per row and thread");
        uint32_t v[5][2] = {};
        const bool h0 = t < need4, h1 = t + 256 < need4;           // one branch per column of the thread, not one per load
        if (h0)
        {
#pragma unroll
            for (int r = 0; r < 5; r++) v[r][0] = reinterpret_cast<const uint32_t *>(g[r] + x0 - LC_HALO)[t];
        }
        if (h1)
        {
#pragma unroll
            for (int r = 0; r < 5; r++) v[r][1] = reinterpret_cast<const uint32_t *>(g[r] + x0 - LC_HALO)[t + 256];
        }
        if (h0)
        {
#pragma unroll
            for (int r = 0; r < 5; r++) reinterpret_cast<uint32_t *>(s_rows[r])[t] = v[r][0];
        }
        if (h1)
        {
#pragma unroll
            for (int r = 0; r < 5; r++) reinterpret_cast<uint32_t *>(s_rows[r])[t + 256] = v[r][1];
        }
    }
    __syncthreads();
    const uint8_t *top = s_rows[0] + LC_HALO - x0, *bot = s_rows[1] + LC_HALO - x0;
    const uint8_t *ot = s_rows[2] + LC_HALO - x0, *ob = s_rows[3] + LC_HALO - x0;
    const uint8_t *dm = s_rows[4] + LC_HALO - x0;
    {
        // every pixel's word, the variance and the edge test included, four pixels per thread out of the rows' dwords
        const int x = x0 + 4 * t;
        if (x < width)
        {
            const uint32_t *T = reinterpret_cast<const uint32_t *>(s_rows[0]) + LC_HALO / 4 + t;   // T[0]: columns x .. x + 3
            const uint32_t *B = reinterpret_cast<const uint32_t *>(s_rows[1]) + LC_HALO / 4 + t;
            const uint32_t *D = reinterpret_cast<const uint32_t *>(s_rows[4]) + LC_HALO / 4 + t;
            const uint32_t tm = T[-1], t4 = T[0], tp = T[1], bm = B[-1], b4 = B[0], bp = B[1];
            const uint32_t d4 = D[0], dn = D[1] & 0xffu;
            uint32_t w[4], base[4];
            int lim[4];
            uint32_t queue = 0, searching = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const int d = (d4 >> (8 * k)) & 0xff, dr = k < 3 ? (int)((d4 >> (8 * k + 8)) & 0xff) : (int)dn;
                const int avg = (int)(((t4 >> (8 * k)) & 0xff) + ((b4 >> (8 * k)) & 0xff) + 1) >> 1;
                lim[k] = s_lim[iabs(d - NEUTRAL) >> 2];
                const bool right = iabs(d - dr) > lim[k];
                if (d != PEAK && x + k < width) searching |= 1u << k;
                base[k] = (uint32_t)avg | ((uint32_t)right << 25);
                w[k] = base[k] | ((uint32_t)avg << 8) | ((uint32_t)NEUTRAL << 16) | (1u << 24);       // the word of a pixel without a direction
            }
            if (searching)                                         // (whole waves have none: the rows above the saturated part of the mask)
            {
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const bool inner = x + k > 1 && x + k < width - 2;
                    const uint32_t ws = k == 0 ? lattice_stage_a4<2>(t4, tm, b4, bm, lim[k], inner, base[k])
                                      : k == 1 ? lattice_stage_a4<3>(t4, tm, b4, bm, lim[k], inner, base[k])
                                      : k == 2 ? lattice_stage_a4<0>(tp, t4, bp, b4, lim[k], inner, base[k])
                                               : lattice_stage_a4<1>(tp, t4, bp, b4, lim[k], inner, base[k]);
                    if ((searching >> k) & 1u)
                    {
                        w[k] = ws & ~LAT_MORE;
                        if (ws & LAT_MORE) queue |= 1u << k;
                    }
                }
            }
            *reinterpret_cast<uint4 *>(&s_cand[4 * t]) = make_uint4(w[0], w[1], w[2], w[3]);
            if (queue)
            {
                int at = atomicAdd(&s_count[1], __popc(queue));
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if ((queue >> k) & 1u) s_list[1][at++] = (uint16_t)(4 * t + k);
            }
        }
    }
    __syncthreads();
    // a stage's lanes all run the same code: the 5-step search, then the short search
    {
        const LatRows R = { reinterpret_cast<const uint32_t *>(s_rows[0]), reinterpret_cast<const uint32_t *>(s_rows[1]),
                            reinterpret_cast<const uint32_t *>(s_rows[2]), reinterpret_cast<const uint32_t *>(s_rows[3]), LC_HALO - x0 };
        for (int i = t, n = s_count[1]; i < n; i += 256)
        {
            const int lx = s_list[1][i];
            const uint32_t w = lattice_stage_b_win(R, top, bot, ot, ob, dm, x0 + lx, width, nt4, nt8, s_cand[lx], s_lim);
            if (w & LAT_MORE) s_list[2][atomicAdd(&s_count[2], 1)] = (uint16_t)lx;
            else s_cand[lx] = w;
        }
    }
    __syncthreads();
#ifdef HBHIP_DEV_STATS
    if (t == 0) { CD_STAT(19, s_count[1]); CD_STAT(20, s_count[2]); CD_STAT(21, 1); }
#endif
    for (int i = t, n = s_count[2]; i < n; i += 256)
    {
        const int lx = s_list[2][i];
        s_cand[lx] = lattice_stage_c(top, bot, dm, x0 + lx, width, pl, nt7, nt, s_cand[lx]);
    }
    __syncthreads();
    {
        const int x = x0 + 4 * t;
        uint32_t *o = cand + (size_t)pl * cand_plane_stride + (size_t)ri * cand_pitch + x;
        if (x + 3 < width && (cand_pitch & 3) == 0) *reinterpret_cast<uint4 *>(o) = *reinterpret_cast<const uint4 *>(&s_cand[4 * t]);
        else for (int k = 0; k < 4 && x + k < width; k++) o[k] = s_cand[4 * t + k];
    }
}

// grid.y = processed rows (+1 for the border-row copy); one workgroup of LR_T threads per row, FOUR pixels per thread
// and pass over the row (LR_PX pixels).  Which outcome a pixel takes depends only on which outcome its left neighbour
// took (that decides the value left standing at x-1, :1194), so every pixel is a 2-state map (bit s = its outcome when
// the left pixel took outcome s).  A thread composes the maps of its four pixels, the threads' maps are composed by a
// prefix scan inside each wave, the wave maps are chained by one thread, and the state entering the next LR_PX pixels
// is the outcome of the last one.  Candidates come in as one 16-byte load, the direction row as a dword, both rows
// leave as dwords (one pixel per thread, byte loads and stores: 144 us per 16 fields).
constexpr int LR_T = 256, LR_PX = 4 * LR_T;

// later o earlier: the map that applies `earlier` first (the thread / wave / pass composition against the pixel-by-pixel
// walk: tests/test_eedi2_identities_cpu.py::test_lattice_resolve_scan_equals_the_serial_walk)
__device__ __forceinline__ unsigned lr_compose(unsigned later, unsigned earlier)
{
    return ((later >> (earlier & 1u)) & 1u) | (((later >> ((earlier >> 1) & 1u)) & 1u) << 1);
}

__global__ __launch_bounds__(LR_T) void k_lattice_resolve(P3 P, const uint32_t *__restrict__ cand, int cand_pitch,
                                                          int cand_plane_stride)
{
    __shared__ uint8_t s_wmap[LR_T / 64];        // composed map of each wave
    __shared__ uint8_t s_win[LR_T / 64];         // resolved state entering each wave
    __shared__ uint8_t s_lim[36];                // eedi2_limlut in LDS: four look-ups per thread and pass
    __shared__ int s_carry;                      // outcome of the last pixel of the previous pass
    FIELD_PLANE(P);
    const int field = tff;
    cand += (size_t)fld * (P.fstride / sizeof(uint32_t));
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nrows = (height - (2 - field)) / 2;                // rows y0, y0+2, ... < height-1
    uint8_t *dst = Q.b;
    if ((int)blockIdx.y >= nrows)
    {
        if ((int)blockIdx.y == nrows)                              // the one-row blit (:1162-1179)
            for (int xx = t; xx < width; xx += LR_T)
            {
                if (field == 1) dst[(size_t)(height - 1) * pitch + xx] = dst[(size_t)(height - 2) * pitch + xx];
                else            dst[xx] = dst[pitch + xx];
            }
        return;
    }
    const int y = (2 - field) + 2 * blockIdx.y;
    uint8_t *mid = dst + (size_t)y * pitch;
    uint8_t *dm = Q.a + (size_t)y * pitch;
    if (maskless)
    {
        // no mask pixel in the plane: every direction is a peak, every pixel of the row the rounded mean of the pixels
        // above and below it (:1192-1197), the direction row stays as it is.  Four pixels per operation:
        // (a + b + 1) >> 1 = (a | b) - (((a ^ b) >> 1) & 0x7f) per byte
        const uint8_t *top = mid - pitch, *bot = mid + pitch;
        for (int x = 4 * t; x < width; x += 4 * LR_T)
        {
            const uint32_t a = *reinterpret_cast<const uint32_t *>(top + x), b = *reinterpret_cast<const uint32_t *>(bot + x);
            const uint32_t v = (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu);
            if (x + 3 < width) *reinterpret_cast<uint32_t *>(mid + x) = v;
            else for (int k = 0; k < 4 && x + k < width; k++) mid[x + k] = (uint8_t)(v >> (8 * k));
        }
        return;
    }
    const uint32_t *cr = cand + (size_t)pl * cand_plane_stride + (size_t)blockIdx.y * cand_pitch;
    const bool cr16 = ((reinterpret_cast<uintptr_t>(cr)) & 15u) == 0;      // block-uniform: the row of candidates starts on 16 bytes
    // value standing at dm[x-1] for x == 0: memory just before the row, never written by this pass
    const int before_row = dm[-1];
    if (t == 0) s_carry = 0;
    if (t < 33) s_lim[t] = c_limlut[t];
    __syncthreads();

    for (int x0 = 0; x0 < width; x0 += LR_PX)
    {
        const int x = x0 + 4 * t;
        const int nlive = min(max(width - x, 0), 4);             // pixels of this thread inside the row
        uint32_t c[4] = { 0u, 0u, 0u, 0u }, d4 = 0u;
        if (nlive == 4 && cr16)
        {
            const uint4 v = *reinterpret_cast<const uint4 *>(cr + x);
            c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
        }
        else
            for (int k = 0; k < nlive; k++) c[k] = cr[x + k];
        if (nlive) d4 = *reinterpret_cast<const uint32_t *>(dm + x);     // the row's pitch is a multiple of 4: the dword is inside it
        // what the last pixel of the thread to the left can leave behind (pa: outcome A, pb: outcome B); the first lane
        // of a wave reads that pixel's candidate word; nothing has been written to this row yet
        const uint32_t cl3 = c[3];
        int pa = __shfl_up((cl3 >> 24) & 1u ? PEAK : NEUTRAL, 1, 64), pb = __shfl_up((int)((cl3 >> 16) & 0xffu), 1, 64);
        if (lane == 0 && nlive)
        {
            if (x == 0) { pa = before_row; pb = before_row; }
            else
            {
                const uint32_t cl = cr[x - 1];
                pa = ((cl >> 24) & 1u) ? PEAK : NEUTRAL;
                pb = (int)((cl >> 16) & 0xffu);
            }
        }
        unsigned m[4], pm[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const uint32_t ck = c[k];
            const int d = (int)((d4 >> (8 * k)) & 0xffu);
            const int lim = s_lim[iabs(d - NEUTRAL) >> 2];
            const bool always_a = (ck >> 24) & 1u, right = (ck >> 25) & 1u;
            if (k >= nlive || always_a) m[k] = 0u;
            else
            {
                const unsigned oa = (right && iabs(d - pa) > lim) ? 0u : 1u;
                const unsigned ob2 = (right && iabs(d - pb) > lim) ? 0u : 1u;
                m[k] = oa | (ob2 << 1);
            }
            pm[k] = k == 0 ? m[0] : lr_compose(m[k], pm[k - 1]);    // the thread's pixels 0..k, earlier first
            pa = always_a ? PEAK : NEUTRAL;                          // what pixel k leaves for k + 1
            pb = (int)((ck >> 16) & 0xffu);
        }
        // inclusive prefix composition of the threads' maps inside the wave (earlier map first)
        unsigned tm = pm[3];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1)
        {
            const unsigned e = __shfl_up(tm, off, 64);
            if (lane >= off) tm = lr_compose(tm, e);
        }
        if (lane == 63) s_wmap[wave] = (uint8_t)tm;
        unsigned before = __shfl_up(tm, 1, 64);                      // everything left of this thread inside the wave
        if (lane == 0) before = 2u;                                  // the identity map
        __syncthreads();
        if (t == 0)
        {
            unsigned state = (unsigned)s_carry;                      // outcome of pixel x0 - 1 (irrelevant for x0 == 0)
            for (int w = 0; w < LR_T / 64; w++)
            {
                s_win[w] = (uint8_t)state;
                state = (s_wmap[w] >> state) & 1u;
            }
        }
        __syncthreads();
        const unsigned sin = (before >> s_win[wave]) & 1u;           // outcome of the pixel left of this thread's first
        if (nlive)
        {
            uint32_t mid4 = 0u, dm4 = 0u;
            unsigned last = 0u;
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const unsigned outcome = (pm[k] >> sin) & 1u;
                const uint32_t ck = c[k];
                const uint32_t val = outcome ? (ck >> 8) & 0xffu : ck & 0xffu;
                const uint32_t nd = outcome ? (ck >> 16) & 0xffu : (((ck >> 24) & 1u) ? (uint32_t)PEAK : (uint32_t)NEUTRAL);
                mid4 |= val << (8 * k);
                dm4 |= nd << (8 * k);
                if (k == nlive - 1) last = outcome;
            }
            if (nlive == 4)
            {
                *reinterpret_cast<uint32_t *>(mid + x) = mid4;
                *reinterpret_cast<uint32_t *>(dm + x) = dm4;
            }
            else
                for (int k = 0; k < nlive; k++) { mid[x + k] = (uint8_t)(mid4 >> (8 * k)); dm[x + k] = (uint8_t)(dm4 >> (8 * k)); }
            if (x + nlive == min(x0 + LR_PX, width)) s_carry = (int)last;
        }
        __syncthreads();
    }
}

// a = nmsk, b = omsk, c = dst (in place, row y from rows y+-1), 4 pixels per thread
__global__ void k_post(P3 P)
{
    XY4_PLANE(P);
    const int y0 = 2 - tff;
    if (x >= width || y >= height - 1 || y < y0 || ((y - y0) & 1)) return;
    const size_t at = (size_t)y * pitch + x;
    const uint32_t nm4 = *reinterpret_cast<const uint32_t *>(Q.a + at), om4 = *reinterpret_cast<const uint32_t *>(Q.b + at);
    uint8_t *d = Q.c + at;
    const uint32_t up4 = *reinterpret_cast<const uint32_t *>(d - pitch), dn4 = *reinterpret_cast<const uint32_t *>(d + pitch);
    const uint32_t cur4 = *reinterpret_cast<const uint32_t *>(d);
    int out[4];
    bool any = false;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        const int nm = (nm4 >> (8 * k)) & 0xffu, om = (om4 >> (8 * k)) & 0xffu;
        const int lim = c_limlut[iabs(nm - NEUTRAL) >> 2];
        const bool fix = iabs(nm - om) > lim && om != PEAK && om != NEUTRAL;
        out[k] = fix ? (int)((((up4 >> (8 * k)) & 0xffu) + ((dn4 >> (8 * k)) & 0xffu) + 1) >> 1) : (int)((cur4 >> (8 * k)) & 0xffu);
        any |= fix;
    }
    if (any) st4(d, out, x, width);
}

// ------------------------------------------------------------------------------------------
// Post-processing 2/3: junctions and corners (eedi2_template.c:1391-1904; decomb_template.c:432-441).
// The reference's three plane threads share ONE set of derivative arrays (decomb.c:398-403), so
// its own result is a data race; what is reproduced here is the defined order "Y, Cb, Cr one
// after the other" on the same flat arrays (oracle/ref_wrap/wrap_decomb.c:hbref_eedi2_run_serial).
// The flat layout matters: the horizontal pass of gaussian_blur_sqrt2 reads src[x+3] instead of
// src[x-3] at x == width-2 (:1589) — the next row, the row padding, or whatever another plane
// left there — so the planes run one after the other and index the arrays exactly as it does.
// Both blurs are symmetric FIRs whose out-of-range taps are mirrored about the centre (written
// in the reference as doubled coefficients on the surviving side).
struct CornerArgs
{
    uint8_t *src, *tmp;          // srcp (blurred in place) and tmpp of one plane
    int     *c[3];               // cx2, cy2, cxy (shared by the planes)
    int     *t[3];               // tmpc, one per array (the reference reuses one; nothing of it outlives a blur)
    int      pitch, width, height;   // half-height geometry of the plane
};

__device__ __forceinline__ int fold_tap(int centre, int d, int n, int &hi)
{
    int lo = centre - d;
    hi = centre + d;
    if (lo < 0) lo = hi;
    if (hi >= n) hi = lo;
    return lo;
}

// eedi2_gaussian_blur1 (:1391-1527), one axis per launch: VERT = false src -> tmp, true tmp -> src
template <bool VERT>
__global__ void k_blur1(CornerArgs A)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= A.width || y >= A.height) return;
    const uint8_t *in = VERT ? A.tmp : A.src;
    uint8_t *out = VERT ? A.src : A.tmp;
    const int W[4] = { 26152, 15862, 3539, 291 };
    int acc = in[(size_t)y * A.pitch + x] * W[0] + 32768;
#pragma unroll
    for (int d = 1; d <= 3; d++)
    {
        int hi;
        const int lo = fold_tap(VERT ? y : x, d, VERT ? A.height : A.width, hi);
        const size_t il = VERT ? (size_t)lo * A.pitch + x : (size_t)y * A.pitch + lo;
        const size_t ih = VERT ? (size_t)hi * A.pitch + x : (size_t)y * A.pitch + hi;
        acc += ((int)in[il] + (int)in[ih]) * W[d];
    }
    out[(size_t)y * A.pitch + x] = (uint8_t)(acc >> 16);
}

// eedi2_calc_derivatives (:1760-1848): differences against clamped neighbours
__global__ void k_derivatives(CornerArgs A)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= A.width || y >= A.height) return;
    const uint8_t *s = A.src + (size_t)y * A.pitch;
    const uint8_t *up = A.src + (size_t)max(y - 1, 0) * A.pitch, *dn = A.src + (size_t)min(y + 1, A.height - 1) * A.pitch;
    const int ix = (int)s[min(x + 1, A.width - 1)] - (int)s[max(x - 1, 0)];
    const int iy = (int)up[x] - (int)dn[x];
    const size_t at = (size_t)y * A.pitch + x;
    A.c[0][at] = (ix * ix) >> 1;
    A.c[1][at] = (iy * iy) >> 1;
    A.c[2][at] = (ix * iy) >> 1;
}

// eedi2_gaussian_blur_sqrt2 (:1539-1748), one axis per launch, the three arrays in blockIdx.z:
// VERT = false c -> t (>> 16, with the x+3 read of :1589), true t -> c (>> 18)
template <bool VERT>
__global__ void k_blur_sqrt2(CornerArgs A)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= A.width || y >= A.height) return;
    const int *in = VERT ? A.t[blockIdx.z] : A.c[blockIdx.z];
    int *out = VERT ? A.c[blockIdx.z] : A.t[blockIdx.z];
    const int W[5] = { 18508, 14415, 6809, 1951, 339 };
    int acc = in[(size_t)y * A.pitch + x] * W[0] + 32768;
#pragma unroll
    for (int d = 1; d <= 4; d++)
    {
        int hi;
        int lo = fold_tap(VERT ? y : x, d, VERT ? A.height : A.width, hi);
        if (!VERT && d == 3 && x == A.width - 2) lo = hi = x + 3;
        const size_t il = VERT ? (size_t)lo * A.pitch + x : (size_t)y * A.pitch + lo;
        const size_t ih = VERT ? (size_t)hi * A.pitch + x : (size_t)y * A.pitch + hi;
        acc += (in[il] + in[ih]) * W[d];
    }
    out[(size_t)y * A.pitch + x] = acc >> (VERT ? 18 : 16);
}

// eedi2_post_process_corner (:1864-1904): msk = tmp2p2, dst = dst2p (row y from rows y+-1, which
// belong to the kept field and are never written here).  The response is evaluated in double, in
// the reference's operation order (int products, 0.09 * s * s, one subtraction, truncation).
__global__ void k_post_corner(CornerArgs A, const uint8_t *msk, uint8_t *dst, int field, int height)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y * blockDim.y + threadIdx.y;
    const int y = 8 - field + 2 * r;
    if (x < 4 || x >= A.width - 4 || y >= height - 7) return;
    const size_t at = (size_t)y * A.pitch + x;
    const int m = msk[at];
    if (m == PEAK || m == NEUTRAL) return;
    bool hit = false;
#pragma unroll
    for (int k = 0; k < 2; k++)
    {
        const size_t i = (size_t)(3 + r + k) * A.pitch + x;
        const int a = A.c[0][i], b = A.c[1][i], c = A.c[2][i];
        const double s = (double)(a + b);
        const double resp = (double)(a * b - c * c) - 0.09 * s * s;
        hit |= (int)resp > 775;
    }
    if (hit) dst[at] = (uint8_t)(((int)dst[at - A.pitch] + (int)dst[at + A.pitch] + 1) >> 1);
}

} // namespace

// ---- MaskChainGuard (eedi2_engine.h): what happens when a wait of the mask chain runs out ------------------------
static std::atomic<int>      g_chain_spin_limit{1 << 20};     // about a second of polling
static std::atomic<uint32_t> g_chain_fallbacks{0};
int  eedi_chain_spin_limit() { return g_chain_spin_limit.load(); }
void eedi_chain_note_fallbacks(uint32_t n) { g_chain_fallbacks += n; }

// Test hook (ABI): spin_limit > 0 sets the polls a chain wait makes before it gives up for every mask launch from now
// on (1 = give up at once: every launch then takes the repair path), 0 restores the default; returns the number of mask
// launches the repair pass has redone in this process so far (as far as the engines' launch() calls have seen).
extern "C" unsigned hbhip_debug_mask_chain(int spin_limit)
{
    if (spin_limit > 0) g_chain_spin_limit = spin_limit;
    else if (spin_limit == 0) g_chain_spin_limit = 1 << 20;
    return g_chain_fallbacks.load();
}

int MaskChainGuard::init(hbhip_ctx *ctx)
{
    HBHIP_CHECK(ctx, hipMalloc((void **)&err, sizeof(uint32_t)));
    HBHIP_CHECK(ctx, hipMemsetAsync(err, 0, sizeof(uint32_t), ctx->stream));
    HBHIP_CHECK(ctx, hipHostMalloc((void **)&count_host, sizeof(uint32_t), hipHostMallocMapped));
    *count_host = 0;
    HBHIP_CHECK(ctx, hipHostGetDevicePointer((void **)&count_dev, count_host, 0));
    return HBHIP_OK;
}

void MaskChainGuard::destroy()
{
    poll("eedi2");
    if (err) (void)hipFree(err);
    if (count_host) (void)hipHostFree(count_host);
    err = nullptr; count_host = nullptr; count_dev = nullptr;
}

void MaskChainGuard::poll(const char *who)
{
    if (!count_host) return;
    const uint32_t now = __atomic_load_n(count_host, __ATOMIC_RELAXED);
    if (now == seen) return;
    if (seen == 0)
        fprintf(stderr, "hbhip: %s: a wait of the mask chain ran out (workgroups were not dispatched in order); the batch's "
                        "masks were recomputed field by field - results are unaffected, that batch was slower\n", who);
    eedi_chain_note_fallbacks(now - seen);
    seen = now;
}

void MaskChainGuard::bind(MaskChain &C) const
{
    C.err = err;
    C.fallbacks = count_dev;
    C.spin_limit = eedi_chain_spin_limit();
}

// ------------------------------------------------------------------- engine
// Fields are queued (add_field) and run together (launch): the edge mask is the only thing one field's run takes from
// the previous one (the lower half of MSKPF keeps the previous run's mask, eedi2_template.c:132), so the mask kernels
// form a chain and every pass behind them takes all queued fields in one launch (blockIdx.z = 3 * field + plane).
// Each field has a slot: its nine scratch frames and its lattice candidates in one slab, slot_bytes_ apart.
// ---- what both engines share (EediEngineBase, eedi2_engine.h)
EediEngineBase::EediEngineBase(hbhip_ctx *ctx, const PicGeometry &geo, const Eedi2Params &p, int capacity, const char *who)
    : ctx_(ctx), geo_(geo), par_(p), who_(who)
{
    cap_ = std::min(std::max(capacity, 1), EEDI_MAX_BATCH);
}

EediEngineBase::~EediEngineBase()
{
    if (slab_) (void)hipFree(slab_);
    if (chain_flags_) (void)hipFree(chain_flags_);
    if (chain_has_) (void)hipFree(chain_has_);
    if (plane_flags_) (void)hipFree(plane_flags_);
    guard_.destroy();
    for (int g = 0; g < MAX_SIDE; g++)
    {
        if (side_[g]) (void)hipStreamDestroy(side_[g]);
        if (ev_join_[g]) (void)hipEventDestroy(ev_join_[g]);
    }
    if (ahead_) (void)hipStreamDestroy(ahead_);
    if (ev_fork_) (void)hipEventDestroy(ev_fork_);
    if (ev_mask_) (void)hipEventDestroy(ev_mask_);
    for (int i = 0; i < 3; i++)
    {
        if (deriv_[i]) (void)hipFree(deriv_[i]);
        if (deriv_tmp_[i]) (void)hipFree(deriv_tmp_[i]);
    }
}

// lays a frame out at `at` bytes into a slot (planes as hb_frame_buffer_init places them, strides in bytes); returns the end
size_t EediEngineBase::place_frame(EediFrame &f, int width, int height, size_t at) const
{
    size_t total = 0;
    for (int c = 0; c < 3; c++)
    {
        f.width[c] = c ? -((-width) >> geo_.log2_cw) : width;
        f.height[c] = c ? -((-height) >> geo_.log2_ch) : height;
        f.stride[c] = hbhip_align_up(f.width[c] * geo_.bps, 64);  // hb_image_stride
        f.plane[c] = reinterpret_cast<uint8_t *>(at + total);  // offset for now, init_slots() adds the slab's address
        total += (size_t)f.stride[c] * f.height[c];
    }
    f.bytes = total;
    return at + total;
}

int EediEngineBase::init_slots(const EediLayout &L)
{
    // the reference overruns its scratch planes when a chroma plane has an odd number
    // of rows (upscale_by_2 writes 2*ceil(h/2) rows); refuse instead of guessing
    if (geo_.height % (2 << geo_.log2_ch) != 0 || geo_.height < 16 || geo_.width < 16)
        return HBHIP_ERR_UNSUPPORTED;
    if (par_.post_processing < 0 || par_.post_processing > 3) return HBHIP_ERR_UNSUPPORTED;
    size_t at = L.guard;
    for (auto &f : half_) at = place_frame(f, geo_.width, geo_.height / 2, at) + L.guard;     // decomb.c:291-296
    for (auto &f : full_) at = place_frame(f, geo_.width, geo_.height, at) + L.guard;         // decomb.c:299-303
    // interpolate_lattice: per-pixel candidate outcomes of the rebuilt rows (every other row of the full-height frame)
    cand_pitch_ = full_[0].stride[0] / geo_.bps;
    cand_plane_stride_ = cand_pitch_ * ((full_[0].height[0] + 1) / 2);
    auto up256 = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t cand_at = up256(at);
    slot_bytes_ = up256(cand_at + L.cand_elem * (size_t)cand_plane_stride_ * 3);
    // one slot more than a batch holds, so that a batch never writes the slot whose mask its first field reads
    const size_t total = slot_bytes_ * (size_t)(cap_ + 1);
    HBHIP_CHECK(ctx_, hipMalloc((void **)&slab_, total));
    HBHIP_CHECK(ctx_, hipMemsetAsync(slab_, 0, total, ctx_->stream));
    for (auto &f : half_) for (int c = 0; c < 3; c++) f.plane[c] = slab_ + (size_t)(uintptr_t)f.plane[c];
    for (auto &f : full_) for (int c = 0; c < 3; c++) f.plane[c] = slab_ + (size_t)(uintptr_t)f.plane[c];
    cand_raw_ = slab_ + cand_at;
    if (cap_ >= 8)
    {
        HBHIP_CHECK(ctx_, hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
        HBHIP_CHECK(ctx_, hipEventCreateWithFlags(&ev_mask_, hipEventDisableTiming));
        // the side streams of a part's later groups of fields: one (two groups; three with two measured + 1.5 % in round 4
        // and nothing now - a stream more than the hardware has queues for once the stages behind decomb have theirs)
        nside_ = std::min(std::max(hbhip_dev_int("HBHIP_EEDI2_GROUPS", 2) - 1, 1), MAX_SIDE);
        if (mask_ahead()) HBHIP_CHECK(ctx_, hipStreamCreateWithFlags(&ahead_, hipStreamNonBlocking));
        for (int g = 0; g < nside_; g++)
        {
            HBHIP_CHECK(ctx_, hipStreamCreateWithFlags(&side_[g], hipStreamNonBlocking));
            HBHIP_CHECK(ctx_, hipEventCreateWithFlags(&ev_join_[g], hipEventDisableTiming));
        }
    }
    HBHIP_CHECK(ctx_, hipMalloc((void **)&plane_flags_, sizeof(uint32_t) * 3 * EEDI_MAX_BATCH));
    HBHIP_CHECK(ctx_, hipMemsetAsync(plane_flags_, 0, sizeof(uint32_t) * 3 * EEDI_MAX_BATCH, ctx_->stream));
    last_slot_ = cap_;                                             // "the previous mask" of the first run: zeros, like the reference's
    {
        const MaskChain C = eedi_mask_chain_tiles(half_[0], L.tile_w, L.tile_h, L.tile_oy);
        chain_ntiles_ = C.ntiles;
        chain_group_ = C.ntiles + C.nupper;
    }
    if (cap_ > 1)
    {
        // one completion flag per lower mask tile and field of a batch (MaskChain); 0 is no launch's number
        const size_t nflags = (size_t)chain_ntiles_ * cap_;
        HBHIP_CHECK(ctx_, hipMalloc((void **)&chain_flags_, sizeof(uint32_t) * nflags));
        HBHIP_CHECK(ctx_, hipMemsetAsync(chain_flags_, 0, sizeof(uint32_t) * nflags, ctx_->stream));
        // and a word per tile (upper ones too) and field: "left a mask sample set" (MaskChain::has)
        HBHIP_CHECK(ctx_, hipMalloc((void **)&chain_has_, sizeof(uint32_t) * (size_t)chain_group_ * cap_));
        HBHIP_CHECK(ctx_, hipMemsetAsync(chain_has_, 0, sizeof(uint32_t) * (size_t)chain_group_ * cap_, ctx_->stream));
        const int grc = guard_.init(ctx_);
        if (grc != HBHIP_OK) return grc;
    }
    if (par_.post_processing > 1)
    {
        // cx2, cy2, cxy: height * stride(luma, bytes) ints each, shared by the planes (decomb.c:398-403);
        // zeroed once — the reference mallocs them, and one element per row is read before anything
        // wrote it (eedi2_template.c:1589)
        const size_t n = sizeof(int) * (size_t)geo_.height * full_[0].stride[0];
        for (int i = 0; i < 3; i++)
        {
            HBHIP_CHECK(ctx_, hipMalloc((void **)&deriv_[i], n));
            HBHIP_CHECK(ctx_, hipMemsetAsync(deriv_[i], 0, n, ctx_->stream));
            HBHIP_CHECK(ctx_, hipMalloc((void **)&deriv_tmp_[i], n));
            HBHIP_CHECK(ctx_, hipMemsetAsync(deriv_tmp_[i], 0, n, ctx_->stream));
        }
    }
    return HBHIP_OK;
}

EediFrame EediEngineBase::at_slot(const EediFrame &f, int slot) const
{
    EediFrame r = f;
    for (int c = 0; c < 3; c++) r.plane[c] = f.plane[c] + (size_t)slot * slot_bytes_;
    return r;
}

// eedi2_planer (decomb_template.c:455-473) for one more field of `cur`; the run itself happens in launch()
int EediEngineBase::add_field(const DevPicture *cur, int tff)
{
    if (n_ >= cap_) return -1;
    for (int c = 0; c < 3; c++)
        if ((cur->pitch[c] & 3) != 0 || ((uintptr_t)cur->plane[c] & 3) != 0) return -1;   // device pictures are 256-byte aligned
    if (n_ == 0) { start_ = last_slot_ == 0 ? 1 : 0; tffbits_ = 0; }
    for (int c = 0; c < 3; c++) { src_frame_[n_][c] = cur->plane[c]; src_pitch_[c] = cur->pitch[c]; }
    if (tff) tffbits_ |= 1u << n_;
    return start_ + n_++;
}

// the number of the next mask launch (chain flags, plane flags)
int EediEngineBase::next_epoch(hbhip_ctx *lc, uint32_t *epoch)
{
    if (chain_epoch_ == 0xffffffffu)
    {
        // 2^32 mask launches later: 0 means "no launch" in the flag arrays and old numbers must not come round again -
        // drain the device, clear the flags and start over at 1 (months of continuous running apart)
        HBHIP_CHECK(lc, hipDeviceSynchronize());
        if (chain_flags_) HBHIP_CHECK(lc, hipMemset(chain_flags_, 0, sizeof(uint32_t) * (size_t)chain_ntiles_ * cap_));
        if (chain_has_) HBHIP_CHECK(lc, hipMemset(chain_has_, 0, sizeof(uint32_t) * (size_t)chain_group_ * cap_));
        HBHIP_CHECK(lc, hipMemset(plane_flags_, 0, sizeof(uint32_t) * 3 * EEDI_MAX_BATCH));
        chain_epoch_ = 0;
    }
    *epoch = ++chain_epoch_;
    return HBHIP_OK;
}

// ---- the 8-bit engine
Eedi2Engine::Eedi2Engine(hbhip_ctx *ctx, const PicGeometry &geo, const Eedi2Params &p, int capacity)
    : EediEngineBase(ctx, geo, p, capacity, "decomb EEDI2")
{
    static_assert(EEDI_MAX_FIELDS == EEDI_MAX_BATCH, "a batch is one launch's fields");
}

Eedi2Engine::~Eedi2Engine()
{
    if (work_list_) (void)hipFree(work_list_);
    if (work_count_) (void)hipFree(work_count_);
}

int Eedi2Engine::init()
{
    if (geo_.bps != 1) return HBHIP_ERR_UNSUPPORTED;
    if (geo_.width >= (1 << 14) || geo_.height >= (1 << 14)) return HBHIP_ERR_UNSUPPORTED;
    const int rc = init_slots({ GUARD, sizeof(uint32_t), MF_W, MF_H, MF_OY });
    if (rc != HBHIP_OK) return rc;
    if (par_.maximum_search_distance > CD_HALO - 2)
    {
        // work list of the calc_directions fallback (every half-height pixel could qualify)
        size_t half_px = 0;
        for (int c = 0; c < 3; c++) half_px += (size_t)half_[0].stride[c] * half_[0].height[c];
        HBHIP_CHECK(ctx_, hipMalloc((void **)&work_list_, sizeof(uint32_t) * half_px));
        HBHIP_CHECK(ctx_, hipMalloc((void **)&work_count_, sizeof(int)));
    }
    HBHIP_CHECK(ctx_, hipStreamSynchronize(ctx_->stream));
    return HBHIP_OK;
}

// the long-search fallback keeps one work list: everything on the caller's stream then
bool Eedi2Engine::may_fork() const { return par_.maximum_search_distance <= CD_HALO - 2; }

// (Round 3 tried the mask chain on a high-priority stream of its own with the passes following it in groups of 8 fields:
// + 2 % on one stream per GPU, and two independent streams per GPU lost half their rate - more streams than hardware
// queues.  What runs since round 4 is the other way round: the mask launch on the caller's stream, the passes of the
// batch's second half beside those of its first, see below.)
int EediEngineBase::launch(hbhip_ctx *lc)
{
    if (n_ == 0) return HBHIP_OK;
    const int n = n_;
    n_ = 0;
    guard_.poll(who_);
    // A batch goes out in parts of at most EEDI_PART fields: one mask launch per part (a chain of that many links), and the
    // passes behind it.
    //  * The passes: every field has its own slot, so the two halves of a part run them beside each other on two streams -
    //    the same kernels half a launch apart fill each other's tails and latency-bound stretches (decomb bob 10 750 ->
    //    11 560, the chain 7 390 -> 7 640 output fps; three quarters / one quarter: 11 100 / 7 540; three groups on three
    //    streams + 1.5 %, four - 9 %).
    //  * The mask launch of the second part goes out on the caller's stream right behind the first part's, in front of the
    //    first part's passes there: it runs beside the other half's passes on the side stream.  (Until round 6 it had a
    //    stream of its own.  HIP spreads its streams over four hardware queues, and a stream that shares its queue with
    //    another waits behind whatever that one waits for: with the mask on the caller's stream a chain keeps three streams
    //    busy - the caller's, the side stream, the stages behind decomb - and measures the same however the process's
    //    other streams were made; with a fourth and fifth it read 8 230 - 8 880 fps by the order of their creation.
    //    8 810 -> 8 916 fps on the chain, 13 390 -> 13 435 on decomb bob: profiles/r6Z_streams_and_queues.log.  The 16-bit
    //    engine keeps the stream of its own: its mask launch is the longer one, and in front of the passes it costs the
    //    10-bit chain 1.5 %, 10-bit decomb bob 3 % - mask_ahead().)
    // Not with post-processing 2 / 3 (its derivative arrays carry values from field to field), not while the profiler
    // brackets launches (its events live on the context's stream) or HBHIP_EEDI2_FORK=0 says so (counter runs), not for
    // the long-search fallback (one work list): then everything goes out on the caller's stream, part after part.
    const bool fork = side_[0] && eedi_fork_enabled() && par_.post_processing < 2 && !lc->profile && may_fork();
    const int parts = (n + EEDI_PART - 1) / EEDI_PART;
    uint32_t epoch[(EEDI_MAX_BATCH + EEDI_PART - 1) / EEDI_PART] = {};
    int rc = enqueue_mask(0, std::min(n, EEDI_PART), lc, lc->stream, &epoch[0]);
    if (!fork)
    {
        for (int p = 0; p < parts && rc == HBHIP_OK; p++)
        {
            const int f0 = p * EEDI_PART, m = std::min(EEDI_PART, n - f0);
            if (p) rc = enqueue_mask(f0, m, lc, lc->stream, &epoch[p]);
            if (rc == HBHIP_OK) rc = enqueue_passes(f0, m, lc, lc->stream, epoch[p]);
        }
    }
    else if (rc == HBHIP_OK)
    {
        // two events order the side streams behind the masks: ev_fork_ = the first part's is out, ev_mask_ = the next part's
        static_assert(EEDI_MAX_BATCH <= 2 * EEDI_PART, "the fork below orders at most two parts with its two mask events");
        bool used[MAX_SIDE] = {}, ahead_used = false;
        // a HIP call that fails in here must not leave the side streams running into slots the caller goes on to reuse:
        // remember it, stop queueing, and fall through to the joins
        hipError_t herr = hipSuccess;
        const char *hwhat = "";
#define FORK_TRY(call) do { if (herr == hipSuccess && (herr = (call)) != hipSuccess) hwhat = #call; } while (0)
        FORK_TRY(hipEventRecord(ev_fork_, lc->stream));                        // the first part's mask is out
        for (int p = 0; p < parts && rc == HBHIP_OK && herr == hipSuccess; p++)
        {
            const int f0 = p * EEDI_PART, m = std::min(EEDI_PART, n - f0);
            hipEvent_t ready = p ? ev_mask_ : ev_fork_;                       // this part's mask
            if (p && ahead_) FORK_TRY(hipStreamWaitEvent(lc->stream, ev_mask_, 0));   // (it ran on the stream of its own)
            if (p + 1 < parts)
            {
                // the next part's mask: behind this part's mask, in front of this part's passes on this stream
                // (the 16-bit engine: on its stream of its own, behind this part's mask - see mask_ahead())
                const int g0 = f0 + EEDI_PART, gm = std::min(EEDI_PART, n - g0);
                hipStream_t ms = ahead_ ? ahead_ : lc->stream;
                if (ahead_) { FORK_TRY(hipStreamWaitEvent(ahead_, ready, 0)); if (herr != hipSuccess) break; ahead_used = true; }
                rc = enqueue_mask(g0, gm, lc, ms, &epoch[p + 1]);
                if (rc != HBHIP_OK) break;
                FORK_TRY(hipEventRecord(ev_mask_, ms));
            }
            if (herr != hipSuccess) break;
            const int groups = m >= 8 ? nside_ + 1 : 1;                       // the caller's stream and the side streams
            for (int g = 0, at = f0; g < groups && rc == HBHIP_OK && herr == hipSuccess; g++)
            {
                const int k = (f0 + m - at) / (groups - g);                   // what is left, evenly
                if (g == 0) rc = enqueue_passes(at, k, lc, lc->stream, epoch[p]);
                else
                {
                    FORK_TRY(hipStreamWaitEvent(side_[g - 1], ready, 0));
                    if (herr != hipSuccess) break;
                    used[g - 1] = true;
                    rc = enqueue_passes(at, k, lc, side_[g - 1], epoch[p]);
                }
                at += k;
            }
        }
#undef FORK_TRY
        if (herr != hipSuccess || rc != HBHIP_OK)
        {
            // whatever was queued on the side streams finishes before the caller sees the error
            for (int g = 0; g < MAX_SIDE; g++) if (used[g]) (void)hipStreamSynchronize(side_[g]);
            if (ahead_used) (void)hipStreamSynchronize(ahead_);
            if (herr != hipSuccess) rc = lc->fail(herr, hwhat);
        }
        else
            for (int g = 0; g < MAX_SIDE; g++)
                if (used[g])
                {
                    HBHIP_CHECK(lc, hipEventRecord(ev_join_[g], side_[g]));
                    HBHIP_CHECK(lc, hipStreamWaitEvent(lc->stream, ev_join_[g], 0));
                }
    }
    last_slot_ = start_ + n - 1;
    return rc;
}

// the five mask passes (+ the field extraction) of the n queued fields: old mask -> new mask.  The part no earlier
// field can influence is one launch, the rest a chain of launches, field after field
int Eedi2Engine::enqueue_mask(int f0, int n, hbhip_ctx *lc, hipStream_t st, uint32_t *epoch_out)
{
    const EediFrame srcp = at_slot(half_[0], start_ + f0), mskp = at_slot(half_[1], start_ + f0),
                    mskp_old = at_slot(half_[1], f0 ? start_ + f0 - 1 : last_slot_);
    P3 P;
    memset(&P, 0, sizeof(P));
    MaskSrc S;
    memset(&S, 0, sizeof(S));
    for (int c = 0; c < 3; c++)
    {
        P.pitch[c] = srcp.stride[c]; P.width[c] = srcp.width[c]; P.height[c] = srcp.height[c];
        P.a[c] = srcp.plane[c]; P.b[c] = mskp_old.plane[c]; P.c[c] = mskp.plane[c];
        S.spitch[c] = src_pitch_[c];
    }
    for (int f = 0; f < n; f++) for (int c = 0; c < 3; c++) S.frame[f][c] = src_frame_[f0 + f][c];
    P.fstride = slot_bytes_;
    P.tffbits = tffbits_ >> f0;
    const int mth = par_.magnitude_threshold * 10, vth = par_.laplacian_threshold * 81, lth = par_.variance_threshold;   // sic: swapped (decomb_template.c:390)
    const unsigned gx = (srcp.width[0] + MF_W - 1) / MF_W, gy = (srcp.height[0] + MF_H - 1) / MF_H;
    uint32_t epoch = 0;
    { const int erc = next_epoch(lc, &epoch); if (erc != HBHIP_OK) return erc; }
    *epoch_out = epoch;
    uint32_t *pflags = plane_flags_ + 3 * f0;
    if (n == 1)
        HBHIP_LAUNCH_ON(lc, st, "eedi2_mask_passes", k_mask_fused4, dim3(gx, gy, 3), dim3(MF_T), 0, P, S, 0, 0, mth, vth, lth,
                        par_.erosion_threshold, par_.dilation_threshold, pflags, epoch);
    else
    {
        // One launch: the tiles no earlier field can influence (upper) and the chain through the fields (lower, MaskChain),
        // field-major - a field's lower tiles, then its upper ones.  As two launches (all upper tiles, then the chain) the
        // chain ran alone at a third of the GPU: it is 16 links of latency, not work (58 + 139 us per 16 fields).
        MaskChain C = eedi_mask_chain_tiles(srcp, MF_W, MF_H, MF_OY);
        C.flags = chain_flags_;
        C.pflags = pflags;
        C.epoch = epoch;
        C.group = C.ntiles + C.nupper;
        C.has = (MF_OPT & 4) ? chain_has_ + (size_t)f0 * C.group : nullptr;
        guard_.bind(C);
        HBHIP_LAUNCH_ON(lc, st, "eedi2_mask_passes", k_mask_chain, dim3((unsigned)(C.group * n)), dim3(MF_T), 0, P, S, C, mth, vth, lth,
                     par_.erosion_threshold, par_.dilation_threshold);
        // the plane flags out of the tiles' words; and one workgroup that returns at once unless a wait above ran out
        // (MaskChain): no abort, no host round trip
        HBHIP_LAUNCH_ON(lc, st, "eedi2_mask_repair", k_mask_chain_repair, dim3(C.has ? 3u * (unsigned)n : 1u), dim3(MF_T), 0, P, S, C, n, mth, vth, lth,
                        par_.erosion_threshold, par_.dilation_threshold);
    }
    HBHIP_CHECK(lc, hipGetLastError());
    return HBHIP_OK;
}

// The pass sequence of eedi2_interpolate_plane (decomb_template.c:366-441) behind the mask passes, for the 3 planes of
// n of the queued fields, on their scratch frames.
int Eedi2Engine::enqueue_passes(int f0, int n, hbhip_ctx *lc, hipStream_t st, uint32_t epoch)
{
    // fields f0 .. f0 + n - 1 of the batch
    const int s0 = start_ + f0;
    const uint32_t tffbits = tffbits_ >> f0;
    const EediFrame srcp = at_slot(half_[0], s0), mskp = at_slot(half_[1], s0), tmpp = at_slot(half_[2], s0),
                    dstp = at_slot(half_[3], s0);
    const EediFrame dst2p = at_slot(full_[0], s0), tmp2p2 = at_slot(full_[1], s0), msk2p = at_slot(full_[2], s0),
                    tmp2p = at_slot(full_[3], s0), dst2mp = at_slot(full_[4], s0);
    uint32_t *cand = reinterpret_cast<uint32_t *>(cand_raw_) + (size_t)s0 * (slot_bytes_ / sizeof(uint32_t));
    const dim3 blk(64, 4);
    const unsigned gz = 3u * (unsigned)n;
    auto grid4_for = [&](const EediFrame &f, bool whole_pitch) {        // kernels with 4 pixels per thread
        const int w = whole_pitch ? f.stride[0] : f.width[0];
        return dim3(hbhip_grid_x((w + 255) / 256), (f.height[0] + 3) / 4, gz);
    };
    bool post_folded = false;
    auto dir_map = [&](const char *name, const EediFrame &f, const P3 &Pv, int step, int expand, int post = 0) {
        // the queueing form pays where few pixels reach the sort (expand: only peak pixels with >= 5 usable neighbours);
        // filter_dir_map sorts at most masked pixels, there the in-place form is ahead
        // step 2: a thread row per PAIR of rows (the rebuilt one and the copied one)
        const int trows = step == 1 ? f.height[0] : (f.height[0] + 1) / 2;             // thread rows
        if (!expand) HBHIP_LAUNCH_ON(lc, st, name, k_dir_map4, dim3(hbhip_grid_x((f.width[0] + 255) / 256), (trows + 3) / 4, gz), blk, 0, Pv, step, expand);
        else
        {
            HBHIP_LAUNCH_ON(lc, st, name, k_dir_map_c, dim3(hbhip_grid_x((f.width[0] + 255) / 256), (trows + DC_ROWS - 1) / DC_ROWS, gz), blk, 0, Pv, step, expand, post);
            post_folded = post != 0;
        }
    };
    auto geom = [&](P3 &P, const EediFrame &f) {
        for (int c = 0; c < 3; c++) { P.pitch[c] = f.stride[c]; P.width[c] = f.width[c]; P.height[c] = f.height[c]; }
    };
    auto bind = [&](uint8_t *(&slot)[3], const EediFrame &f) { for (int c = 0; c < 3; c++) slot[c] = f.plane[c]; };

    P3 P;
    memset(&P, 0, sizeof(P));
    P.fstride = slot_bytes_;
    P.tffbits = tffbits;
    P.pflags = plane_flags_ + 3 * f0;                             // every kernel behind FIELD_PLANE reads it: never null here
    P.pepoch = epoch;

    // half-height passes
    geom(P, srcp);
    // filter_dir_map and expand_dir_map as one launch (k_dir_map_fe): calc_directions then writes dstp, so that the fused
    // pass leaves the expanded map in tmpp, where the reference has it
    // (the padding of the rows with it: calc_directions' memset leaves 255 there, which the fused pass writes into tmpp,
    // while dstp keeps the zeros the passes of the reference never touch)
    const bool fused = par_.maximum_search_distance <= CD_HALO - 2 && hbhip_dev_int("HBHIP_EEDI2_FUSE_DIRMAP", 1) != 0;
    const uint32_t padv = fused ? 0u : 0xffffffffu;
    bind(P.a, mskp); bind(P.b, srcp); bind(P.c, fused ? dstp : tmpp);
    const int nt13 = (par_.noise_threshold * 13) & 0xff, nt19 = (par_.noise_threshold * 19) & 0xff;      // typed `pixel` in the reference
    if (par_.maximum_search_distance <= CD_HALO - 2)
    {
        // 256 columns x 4 rows per block: blocks with at least half of their pixels listed search in the dense form
        // (calc_dir_dense), the others walk their list.  Measured on the decomb bob workload, us per 16 fields: the list form
        // alone 505 (2 rows) / 673 (8 rows); with the dense form 349 at 4 rows (120 registers, four waves per SIMD), 384 at 6
        // (161), 417 at 8 (202 registers, two waves: fewer instructions - 2.75 SADs a pixel-step against 3.5 - but the
        // LDS latency of a trip shows; forced to 128 registers with spills: 334).  Thresholds of 3/8 .. 3/4 of a block's
        // pixels measure alike (the blocks are either nearly full or far from it); below 1/4 the dense form loses.
        int dense_min = CD_W * 4 / 2;
#ifdef HBHIP_DEV
        const int rows = hbhip_dev_int("HBHIP_EEDI2_CALCDIR_ROWS", 4);
        dense_min = hbhip_dev_int("HBHIP_EEDI2_CALCDIR_DENSE_MIN", CD_W * rows / 2);
        if (rows == 8)
            HBHIP_LAUNCH_ON(lc, st, "eedi2_calc_directions", k_calc_dir_rows<8>,
                         dim3(hbhip_grid_x((srcp.stride[0] + CD_W - 1) / CD_W), (srcp.height[0] + 7) / 8, gz), dim3(CD_W), 0, P,
                         par_.maximum_search_distance, nt13, nt19, dense_min, padv);
        else if (rows == 6)
            HBHIP_LAUNCH_ON(lc, st, "eedi2_calc_directions", k_calc_dir_rows<6>,
                         dim3(hbhip_grid_x((srcp.stride[0] + CD_W - 1) / CD_W), (srcp.height[0] + 5) / 6, gz), dim3(CD_W), 0, P,
                         par_.maximum_search_distance, nt13, nt19, dense_min, padv);
        else if (rows == 2)
            HBHIP_LAUNCH_ON(lc, st, "eedi2_calc_directions", k_calc_dir_rows<2>,
                         dim3(hbhip_grid_x((srcp.stride[0] + CD_W - 1) / CD_W), (srcp.height[0] + 1) / 2, gz), dim3(CD_W), 0, P,
                         par_.maximum_search_distance, nt13, nt19, dense_min, padv);
        else
#endif
        HBHIP_LAUNCH_ON(lc, st, "eedi2_calc_directions", k_calc_dir_rows<4>,
                     dim3(hbhip_grid_x((srcp.stride[0] + CD_W - 1) / CD_W), (srcp.height[0] + 3) / 4, gz), dim3(CD_W), 0, P,
                     par_.maximum_search_distance, nt13, nt19, dense_min, padv);
    }
    else
    {
        // search distances beyond the LDS halo: a work list of the edge pixels, field after field
        size_t half_px = 0;
        for (int c = 0; c < 3; c++) half_px += (size_t)srcp.width[c] * srcp.height[c];
        P3 Q = P;
        Q.fstride = 0;
        for (int f = 0; f < n; f++)
        {
            for (int c = 0; c < 3; c++)
            {
                Q.a[c] = P.a[c] + (size_t)f * slot_bytes_; Q.b[c] = P.b[c] + (size_t)f * slot_bytes_; Q.c[c] = P.c[c] + (size_t)f * slot_bytes_;
            }
            Q.pflags = P.pflags + 3 * f;                              // (the launch's field 0 is the batch's field f)
            HBHIP_CHECK(lc, hipMemsetAsync(work_count_, 0, sizeof(int), st));
            HBHIP_LAUNCH_ON(lc, st, "eedi2_calc_directions_mark", k_calc_dir_mark, dim3((srcp.stride[0] + 63) / 64, (srcp.height[0] + 3) / 4, 3), blk, 0,
                         Q, work_list_, work_count_);
            HBHIP_LAUNCH_ON(lc, st, "eedi2_calc_directions_work", k_calc_dir_work, dim3((unsigned)((half_px + 255) / 256)), dim3(256), 0, Q,
                         (const uint32_t *)work_list_, (const int *)work_count_, par_.maximum_search_distance, nt13, nt19);
        }
    }
    if (fused)
    {
        bind(P.a, mskp); bind(P.b, dstp); bind(P.c, tmpp);
        HBHIP_LAUNCH_ON(lc, st, "eedi2_filter_expand_dir_map", (k_dir_map_fe<1, false>),
                        dim3(hbhip_grid_x((srcp.stride[0] + 255) / 256), (srcp.height[0] + FE_R - 1) / FE_R, gz), blk, 0, P, 0xffffffffu);
    }
    else
    {
        bind(P.a, mskp); bind(P.b, tmpp); bind(P.c, dstp);
        dir_map("eedi2_filter_dir_map", srcp, P, 1, 0);
        bind(P.a, mskp); bind(P.b, dstp); bind(P.c, tmpp);
        dir_map("eedi2_expand_dir_map", srcp, P, 1, 1);
    }
    bind(P.a, mskp); bind(P.b, tmpp); bind(P.c, dstp);
    HBHIP_LAUNCH_ON(lc, st, "eedi2_filter_map", k_filter_map, grid4_for(srcp, false), blk, 0, P);
    // line doubling of srcp / dstp / mskp + mark_directions_2x in one launch (full-height geometry)
    geom(P, dst2p);
    bind(P.g, srcp); bind(P.b, dstp); bind(P.a, mskp);
    // the pair of _2x dir-map passes behind it as one launch too (k_dir_map_fe<2>): the marked map then goes to dst2mp
    const bool fused2 = hbhip_dev_int("HBHIP_EEDI2_FUSE_DIRMAP_2X", 1) != 0;
    // ... and the pair in front of post_process (k_dir_map_fe<2, true>).  That pass reads the map the lattice leaves and the
    // reference has it write the new map over it, with the eedi2_bit_blit in front keeping a copy in tmp2p2
    // (decomb_template.c:426-429); one launch cannot read a plane's neighbourhoods and write that plane.  So the two planes
    // change places from mark_directions_2x on: the direction map lives in tmp2p2 - where the copy would go: no blit - and the
    // doubled half-height map (the lattice's omsk, dead behind it) in tmp2p, which the fused pass then overwrites with the new
    // map.  Every plane ends as the reference leaves it, the padding of its rows included.
    const bool post1 = par_.post_processing == 1 || par_.post_processing == 3;
    const bool swapped = fused2 && post1 && hbhip_dev_int("HBHIP_EEDI2_FUSE_DIRMAP_POST", 1) != 0;
    const EediFrame &map2 = swapped ? tmp2p2 : tmp2p, &omsk2 = swapped ? tmp2p : tmp2p2;
    bind(P.d, dst2p); bind(P.e, omsk2); bind(P.f, msk2p); bind(P.c, fused2 ? dst2mp : map2);
    HBHIP_LAUNCH_ON(lc, st, "eedi2_mark_directions_2x", k_mark_2x4,                                      // a thread row per PAIR of full-height rows
                 dim3(hbhip_grid_x((dst2p.stride[0] + 255) / 256), ((dst2p.height[0] + 1) / 2 + 3) / 4, gz), blk, 0, P, fused2 ? 0u : 0xffffffffu);
    for (int c = 0; c < 3; c++) P.d[c] = P.e[c] = P.f[c] = P.g[c] = nullptr;    // slot d doubles as the dir-map kernels' optional copy target
    const dim3 fe2_grid(hbhip_grid_x((dst2p.stride[0] + 255) / 256), ((dst2p.height[0] + 1) / 2 + FE_R - 1) / FE_R, gz);
    if (fused2)
    {
        bind(P.a, msk2p); bind(P.b, dst2mp); bind(P.c, map2);
        HBHIP_LAUNCH_ON(lc, st, "eedi2_filter_expand_dir_map_2x", (k_dir_map_fe<2, false>), fe2_grid, blk, 0, P, swapped ? 0u : 0xffffffffu);
    }
    else
    {
        bind(P.a, msk2p); bind(P.b, tmp2p); bind(P.c, dst2mp);
        dir_map("eedi2_filter_dir_map_2x", dst2p, P, 2, 0);
        bind(P.a, msk2p); bind(P.b, dst2mp); bind(P.c, tmp2p);
        dir_map("eedi2_expand_dir_map_2x", dst2p, P, 2, 1);
    }
    // (a workgroup per row here: with the copied row of a pair folded into the workgroup of the rebuilt one, as in the dir-map
    // passes, this kernel went from 131 to 163-165 us per launch)
    const dim3 fg_grid(hbhip_grid_x((dst2p.width[0] + FG_W - 1) / FG_W), (dst2p.height[0] + 2 * FG_R - 1) / (2 * FG_R), gz);
    bind(P.a, msk2p); bind(P.b, map2); bind(P.c, dst2mp);
    HBHIP_LAUNCH_ON(lc, st, "eedi2_fill_gaps_2x", k_fill_gaps_b, fg_grid, dim3(FG_T), 0, P);
    bind(P.a, msk2p); bind(P.b, dst2mp); bind(P.c, map2);
    HBHIP_LAUNCH_ON(lc, st, "eedi2_fill_gaps_2x", k_fill_gaps_b, fg_grid, dim3(FG_T), 0, P);
    // lattice
    bind(P.a, map2); bind(P.b, dst2p); bind(P.c, omsk2);
    {
        const int nrows = (dst2p.height[0] - 1) / 2;      // rows y0, y0 + 2, ... < height - 1 for either parity (the heights are even)
        const int nt = par_.noise_threshold;
        HBHIP_LAUNCH_ON(lc, st, "eedi2_lattice_candidates", k_lattice_cand_q, dim3(hbhip_grid_x((dst2p.width[0] + LQ_W - 1) / LQ_W), nrows, gz),
                     dim3(256), 0, P, cand, cand_pitch_, cand_plane_stride_, (nt * 4) & 0xff, (nt * 7) & 0xff, (nt * 8) & 0xff, nt);
        HBHIP_LAUNCH_ON(lc, st, "eedi2_lattice_resolve", k_lattice_resolve, dim3(1, nrows + 1, gz), dim3(LR_T), 0, P,
                     (const uint32_t *)cand, cand_pitch_, cand_plane_stride_);
    }
    if (swapped)
    {
        // filter_dir_map_2x, expand_dir_map_2x and post_process in one launch: tmp2p2 (the map, and what the reference's copy of
        // it would hold) -> tmp2p, the filtered map to dst2mp, the corrections to dst2p
        bind(P.a, msk2p); bind(P.b, tmp2p2); bind(P.c, tmp2p); bind(P.d, dst2mp); bind(P.f, dst2p);
        HBHIP_LAUNCH_ON(lc, st, "eedi2_filter_expand_dir_map_2x_post", (k_dir_map_fe<2, true>), fe2_grid, blk, 0, P, 0xffffffffu);
        for (int c = 0; c < 3; c++) P.d[c] = P.f[c] = nullptr;
    }
    else if (post1)
    {
        // eedi2_bit_blit(tmp2p -> tmp2p2) keeps the pre-filter direction map for post_process
        // (decomb_template.c:426); the filter that follows reads every byte of tmp2p the blit copies,
        // so it writes that copy itself (slot d) and the separate launch is saved
        bind(P.a, msk2p); bind(P.b, tmp2p); bind(P.c, dst2mp); bind(P.d, tmp2p2);
        dir_map("eedi2_filter_dir_map_2x", dst2p, P, 2, 0);
        for (int c = 0; c < 3; c++) P.d[c] = nullptr;
        bind(P.a, msk2p); bind(P.b, dst2mp); bind(P.c, tmp2p); bind(P.e, tmp2p2); bind(P.f, dst2p);
        dir_map("eedi2_expand_dir_map_2x", dst2p, P, 2, 1, 1);                          // + post_process where the kernel can carry it
        for (int c = 0; c < 3; c++) P.e[c] = P.f[c] = nullptr;
        bind(P.a, tmp2p); bind(P.b, tmp2p2); bind(P.c, dst2p);
        if (!post_folded) HBHIP_LAUNCH_ON(lc, st, "eedi2_post_process", k_post, grid4_for(dst2p, false), blk, 0, P);
    }
    if (par_.post_processing == 2 || par_.post_processing == 3)
    {
        // junctions and corners, field after field and plane after plane (see CornerArgs: the derivative arrays carry
        // values from plane to plane and from field to field)
        for (int f = 0; f < n; f++)
        {
            const int tff = (int)((tffbits >> f) & 1u);
            const size_t foff = (size_t)f * slot_bytes_;
            for (int c = 0; c < 3; c++)
            {
                CornerArgs A;
                A.src = srcp.plane[c] + foff; A.tmp = tmpp.plane[c] + foff;
                for (int i = 0; i < 3; i++) { A.c[i] = deriv_[i]; A.t[i] = deriv_tmp_[i]; }
                A.pitch = srcp.stride[c]; A.width = srcp.width[c]; A.height = srcp.height[c];
                const dim3 g1((A.width + 63) / 64, (A.height + 3) / 4, 1), g3(g1.x, g1.y, 3);
                HBHIP_LAUNCH_ON(lc, st, "eedi2_gaussian_blur1_h", k_blur1<false>, g1, blk, 0, A);
                HBHIP_LAUNCH_ON(lc, st, "eedi2_gaussian_blur1_v", k_blur1<true>, g1, blk, 0, A);
                HBHIP_LAUNCH_ON(lc, st, "eedi2_calc_derivatives", k_derivatives, g1, blk, 0, A);
                HBHIP_LAUNCH_ON(lc, st, "eedi2_gaussian_blur_sqrt2_h", k_blur_sqrt2<false>, g3, blk, 0, A);
                HBHIP_LAUNCH_ON(lc, st, "eedi2_gaussian_blur_sqrt2_v", k_blur_sqrt2<true>, g3, blk, 0, A);
                const int rows = (dst2p.height[c] - 7 - (8 - tff) + 1) / 2;      // y = 8-field, 10-field, ... < height-7
                if (rows > 0)
                    HBHIP_LAUNCH_ON(lc, st, "eedi2_post_process_corner", k_post_corner, dim3((A.width + 63) / 64, (rows + 3) / 4, 1), blk, 0, A,
                                 (const uint8_t *)(tmp2p2.plane[c] + foff), dst2p.plane[c] + foff, tff, dst2p.height[c]);
            }
        }
    }
    HBHIP_CHECK(lc, hipGetLastError());
    return HBHIP_OK;
}

#ifdef HBHIP_DEV_STATS
// development builds with -DHBHIP_DEV_STATS: the search-schedule counters of k_calc_dir_rows (read and cleared)
extern "C" int hbhip_dev_eedi2_stats(unsigned long long *out, int n)
{
    unsigned long long h[24] = { 0 };
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cd_stats), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < n && i < 24; i++) out[i] = h[i];
    unsigned long long z[24] = { 0 };
    return hipMemcpyToSymbol(HIP_SYMBOL(g_cd_stats), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
