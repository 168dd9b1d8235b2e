// nlmeans.hip — Non-local-means denoise for gfx950.
//
// Replaces the CPU hot loop of libhb/templates/nlmeans_template.c:545-717
// (build_integral_* + nlmeans_plane_8) and the bordered-copy of :20-101.
//
// Algorithm per output pixel (identical integers / identical float sequence to
// the reference, SURVEY §6b hazards 1-8):
//   for f in frames (f=0 is the frame itself, f>0 the FOLLOWING frames)
//     for dy, dx in [-r/2, r/2]^2                      (nlmeans_template.c:629-640)
//       origin (f=0,0,0): sums += origin_tune (in double)           (:644-655)
//       diff  = sum over the n x n patch of (src - cmp(+dx,+dy))^2   (:573-574,:682)
//       if diff < diff_max: w = exptable[(int)(diff * wft)]          (:685-690)
//            weight_sum += w ; pixel_sum += w * cmp(+dx,+dy)         (:692-693)
//   out = (u8)(pixel_sum / weight_sum), 0 -> source pixel            (:710-711)
//
// GPU mapping (no integral image): one workgroup (32 x 8 lanes) owns a 120 x 64 tile.  Frame 0's
// tile with the patch + search halo stays in LDS for the whole kernel (it is the source patch tile
// and the compare tile of f = 0); each following frame's tile is staged in turn; the reference's
// mirrored borders (nlmeans_template.c:29-41) are applied by index reflection while staging, so no
// bordered copy of a frame is ever materialised in HBM.  A lane owns 4 adjacent columns x 8 rows.
// Per displacement it walks 8+n-1 rows keeping prefix sums of the squared differences down its own
// 4 columns; on each of its 8 output rows the n-row column sums are combined into the 4 horizontal
// n-sums with DPP wave shifts (the needed prefix / suffix sums of the neighbouring lanes), so no
// square is computed twice along x.  Weight / pixel accumulators of the 32 pixels live in registers
// across all displacements and frames.  HBM traffic is ~(nframes + 1) bytes per pixel; the kernel
// is VALU-bound (see DESIGN.md for the instruction budget).
#include "hbhip_internal.h"

#include <algorithm>
#include <type_traits>
#include <unordered_map>

namespace {

constexpr int PX = 4;            // pixels per thread along x (one dword)
constexpr int RY = 8;            // rows per thread
constexpr int TXN = 32;          // threads along x
constexpr int TYN = 8;           // threads along y
constexpr int TH = TYN * RY;     // 64
constexpr int NLM_BORDER = 16;   // nlmeans.c:529 for every patch size <= 29
#ifndef NLM16_SYM_PAIRS
#define NLM16_SYM_PAIRS 3        // 16-bit samples: how many of frame 0's four displacement pairs share their distances (3 | 4)
#endif

struct alignas(16) NlmJob
{
    const uint8_t *frame[HBHIP_NLMEANS_FRAMES_MAX];
    int            fpitch[HBHIP_NLMEANS_FRAMES_MAX];   // row pitch of each temporal frame
    // prefilter != 0 only (nlmeans_template.c:428-543): the prefiltered twin of every frame, which
    // the patch distances are taken on, and the plane frame 0's patches are read from (src_pre,
    // latched at :615 -- the raw frame until the frame has been prefiltered by an earlier call)
    const uint8_t *frame_pre[HBHIP_NLMEANS_FRAMES_MAX];
    int            ppitch[HBHIP_NLMEANS_FRAMES_MAX];
    const uint8_t *src_pre;
    int            src_pre_pitch;
    uint8_t       *dst;
    const float   *exptable;
    double         origin_tune;
    float          wft;
    int            diff_max;
    int            diff_cap;       // FAST gate: smallest diff whose table index is 127 (weight 0)
    uint32_t       imul4;          // FAST 2: weight_fact * 2^(34 + ishift) when ((diff * that) >> 32 >> ishift) & ~3 == 4 * (int)(diff * wft)
    int            ishift;         // for every diff <= diff_cap (checked on the host), else 0.  ishift is 0 at 8 bits (the 8-bit kernel
                                   // takes nothing else); small weight factors (10 / 12 bits) need it for the product to be an integer
    int            w, h, dst_pitch;
    int            nframes, r_half;
    int            tiles_x, tile_start;
};

// the job table's way to the device (see launch_views): src = the pinned host buffer, read over the bus
__global__ void job_table_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, int n16)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

struct __attribute__((packed, aligned(1))) u32_unaligned { uint32_t v; };
typedef float f2 __attribute__((ext_vector_type(2)));

// mirrored coordinate (edge pixel repeated), then clamped so far-out halo
// reads of tiny planes stay inside the allocation.
__device__ __forceinline__ int reflect(int x, int n)
{
    x = x < 0 ? -1 - x : x;
    x = x >= n ? 2 * n - 1 - x : x;
    return min(max(x, 0), n - 1);
}

__device__ __forceinline__ uint32_t byte_of(uint32_t v, int k) { return (v >> (8 * k)) & 0xffu; }

// ---------------------------------------------------------------------------------------------
// Lane sharing.  A thread that squared the differences of its whole 4+n-1 byte window would redo
// (n-1)/4 of its neighbours' work.  Here a lane squares only its own 4 bytes per row, and the
// horizontal n-sums are completed with DPP wave shifts: pixel p of lane l needs the last
// (n/2 - p) squares of lane l-1 and the first (p + n/2 - 3) squares of lane l+1.  The two outer
// lanes of each 32-lane tile row only feed their neighbours (they own no output), so a tile is
// (32-2)*4 = 120 pixels wide; n <= 9 keeps every window inside the adjacent lanes.
constexpr int LTXA = TXN - 2;            // lanes of a tile row that own output pixels
constexpr int LTW = LTXA * PX;           // 120

// Stage a (rows x dwords*4)-byte window of `plane` whose top-left pixel is (x0, y0) into LDS
// (row pitch `pitch` dwords) with the reference's mirrored borders (nlmeans_template.c:29-41).
// A thread keeps one column for the whole tile, so the column reflection and the index split
// are computed once; the per-row work is a row reflection, one address and one load.
__device__ __forceinline__ void load_tile_p(uint32_t *lds, int pitch, int dwords, int rows,
                                            const uint8_t *__restrict__ plane, int src_pitch,
                                            int w, int h, int x0, int y0)
{
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                      // keep this cold code out of LICM's reach
    const int rpp = (TXN * TYN) / dwords;              // rows per pass
    const int r0 = (int)(((float)tid + 0.5f) * (1.0f / (float)dwords));   // tid / dwords (tid < 2^10)
    const int c = tid - r0 * dwords;
    if (r0 >= rpp) return;
    const int x = x0 + 4 * c;
    const bool whole = x >= 0 && x + 3 < w;
    const int o0 = reflect(x, w), o1 = reflect(x + 1, w), o2 = reflect(x + 2, w), o3 = reflect(x + 3, w);
    uint32_t *out = lds + r0 * pitch + c;
    if (y0 >= 0 && y0 + rows <= h && x0 >= 0 && x0 + 4 * dwords <= w)
    {
        // tile and halo wholly inside the plane (most tiles): no reflection, one pointer bump per row
        const uint8_t *ptr = plane + (size_t)(y0 + r0) * src_pitch + x;
        const size_t step = (size_t)rpp * src_pitch;
#pragma nounroll
        for (int r = r0; r < rows; r += rpp, out += rpp * pitch, ptr += step)
            *out = reinterpret_cast<const u32_unaligned *>(ptr)->v;
        return;
    }
#pragma nounroll
    for (int r = r0; r < rows; r += rpp, out += rpp * pitch)
    {
        const uint8_t *row = plane + (size_t)reflect(y0 + r, h) * src_pitch;
        uint32_t v;
        if (whole)
            v = reinterpret_cast<const u32_unaligned *>(row + x)->v;
        else
            v = (uint32_t)row[o0] | ((uint32_t)row[o1] << 8) | ((uint32_t)row[o2] << 16) | ((uint32_t)row[o3] << 24);
        *out = v;
    }
}

// DPP wave shifts (all 64 lanes are active wherever these are used).
__device__ __forceinline__ uint32_t from_lane_below(uint32_t x)   // lane l <- lane l-1
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t from_lane_above(uint32_t x)   // lane l <- lane l+1
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}

// a kernel whose code object declares no static LDS: its dynamic LDS block starts at LDS address 0
static bool nlm_no_static_lds(const void *kernel)
{
    hipFuncAttributes a;
    memset(&a, 0, sizeof(a));
    if (hipFuncGetAttributes(&a, kernel) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.sharedSizeBytes == 0;
}

// FAST: how a patch distance becomes a table index.  0: the reference's expression with its gate (diff < diff_max).
// 1: the gate folded into a clamp (exactly equivalent when the table ends in 0 at the index the cap maps to).  2: the
// clamp, and the float product (int)(diff * wft) replaced by an integer multiply-high: wft is a float, so wft * 2^32 is
// an integer M, and (diff * M) >> 32 is the exact floor of the real product - the float product can only differ from
// it where rounding carries it across an integer, which the host rules out for every diff up to the cap before it
// picks this form (three instructions per pixel instead of five: min, mul_hi, and).
// 3: form 2, and inside frame 0 the patch distance of a displacement and of its mirror computed once (range 3 only, no
// prefilter).  The distance between the patches at p and p + d IS the distance between the patches at p + d and
// (p + d) - d: W(-d)(p) = W(d)(p - d) - taken on the same mirrored picture, so it holds at the borders too.  The
// displacements of frame 0 come in the order (-1,-1) (-1,0) (-1,1) (0,-1) (0,0) (0,1) (1,-1) (1,0) (1,1): the first
// four are computed as ever and leave the table index of every pixel of the lane (4 x 7 bits per dword and row; one
// row more than the lane puts out for the three with dy = -1) in LDS, in the space the tile of the NEXT frame takes
// later plus 26.5 KB; the last four read the index of their mirror one row down and / or one pixel across (a
// neighbouring lane's dword of the same wave) and only look the weight up, convert the pixel and accumulate - in the
// reference's order of displacements, with the reference's weights.
template <int N, int FAST, int CPD, bool PRE>
__global__ __launch_bounds__(TXN * TYN, 3) void nlmeans_lanes_kernel(const NlmJob *__restrict__ jobs, int njobs,
                                                                     int cmp_rows, int rq)
{
    constexpr int NH = N / 2;
    constexpr int ROWS = RY + N - 1;
    constexpr bool INTIDX = FAST >= 2;
    constexpr bool SYM = FAST == 3 && !PRE;
    static_assert(FAST != 3 || N <= 7, "the pairs' sharing reads a neighbouring lane's edge pixel: patch sizes up to 7");
    constexpr int ST_ROWS = RY + 1, ST_SLOT = ST_ROWS * TXN * TYN;        // the stash: [slot 0..3][row 0..RY][thread] dwords
    static_assert(NH <= PX, "patch must not reach past the adjacent lane");
    static_assert(CPD % 8 == 4, "the two tile rows of a wave must sit 32 banks apart");

    // Tiles of CPD x cmp_rows dwords.  Without a prefilter: s_t0 holds frame 0 with the patch +
    // search halo for the whole kernel (it is both the source patch tile and the compare tile of
    // f = 0), s_tc the following frames in turn.  With a prefilter (PRE) the patch distances are
    // taken between src_pre (s_t0) and the prefiltered frame f (s_tc), while the pixel values
    // that are averaged come from the raw frames: s_r0 (frame 0, also the origin term and the
    // zero fallback) and s_rc (frame f > 0).  CPD is a template constant so every LDS read in
    // the row walk uses an immediate offset.
    // The weight table sits first, at a fixed LDS address, so a table read is one ds_read with the address in its
    // immediate offset; the tiles follow it (128 dwords on: the same banks as without it).
    extern __shared__ uint32_t smem[];
    const int tile_dwords = CPD * cmp_rows + 4;
    float *s_exp = reinterpret_cast<float *>(smem);
    uint32_t *s_t0 = smem + 128;
    uint32_t *s_tc = s_t0 + tile_dwords;
    uint32_t *s_r0 = PRE ? s_tc + tile_dwords : s_t0;
    uint32_t *s_rc = PRE ? s_r0 + tile_dwords : s_tc;
    uint32_t *s_stash = s_tc;                                     // SYM: over the next frame's tile (free while f == 0) and beyond

    // which (frame, plane) job owns this tile: binary search over the jobs' first tile indices
    int j = 0;
    for (int hi = njobs - 1; j < hi;)
    {
        const int mid = (j + hi + 1) >> 1;
        if ((int)blockIdx.x >= jobs[mid].tile_start) j = mid; else hi = mid - 1;
    }
    const NlmJob &job = jobs[j];
    const int tile = blockIdx.x - job.tile_start;
    const int tile_y = tile / job.tiles_x;
    int tile_x = tile - tile_y * job.tiles_x;
    // an even number of tiles per row (1920 / 120 = 16) would keep every column of tiles on one XCD, or on a few (workgroups
    // go to the XCDs round robin): each row of tiles starts one column further on instead, so that the tile below a tile is an
    // odd number of workgroups away (hbhip_internal.h: hbhip_grid_x has the numbers)
    if (HBHIP_GRID_ROTATE && (job.tiles_x & 1) == 0) { tile_x += tile_y % job.tiles_x; if (tile_x >= job.tiles_x) tile_x -= job.tiles_x; }
    const int tx0 = tile_x * LTW, ty0 = tile_y * TH;
    const int w = job.w, h = job.h;
    const int RH = job.r_half;

    const int tx = threadIdx.x & (TXN - 1);
    const int ty = threadIdx.x / TXN;

    // FAST 2 reads the table by its LDS address, taken to be 0: this kernel has no static LDS, so the dynamic block -
    // and the table at its head - starts there (the compiler leaves "+ &smem" as an add of 0 per read otherwise)
    typedef __attribute__((address_space(3))) const float lds_cfloat;
    // (checked on the host before the first launch of every such instantiation: nlm_no_static_lds)
    __builtin_assume(!INTIDX || reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint32_t *)smem) == 0);
    if (threadIdx.x < 128) s_exp[threadIdx.x] = job.exptable[threadIdx.x];
    // lane tx holds the source pixels tx0 + 4*(tx-1) .. +3: the tiles start one lane (and rq
    // dwords of search halo) left of tx0
    if (PRE)
    {
        load_tile_p(s_t0, CPD, CPD, cmp_rows, job.src_pre, job.src_pre_pitch, w, h,
                    tx0 - PX - 4 * rq, ty0 - NH - RH);
        load_tile_p(s_r0, CPD, CPD, cmp_rows, job.frame[0], job.fpitch[0], w, h,
                    tx0 - PX - 4 * rq, ty0 - NH - RH);
    }
    else
    {
        load_tile_p(s_t0, CPD, CPD, cmp_rows, job.frame[0], job.fpitch[0], w, h,
                    tx0 - PX - 4 * rq, ty0 - NH - RH);
    }
    const int own_off = (ty * RY + RH) * CPD + tx + rq;
    const uint32_t *own = s_t0 + own_off;       // this lane's source-patch dwords, row 0 of its walk
    const uint32_t *own_raw = s_r0 + own_off;   // the same pixels of the raw frame being filtered

    // weight / weighted-pixel accumulators as float pairs: the adds and the w*pixel product
    // below are packed (v_pk_add_f32 / v_pk_mul_f32, each lane of a pair rounded like the scalar op)
    f2 aw[RY][PX / 2], ap[RY][PX / 2];
#pragma unroll
    for (int o = 0; o < RY; o++)
#pragma unroll
        for (int p = 0; p < PX / 2; p++) { aw[o][p] = f2{0.f, 0.f}; ap[o][p] = f2{0.f, 0.f}; }
    const f2 wft2 = {job.wft, job.wft};

    const float wft = job.wft;
    const int diff_max = job.diff_max;
    const int diff_cap = job.diff_cap;
    const uint32_t imul4 = job.imul4;
    const double origin_tune = job.origin_tune;
    // a wave (two tile rows of lanes) whose 16 output rows all lie below the plane has nothing to do
    const bool wave_live = ty0 + (ty & ~1) * RY < h;

    for (int f = 0; f < job.nframes; f++)
    {
        if (PRE || f > 0)
        {
            __syncthreads();   // everyone is done with the previous compare tile
            if (PRE)
            {
                load_tile_p(s_tc, CPD, CPD, cmp_rows, job.frame_pre[f], job.ppitch[f], w, h,
                            tx0 - PX - 4 * rq, ty0 - NH - RH);
                if (f > 0)
                    load_tile_p(s_rc, CPD, CPD, cmp_rows, job.frame[f], job.fpitch[f], w, h,
                                tx0 - PX - 4 * rq, ty0 - NH - RH);
            }
            else
            {
                load_tile_p(s_tc, CPD, CPD, cmp_rows, job.frame[f], job.fpitch[f], w, h,
                            tx0 - PX - 4 * rq, ty0 - NH - RH);
            }
        }
        __syncthreads();
        if (!wave_live) continue;
        const uint32_t *cmp_tile = (PRE || f > 0) ? s_tc : s_t0;     // what the patch distances are taken against
        const uint32_t *pix_tile = f > 0 ? s_rc : s_r0;              // what is averaged (== cmp_tile without a prefilter)

        // one displacement; F0: frame 0 of a launch that shares the pairs' distances (a compile-time flag, so that the
        // row walk of every other frame carries none of the stash code)
        auto displacement = [&](auto f0c, int dy, int dx) __attribute__((always_inline))
        {
            constexpr bool F0 = decltype(f0c)::value;
            {
                if (f == 0 && dx == 0 && dy == 0)
                {
#pragma unroll
                    for (int o = 0; o < RY; o++)
                    {
                        const uint32_t cpx = own_raw[(o + NH) * CPD];
#pragma unroll
                        for (int p = 0; p < PX; p++)
                        {
                            aw[o][p / 2][p & 1] = (float)((double)aw[o][p / 2][p & 1] + origin_tune);
                            ap[o][p / 2][p & 1] = (float)((double)ap[o][p / 2][p & 1] + origin_tune * (double)(int)byte_of(cpx, p));
                            __builtin_amdgcn_sched_barrier(0);   // cold block: keep its f64 temporaries few
                        }
                    }
                    return;
                }

                const int s = dx + 4 * rq;                   // >= 0, wave-uniform
                const int sh = s & 3;
                if (SYM && F0 && (dy > 0 || (dy == 0 && dx > 0)))
                {
                    // the second of a pair: the index its mirror (-dy, -dx) left for the pixel at (row + dy, x + dx)
                    const uint32_t *st = s_stash + ((-dy + 1) * 3 + (-dx + 1)) * ST_SLOT + dy * (TXN * TYN) + (int)threadIdx.x;
                    const uint32_t *prow = pix_tile + (ty * RY + dy + RH + NH) * CPD + tx + (s >> 2);
#pragma unroll
                    for (int o = 0; o < RY; o++)
                    {
                        const uint32_t w0 = st[o * (TXN * TYN)];
                        uint32_t id4 = w0;
                        if (dx > 0) id4 = __builtin_amdgcn_alignbyte(st[o * (TXN * TYN) + 1], w0, 1);
                        else if (dx < 0) id4 = __builtin_amdgcn_alignbyte(w0, st[o * (TXN * TYN) - 1], 3);
                        const uint32_t pix = __builtin_amdgcn_alignbyte(prow[o * CPD + 1], prow[o * CPD], sh);
                        const uint32_t o0 = (id4 << 2) & 0x1fcu, o1 = (id4 >> 6) & 0x1fcu, o2 = (id4 >> 14) & 0x1fcu, o3 = (id4 >> 22) & 0x1fcu;
                        const f2 wa = {*reinterpret_cast<lds_cfloat *>(o0), *reinterpret_cast<lds_cfloat *>(o1)};
                        const f2 wb = {*reinterpret_cast<lds_cfloat *>(o2), *reinterpret_cast<lds_cfloat *>(o3)};
                        const f2 pa = {(float)(int)byte_of(pix, 0), (float)(int)byte_of(pix, 1)};
                        const f2 pb = {(float)(int)byte_of(pix, 2), (float)(int)byte_of(pix, 3)};
                        aw[o][0] += wa; ap[o][0] += wa * pa;
                        aw[o][1] += wb; ap[o][1] += wb * pb;
                    }
                    return;
                }
                // the first of a pair leaves its indices behind; with dy < 0 its mirror reads them one row down
                constexpr bool stash_on = SYM && F0;        // (here: dy < 0 or dy == 0 && dx < 0)
                const bool extra_row = stash_on && dy < 0;
                uint32_t *stw = s_stash + ((dy + 1) * 3 + (dx + 1)) * ST_SLOT + (int)threadIdx.x;
                const uint32_t *srow = own;
                const uint32_t *crow = cmp_tile + (ty * RY + dy + RH) * CPD + tx + (s >> 2);

                // Column sums first: C[q] is the running (prefix) sum down this lane's 4 columns of the
                // squared differences, hist[] keeps the first RY-1 of them, so the n-row window sum of
                // output row o is C(after row o+n-1) - C(after row o-1).  Only output rows then pay
                // for the horizontal n-sum across lanes.
                uint32_t C[PX], hist[RY - 1 + (SYM ? 1 : 0)][PX], v[PX];
                uint32_t centre[NH + 1];                     // compare-frame dwords of the last NH+1 rows
                uint32_t pixq = 0;                           // pixels and ...
                f2 wq[PX / 2];                               // ... table weights in flight for the previous output row
                const uint32_t *prow = pix_tile + (ty * RY + dy + RH + NH) * CPD + tx + (s >> 2);   // PRE only
#pragma unroll
                for (int q = 0; q < PX; q++) C[q] = 0;

                uint32_t a_n = srow[0], b_n0 = crow[0], b_n1 = crow[1];
#pragma unroll
                for (int i = 0; i < ROWS + (SYM ? 1 : 0); i++)
                {
                    // (the trip past ROWS only exists for the pairs' sake; a uniform branch around it, not a break: the
                    // loop has to unroll for hist[] / centre[] to stay in registers)
                    if (!(SYM && i == ROWS) || extra_row)
                    {
                    const uint32_t a = a_n;
                    const uint32_t bw = __builtin_amdgcn_alignbyte(b_n1, b_n0, sh);
                    if (i + 1 < ROWS || (SYM && i + 1 == ROWS && extra_row))
                    {
                        a_n = srow[(i + 1) * CPD];
                        b_n0 = crow[(i + 1) * CPD];
                        b_n1 = crow[(i + 1) * CPD + 1];
                    }
                    if (!PRE) centre[i % (NH + 1)] = bw;
#pragma unroll
                    for (int q = 0; q < PX; q++)
                    {
                        const int d = (int)byte_of(a, q) - (int)byte_of(bw, q);
                        C[q] += (uint32_t)(d * d);
                        if (i < RY - 1 + (SYM ? 1 : 0)) hist[i][q] = C[q];
                    }
                    if (i >= N - 1)
                    {
                        const int o = i - (N - 1);
                        uint32_t V[PX], pre[PX + 1], suf[PX + 1];
#pragma unroll
                        for (int q = 0; q < PX; q++) V[q] = o > 0 ? C[q] - hist[o > 0 ? o - 1 : 0][q] : C[q];
                        pre[0] = 0; suf[0] = 0;
#pragma unroll
                        for (int q = 0; q < PX; q++)
                        {
                            pre[q + 1] = pre[q] + V[q];
                            suf[q + 1] = suf[q] + V[PX - 1 - q];
                        }
#pragma unroll
                        for (int p = 0; p < PX; p++)
                        {
                            const int lo = p - NH, hi = p + NH;
                            uint32_t t;
                            if (lo <= 0 && hi >= PX - 1) t = pre[PX];
                            else if (lo <= 0) t = pre[hi + 1];
                            else if (hi >= PX - 1) t = suf[PX - lo];
                            else t = pre[hi + 1] - pre[lo];
                            // one VOP2 add per neighbour so that each folds its DPP move (an add3 cannot)
                            if (lo < 0) { t += from_lane_below(suf[-lo]); asm volatile("" : "+v"(t)); }
                            if (hi > PX - 1) { t += from_lane_above(pre[hi - (PX - 1)]); asm volatile("" : "+v"(t)); }
                            v[p] = t;
                        }
                    }
                    // The table reads issued for the previous output row have had this row's integer
                    // work to complete: fold them in now, then issue this row's.
                    if (i >= N && i < ROWS)
                    {
                        const int o = i - N;
                        const uint32_t pix = pixq;
#pragma unroll
                        for (int pp = 0; pp < PX / 2; pp++)
                        {
                            const f2 pv = {(float)(int)byte_of(pix, 2 * pp), (float)(int)byte_of(pix, 2 * pp + 1)};
                            aw[o][pp] += wq[pp];
                            ap[o][pp] += wq[pp] * pv;
                        }
                    }
                    if (i >= N - 1)
                    {
                        // byte offsets into the table: (diff * 4M) >> 32 = floor(4 * diff * wft), and clearing its two low
                        // bits gives 4 * floor(diff * wft); the mask also tells the compiler the offset is < 512, so the
                        // table's LDS address goes into the read's immediate offset
                        uint32_t offs[PX];
                        if (INTIDX)
                        {
#pragma unroll
                            for (int q = 0; q < PX; q++) offs[q] = __umulhi(min(v[q], (uint32_t)diff_cap), imul4) & 0x1fcu;
                            if (SYM && stash_on)                 // (uniform) four 7-bit indices per dword: byte q = offs[q] >> 2
                                stw[(i - (N - 1)) * (TXN * TYN)] = (offs[0] >> 2) | (offs[1] << 6) | (offs[2] << 14) | (offs[3] << 22);
                        }
                        if (!(SYM && i == ROWS))                  // (the row below the lane's last: only its indices were wanted)
                        {
                        // the pixels that go with these weights: the compare frame's row i - NH
                        if (PRE)
                        {
                            const int o = i - (N - 1);
                            pixq = __builtin_amdgcn_alignbyte(prow[o * CPD + 1], prow[o * CPD], sh);
                        }
                        else
                            pixq = centre[(i - NH) % (NH + 1)];
#pragma unroll
                        for (int pp = 0; pp < PX / 2; pp++)
                        {
                            if (INTIDX)
                                wq[pp] = f2{*reinterpret_cast<lds_cfloat *>(offs[2 * pp]), *reinterpret_cast<lds_cfloat *>(offs[2 * pp + 1])};
                            else
                            {
                                int idx[2];
                                if (FAST == 1)
                                {
                                    const f2 fd = {(float)(int)min(v[2 * pp], (uint32_t)diff_cap),
                                                   (float)(int)min(v[2 * pp + 1], (uint32_t)diff_cap)};
                                    const f2 fi = fd * wft2;
                                    idx[0] = (int)fi.x;
                                    idx[1] = (int)fi.y;
                                }
                                else
                                {
#pragma unroll
                                    for (int e = 0; e < 2; e++)
                                    {
                                        const int diff = (int)v[2 * pp + e];
                                        int ix = (int)((float)diff * wft);
                                        ix = diff < diff_max ? ix : 127;
                                        idx[e] = min(ix, 127);
                                    }
                                }
                                wq[pp] = f2{s_exp[idx[0]], s_exp[idx[1]]};
                            }
                        }
                        }
                    }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                {
                    // last output row of this displacement
                    const uint32_t pix = pixq;
#pragma unroll
                    for (int pp = 0; pp < PX / 2; pp++)
                    {
                        const f2 pv = {(float)(int)byte_of(pix, 2 * pp), (float)(int)byte_of(pix, 2 * pp + 1)};
                        aw[RY - 1][pp] += wq[pp];
                        ap[RY - 1][pp] += wq[pp] * pv;
                    }
                }
            }
        };
        for (int dy = -RH; dy <= RH; dy++)
            for (int dx = -RH; dx <= RH; dx++)
            {
                if (SYM && f == 0) displacement(std::true_type{}, dy, dx);
                else displacement(std::false_type{}, dy, dx);
            }
    }

    // normalise + store (the outer lanes own no pixels)
    const int x = tx0 + (tx - 1) * PX;
    if (tx == 0 || tx == TXN - 1 || x >= w) return;
#pragma unroll
    for (int o = 0; o < RY; o++)
    {
        const int y = ty0 + ty * RY + o;
        if (y >= h) break;
        uint32_t packed = 0;
        const uint32_t cpx = own_raw[(o + NH) * CPD];
#pragma unroll
        for (int p = 0; p < PX; p++)
        {
            const float q = ap[o][p / 2][p & 1] / aw[o][p / 2][p & 1];
            uint32_t r = (uint32_t)(int)q & 0xffu;
            if (r == 0) r = byte_of(cpx, p);
            packed |= r << (8 * p);
            __builtin_amdgcn_sched_barrier(0);
        }
        uint8_t *out = job.dst + (size_t)y * job.dst_pitch + x;
        if (x + 3 < w)
        {
            *reinterpret_cast<uint32_t *>(out) = packed;
        }
        else
        {
            for (int p = 0; p < PX && x + p < w; p++) out[p] = (uint8_t)(packed >> (8 * p));
        }
    }
}

// ------------------------------------------------------------------------------- 16-bit samples
// The same kernel for pixel = uint16_t (nlmeans_plane_16 etc., nlmeans.c:253-262; depth 10 or 12
// in 16-bit containers; prefilter = 0).  A lane still owns 4 pixels x 8 rows, now 2 dwords per
// row; tiles hold 2 pixels per dword.  Squared differences stay below 2^24 and the patch sums
// below 2^31 for depths <= 12, so the integer path is unchanged.
__device__ __forceinline__ void load_tile16(uint32_t *lds, int pitch, int dwords, int rows,
                                            const uint8_t *__restrict__ plane, int src_pitch_bytes,
                                            int w, int h, int x0, int y0)
{
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int rpp = (TXN * TYN) / dwords;
    const int r0 = (int)(((float)tid + 0.5f) * (1.0f / (float)dwords));
    const int c = tid - r0 * dwords;
    if (r0 >= rpp) return;
    const int x = x0 + 2 * c;                                  // first of the dword's two pixels
    const bool whole = x >= 0 && x + 1 < w;
    const int o0 = reflect(x, w), o1 = reflect(x + 1, w);
    // dword c of a row goes to the row's even half (c / 2) or odd half (pitch / 2 + c / 2): see nlmeans_lanes16_kernel
    uint32_t *out = lds + r0 * pitch + (c >> 1) + (c & 1) * (pitch >> 1);
#pragma nounroll
    for (int r = r0; r < rows; r += rpp, out += rpp * pitch)
    {
        const uint16_t *row = reinterpret_cast<const uint16_t *>(plane + (size_t)reflect(y0 + r, h) * src_pitch_bytes);
        uint32_t v;
        if (whole)
            v = reinterpret_cast<const u32_unaligned *>(row + x)->v;
        else
            v = (uint32_t)row[o0] | ((uint32_t)row[o1] << 16);
        *out = v;
    }
}

__device__ __forceinline__ uint32_t half_of(uint32_t v, int k) { return (v >> (16 * k)) & 0xffffu; }

// (a.lo - b.lo, a.hi - b.hi) as two int16, and c + d.lo^2 / c + d.hi^2 (the compiler extracts the halves with a v_bfe and
// a shift before a 24-bit multiply-add; the 16-bit multiply-add takes either half of its operands itself)
typedef short nlm_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (nlm_s2)(__builtin_bit_cast(nlm_s2, a) - __builtin_bit_cast(nlm_s2, b)));
}
__device__ __forceinline__ uint32_t sq_acc_lo(uint32_t d, uint32_t c)
{
    uint32_t r;
    asm("v_mad_i32_i16 %0, %1, %1, %2" : "=v"(r) : "v"(d), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t sq_acc_hi(uint32_t d, uint32_t c)
{
    uint32_t r;
    asm("v_mad_i32_i16 %0, %1, %1, %2 op_sel:[1,1,0,0]" : "=v"(r) : "v"(d), "v"(c));
    return r;
}

// FAST 0 / 1 / 2 / 3 as in the 8-bit kernel (2: the table index by an integer multiply-high, the table at LDS address 0;
// 3: 2, and inside frame 0 the patch distance of a displacement and of its mirror computed once - SYM_PAIRS of the four
// pairs: all four need 59 KB of LDS per workgroup (two workgroups per CU), the three with dy != 0 fit three)
template <int N, int FAST, int CPD, bool PRE, int SYM_PAIRS = 4>
__global__ __launch_bounds__(TXN * TYN, 3) void nlmeans_lanes16_kernel(const NlmJob *__restrict__ jobs, int njobs,
                                                                       int cmp_rows, int rq)
{
    constexpr int NH = N / 2;
    constexpr int ROWS = RY + N - 1;
    constexpr bool SYM = FAST == 3 && !PRE;
    static_assert(FAST != 3 || N <= 7, "the pairs' sharing reads a neighbouring lane's edge pixel: patch sizes up to 7");
    constexpr int ST_ROWS = RY + 1, ST_SLOT = ST_ROWS * TXN * TYN;        // the stash: [slot][row 0..RY][thread] dwords
    static_assert(NH <= PX, "patch must not reach past the adjacent lane");
    static_assert(CPD % 8 == 4, "the two tile rows of a wave must sit 32 banks apart");
    // A lane's four samples are two dwords, so a plain row layout has the lanes of a group read with a stride of two dwords:
    // under the 32-bank rule of ds_read_b32 / ds_read2_b32 lanes l and l + 16 then meet on a bank, and every row read of the
    // walk took twice its cycles (46 % of the kernel's LDS cycles were conflict cycles, and the LDS was what bound it).  A tile
    // row is therefore stored as its even dwords followed by its odd dwords (H = CPD / 2 each): the lanes of a group read
    // consecutive dwords of one half, whichever parity the compare window's offset has.
    constexpr int H = CPD / 2;

    extern __shared__ uint32_t smem[];
    const int tile_dwords = CPD * cmp_rows + 4;
    float *s_exp = reinterpret_cast<float *>(smem);             // the weight table first (see the 8-bit kernel)
    uint32_t *s_t0 = smem + 128;                                 // frame 0: source patches and compare tile of f = 0
    uint32_t *s_tc = s_t0 + tile_dwords;                         // frame f > 0
    // PRE (a prefilter is on, nlmeans.c:253-262 `_16` instantiation of nlmeans_template.c:428-543): distances between
    // src_pre (s_t0) and the prefiltered frame f (s_tc); the samples that are averaged come from the raw frames,
    // s_r0 (frame 0: also the origin term and the zero fallback) and s_rc (frame f > 0) - as in the 8-bit kernel
    uint32_t *s_r0 = PRE ? s_tc + tile_dwords : s_t0;
    uint32_t *s_rc = PRE ? s_r0 + tile_dwords : s_tc;
    // SYM: over the next frame's tile (free while f == 0) and beyond; slot = (dy + 1) * 3 + (dx + 1) of the pair's first:
    // 0..3, or with three pairs 0..2 (the (0, -1) slot, 3, is never touched)
    uint32_t *s_stash = s_tc;
    typedef __attribute__((address_space(3))) const float lds_cfloat;
    __builtin_assume(FAST < 2 || reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint32_t *)smem) == 0);   // nlm_no_static_lds

    int j = 0;
    for (int hi = njobs - 1; j < hi;)
    {
        const int mid = (j + hi + 1) >> 1;
        if ((int)blockIdx.x >= jobs[mid].tile_start) j = mid; else hi = mid - 1;
    }
    const NlmJob &job = jobs[j];
    const int tile = blockIdx.x - job.tile_start;
    const int tile_y = tile / job.tiles_x;
    int tile_x = tile - tile_y * job.tiles_x;
    // an even number of tiles per row (1920 / 120 = 16) would keep every column of tiles on one XCD, or on a few (workgroups
    // go to the XCDs round robin): each row of tiles starts one column further on instead, so that the tile below a tile is an
    // odd number of workgroups away (hbhip_internal.h: hbhip_grid_x has the numbers)
    if (HBHIP_GRID_ROTATE && (job.tiles_x & 1) == 0) { tile_x += tile_y % job.tiles_x; if (tile_x >= job.tiles_x) tile_x -= job.tiles_x; }
    const int tx0 = tile_x * LTW, ty0 = tile_y * TH;
    const int w = job.w, h = job.h;
    const int RH = job.r_half;

    const int tx = threadIdx.x & (TXN - 1);
    const int ty = threadIdx.x / TXN;

    if (threadIdx.x < 128) s_exp[threadIdx.x] = job.exptable[threadIdx.x];
    if (PRE)
    {
        load_tile16(s_t0, CPD, CPD, cmp_rows, job.src_pre, job.src_pre_pitch, w, h, tx0 - PX - 4 * rq, ty0 - NH - RH);
        load_tile16(s_r0, CPD, CPD, cmp_rows, job.frame[0], job.fpitch[0], w, h, tx0 - PX - 4 * rq, ty0 - NH - RH);
    }
    else
        load_tile16(s_t0, CPD, CPD, cmp_rows, job.frame[0], job.fpitch[0], w, h, tx0 - PX - 4 * rq, ty0 - NH - RH);
    // lane tx's own 4 pixels = dwords 2*tx + 2*rq, +1 of a tile row = entry tx + rq of the row's even and of its odd half
    const int own_off = (ty * RY + RH) * CPD + tx + rq;
    const uint32_t *own = s_t0 + own_off;
    const uint32_t *own_raw = s_r0 + own_off;                    // the same samples of the raw frame being filtered

    f2 aw[RY][PX / 2], ap[RY][PX / 2];
#pragma unroll
    for (int o = 0; o < RY; o++)
#pragma unroll
        for (int p = 0; p < PX / 2; p++) { aw[o][p] = f2{0.f, 0.f}; ap[o][p] = f2{0.f, 0.f}; }
    const f2 wft2 = {job.wft, job.wft};
    const float wft = job.wft;
    const int diff_max = job.diff_max;
    const int diff_cap = job.diff_cap;
    const uint32_t imul4 = job.imul4;
    const int ishift = job.ishift;
    const double origin_tune = job.origin_tune;
    const bool wave_live = ty0 + (ty & ~1) * RY < h;

    for (int f = 0; f < job.nframes; f++)
    {
        if (PRE || f > 0)
        {
            __syncthreads();
            if (PRE)
            {
                load_tile16(s_tc, CPD, CPD, cmp_rows, job.frame_pre[f], job.ppitch[f], w, h,
                            tx0 - PX - 4 * rq, ty0 - NH - RH);
                if (f > 0)
                    load_tile16(s_rc, CPD, CPD, cmp_rows, job.frame[f], job.fpitch[f], w, h,
                                tx0 - PX - 4 * rq, ty0 - NH - RH);
            }
            else
                load_tile16(s_tc, CPD, CPD, cmp_rows, job.frame[f], job.fpitch[f], w, h,
                            tx0 - PX - 4 * rq, ty0 - NH - RH);
        }
        __syncthreads();
        if (!wave_live) continue;
        const uint32_t *cmp_tile = (PRE || f > 0) ? s_tc : s_t0;
        const uint32_t *pix_tile = f > 0 ? s_rc : s_r0;          // what is averaged (== cmp_tile without a prefilter)

        // one displacement; F0: frame 0 of a launch that shares the pairs' distances (FAST 3, as in the 8-bit kernel: the first
        // of a pair (-dy, -dx) / (dy, dx) leaves the table index of every pixel of the lane in LDS - four 7-bit indices per
        // dword and row -, the second reads the index of its mirror one row down and / or one pixel across)
        auto displacement = [&](auto f0c, int dy, int dx) __attribute__((always_inline))
        {
            constexpr bool F0 = decltype(f0c)::value;
            if (f == 0 && dx == 0 && dy == 0)
            {
#pragma unroll
                for (int o = 0; o < RY; o++)
                {
                    const uint32_t c0 = own_raw[(o + NH) * CPD], c1 = own_raw[(o + NH) * CPD + H];
#pragma unroll
                    for (int p = 0; p < PX; p++)
                    {
                        const int sv = (int)half_of(p < 2 ? c0 : c1, p & 1);
                        aw[o][p / 2][p & 1] = (float)((double)aw[o][p / 2][p & 1] + origin_tune);
                        ap[o][p / 2][p & 1] = (float)((double)ap[o][p / 2][p & 1] + origin_tune * (double)sv);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                return;
            }

            const int s = dx + 4 * rq;                   // pixel offset of the compare window, >= 0, uniform
            const int sh = 2 * (s & 1);                  // byte shift inside the dword pair
            // the three dwords 2 tx + m .. + 2 (m = s / 2) of a row, each in its half: entry e of half (m + k) & 1
            const int m = s >> 1;
            const int e0 = (m >> 1) + (m & 1) * H, e1 = ((m + 1) >> 1) + ((m + 1) & 1) * H, e2 = ((m + 2) >> 1) + (m & 1) * H;
            const uint32_t *prow = pix_tile + (ty * RY + dy + RH + NH) * CPD + tx;   // the row whose pixels are averaged
            const uint32_t *prow0 = prow + e0, *prow1 = prow + e1, *prow2 = prow + e2;
            // which pairs share: all four, or (SYM_PAIRS == 3) the three with dy != 0 - their stash then fits beside the
            // tiles of three workgroups per CU
            const bool shared_pair = SYM_PAIRS == 4 || dy != 0;
            if (SYM && F0 && shared_pair && (dy > 0 || (dy == 0 && dx > 0)))
            {
                const uint32_t *st = s_stash + ((-dy + 1) * 3 + (-dx + 1)) * ST_SLOT + dy * (TXN * TYN) + (int)threadIdx.x;
#pragma unroll
                for (int o = 0; o < RY; o++)
                {
                    const uint32_t w0 = st[o * (TXN * TYN)];
                    uint32_t id4 = w0;
                    if (dx > 0) id4 = __builtin_amdgcn_alignbyte(st[o * (TXN * TYN) + 1], w0, 1);
                    else if (dx < 0) id4 = __builtin_amdgcn_alignbyte(w0, st[o * (TXN * TYN) - 1], 3);
                    const uint32_t p0 = prow0[o * CPD], p1 = prow1[o * CPD], p2 = prow2[o * CPD];
                    const uint32_t pix0 = __builtin_amdgcn_alignbyte(p1, p0, sh), pix1 = __builtin_amdgcn_alignbyte(p2, p1, sh);
                    const uint32_t o0 = (id4 << 2) & 0x1fcu, o1 = (id4 >> 6) & 0x1fcu, o2 = (id4 >> 14) & 0x1fcu, o3 = (id4 >> 22) & 0x1fcu;
                    const f2 wa = {*reinterpret_cast<lds_cfloat *>(o0), *reinterpret_cast<lds_cfloat *>(o1)};
                    const f2 wb = {*reinterpret_cast<lds_cfloat *>(o2), *reinterpret_cast<lds_cfloat *>(o3)};
                    const f2 pa = {(float)(int)half_of(pix0, 0), (float)(int)half_of(pix0, 1)};
                    const f2 pb = {(float)(int)half_of(pix1, 0), (float)(int)half_of(pix1, 1)};
                    aw[o][0] += wa; ap[o][0] += wa * pa;
                    aw[o][1] += wb; ap[o][1] += wb * pb;
                }
                return;
            }
            constexpr bool stash_can = SYM && F0;            // (here: dy < 0, or dy == 0 && dx < 0)
            const bool stash_on = stash_can && shared_pair;
            const bool extra_row = stash_on && dy < 0;
            uint32_t *stw = s_stash + ((dy + 1) * 3 + (dx + 1)) * ST_SLOT + (int)threadIdx.x;
            const uint32_t *srow = own;
            const uint32_t *crow = cmp_tile + (ty * RY + dy + RH) * CPD + tx;
            const uint32_t *crow0 = crow + e0, *crow1 = crow + e1, *crow2 = crow + e2;

            uint32_t C[PX], hist[RY - 1 + (SYM ? 1 : 0)][PX], v[PX];
            uint32_t centre0[NH + 1], centre1[NH + 1];           // !PRE: the compare window's dwords of the last NH + 1 rows
            uint32_t pixq0 = 0, pixq1 = 0;
            f2 wq[PX / 2];
#pragma unroll
            for (int q = 0; q < PX; q++) C[q] = 0;

            uint32_t a_n0 = srow[0], a_n1 = srow[H], b_n0 = crow0[0], b_n1 = crow1[0], b_n2 = crow2[0];
#pragma unroll
            for (int i = 0; i < ROWS + (SYM ? 1 : 0); i++)
            {
                // (the trip past ROWS only exists for the pairs' sake: a uniform branch around it, not a break - the loop has
                // to unroll for hist[] to stay in registers)
                if (!(SYM && i == ROWS) || extra_row)
                {
                const uint32_t a0 = a_n0, a1 = a_n1;
                const uint32_t bw0 = __builtin_amdgcn_alignbyte(b_n1, b_n0, sh);
                const uint32_t bw1 = __builtin_amdgcn_alignbyte(b_n2, b_n1, sh);
                if (i + 1 < ROWS || (SYM && i + 1 == ROWS && extra_row))
                {
                    a_n0 = srow[(i + 1) * CPD]; a_n1 = srow[(i + 1) * CPD + H];
                    b_n0 = crow0[(i + 1) * CPD]; b_n1 = crow1[(i + 1) * CPD]; b_n2 = crow2[(i + 1) * CPD];
                }
                if (!PRE) { centre0[i % (NH + 1)] = bw0; centre1[i % (NH + 1)] = bw1; }
                // two differences per v_pk_sub_i16 (samples < 2^15), then C += d * d on either half of the pair by
                // v_mad_i32_i16 with op_sel: six instructions per row of four pixels instead of eight
                const uint32_t d01 = pk_sub_i16(a0, bw0), d23 = pk_sub_i16(a1, bw1);
                C[0] = sq_acc_lo(d01, C[0]); C[1] = sq_acc_hi(d01, C[1]);
                C[2] = sq_acc_lo(d23, C[2]); C[3] = sq_acc_hi(d23, C[3]);
                if (i < RY - 1 + (SYM ? 1 : 0))
                {
#pragma unroll
                    for (int q = 0; q < PX; q++) hist[i][q] = C[q];
                }
                if (i >= N - 1)
                {
                    const int o = i - (N - 1);
                    uint32_t V[PX], pre[PX + 1], suf[PX + 1];
#pragma unroll
                    for (int q = 0; q < PX; q++) V[q] = o > 0 ? C[q] - hist[o > 0 ? o - 1 : 0][q] : C[q];
                    pre[0] = 0; suf[0] = 0;
#pragma unroll
                    for (int q = 0; q < PX; q++)
                    {
                        pre[q + 1] = pre[q] + V[q];
                        suf[q + 1] = suf[q] + V[PX - 1 - q];
                    }
#pragma unroll
                    for (int p = 0; p < PX; p++)
                    {
                        const int lo = p - NH, hi = p + NH;
                        uint32_t t;
                        if (lo <= 0 && hi >= PX - 1) t = pre[PX];
                        else if (lo <= 0) t = pre[hi + 1];
                        else if (hi >= PX - 1) t = suf[PX - lo];
                        else t = pre[hi + 1] - pre[lo];
                        if (lo < 0) { t += from_lane_below(suf[-lo]); asm volatile("" : "+v"(t)); }
                        if (hi > PX - 1) { t += from_lane_above(pre[hi - (PX - 1)]); asm volatile("" : "+v"(t)); }
                        v[p] = t;
                    }
                }
                if (i >= N && i < ROWS)
                {
                    const int o = i - N;
#pragma unroll
                    for (int pp = 0; pp < PX / 2; pp++)
                    {
                        const uint32_t pix = pp ? pixq1 : pixq0;
                        const f2 pv = {(float)(int)half_of(pix, 0), (float)(int)half_of(pix, 1)};
                        aw[o][pp] += wq[pp];
                        ap[o][pp] += wq[pp] * pv;
                    }
                }
                if (i >= N - 1)
                {
                    const int o = i - (N - 1);
                    uint32_t offs[PX];
                    if (FAST >= 2)
                    {
#pragma unroll
                        for (int q = 0; q < PX; q++) offs[q] = (__umulhi(min(v[q], (uint32_t)diff_cap), imul4) >> ishift) & 0x1fcu;
                        if (stash_can && stash_on)           // (uniform) four 7-bit indices per dword: byte q = offs[q] >> 2
                            stw[o * (TXN * TYN)] = (offs[0] >> 2) | (offs[1] << 6) | (offs[2] << 14) | (offs[3] << 22);
                    }
                    if (!(SYM && i == ROWS))                  // (the row below the lane's last: only its indices were wanted)
                    {
                    // the samples that go with these weights: the compare frame's row i - NH - without a prefilter the
                    // dwords the walk had in its hands NH rows ago
                    if (PRE)
                    {
                        const uint32_t p0 = prow0[o * CPD], p1 = prow1[o * CPD], p2 = prow2[o * CPD];
                        pixq0 = __builtin_amdgcn_alignbyte(p1, p0, sh);
                        pixq1 = __builtin_amdgcn_alignbyte(p2, p1, sh);
                    }
                    else
                    {
                        pixq0 = centre0[(i - NH) % (NH + 1)];
                        pixq1 = centre1[(i - NH) % (NH + 1)];
                    }
#pragma unroll
                    for (int pp = 0; pp < PX / 2; pp++)
                    {
                        if (FAST >= 2)
                        {
                            wq[pp] = f2{*reinterpret_cast<lds_cfloat *>(offs[2 * pp]), *reinterpret_cast<lds_cfloat *>(offs[2 * pp + 1])};
                            continue;
                        }
                        int idx[2];
                        if (FAST == 1)
                        {
                            const f2 fd = {(float)(int)min(v[2 * pp], (uint32_t)diff_cap),
                                           (float)(int)min(v[2 * pp + 1], (uint32_t)diff_cap)};
                            const f2 fi = fd * wft2;
                            idx[0] = (int)fi.x;
                            idx[1] = (int)fi.y;
                        }
                        else
                        {
#pragma unroll
                            for (int e = 0; e < 2; e++)
                            {
                                const int diff = (int)v[2 * pp + e];
                                int ix = (int)((float)diff * wft);
                                ix = diff < diff_max ? ix : 127;
                                idx[e] = min(ix, 127);
                            }
                        }
                        wq[pp] = f2{s_exp[idx[0]], s_exp[idx[1]]};
                    }
                    }
                }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int pp = 0; pp < PX / 2; pp++)
            {
                const uint32_t pix = pp ? pixq1 : pixq0;
                const f2 pv = {(float)(int)half_of(pix, 0), (float)(int)half_of(pix, 1)};
                aw[RY - 1][pp] += wq[pp];
                ap[RY - 1][pp] += wq[pp] * pv;
            }
        };
        for (int dy = -RH; dy <= RH; dy++)
            for (int dx = -RH; dx <= RH; dx++)
            {
                if (SYM && f == 0) displacement(std::true_type{}, dy, dx);
                else displacement(std::false_type{}, dy, dx);
            }
    }

    const int x = tx0 + (tx - 1) * PX;
    if (tx == 0 || tx == TXN - 1 || x >= w) return;
#pragma unroll
    for (int o = 0; o < RY; o++)
    {
        const int y = ty0 + ty * RY + o;
        if (y >= h) break;
        const uint32_t c0 = own_raw[(o + NH) * CPD], c1 = own_raw[(o + NH) * CPD + H];
        uint32_t r[PX];
#pragma unroll
        for (int p = 0; p < PX; p++)
        {
            const float q = ap[o][p / 2][p & 1] / aw[o][p / 2][p & 1];
            uint32_t rv = (uint32_t)(int)q & 0xffffu;
            if (rv == 0) rv = half_of(p < 2 ? c0 : c1, p & 1);
            r[p] = rv;
            __builtin_amdgcn_sched_barrier(0);
        }
        uint16_t *out = reinterpret_cast<uint16_t *>(job.dst + (size_t)y * job.dst_pitch) + x;
        if (x + 3 < w)
        {
            reinterpret_cast<uint32_t *>(out)[0] = r[0] | (r[1] << 16);
            reinterpret_cast<uint32_t *>(out)[1] = r[2] | (r[3] << 16);
        }
        else
        {
            for (int p = 0; p < PX && x + p < w; p++) out[p] = (uint16_t)r[p];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The generic form: every patch size and search range the reference takes (nlmeans.c:284-334 keeps any odd patch size
// >= 1 and any odd range; what bounds them is the 16-pixel mirrored border, nlmeans.c:529) that the lane-sharing kernels
// above have no instantiation for - patch 1, 11, 13, 15, ... - at 8 and 10 / 12 bits, with or without a prefilter.  Not
// tuned: it exists so that no valid setting drops a job to the CPU filter (VERDICT r05 "missing" 3), and as a second
// implementation the tuned kernels can be held against (tests/test_nlmeans_gpu.py runs patch 7 through both).
// A workgroup of 32 x 8 threads owns 32 x 8 pixels, one each.  Per displacement the squared differences are summed along
// x first (one n-tap row sum per pixel of the 8 + n - 1 rows the workgroup's patches cover, through LDS), then n row
// sums down each pixel's column: 2 n operations per pixel and displacement instead of n * n.  The same integers as the
// reference's integral image gives (nlmeans_template.c:545-591, :672-675), the same float sequence per pixel (:677-686).
constexpr int GW = 32, GH = 8;

template <typename PIX>
__device__ __forceinline__ void generic_stage(PIX *lds, int tw, int th, const uint8_t *__restrict__ plane, int pitch,
                                              int w, int h, int x0, int y0)
{
    for (int i = threadIdx.x; i < tw * th; i += GW * GH)
    {
        const int r = i / tw, c = i - r * tw;
        lds[i] = reinterpret_cast<const PIX *>(plane + (size_t)reflect(y0 + r, h) * pitch)[reflect(x0 + c, w)];
    }
}

template <typename PIX>
__global__ __launch_bounds__(GW * GH) void nlmeans_generic_kernel(const NlmJob *__restrict__ jobs, int njobs, int n, int max_rh, int pre)
{
    extern __shared__ uint32_t gsm[];
    const int nh = n / 2;
    const int aw_ = GW + 2 * nh, ah_ = GH + 2 * nh;                    // source-patch tile
    const int bw_ = aw_ + 2 * max_rh, bh_ = ah_ + 2 * max_rh;          // compare tile: the search halo around it
    uint32_t *rs = gsm;                                                // row sums: ah_ rows of GW
    PIX *ta = reinterpret_cast<PIX *>(rs + ah_ * GW);
    PIX *tb = ta + ((aw_ * ah_ + 1) & ~1);

    int j = 0;
    for (int hi = njobs - 1; j < hi;)
    {
        const int mid = (j + hi + 1) >> 1;
        if ((int)blockIdx.x >= jobs[mid].tile_start) j = mid; else hi = mid - 1;
    }
    const NlmJob &job = jobs[j];
    const int tile = blockIdx.x - job.tile_start;
    const int tile_y = tile / job.tiles_x, tile_x = tile - tile_y * job.tiles_x;
    const int tx0 = tile_x * GW, ty0 = tile_y * GH;
    const int w = job.w, h = job.h, RH = job.r_half;
    const int lx = threadIdx.x % GW, ly = threadIdx.x / GW;
    const int x = tx0 + lx, y = ty0 + ly;
    const bool live = x < w && y < h;

    generic_stage<PIX>(ta, aw_, ah_, pre ? job.src_pre : job.frame[0], pre ? job.src_pre_pitch : job.fpitch[0], w, h, tx0 - nh, ty0 - nh);
    const uint8_t *raw0 = job.frame[0];
    const int own = live ? (int)reinterpret_cast<const PIX *>(raw0 + (size_t)y * job.fpitch[0])[x] : 0;
    float wsum = 0.f, psum = 0.f;
    for (int f = 0; f < job.nframes; f++)
    {
        __syncthreads();                                               // the previous frame's compare tile is done with
        generic_stage<PIX>(tb, bw_, bh_, pre ? job.frame_pre[f] : job.frame[f], pre ? job.ppitch[f] : job.fpitch[f], w, h,
                           tx0 - nh - max_rh, ty0 - nh - max_rh);
        __syncthreads();
        for (int dy = -RH; dy <= RH; dy++)
            for (int dx = -RH; dx <= RH; dx++)
            {
                if (f == 0 && dx == 0 && dy == 0)
                {
                    wsum = (float)((double)wsum + job.origin_tune);                       // nlmeans_template.c:644-655
                    psum = (float)((double)psum + job.origin_tune * (double)own);
                    continue;
                }
                for (int i = threadIdx.x; i < ah_ * GW; i += GW * GH)
                {
                    const int r = i / GW, c = i - r * GW;
                    const PIX *pa = ta + r * aw_ + c;
                    const PIX *pb = tb + (r + dy + max_rh) * bw_ + c + dx + max_rh;
                    uint32_t acc = 0;
                    for (int k = 0; k < n; k++)
                    {
                        const int d = (int)pa[k] - (int)pb[k];
                        acc += (uint32_t)(d * d);
                    }
                    rs[i] = acc;
                }
                __syncthreads();
                uint32_t du = 0;
                for (int k = 0; k < n; k++) du += rs[(ly + k) * GW + lx];
                __syncthreads();                                       // (the next displacement rewrites rs)
                const int diff = (int)du;                              // :672-675
                if (live && diff < job.diff_max)
                {
                    const int idx = (int)((float)diff * job.wft);
                    const float wgt = job.exptable[min(max(idx, 0), 127)];
                    const int cmp = (int)reinterpret_cast<const PIX *>(job.frame[f] + (size_t)reflect(y + dy, h) * job.fpitch[f])[reflect(x + dx, w)];
                    wsum += wgt;
                    psum += wgt * (float)cmp;
                }
            }
    }
    if (!live) return;
    const float q = psum / wsum;
    PIX r = (PIX)(int)q;                                               // :710-711
    if (r == 0) r = (PIX)own;
    reinterpret_cast<PIX *>(job.dst + (size_t)y * job.dst_pitch)[x] = r;
}

// ------------------------------------------------------------------------------- prefilters
// nlmeans_prefilter (nlmeans_template.c:428-543) on one plane.  The reference filters the bordered
// image and re-mirrors the result, so a prefiltered plane is fully described by its w x h interior;
// reads outside the plane mirror (the 16-pixel border is wider than any window here).
constexpr int PF_MEAN3 = 1, PF_MEAN5 = 2, PF_MEDIAN3 = 4, PF_MEDIAN5 = 8, PF_CSM3 = 16, PF_CSM5 = 32,
              PF_REDUCE25 = 256, PF_REDUCE50 = 512, PF_EDGEBOOST = 1024, PF_PASSTHRU = 2048,
              PF_BASE = PF_MEAN3 | PF_MEAN5 | PF_MEDIAN3 | PF_MEDIAN5 | PF_CSM3 | PF_CSM5;

// PIX = uint8_t, or uint16_t for the `_16` instantiation (nlmeans.c:253-262: pixel = uint16_t, pixel_2 = uint32_t);
// pitches are in BYTES.  What the wider types change is noted where it matters (oracle/nlmeans_prefilter16.h).
template <typename PIX>
__device__ __forceinline__ int pf_mix(int pre, int src, int wet, int dry)
{
    return dry > 0 ? (int)(PIX)((wet * pre + dry * src) / (wet + dry)) : pre;     // :498-525
}

template <typename PIX>
__device__ __forceinline__ const PIX *pf_row(const uint8_t *base, int pitch, int y)
{
    return reinterpret_cast<const PIX *>(base + (size_t)y * pitch);
}

// base filter (+ the wet/dry blend when no edge boost comes in between)
template <typename PIX>
__global__ __launch_bounds__(256) void nlm_prefilter_kernel(const uint8_t *__restrict__ src, int spitch,
                                                            uint8_t *__restrict__ dst, int dpitch,
                                                            int w, int h, int type, int wet, int dry)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    auto at = [&](int xx, int yy) -> int { return pf_row<PIX>(src, spitch, reflect(yy, h))[reflect(xx, w)]; };
    const int centre = at(x, y);
    int out = centre;
    // one base filter, picked in the reference's order of precedence (:459-496)
    const int kind = (type & PF_CSM5) ? PF_CSM5 : (type & PF_CSM3) ? PF_CSM3 : (type & PF_MEDIAN5) ? PF_MEDIAN5
                   : (type & PF_MEDIAN3) ? PF_MEDIAN3 : (type & PF_MEAN5) ? PF_MEAN5 : PF_MEAN3;
    const int size = (kind & (PF_CSM5 | PF_MEDIAN5 | PF_MEAN5)) ? 5 : 3;
    const int lo = -((size - 1) / 2), hi = (size + 1) / 2;
    if (kind & (PF_CSM5 | PF_CSM3))
    {
        // nlmeans_filter_csm (:232-323): the reference leaves its row loop with `goto end` at the
        // first neighbour and at the origin, so the first column contributes only its top pixel
        // and the centre column only the pixels above the origin.
        int vmin = 0, vmax = 0;
        for (int dx = lo; dx < hi; dx++)
            for (int dy = lo; dy < hi; dy++)
            {
                if (dx == 0 && dy == 0) break;
                const int v = at(x + dx, y + dy);
                if (dx == lo && dy == lo) { vmin = vmax = v; break; }
                vmin = min(vmin, v);
                vmax = max(vmax, v);
            }
        const int mid = (vmin + vmax) / 2;
        const int min2 = (vmin + mid) / 2, max2 = (vmax + mid) / 2;
        const int min3 = (min2 + mid) / 2, max3 = (max2 + mid) / 2;
        if      (centre < vmin) out = vmin;
        else if (centre > vmax) out = vmax;
        else if (centre < min2) out = min2;
        else if (centre > max2) out = max2;
        else if (centre < min3) out = min3;
        else if (centre > max3) out = max3;
    }
    else if (kind & (PF_MEDIAN5 | PF_MEDIAN3))
    {
        // nlmeans_filter_median (:135-230): the networks return the true median = the element of
        // rank n/2; found here by counting, for every candidate, the elements that sort before it.
        const int n = size * size;
        for (int i = 0; i < n; i++)
        {
            const int vi = at(x + lo + i / size, y + lo + i % size);
            int rank = 0;
            for (int j = 0; j < n; j++)
            {
                const int vj = at(x + lo + j / size, y + lo + j % size);
                rank += (vj < vi) || (vj == vi && j < i);
            }
            if (rank == n / 2) out = vi;
        }
    }
    else
    {
        // nlmeans_filter_mean (:103-133): window sum in pixel_2 (uint16 / uint32) scaled by a double, truncated
        int sum = 0;
        for (int dx = lo; dx < hi; dx++)
            for (int dy = lo; dy < hi; dy++) sum += at(x + dx, y + dy);
        if (sizeof(PIX) == 1) out = (int)(uint8_t)((double)(sum & 0xffff) * (1.0 / (double)(size * size)));
        else                  out = (int)(uint16_t)((double)(uint32_t)sum * (1.0 / (double)(size * size)));
    }
    if (!(type & PF_EDGEBOOST)) out = pf_mix<PIX>(out, centre, wet, dry);
    reinterpret_cast<PIX *>(dst + (size_t)y * dpitch)[x] = (PIX)out;
}

// nlmeans_filter_edgeboost, first pass (:335-377): Sobel-like gradients in pixel_2 = uint16 / uint32 (negative
// sums wrap, as in the reference), classified into {16, 128, 235} - thresholds NOT scaled with the depth
// (:367-378), so on 10 / 12-bit data nearly every sample classifies as a strong edge.  The mask is a byte plane.
template <typename PIX>
__global__ __launch_bounds__(256) void nlm_edge_mask_kernel(const uint8_t *__restrict__ src, int spitch,
                                                            uint8_t *__restrict__ mask, int mpitch, int w, int h)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    auto at = [&](int xx, int yy) -> int { return pf_row<PIX>(src, spitch, reflect(yy, h))[reflect(xx, w)]; };
    const int kern[3][3] = { {-31, 0, 31}, {-44, 0, 44}, {-31, 0, 31} };
    int g1 = 0, g2 = 0;
#pragma unroll
    for (int dx = -1; dx <= 1; dx++)
#pragma unroll
        for (int dy = -1; dy <= 1; dy++)
        {
            const int v = at(x + dx, y + dy);
            g1 += kern[dy + 1][dx + 1] * v;
            g2 += kern[dx + 1][dy + 1] * v;
        }
    const double coef = 1.0 / 126.42;
    uint32_t m;
    if (sizeof(PIX) == 1)
    {
        const uint32_t a = (uint32_t)(uint16_t)(int)(((double)(g1 & 0xffff) * coef) + 128);
        const uint32_t b = (uint32_t)(uint16_t)(int)(((double)(g2 & 0xffff) * coef) + 128);
        m = (a + b) & 0xff;
    }
    else
    {
        const uint32_t a = (uint32_t)(((double)(uint32_t)g1 * coef) + 128);
        const uint32_t b = (uint32_t)(((double)(uint32_t)g2 * coef) + 128);
        m = (a + b) & 0xffff;
    }
    mask[(size_t)y * mpitch + x] = m > 160 ? 235 : m > 16 ? 128 : 16;
}

// nlmeans_filter_edgeboost, second pass (:379-424) + the wet/dry blend.  The reference edits the
// mask in place in raster order, so whether a pixel counts as "edge" for its right and lower
// neighbours depends on whether it was itself demoted just before: a chain through the whole
// plane.  One workgroup walks the rows in order; within a row the only unknown of a pixel is
// whether its left neighbour survived, so each pixel is a map {left dead, left alive} -> {dead,
// alive}, and the row is resolved with a prefix composition of those maps (threads compose their
// own run of pixels, then a block scan joins the runs).
constexpr int EB_THREADS = 256;
template <typename PIX>
__global__ __launch_bounds__(EB_THREADS) void nlm_edge_apply_kernel(const uint8_t *__restrict__ src, int spitch,
                                                                     uint8_t *__restrict__ mask, int mpitch,
                                                                     uint8_t *__restrict__ pre, int ppitch,
                                                                     int w, int h)
{
    __shared__ uint8_t s_map[EB_THREADS];      // bit0 = outcome if the run's left neighbour is dead, bit1 = if alive
    __shared__ uint8_t s_in[EB_THREADS];       // resolved state entering each thread's run
    const int t = threadIdx.x;
    const int per = (w + EB_THREADS - 1) / EB_THREADS;
    const int x0 = t * per, x1 = min(w, x0 + per);
    for (int y = 0; y < h; y++)
    {
        uint8_t *mrow = mask + (size_t)y * mpitch;
        const uint8_t *mup = mrow - mpitch, *mdn = mrow + mpitch;
        // everything but the left neighbour: the row above is final, this row's right neighbour
        // and the row below still hold their first-pass values; outside the plane counts as 0
        // the pixel right of this thread's run belongs to the next thread, which may demote it
        // while this thread still needs its first-pass value
        const int right_of_run = x1 < w ? mrow[x1] : 0;
        auto others = [&](int x) -> int {
            int c = 1;                                                      // the pixel itself (> 16, else no test)
            if (x + 1 < w) c += (x + 1 == x1 ? right_of_run : (int)mrow[x + 1]) > 16;
            if (y > 0)     c += (x > 0 && mup[x - 1] > 16) + (mup[x] > 16) + (x + 1 < w && mup[x + 1] > 16);
            if (y + 1 < h) c += (x > 0 && mdn[x - 1] > 16) + (mdn[x] > 16) + (x + 1 < w && mdn[x + 1] > 16);
            return c;
        };
        // compose this thread's run: state s = "pixel x-1 is an edge after its own test"
        int f0 = 0, f1 = 1;                                                 // identity map
        for (int x = x0; x < x1; x++)
        {
            int g0, g1;
            if (mrow[x] <= 16) { g0 = 0; g1 = 0; }
            else { const int c = others(x); g0 = c >= 3; g1 = c + 1 >= 3; }
            f0 = f0 ? g1 : g0;
            f1 = f1 ? g1 : g0;
        }
        s_map[t] = (uint8_t)(f0 | (f1 << 1));
        __syncthreads();
        if (t == 0)
        {
            int state = 0;                                                  // nothing left of column 0
            for (int i = 0; i < EB_THREADS; i++)
            {
                s_in[i] = (uint8_t)state;
                state = (s_map[i] >> state) & 1;
            }
        }
        __syncthreads();
        int state = s_in[t];
        for (int x = x0; x < x1; x++)
        {
            const int m = mrow[x];
            if (m <= 16) { state = 0; continue; }
            const int c = others(x) + state;
            if (c < 3)
            {
                mrow[x] = 16;
                state = 0;
                continue;
            }
            state = 1;
            const int sv = pf_row<PIX>(src, spitch, y)[x];
            PIX *o = reinterpret_cast<PIX *>(pre + (size_t)y * ppitch) + x;
            *o = (PIX)(m == 235 ? (3 * sv + (int)*o) / 4 : (2 * sv + 3 * (int)*o) / 5);
        }
        __threadfence_block();
        __syncthreads();                                                    // row y is final before row y+1 reads it
    }
}

// the wet/dry blend (:498-525) when the edge boost had to run between it and the base filter
template <typename PIX>
__global__ __launch_bounds__(256) void nlm_mix_kernel(const uint8_t *__restrict__ src, int spitch,
                                                      uint8_t *__restrict__ pre, int ppitch, int w, int h, int wet, int dry)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    PIX *o = reinterpret_cast<PIX *>(pre + (size_t)y * ppitch) + x;
    *o = (PIX)pf_mix<PIX>(*o, pf_row<PIX>(src, spitch, y)[x], wet, dry);
}

__global__ void copy_plane_kernel(uint8_t *dst, int dst_pitch, const uint8_t *src, int src_pitch,
                                  int row_bytes, int rows)
{
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y;
    if (y >= rows || x >= row_bytes) return;
    const uint8_t *s = src + (size_t)y * src_pitch + x;
    uint8_t *d = dst + (size_t)y * dst_pitch + x;
    if (x + 3 < row_bytes)
        *reinterpret_cast<uint32_t *>(d) = *reinterpret_cast<const uint32_t *>(s);
    else
        for (int i = 0; x + i < row_bytes; i++) d[i] = s[i];
}

// ------------------------------------------------------------------- host side
class NlmFilter : public hbhip_filter
{
public:
    NlmFilter(hbhip_ctx *c, const hbhip_nlmeans_params &p) : hbhip_filter(c), par(p) {}
    ~NlmFilter() override
    {
        if (d_exp) (void)hipFree(d_exp);
        if (d_jobs) (void)hipFree(d_jobs);
        if (h_jobs) (void)hipHostFree(h_jobs);
        for (int i = 0; i < NTABLES; i++)
            if (table_ev[i]) (void)hipEventDestroy(table_ev[i]);
    }

    int setup(int width, int height, int depth, int lcw, int lch)
    {
        in_geo.set(width, height, depth, lcw, lch);
        out_geo = in_geo;
        pool.configure(ctx, in_geo);
        max_frames = 1;
        for (int c = 0; c < 3; c++)
        {
            // a prefilter only exists if one of the base filters is selected (:438-443)
            pf_type[c] = (par.prefilter[c] & PF_BASE) ? par.prefilter[c] : 0;
            passthru[c] = (par.prefilter[c] & PF_PASSTHRU) != 0;
            any_pre |= pf_type[c] != 0;
            if (passthru[c]) continue;                 // the plane is not denoised at all (nlmeans.c:485-492)
            max_frames = std::max(max_frames, par.nframes[c]);
            if (par.strength[c] == 0) continue;
            const int n = par.patch_size[c];
            if (n < 1 || (n & 1) == 0) return HBHIP_ERR_ARG;                        // (the drop-in sanitises as nlmeans.c:329-330)
            if (n / 2 + par.range[c] / 2 > NLM_BORDER) return HBHIP_ERR_UNSUPPORTED;  // past the reference's own mirrored border
            if (par.nframes[c] < 1 || par.nframes[c] > HBHIP_NLMEANS_FRAMES_MAX) return HBHIP_ERR_ARG;
            if (in_geo.pw[c] < NLM_BORDER || in_geo.ph[c] < NLM_BORDER) return HBHIP_ERR_UNSUPPORTED;
        }
        for (int c = 0; c < 3; c++)
        {
            // smallest diff whose index reaches 127; usable only if that index is exactly 127,
            // it does not exceed diff_max, and the table really ends in 0
            diff_cap[c] = -1;
            const float wft = par.weight_fact_table[c];
            if (par.strength[c] == 0 || par.exptable[c][127] != 0.f || !(wft > 0.f)) continue;
            for (int d = 0; d <= par.diff_max[c]; d++)
                if ((int)((float)d * wft) >= 127)
                {
                    if ((int)((float)d * wft) == 127) diff_cap[c] = d;
                    break;
                }
            // the integer form of the index (FAST 2): wft is a float, so wft * 2^(34 + s) is an integer M for some small s;
            // usable when M fits 32 bits and (d * M) >> (34 + s) equals the float expression for every d up to the cap.
            // The kernels take (mul_hi(d, M) >> s) & 0x1fc = 4 * index as the table's byte offset.
            imul4[c] = 0;
            ishift[c] = 0;
            if (diff_cap[c] >= 0)
                for (int sft = 0; sft <= 12; sft++)
                {
                    const double m = std::ldexp((double)wft, 34 + sft);
                    if (!(m > 0.0 && m < 4294967296.0)) break;
                    if (m != std::floor(m)) continue;
                    const uint64_t M = (uint64_t)m;
                    bool same = true;
                    for (int d = 0; d <= diff_cap[c] && same; d++)
                        same = (int)(((uint64_t)d * M) >> (34 + sft)) == (int)((float)d * wft);
                    if (same) { imul4[c] = (uint32_t)M; ishift[c] = sft; }
                    break;
                }
        }
        if (any_pre)
        {
            pre_pool.configure(ctx, in_geo);
            if (par.prefilter[0] & PF_EDGEBOOST || par.prefilter[1] & PF_EDGEBOOST || par.prefilter[2] & PF_EDGEBOOST)
            {
                mask_pic = pre_pool.acquire();
                if (!mask_pic) return HBHIP_ERR_NOMEM;
            }
        }
        HBHIP_CHECK(ctx, hipMalloc((void **)&d_exp, sizeof(float) * 3 * 128));
        HBHIP_CHECK(ctx, hipMemcpyAsync(d_exp, par.exptable, sizeof(float) * 3 * 128,
                                        hipMemcpyHostToDevice, ctx->stream));
        HBHIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        return HBHIP_OK;
    }

    DevPicture *acquire_input() override { return pool.acquire(); }
    int use_frames() override { pool.use_frames(true); frames_mode = true; return HBHIP_OK; }

    int submit(DevPicture *pic) override
    {
        if (any_pre)
        {
            int rc = prefilter_frame(pic);
            if (rc != HBHIP_OK) return rc;
        }
        in.push_back(pic);
        return schedule(false);
    }

    int flush() override { return schedule(true); }
    int pending() override { return (int)out.size(); }
    // fused chain: gather the frames of a chain batch, launch once for all that are ready
    void defer_launches(bool on) override { deferred = on; }
    int kick() override { return schedule(false, true); }

    DevPicture *pop_output() override
    {
        if (out.empty()) return nullptr;
        DevPicture *p = out.front();
        out.pop_front();
        return p;
    }
    void recycle_output(DevPicture *p) override { pool.release(p); }

    int batch = 1;
    bool deferred = false;
    int diff_cap[3] = {-1, -1, -1};
    uint32_t imul4[3] = {0, 0, 0};
    int ishift[3] = {0, 0, 0};
    int pf_type[3] = {0, 0, 0};          // effective prefilter bits per plane (0 = none)
    bool passthru[3] = {false, false, false};
    bool any_pre = false;
    // HBHIP_NLMEANS_GENERIC=1: every patch size through nlmeans_generic_kernel (a verification switch: the tuned kernels and
    // the generic one are two implementations of the same integers and the same float sequence)
    bool force_generic = [] { const char *e = getenv("HBHIP_NLMEANS_GENERIC"); return e != nullptr && atoi(e) != 0; }();

private:
    int ensure_jobs(int n)
    {
        if (n <= jobs_cap) return HBHIP_OK;
        if (d_jobs) (void)hipFree(d_jobs);
        if (h_jobs) (void)hipHostFree(h_jobs);
        d_jobs = nullptr; h_jobs = nullptr; jobs_cap = 0;
        HBHIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        // a ring of job tables; a slot is rewritten only after the upload that
        // last read it has completed (event per slot)
        HBHIP_CHECK(ctx, hipMalloc((void **)&d_jobs, sizeof(NlmJob) * n * NTABLES));
        HBHIP_CHECK(ctx, hipHostMalloc((void **)&h_jobs, sizeof(NlmJob) * n * NTABLES, hipHostMallocDefault));
        HBHIP_CHECK(ctx, hipHostGetDevicePointer((void **)&h_jobs_dev, h_jobs, 0));
        for (int i = 0; i < NTABLES; i++)
        {
            if (!table_ev[i]) HBHIP_CHECK(ctx, hipEventCreateWithFlags(&table_ev[i], hipEventDisableTiming));
            table_used[i] = false;
        }
        jobs_cap = n;
        return HBHIP_OK;
    }

    // pre*: the prefiltered twin of the frame (== the raw plane where a plane has no prefilter);
    // first: this is the first frame of the stream (its src_pre is latched raw, see NlmJob)
    struct View { uint8_t *plane[3]; int pitch[3]; uint8_t *pre[3]; int ppitch[3]; bool first; };

    View view_of(const DevPicture *p)
    {
        View v;
        const DevPicture *q = nullptr;
        if (any_pre)
        {
            auto it = pre_of.find(p);
            if (it != pre_of.end()) q = it->second;
        }
        for (int c = 0; c < 3; c++)
        {
            v.plane[c] = p->plane[c]; v.pitch[c] = p->pitch[c];
            const bool has = q && pf_type[c];
            v.pre[c] = has ? q->plane[c] : p->plane[c];
            v.ppitch[c] = has ? q->pitch[c] : p->pitch[c];
        }
        v.first = p->aux == 1;
        return v;
    }
    static View view_of(const hbhip_dev_frame &f)
    {
        View v;
        for (int c = 0; c < 3; c++)
        {
            v.plane[c] = (uint8_t *)f.plane[c]; v.pitch[c] = f.stride[c];
            v.pre[c] = v.plane[c]; v.ppitch[c] = v.pitch[c];
        }
        v.first = false;
        return v;
    }

    void release_input(DevPicture *p)
    {
        if (any_pre)
        {
            auto it = pre_of.find(p);
            if (it != pre_of.end())
            {
                pre_pool.release(it->second);
                pre_of.erase(it);
            }
        }
        hbhip_pic_release(p, ctx);     // possibly another filter's picture (fused chain)
    }

    // nlmeans_prefilter (nlmeans_template.c:428-543) of every plane that has one, into the frame's twin
    int prefilter_frame(DevPicture *pic)
    {
        DevPicture *q = pre_pool.acquire();
        if (!q) return HBHIP_ERR_NOMEM;
        pre_of[pic] = q;
        pic->aux = frames_seen++ == 0 ? 1 : 0;
        for (int c = 0; c < 3; c++)
        {
            const int type = pf_type[c];
            if (!type) continue;
            const int w = in_geo.pw[c], h = in_geo.ph[c];
            int wet = 1, dry = 0;
            if ((type & PF_REDUCE50) && (type & PF_REDUCE25)) { wet = 1; dry = 3; }
            else if (type & PF_REDUCE50)                      { wet = 1; dry = 1; }
            else if (type & PF_REDUCE25)                      { wet = 3; dry = 1; }
            const dim3 grid((w + 63) / 64, (h + 3) / 4), block(256);
#define PF_LAUNCH(NAME, KERNEL, GRID, BLOCK, ...) do { \
                if (in_geo.bps == 2) HBHIP_LAUNCH(ctx, NAME, KERNEL<uint16_t>, GRID, BLOCK, 0, __VA_ARGS__); \
                else                 HBHIP_LAUNCH(ctx, NAME, KERNEL<uint8_t>, GRID, BLOCK, 0, __VA_ARGS__); } while (0)
            PF_LAUNCH("nlmeans_prefilter", nlm_prefilter_kernel, grid, block,
                      (const uint8_t *)pic->plane[c], pic->pitch[c], q->plane[c], q->pitch[c], w, h, type, wet, dry);
            if (type & PF_EDGEBOOST)
            {
                PF_LAUNCH("nlmeans_edge_mask", nlm_edge_mask_kernel, grid, block,
                          (const uint8_t *)pic->plane[c], pic->pitch[c], mask_pic->plane[c], mask_pic->pitch[c], w, h);
                PF_LAUNCH("nlmeans_edge_apply", nlm_edge_apply_kernel, dim3(1), dim3(EB_THREADS),
                          (const uint8_t *)pic->plane[c], pic->pitch[c], mask_pic->plane[c], mask_pic->pitch[c],
                          q->plane[c], q->pitch[c], w, h);
                if (dry > 0)
                    PF_LAUNCH("nlmeans_prefilter_mix", nlm_mix_kernel, grid, block,
                              (const uint8_t *)pic->plane[c], pic->pitch[c], q->plane[c], q->pitch[c], w, h, wet, dry);
            }
#undef PF_LAUNCH
        }
        HBHIP_CHECK(ctx, hipGetLastError());
        return HBHIP_OK;
    }

    // Filter frames ins[0..ready) (ins beyond `ready` are look-ahead only) into outs[0..ready).
    int launch_views(const std::vector<View> &ins, int ready, const std::vector<View> &outs)
    {
        int rc = ensure_jobs(ready * 3);
        if (rc != HBHIP_OK) return rc;
        const int total = (int)ins.size();

        // group jobs by patch size (one launch per distinct n)
        std::vector<int> sizes;
        for (int c = 0; c < 3; c++)
            if (!passthru[c] && par.strength[c] != 0 && std::find(sizes.begin(), sizes.end(), par.patch_size[c]) == sizes.end())
                sizes.push_back(par.patch_size[c]);
        std::sort(sizes.begin(), sizes.end());
        for (int n : sizes)
        {
            // patch sizes without a lane-sharing instantiation (1, 11, 13, 15, ...) - or all of them, when asked for the
            // second implementation - go to nlmeans_generic_kernel
            const bool generic = force_generic || !(n == 3 || n == 5 || n == 7 || n == 9);
            bool any = false, pre = false;
            for (int c = 0; c < 3; c++)
                if (!passthru[c] && par.strength[c] != 0 && par.patch_size[c] == n)
                {
                    any = true;
                    pre |= pf_type[c] != 0;
                }
            if (!any) continue;
            table = (table + 1) % NTABLES;
            if (table_used[table]) HBHIP_CHECK(ctx, hipEventSynchronize(table_ev[table]));
            NlmJob *hj = h_jobs + (size_t)table * jobs_cap;
            NlmJob *dj = d_jobs + (size_t)table * jobs_cap;
            int nj = 0, tiles = 0, max_rh = 0;
            bool all_rh1 = true;
            bool fast = true, fast_int = true;
            for (int t = 0; t < ready; t++)
                for (int c = 0; c < 3; c++)
                {
                    if (passthru[c] || par.strength[c] == 0 || par.patch_size[c] != n) continue;
                    NlmJob &jb = hj[nj++];
                    jb.nframes = std::min(par.nframes[c], total - t);
                    for (int f = 0; f < jb.nframes; f++)
                    {
                        jb.frame[f] = ins[t + f].plane[c];
                        jb.fpitch[f] = ins[t + f].pitch[c];
                        jb.frame_pre[f] = ins[t + f].pre[c];
                        jb.ppitch[f] = ins[t + f].ppitch[c];
                    }
                    // src_pre is latched before the frame's own prefilter call (nlmeans_template.c:615
                    // vs :631): it is the prefiltered plane only if an earlier frame already used this
                    // one as a compare frame, i.e. not for the first frame of the stream and never
                    // with a single-frame window.  (With threads > 1 the reference races here; this
                    // is its single-threaded order.)
                    const bool latched_raw = ins[t].first || par.nframes[c] < 2;
                    jb.src_pre = latched_raw ? ins[t].plane[c] : ins[t].pre[c];
                    jb.src_pre_pitch = latched_raw ? ins[t].pitch[c] : ins[t].ppitch[c];
                    jb.dst = outs[t].plane[c];
                    jb.exptable = d_exp + 128 * c;
                    jb.origin_tune = par.origin_tune[c];
                    jb.wft = par.weight_fact_table[c];
                    jb.diff_max = par.diff_max[c];
                    jb.diff_cap = diff_cap[c];
                    jb.imul4 = imul4[c];
                    jb.ishift = ishift[c];
                    fast &= diff_cap[c] >= 0;
                    fast_int &= imul4[c] != 0 && (in_geo.bps == 2 || ishift[c] == 0);
                    jb.w = in_geo.pw[c];
                    jb.h = in_geo.ph[c];
                    jb.dst_pitch = outs[t].pitch[c];
                    jb.r_half = (par.range[c] - 1) / 2;
                    jb.tiles_x = generic ? (jb.w + GW - 1) / GW : (jb.w + LTW - 1) / LTW;
                    jb.tile_start = tiles;
                    tiles += jb.tiles_x * (generic ? (jb.h + GH - 1) / GH : (jb.h + TH - 1) / TH);
                    max_rh = std::max(max_rh, jb.r_half);
                    all_rh1 &= jb.r_half == 1;
                }
            if (nj == 0) continue;
            // the table travels by a copy kernel reading the pinned host buffer: hipMemcpyAsync of this size holds the
            // calling thread until the stream gets to it (measured 1.4 ms per call in a busy chain)
            {
                static_assert(sizeof(NlmJob) % 16 == 0, "NlmJob is copied in 16-byte units");
                const int n16 = (int)(sizeof(NlmJob) * nj / 16);
                HBHIP_LAUNCH(ctx, "nlmeans_job_table", job_table_kernel, dim3((n16 + 255) / 256), dim3(256), 0,
                             reinterpret_cast<uint4 *>(dj), reinterpret_cast<const uint4 *>(h_jobs_dev + (size_t)table * jobs_cap), n16);
            }
            HBHIP_CHECK(ctx, hipEventRecord(table_ev[table], ctx->stream));
            table_used[table] = true;
            const int nh = n / 2;
            if (generic)
            {
                const int aw_ = GW + 2 * nh, ah_ = GH + 2 * nh, bw_ = aw_ + 2 * max_rh, bh_ = ah_ + 2 * max_rh;
                const size_t shmem = sizeof(uint32_t) * ah_ * GW + (size_t)in_geo.bps * (((aw_ * ah_ + 1) & ~1) + bw_ * bh_ + 2);
                if (in_geo.bps == 1)
                    HBHIP_LAUNCH(ctx, "nlmeans_plane_generic", nlmeans_generic_kernel<uint8_t>, dim3(tiles), dim3(GW * GH), shmem, dj, nj, n, max_rh, (int)pre);
                else
                    HBHIP_LAUNCH(ctx, "nlmeans_plane_generic", nlmeans_generic_kernel<uint16_t>, dim3(tiles), dim3(GW * GH), shmem, dj, nj, n, max_rh, (int)pre);
                HBHIP_CHECK(ctx, hipGetLastError());
                continue;
            }
            const int cmp_rows = TH + 2 * (nh + max_rh);
            dim3 grid(tiles), block(TXN * TYN);
            // tiles: 32 lanes + rq dwords of search halo either side (+1 for the alignbyte high
            // word), pitch = 4 (mod 8) dwords
            const int rq = (max_rh + 3) / 4;
            const bool wide = in_geo.bps == 2;                 // 16-bit samples: 2 pixels per tile dword
            const int cpd = wide ? (rq <= 2 ? 76 : 84) : (rq <= 1 ? 36 : 44);
            // the pairs of frame 0 share their patch distances (FAST 3) when every plane of the launch searches 3 x 3
            // (patch sizes up to 7: the mirrored index of a lane's edge pixel comes from the neighbouring lane's edge pixel,
            // whose own window must not reach past ITS neighbour - the outer lanes of a tile row have none)
            const bool sym = fast && fast_int && !pre && all_rh1 && n <= 7;
            size_t shmem = sizeof(uint32_t) * (pre ? 4 : 2) * (cpd * cmp_rows + 4) + 512;
            // (16-bit samples: NLM16_SYM_PAIRS of the four pairs share - the stash of three fits three workgroups on a CU)
            const int sym_slots = wide ? NLM16_SYM_PAIRS : 4;
            if (sym) shmem = 512 + sizeof(uint32_t) * ((size_t)(cpd * cmp_rows + 4) + std::max<size_t>(cpd * cmp_rows + 4, (size_t)sym_slots * (RY + 1) * TXN * TYN));
            // the widest search ranges need more than the default 64 KB of dynamic LDS
            // The integer-index kernels (FAST >= 2) read the weight table at LDS address 0: true when the kernel has no static
            // LDS in front of its dynamic block.  Asked of the code object once per instantiation; a kernel that fails it is
            // refused here (the filter reports an error) rather than launched.
#define NLM_LAUNCH(KERNEL) do { \
                static const bool lds_free = nlm_no_static_lds((const void *)KERNEL); \
                if ((fast && fast_int) && !lds_free) return ctx->fail(hipErrorInvalidValue, "nlmeans: kernel has static LDS in front of its table"); \
                if (shmem > 65536) \
                    HBHIP_CHECK(ctx, hipFuncSetAttribute((const void *)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
                HBHIP_LAUNCH(ctx, kname, (KERNEL), grid, block, shmem, dj, nj, cmp_rows, rq); \
            } while (0)
#define NLM_GO(NN, FF, CC, PP) NLM_LAUNCH((nlmeans_lanes_kernel<NN, FF, CC, PP>))
#define NLM_PRE(NN, FF, CC) do { if (pre) NLM_GO(NN, FF, CC, true); else NLM_GO(NN, FF, CC, false); } while (0)
#define NLM_16P(NN, FF, PP) do { if (cpd == 76) NLM_LAUNCH((nlmeans_lanes16_kernel<NN, FF, 76, PP>)); \
                                else NLM_LAUNCH((nlmeans_lanes16_kernel<NN, FF, 84, PP>)); } while (0)
#define NLM_16(NN, FF) do { if (pre) NLM_16P(NN, FF, true); else NLM_16P(NN, FF, false); } while (0)
#define NLM_16S(NN) NLM_LAUNCH((nlmeans_lanes16_kernel<NN, (NN <= 7 ? 3 : 2), 76, false, NLM16_SYM_PAIRS>))
#define NLM_VAR(NN) do { const char *kname = "nlmeans_plane_n" #NN; \
                     if (wide) { if (sym && NN <= 7 && cpd == 76) NLM_16S(NN); else if (fast && fast_int) NLM_16(NN, 2); else if (fast) NLM_16(NN, 1); else NLM_16(NN, 0); } \
                     else if (cpd == 36) { if (sym && NN <= 7) NLM_GO(NN, (NN <= 7 ? 3 : 2), 36, false); else if (fast && fast_int) NLM_PRE(NN, 2, 36); else if (fast) NLM_PRE(NN, 1, 36); else NLM_PRE(NN, 0, 36); } \
                     else { if (sym && NN <= 7) NLM_GO(NN, (NN <= 7 ? 3 : 2), 44, false); else if (fast && fast_int) NLM_PRE(NN, 2, 44); else if (fast) NLM_PRE(NN, 1, 44); else NLM_PRE(NN, 0, 44); } } while (0)
            switch (n)
            {
                case 3: NLM_VAR(3); break;
                case 5: NLM_VAR(5); break;
                case 7: NLM_VAR(7); break;
                case 9: NLM_VAR(9); break;
            }
#undef NLM_VAR
#undef NLM_16
#undef NLM_16P
#undef NLM_16S
#undef NLM_PRE
#undef NLM_GO
#undef NLM_LAUNCH
            HBHIP_CHECK(ctx, hipGetLastError());
        }

        // prefilter "passthru" planes output their prefiltered twin (nlmeans.c:485-492, with
        // nlmeans_template.c:533-537), strength == 0 planes the frame itself (nlmeans.c:493-499)
        for (int t = 0; t < ready; t++)
            for (int c = 0; c < 3; c++)
                if (passthru[c] || par.strength[c] == 0)
                {
                    const int row = in_geo.pw[c] * in_geo.bps;
                    dim3 grid((row / 4 + 255) / 256 + 1, in_geo.ph[c]);
                    HBHIP_LAUNCH(ctx, "nlmeans_copy_plane", copy_plane_kernel, grid, dim3(256), 0,
                                 outs[t].plane[c], outs[t].pitch[c],
                                 (const uint8_t *)(passthru[c] ? ins[t].pre[c] : ins[t].plane[c]),
                                 passthru[c] ? ins[t].ppitch[c] : ins[t].pitch[c], row, in_geo.ph[c]);
                }
        return HBHIP_OK;
    }

    // Filter as many queued frames as are ready (all of them when draining).
    int schedule(bool draining, bool kicked = false)
    {
        int ready = draining ? (int)in.size() : (int)in.size() - (max_frames - 1);
        if (ready <= 0) return HBHIP_OK;
        if (!draining && !kicked && (deferred || ready < batch)) return HBHIP_OK;

        std::vector<DevPicture *> outs(ready, nullptr);
        std::vector<View> vin, vout;
        for (DevPicture *p : in) vin.push_back(view_of(p));
        for (int t = 0; t < ready; t++)
        {
            outs[t] = pool.acquire();
            if (!outs[t])
            {
                for (int k = 0; k < t; k++) pool.release(outs[k]);      // nothing was launched into them
                return HBHIP_ERR_NOMEM;
            }
            outs[t]->tag = in[t]->tag;
            vout.push_back(view_of(outs[t]));
        }
        int rc = launch_views(vin, ready, vout);
        if (rc != HBHIP_OK)
        {
            for (DevPicture *o : outs) pool.release(o);
            return rc;
        }
        for (int t = 0; t < ready; t++)
        {
            out.push_back(outs[t]);
            release_input(in.front());   // stream-ordered reuse
            in.pop_front();
        }
        return HBHIP_OK;
    }

public:
    // Zero-copy batch: the kernel reads the caller's frames and writes the caller's output
    // frames; only the max_frames-1 look-ahead frames are copied into the ring for the next call.
    int process_dev_batch(const hbhip_dev_frame *fin, int n_in, int64_t tag0,
                          const hbhip_dev_frame *fout, int out_cap, int *n_out) override
    {
        const int keep = max_frames - 1;
        const int total = (int)in.size() + n_in;
        const int ready = total - keep;
        bool direct = !any_pre && out.empty() && n_in >= keep && ready > 0 && ready <= out_cap && (int)in.size() <= ready;
        for (int i = 0; direct && i < n_in; i++)
            for (int c = 0; c < 3; c++)
                if ((fin[i].stride[c] & 3) || ((uintptr_t)fin[i].plane[c] & 3)) direct = false;
        for (int i = 0; direct && i < ready; i++)
            for (int c = 0; c < 3; c++)
                if ((fout[i].stride[c] & 3) || ((uintptr_t)fout[i].plane[c] & 3)) direct = false;
        if (!direct)
            return hbhip_filter::process_dev_batch(fin, n_in, tag0, fout, out_cap, n_out);

        std::vector<View> vin, vout;
        for (DevPicture *p : in) vin.push_back(view_of(p));
        for (int i = 0; i < n_in; i++) vin.push_back(view_of(fin[i]));
        for (int t = 0; t < ready; t++) vout.push_back(view_of(fout[t]));
        int rc = launch_views(vin, ready, vout);
        if (rc != HBHIP_OK) return rc;
        while (!in.empty())
        {
            release_input(in.front());
            in.pop_front();
        }
        for (int i = n_in - keep; i < n_in; i++)
        {
            DevPicture *p = pool.acquire();
            if (!p) return HBHIP_ERR_NOMEM;
            p->tag = tag0 + i;
            rc = hbhip_copy_d2d_in(ctx, p, &fin[i]);
            if (rc != HBHIP_OK) return rc;
            in.push_back(p);
        }
        *n_out = ready;
        return HBHIP_OK;
    }

private:
    hbhip_nlmeans_params par;
    PicturePool pool;
    PicturePool pre_pool;                                    // prefiltered twins (+ the edge-boost mask)
    std::unordered_map<const DevPicture *, DevPicture *> pre_of;
    DevPicture *mask_pic = nullptr;
    long frames_seen = 0;
    std::deque<DevPicture *> in, out;
    int max_frames = 1;
    float *d_exp = nullptr;
    NlmJob *d_jobs = nullptr, *h_jobs = nullptr, *h_jobs_dev = nullptr;   // h_jobs_dev: the device's address of h_jobs
    int jobs_cap = 0, table = 0;
    static constexpr int NTABLES = 8;
    hipEvent_t table_ev[NTABLES] = {};
    bool table_used[NTABLES] = {};
};

} // namespace

extern "C" int hbhip_nlmeans_create(hbhip_ctx *ctx, const hbhip_nlmeans_params *p,
                                    int width, int height, int depth,
                                    int log2_chroma_w, int log2_chroma_h, hbhip_filter **out)
{
    if (!ctx || !p || !out) return HBHIP_ERR_ARG;
    *out = nullptr;
    if (depth != 8 && depth != 10 && depth != 12) return HBHIP_ERR_UNSUPPORTED;
    if (width < 1 || height < 1) return HBHIP_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    NlmFilter *f = new (std::nothrow) NlmFilter(ctx, *p);
    if (!f) return HBHIP_ERR_NOMEM;
    int rc = f->setup(width, height, depth, log2_chroma_w, log2_chroma_h);
    if (rc != HBHIP_OK)
    {
        delete f;
        return rc;
    }
    *out = f;
    return HBHIP_OK;
}

extern "C" int hbhip_nlmeans_set_batch(hbhip_filter *f, int frames)
{
    NlmFilter *n = dynamic_cast<NlmFilter *>(f);
    if (!n || frames < 1) return HBHIP_ERR_ARG;
    n->batch = frames;
    return HBHIP_OK;
}
