// eedi2_16.hip — EEDI2 for 10 / 12-bit samples (eedi2_template.c instantiated with pixel = uint16_t,
// decomb.c:324-331), first correct form: one thread per sample and pass, no LDS tiling, the
// in-place lattice pass walked serially per row.  The tuned 8-bit kernels of eedi2.hip lean on byte
// packing (packed SAD, 4 samples per dword) and do not carry over; this file follows the pinned
// restatement oracle/eedi2_16_oracle.c pass by pass instead.
//
// What the reference does differently above 8 bits (all marked "16:" in the oracle): thresholds shifted
// by depth-8 or typed `pixel` = uint16 so that they wrap at 16 bits, limlut << (depth-8), PEAK / NEUTRAL
// from the depth, sums and squares taken on samples >> (depth-8).  Scratch frames keep the sample
// layout hb_frame_buffer_init gives a 16-bit frame (stride = 2*width rounded up to 64 bytes), inside
// zeroed guards, because the passes index flat buffers and read outside rows / planes.
// All pitches inside the kernels are in SAMPLES.
#include "eedi2_engine.h"
#include <algorithm>

namespace {

constexpr size_t GUARD16 = 32768;          // samples

struct K16
{
    int peak, neutral, shift;
    int limlut[33];                        // eedi2_init_limlut (:23-33): (pixel)eedi2_limlut[i] << shift, stored as pixel
};

struct Q3
{
    uint16_t *a[3], *b[3], *c[3];          // pass specific roles, see each kernel
    uint16_t *d[3], *e[3], *f[3], *g[3];   // q_mark_2x's line doublings (picked by that kernel itself)
    int pitch[3], width[3], height[3];
    size_t   fstride;                      // field batching (see eedi2.hip): samples between the slots of consecutive fields
    uint32_t tffbits;                      // bit f: pv->tff of field f of the launch
    const uint32_t *pflags;                // [field][plane] of the launch: == pepoch when the plane's edge mask has a sample set
    uint32_t pepoch;                       // (eedi2.hip: P3::pflags - a plane without one is filled / copied by the shortest way)
};

// blockIdx.z = 3 * field + plane: the block's pointers, picked once from the arguments (never written back, eedi2.hip)
struct QL { uint16_t *a, *b, *c; };
__device__ __forceinline__ QL plane_ptrs16(const Q3 &P, int pl, size_t off)
{
    return QL{ P.a[pl] + off, P.b[pl] + off, P.c[pl] + off };     // none is ever tested for "not bound" (eedi2.hip: plane_ptrs)
}
#define FIELD16(P)                                                           \
    const int fld = (int)blockIdx.z / 3, pl = (int)blockIdx.z - 3 * fld;     \
    const int tff = (int)(((P).tffbits >> fld) & 1u);                        \
    const QL Q = plane_ptrs16((P), pl, (size_t)fld * (P).fstride);           \
    const bool maskless = (P).pflags[blockIdx.z] != (P).pepoch;              \
    (void)tff; (void)maskless

#define XY16(P)                                                              \
    FIELD16(P);                                                              \
    const int x = blockIdx.x * blockDim.x + threadIdx.x;                     \
    const int y = blockIdx.y * blockDim.y + threadIdx.y;                     \
    const int pitch = (P).pitch[pl], width = (P).width[pl], height = (P).height[pl]; \
    (void)width; (void)height; (void)pitch

__device__ __forceinline__ int iabs16(int v) { return v < 0 ? -v : v; }

__device__ __forceinline__ int sad3w(const uint16_t *a, int ai, const uint16_t *b, int bi)
{
    return iabs16((int)a[ai - 1] - (int)b[bi - 1]) + iabs16((int)a[ai] - (int)b[bi]) + iabs16((int)a[ai + 1] - (int)b[bi + 1]);
}

// insertion sort + midpoint rule (eedi2.c:65-80)
__device__ __forceinline__ int sorted_mid16(int *v, int n)
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

// ---- the field extraction and the five mask passes in one launch, as for 8-bit samples (eedi2.hip: k_mask_fused4) ----
// build_edge_mask -> erode -> dilate -> erode -> remove_small_gaps each look one sample (three along x for the last)
// around themselves, so a workgroup carries a 128 x 16 tile of the final mask through all of them in LDS with a
// shrinking halo.  Only build_edge_mask sees samples; the mask itself is 0 / peak at any depth, so inside the kernel a
// mask cell is a byte holding 0 / 1 and the morphology runs on four cells per 32-bit operation exactly as in the 8-bit
// kernel.  a = SRCPF (written: the tile's part of the extracted field), b = the finished mask of the field before
// field 0 of the launch, c = MSKPF.  `part`: 0 = every tile, 1 = only the tiles whose LDS frame stays above height / 2
// (independent of the previous field: all fields of a batch in one launch), 2 = only the others (the chain).
#ifndef QM_TILE_W
#define QM_TILE_W 128
#endif
#ifndef QM_THREADS
#define QM_THREADS 512
#endif
constexpr int QM_W = QM_TILE_W, QM_H = 16, QM_OX = 8, QM_OY = 4;
constexpr int QM_LP = QM_W + 2 * QM_OX, QM_LR = QM_H + 2 * QM_OY;      // 144 x 24
constexpr int QM_DW = QM_LP / 4, QM_DP = QM_DW + 2, QM_SR = 2, QM_T = QM_THREADS;
static_assert(((QM_LR - 2 + QM_SR - 1) / QM_SR) * QM_DW <= QM_T, "a thread per strip and dword column");

__device__ __forceinline__ uint32_t qm_bytes_in(int X, int lo, int hi)        // 0xff in byte k when lo <= X + k < hi
{
    uint32_t m = 0xffffffffu;
    const int a = lo - X, b = hi - X;
    if (a > 0) m = a >= 4 ? 0u : (m << (8 * a));
    if (b < 4) m = b <= 0 ? 0u : (m & (0xffffffffu >> (8 * (4 - b))));
    return m;
}

template <bool GROW>
__device__ __forceinline__ void qm_morph4(const uint32_t (*src)[QM_DP], uint32_t (*dst)[QM_DP], int c4, int strip,
                                          int ra, int rb, int thr, uint32_t px1, int fy, int height)
{
    const int r0 = ra + strip * QM_SR;
    if (r0 <= rb)
    {
        const uint32_t K = (uint32_t)(0x80 - min(max(thr, 0), 9)) * 0x01010101u;
        uint32_t S2[QM_SR + 2], S3[QM_SR + 2], C[QM_SR + 2];
#pragma unroll
        for (int i = 0; i < QM_SR + 2; i++)
        {
            const int r = min(r0 - 1 + i, QM_LR - 1);
            const uint32_t l = src[r][c4], c = src[r][c4 + 1], rr = src[r][c4 + 2];
            C[i] = c;
            S2[i] = __builtin_amdgcn_alignbyte(c, l, 3) + __builtin_amdgcn_alignbyte(rr, c, 1);
            S3[i] = S2[i] + c;
        }
#pragma unroll
        for (int i = 0; i < QM_SR; i++)
        {
            const int r = r0 + i;
            if (r > rb) break;
            const int y = fy + r;
            const uint32_t count = S3[i] + S2[i + 1] + S3[i + 2];
            const uint32_t ge = ((count + K) >> 7) & 0x01010101u;          // count >= thr, per cell
            const uint32_t pm = (y >= 1 && y < height - 1) ? px1 : 0u;
            const uint32_t c = C[i + 1];
            dst[r][c4 + 1] = GROW ? (c | (ge & pm)) : (c & ~((ge ^ 0x01010101u) & pm));
        }
    }
    __syncthreads();
}

struct MaskSrc16 { const uint16_t *frame[EEDI_MAX_BATCH][3]; int sp[3]; };     // sp: frame pitch in samples

// CHAIN: the tile is one of q_mask_chain's (MaskChain, eedi2_engine.h): it waits for the previous field's tiles around it
// before it reads their mask, reads and writes mask words as agent-scope atomics, and publishes itself at the end
template <bool CHAIN>
__device__ __forceinline__ void qmask_tile(const Q3 &P, const MaskSrc16 &S, const K16 &k, const MaskChain &C, int fld, int pl, int bx, int by,
                                           int mth, int vth, int lth, int erode_thr, int dilate_thr,
                                           uint16_t (*s_src)[QM_LP + 8], uint32_t (*s_a)[QM_DP], uint32_t (*s_b)[QM_DP])
{
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    const int x0 = bx * QM_W, y0 = by * QM_H;
    const bool upper = y0 + QM_H + QM_OY <= height / 2;            // no row of the LDS frame reaches the kept half
    const size_t foff = (size_t)fld * P.fstride;
    const uint16_t *oldm = fld == 0 ? P.b[pl] : P.c[pl] + foff - P.fstride;
    const uint16_t *frame = S.frame[fld][pl];
    const int start_line = (int)(((P.tffbits >> fld) & 1u) ^ 1u);
    uint16_t *srcp = P.a[pl] + foff, *newm = P.c[pl] + foff;
    const int t = threadIdx.x, fx = x0 - QM_OX, fy = y0 - QM_OY;
    const int peak = k.peak, sh = k.shift;

    for (int i = t; i < QM_LR * QM_DW; i += QM_T)
    {
        const int r = i / QM_DW, c4 = i - r * QM_DW;
        const int y = fy + r, x = fx + 4 * c4;
        uint2 sv = make_uint2(0u, 0u);
        if (y >= 0 && y < height && x >= 0 && x < pitch)
        {
            if (x < width)
            {
                sv = *reinterpret_cast<const uint2 *>(frame + (size_t)(start_line + 2 * y) * S.sp[pl] + x);
                const int n = width - x;                           // samples at x >= width read as 0
                if (n < 4)
                {
                    if (n <= 2) sv.y = 0;
                    if (n == 3) sv.y &= 0xffffu;
                    if (n == 1) sv.x &= 0xffffu;
                }
            }
            // the tile's own cells go out as SRCPF (every cell of the plane belongs to exactly one tile)
            if (r >= QM_OY && r < QM_OY + QM_H && c4 >= QM_OX / 4 && c4 < (QM_OX + QM_W) / 4)
                *reinterpret_cast<uint2 *>(srcp + (size_t)y * pitch + x) = sv;
        }
        *reinterpret_cast<uint2 *>(&s_src[r][4 * c4 + 4]) = sv;
    }
    if (CHAIN && fld > 0) eedi_chain_wait(C, fld, pl, bx, by);    // (the source rows above are already on their way)
    for (int i = t; i < QM_LR * QM_DW; i += QM_T)
    {
        const int r = i / QM_DW, c4 = i - r * QM_DW;
        const int y = fy + r, x = fx + 4 * c4;
        uint32_t mv = 0;
        // (only the rows of the kept half are used, and those were written by lower tiles)
        if (!upper && y >= 0 && y < height && x >= 0 && x < pitch)
        {
            const uint32_t *m = reinterpret_cast<const uint32_t *>(oldm + (size_t)y * pitch + x);
            uint2 ov;
            if (CHAIN)
            {
                ov.x = __hip_atomic_load(m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ov.y = __hip_atomic_load(m + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            else ov = *reinterpret_cast<const uint2 *>(m);
            mv = ((ov.x & 0xffffu) == (uint32_t)peak ? 1u : 0u) | ((ov.x >> 16) == (uint32_t)peak ? 0x100u : 0u) |
                 ((ov.y & 0xffffu) == (uint32_t)peak ? 0x10000u : 0u) | ((ov.y >> 16) == (uint32_t)peak ? 0x1000000u : 0u);
        }
        s_a[r][c4 + 1] = mv;
    }
    __syncthreads();

    const int c4 = t % QM_DW, strip = t / QM_DW;               // strips past the frame have no rows in any pass
    const int X = fx + 4 * c4;
    const uint32_t px1 = qm_bytes_in(X, 1, width - 1) & 0x01010101u;

    // build_edge_mask (:122-195), in place on the old mask; LDS rows 1 .. 22
    {
        const int r0 = 1 + strip * QM_SR;
        if (r0 <= QM_LR - 2)
        {
            const int ten = (uint16_t)(10 << sh);
            int b[QM_SR + 2][6], q[QM_SR + 2][6];
#pragma unroll
            for (int i = 0; i < QM_SR + 2; i++)
            {
                const int r = min(r0 - 1 + i, QM_LR - 1);
                const uint16_t *row = &s_src[r][4 * c4 + 4];      // samples X .. X + 3 at row[0 .. 3]
                const uint2 c = *reinterpret_cast<const uint2 *>(row);
                b[i][0] = (int)row[-1];                            // column X - 1 (c4 = 0: a pad column, its cells are masked out)
                b[i][1] = (int)(c.x & 0xffffu); b[i][2] = (int)(c.x >> 16); b[i][3] = (int)(c.y & 0xffffu); b[i][4] = (int)(c.y >> 16);
                b[i][5] = (int)row[4];
#pragma unroll
                for (int j = 0; j < 6; j++) q[i][j] = (b[i][j] >> sh) * (b[i][j] >> sh);
            }
#pragma unroll
            for (int i = 0; i < QM_SR; i++)
            {
                const int r = r0 + i;
                if (r > QM_LR - 2) break;
                const int y = fy + r;
                const int (&Pr)[6] = b[i], (&Cr)[6] = b[i + 1], (&Nr)[6] = b[i + 2];
                // (max - min of a column's three samples serves the flatness test and Iy: eedi2.hip, mask_tile)
                int cs[6], cq[6], rng[6];
                bool fl[6];
#pragma unroll
                for (int j = 0; j < 6; j++)
                {
                    cs[j] = Pr[j] + Cr[j] + Nr[j];
                    cq[j] = q[i][j] + q[i + 1][j] + q[i + 2][j];
                    rng[j] = max(max(Pr[j], Cr[j]), Nr[j]) - min(min(Pr[j], Cr[j]), Nr[j]);
                    fl[j] = rng[j] < ten;
                }
                uint32_t edge = 0;
#pragma unroll
                for (int kk = 0; kk < 4; kk++)
                {
                    const int sum = (cs[kk] + cs[kk + 1] + cs[kk + 2]) >> sh, sumsq = cq[kk] + cq[kk + 1] + cq[kk + 2];
                    const int C0 = Cr[kk], C1 = Cr[kk + 1], C2 = Cr[kk + 2], P1 = Pr[kk + 1], N1 = Nr[kk + 1];
                    const int ix = (C2 - C0) >> sh;
                    const int iy = rng[kk + 1] >> sh;
                    const int ixx = (C0 - 2 * C1 + C2) >> sh, iyy = (P1 - 2 * C1 + N1) >> sh;
                    // (no short-circuit: as `&&` / `||` the tests become exec-mask branches)
                    const bool notflat = !(fl[kk + 1] | (fl[kk] & fl[kk + 2]));
                    const bool var = !(9 * sumsq - sum * sum < vth);
                    const bool mag = ix * ix + iy * iy >= mth;
                    const bool e = notflat & var & (mag | (iabs16(ixx) + iabs16(iyy) >= lth));
                    edge |= (e ? 1u : 0u) << (8 * kk);
                }
                const uint32_t keep = (y < height / 2) ? 0u : s_a[r][c4 + 1];
                const uint32_t pm = (y >= 1 && y < height - 1) ? px1 : 0u;
                s_a[r][c4 + 1] = keep | (edge & pm);
            }
        }
    }
    __syncthreads();

    qm_morph4<false>(s_a, s_b, c4, strip, 2, QM_LR - 3, erode_thr, px1, fy, height);
    qm_morph4<true>(s_b, s_a, c4, strip, 3, QM_LR - 4, dilate_thr, px1, fy, height);
    qm_morph4<false>(s_a, s_b, c4, strip, 4, QM_LR - 5, erode_thr, px1, fy, height);

    // remove_small_gaps (:308-342) on the tile's 16 rows x 32 cell dwords, straight to the new mask
    uint32_t anyset = 0;
    for (int i = t; i < QM_H * (QM_W / 4); i += QM_T)
    {
        const int r = QM_OY + i / (QM_W / 4), g4 = QM_OX / 4 + (i & (QM_W / 4 - 1));
        const int y = fy + r, x = fx + 4 * g4;
        if (y >= height || x >= width) continue;
        const uint32_t l = s_b[r][g4], c = s_b[r][g4 + 1], rr = s_b[r][g4 + 2];
        const uint32_t a1 = __builtin_amdgcn_alignbyte(c, l, 3), a2 = __builtin_amdgcn_alignbyte(c, l, 2), a3 = __builtin_amdgcn_alignbyte(c, l, 1);
        const uint32_t b1 = __builtin_amdgcn_alignbyte(rr, c, 1), b2 = __builtin_amdgcn_alignbyte(rr, c, 2), b3 = __builtin_amdgcn_alignbyte(rr, c, 3);
        const uint32_t a12 = a1 | a2, a123 = a12 | a3;
        const uint32_t set = c & (a123 | b1 | b2 | b3);                               // a set cell survives with any neighbour set
        const uint32_t fill = ((b1 & a123) | (b2 & a12) | (b3 & a1)) & (c ^ 0x01010101u);
        const uint32_t pm = (y >= 1 && y < height - 1) ? (qm_bytes_in(x, 3, width - 3) & 0x01010101u) : 0u;
        const uint32_t res = ((set | fill) & pm) | (c & ~pm);                         // 0 / 1 per cell
        anyset |= x + 3 < width ? res : res & (0xffffffffu >> (8 * (x + 4 - width)));
        const uint32_t pk = (uint32_t)peak;
        const uint2 out = make_uint2(((res & 1u) ? pk : 0u) | ((res & 0x100u) ? pk << 16 : 0u),
                                     ((res & 0x10000u) ? pk : 0u) | ((res & 0x1000000u) ? pk << 16 : 0u));
        uint16_t *d = newm + (size_t)y * pitch + x;
        if (x + 3 < width)
        {
            if (CHAIN)
            {
                __hip_atomic_store(reinterpret_cast<uint32_t *>(d), out.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(reinterpret_cast<uint32_t *>(d) + 1, out.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            else *reinterpret_cast<uint2 *>(d) = out;
        }
        else
        {
            const uint16_t o4[4] = { (uint16_t)(out.x & 0xffffu), (uint16_t)(out.x >> 16), (uint16_t)(out.y & 0xffffu), (uint16_t)(out.y >> 16) };
            for (int kk = 0; kk < 4 && x + kk < width; kk++)
            {
                if (CHAIN) __hip_atomic_store(d + kk, o4[kk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else d[kk] = o4[kk];
            }
        }
    }
    // a plane with a mask sample somewhere: say so (eedi2.hip: mask_tile)
    const bool has = CHAIN ? eedi_chain_signal(C, fld, pl, bx, by, anyset != 0u) : (bool)__syncthreads_or(anyset != 0u);
    eedi_chain_note_has(C, fld, pl, bx, by, has);
}

__global__ __launch_bounds__(QM_T) void q_mask_fused(Q3 P, MaskSrc16 S, K16 k, int f0, int part, int mth, int vth, int lth,
                                                     int erode_thr, int dilate_thr, uint32_t *pflags, uint32_t epoch)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_src[QM_LR][QM_LP + 8];   // sample column = frame column + 4
    __shared__ uint32_t s_a[QM_LR][QM_DP];
    __shared__ uint32_t s_b[QM_LR][QM_DP];
    const int zf = (int)blockIdx.z / 3, pl = (int)blockIdx.z - 3 * zf, fld = f0 + zf;   // f0: first field of this launch
    const int x0 = blockIdx.x * QM_W, y0 = blockIdx.y * QM_H;
    if (x0 >= P.width[pl] || y0 >= P.height[pl]) return;
    const bool upper = y0 + QM_H + QM_OY <= P.height[pl] / 2;
    if (part != 0 && upper != (part == 1)) return;
    MaskChain none;
    none.pflags = pflags; none.epoch = epoch; none.has = nullptr;
    qmask_tile<false>(P, S, k, none, fld, pl, (int)blockIdx.x, (int)blockIdx.y, mth, vth, lth, erode_thr, dilate_thr, s_src, s_a, s_b);
}

// blockIdx.x = field * C.ntiles + tile: field-major, see MaskChain
__global__ __launch_bounds__(QM_T) void q_mask_chain(Q3 P, MaskSrc16 S, K16 k, MaskChain C, int mth, int vth, int lth,
                                                     int erode_thr, int dilate_thr)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_src[QM_LR][QM_LP + 8];
    __shared__ uint32_t s_a[QM_LR][QM_DP];
    __shared__ uint32_t s_b[QM_LR][QM_DP];
    int fld, pl, bx, by;
    if (eedi_chain_tile(C, fld, pl, bx, by))                      // block-uniform: a link of the chain, or an upper tile riding along
        qmask_tile<true>(P, S, k, C, fld, pl, bx, by, mth, vth, lth, erode_thr, dilate_thr, s_src, s_a, s_b);
    else
        qmask_tile<false>(P, S, k, C, fld, pl, bx, by, mth, vth, lth, erode_thr, dilate_thr, s_src, s_a, s_b);
}

// the pass behind q_mask_chain (k_mask_chain_repair, eedi2.hip, says what it is for): the plane flags out of the tiles'
// words, a workgroup per field and plane; the first one then repairs - a no-op unless a wait of the chain ran out
__global__ __launch_bounds__(QM_T) void q_mask_chain_repair(Q3 P, MaskSrc16 S, K16 k, MaskChain C, int nfields, int mth, int vth, int lth,
                                                            int erode_thr, int dilate_thr)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_src[QM_LR][QM_LP + 8];
    __shared__ uint32_t s_a[QM_LR][QM_DP];
    __shared__ uint32_t s_b[QM_LR][QM_DP];
    eedi_chain_fold_has(C, QM_T);
    if (blockIdx.x != 0 || __hip_atomic_load(C.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;      // block-uniform
    MaskChain R = C;
    R.has = nullptr;                                               // the repaired tiles raise the plane flags themselves
    for (int fld = 0; fld < nfields; fld++)
        for (int tile = 0; tile < C.ntiles; tile++)
        {
            int pl, bx, by;
            eedi_chain_lower_tile(C, tile, pl, bx, by);
            qmask_tile<true>(P, S, k, R, fld, pl, bx, by, mth, vth, lth, erode_thr, dilate_thr, s_src, s_a, s_b);
            __syncthreads();
        }
    if (threadIdx.x == 0)
    {
        __hip_atomic_store(C.err, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(C.fallbacks, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// eedi2_calc_directions (:358-525): a = mskp, b = srcp, c = out (whole pitch pre-filled with PEAK)
__global__ void q_calc_dir(Q3 P, K16 k, int maxd, int nt)
{
    XY16(P);
    if (x >= pitch || y >= height) return;
    const int peak = k.peak;
    int out = peak;
    const uint16_t *mc = Q.a + (size_t)y * pitch, *mp = mc - pitch, *mn = mc + pitch;
    if (x >= 1 && x < width - 1 && y >= 1 && y < height - 1 && mc[x] == peak && (mc[x - 1] == peak || mc[x + 1] == peak))
    {
        const uint16_t *sc = Q.b + (size_t)y * pitch, *sp = sc - pitch, *s2p = sp - pitch, *sn = sc + pitch, *s2n = sn + pitch;
        const int nt13 = (uint16_t)((nt << k.shift) * 13), nt19 = (uint16_t)((nt << k.shift) * 19);
        const int maxdt = pl == 0 ? maxd : (maxd >> 1);
        const int startu = max(-x + 1, -maxdt), stopu = min(width - 2 - x, maxdt);
        const int vert = iabs16((int)sc[x] - (int)sn[x]) + iabs16((int)sc[x] - (int)sp[x]);
        int minb = min(nt13, vert * 6), mina = min(nt19, vert * 9);
        int minc = mina, mind = minb, mine = minb;
        int dira = -5000, dirb = -5000, dirc = -5000, dird = -5000, dire = -5000;
        for (int u = startu; u <= stopu; u++)
        {
            if (!(y == 1 || mp[x - 1 + u] == peak || mp[x + u] == peak || mp[x + 1 + u] == peak)) continue;
            if (!(y == height - 2 || mn[x - 1 - u] == peak || mn[x - u] == peak || mn[x + 1 - u] == peak)) continue;
            const int diffsn = sad3w(sc, x, sn, x - u);
            const int diffsp = sad3w(sc, x, sp, x + u);
            const int diffps = sad3w(sp, x, sc, x - u);
            const int diffns = sad3w(sn, x, sc, x + u);
            const int diff = diffsn + diffsp + diffps + diffns;
            int diffd = diffsp + diffns, diffe = diffsn + diffps;
            if (diff < minb) { dirb = u; minb = diff; }
            if (y > 1)
            {
                const int diff2pp = sad3w(s2p, x, sp, x - u);
                const int diffp2p = sad3w(sp, x, s2p, x + u);
                const int diffa = diff + diff2pp + diffp2p;
                diffd += diffp2p;
                diffe += diff2pp;
                if (diffa < mina) { dira = u; mina = diffa; }
            }
            if (y < height - 2)
            {
                const int diff2nn = sad3w(s2n, x, sn, x + u);
                const int diffn2n = sad3w(sn, x, s2n, x - u);
                const int diffc = diff + diff2nn + diffn2n;
                diffd += diff2nn;
                diffe += diffn2n;
                if (diffc < minc) { dirc = u; minc = diffc; }
            }
            if (diffd < mind) { dird = u; mind = diffd; }
            if (diffe < mine) { dire = u; mine = diffe; }
        }
        int order[5], n = 0;
        if (dira != -5000) order[n++] = dira;
        if (dirb != -5000) order[n++] = dirb;
        if (dirc != -5000) order[n++] = dirc;
        if (dird != -5000) order[n++] = dird;
        if (dire != -5000) order[n++] = dire;
        out = k.neutral;
        if (n > 1)
        {
            const int mid = sorted_mid16(order, n);
            const int tlim = max(k.limlut[iabs16(mid)] >> 2, 2);
            int sum = 0, count = 0;
            for (int i = 0; i < n; i++)
                if (iabs16(order[i] - mid) <= tlim) { count++; sum += order[i]; }
            if (count > 1) out = (uint16_t)(k.neutral + ((int)((float)sum / (float)count) << (2 + k.shift)));
        }
    }
    Q.c[(size_t)y * pitch + x] = (uint16_t)out;
}

// calc_directions for search distances <= 30, the form that runs: the structure of k_calc_dir_rows (eedi2.hip) on 16-bit
// samples.  A block takes 256 columns x R rows: R + 4 source and R + 2 mask rows staged in LDS (flat addressing, as the
// reference's pointers), the three samples every search step needs of a row (columns i .. i+2) formed once into a table
// of dword pairs ({s[i] | s[i+1] << 16, s[i+2]}: a 3-sample SAD is two v_sad_u16), the mask rows reduced to peak
// bitmaps (ballots) from which each masked pixel cuts its step set as a 64-bit word, the masked pixels listed in column
// order, one lane each.  A running minimum and its offset are one integer, (sum << 6) | (u + 32) (sums of 18 samples
// stay below 2^26); values and comparisons are q_calc_dir's (:358-525).
constexpr int QW = 256, QHALO = 32, QLW = QW + 2 * QHALO;

__device__ __forceinline__ uint32_t sad3q(uint2 a, uint2 b, uint32_t acc)
{
    return __builtin_amdgcn_sad_u16(a.x, b.x, __builtin_amdgcn_sad_u16(a.y, b.y, acc));
}

// the vote over the five offsets (:486-517): ta .. te = u + 32, 0 = never set (the reference's -5000)
__device__ __forceinline__ int calc_dir_vote16(int ta, int tb, int tc, int td, int te, const int *limlut, int neutral, int shift)
{
    // the offsets that were set, sorted (unset ones as a large sentinel at the end): 9-exchange network on 5 values
    constexpr int BIG = 1 << 20;
    int v0 = ta ? ta - 32 : BIG, v1 = tb ? tb - 32 : BIG, v2 = tc ? tc - 32 : BIG, v3 = td ? td - 32 : BIG, v4 = te ? te - 32 : BIG;
    const int n = (ta != 0) + (tb != 0) + (tc != 0) + (td != 0) + (te != 0);
#define Q_CX(a, b) { const int lo_ = min(a, b), hi_ = max(a, b); a = lo_; b = hi_; }
    Q_CX(v0, v1) Q_CX(v3, v4) Q_CX(v2, v4) Q_CX(v2, v3) Q_CX(v0, v3) Q_CX(v0, v2) Q_CX(v1, v4) Q_CX(v1, v3) Q_CX(v1, v2)
#undef Q_CX
    int out = neutral;
    if (n > 1)
    {
        // sorted_mid16's midpoint rule: odd n -> v[n/2], even n -> (v[(n-1)/2] + v[n/2] + 1) >> 1; one formula serves both
        const int lo = n == 2 ? v0 : (n == 5 ? v2 : v1);
        const int hi = n >= 4 ? v2 : v1;
        const int mid = (lo + hi + 1) >> 1;
        const int tlim = max(limlut[iabs16(mid)] >> 2, 2);
        int sum = 0, cnt = 0;
        if (iabs16(v0 - mid) <= tlim) { cnt++; sum += v0; }          // the sentinels fail the test by themselves
        if (iabs16(v1 - mid) <= tlim) { cnt++; sum += v1; }
        if (iabs16(v2 - mid) <= tlim) { cnt++; sum += v2; }
        if (iabs16(v3 - mid) <= tlim) { cnt++; sum += v3; }
        if (iabs16(v4 - mid) <= tlim) { cnt++; sum += v4; }
        if (cnt > 1) out = (uint16_t)(neutral + ((int)((float)sum / (float)cnt) << (2 + shift)));
    }
    return out;
}

// The dense form of the search (calc_dir_dense in eedi2.hip has the derivation: the mask's lower half saturates, and the
// 3-sample SADs of a step are shared between a column's rows and between u and -u).  On 16-bit samples a SAD is two
// v_sad_u16 and does not come shifted, so P_r / Q_r are moved into place with their tag by one v_lshl_add each:
// keys (sum << 9) | tags, tags = 4 (u + 32) for b / d / e and 6 (u + 32) for a / c (sums of 18 samples of at most 12 bits
// stay below 2^17), a step a pixel does not take gets 2^30 added.
template <int R, bool PRED>
__device__ __forceinline__ void calc_dir_dense16(const uint2 *tr, int maxdt, const uint32_t (&up)[R], const uint32_t (&dn)[R],
                                                 int nt13, int nt19, uint32_t (&ka)[R], uint32_t (&kb)[R], uint32_t (&kc)[R],
                                                 uint32_t (&kd)[R], uint32_t (&ke)[R])
{
    constexpr int NS = R + 4, NE = R + 3;
    uint2 T[NS];
#pragma unroll
    for (int r = 0; r < NS; r++) T[r] = tr[r * QLW];
#pragma unroll
    for (int j = 0; j < R; j++)
    {
        const int ctr = (int)(T[j + 2].x >> 16);
        const int vert = iabs16(ctr - (int)(T[j + 3].x >> 16)) + iabs16(ctr - (int)(T[j + 1].x >> 16));
        kb[j] = (uint32_t)min(nt13, vert * 6) << 9; ka[j] = (uint32_t)min(nt19, vert * 9) << 9;
        kc[j] = ka[j]; kd[j] = kb[j]; ke[j] = kb[j];
    }
    auto term = [](uint2 a, uint2 b, uint32_t tag) { return (sad3q(a, b, 0u) << 9) + tag; };
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
        uint32_t Pv[NE], ca[R], cb[R], cc[R], cd[R], ce[R];
#pragma unroll
        for (int r = 0; r < NE; r++) Pv[r] = term(T[r], T[r + 1], 32u);
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
        const uint2 *tp = tr + d, *tm = tr - d;
        uint2 Pl[NS], Mi[NS];
#pragma unroll
        for (int r = 0; r < NS; r++) { Pl[r] = tp[r * QLW]; Mi[r] = tm[r * QLW]; }
        const uint32_t t1 = 32u + (uint32_t)d, t2 = 32u - (uint32_t)d;
        uint32_t P1[NE], Q1[NE], Pn[NE], Qn[NE];
#pragma unroll
        for (int r = 0; r < NE; r++)
        {
            P1[r] = term(T[r], Mi[r + 1], t1); Q1[r] = term(Pl[r], T[r + 1], t1);      // u = +d
            Pn[r] = term(T[r], Pl[r + 1], t2); Qn[r] = term(Mi[r], T[r + 1], t2);      // u = -d
        }
        if (PRED)
        {
            const uint32_t sh = 30u - (uint32_t)d;
#pragma unroll
            for (int j = 0; j < R; j++) { X1[j] = (up[j] << sh) & 0x40000000u; X2[j] = (dn[j] << sh) & 0x40000000u; }
        }
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
}

// bit t of a mask row's window: a peak among columns start + t .. + 2
__device__ __forceinline__ uint64_t calc_dir_window16(const uint64_t *bits, int start, uint64_t lenmask)
{
    const int wq = start >> 6, sh = start & 63;
    const uint64_t lo = bits[wq], hi = bits[wq + 1];
    const uint64_t w = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
    return (w | (w >> 1) | (w >> 2)) & lenmask;
}

template <bool EDGE>
__device__ __forceinline__ int calc_dir_search16(const uint2 *tr, uint64_t pass, int maxdt, bool first, bool last, int nt13, int nt19,
                                                 const int *limlut, int neutral, int shift)
{
    // tr = &s_tri[j][b]: row r of the table is r * QLW further (r = 0..4: rows y-2 .. y+2)
    const uint2 F2p = tr[0], Fp = tr[QLW], Fc = tr[2 * QLW], Fn = tr[3 * QLW], F2n = tr[4 * QLW];
    const int ctr = (int)(Fc.x >> 16);
    const int vert = iabs16(ctr - (int)(Fn.x >> 16)) + iabs16(ctr - (int)(Fp.x >> 16));
    // keys: (running minimum << 6) | (u + 32), low six bits 0 = unset (the reference's -5000)
    uint32_t kb = (uint32_t)min(nt13, vert * 6) << 6, ka = (uint32_t)min(nt19, vert * 9) << 6;
    uint32_t kc = ka, kd = kb, ke = kb;
    while (pass)
    {
        const int jj = __ffsll((unsigned long long)pass) - 1;
        pass &= pass - 1ull;
        const int u = jj - maxdt;
        const uint32_t ub = (uint32_t)(u + 32);
        const uint2 *tp = tr + u, *tm = tr - u;
        const uint32_t e1 = sad3q(Fp, tm[2 * QLW], sad3q(Fc, tm[3 * QLW], 0u));     // diffsn + diffps
        const uint32_t d1 = sad3q(Fn, tp[2 * QLW], sad3q(Fc, tp[1 * QLW], 0u));     // diffsp + diffns
        const uint32_t diff = e1 + d1;
        uint32_t diffd = d1, diffe = e1;
        kb = min(kb, (diff << 6) | ub);
        if (!EDGE || !first)
        {
            const uint32_t diff2pp = sad3q(F2p, tm[1 * QLW], 0u);
            const uint32_t diffp2p = sad3q(Fp, tp[0 * QLW], 0u);
            diffd += diffp2p;
            diffe += diff2pp;
            ka = min(ka, ((diff + diff2pp + diffp2p) << 6) | ub);
        }
        if (!EDGE || !last)
        {
            const uint32_t diff2nn = sad3q(F2n, tp[3 * QLW], 0u);
            const uint32_t diffn2n = sad3q(Fn, tm[4 * QLW], 0u);
            diffd += diff2nn;
            diffe += diffn2n;
            kc = min(kc, ((diff + diff2nn + diffn2n) << 6) | ub);
        }
        kd = min(kd, (diffd << 6) | ub);
        ke = min(ke, (diffe << 6) | ub);
    }
    return calc_dir_vote16((int)(ka & 63u), (int)(kb & 63u), (int)(kc & 63u), (int)(kd & 63u), (int)(ke & 63u), limlut, neutral, shift);
}

// the two samples at columns X, X + 1 with those at or beyond `width` (the row's padding) replaced by padv: the reference's
// fill of calc_directions' output covers the padding, the bit_blit of the passes behind it does not (eedi2.hip: pad_bytes)
__device__ __forceinline__ uint32_t pad_pair16(uint32_t v, int X, int width, int padv)
{
    if (X >= width) return (uint32_t)padv * 0x00010001u;
    if (X + 1 >= width) return (v & 0xffffu) | ((uint32_t)padv << 16);
    return v;
}

// dense_min: a block with at least this many listed pixels (none of them on the plane's first / last row) searches in the
// dense form (R >= 4, depths up to 12 bits)

template <int R>
__global__ __launch_bounds__(QW) void q_calc_dir_rows(Q3 P, K16 k, int maxd, int nt, int dense_min, int padv)
{
    constexpr int NS = R + 4, NM = R + 2, RWD = QLW / 2;                      // RWD: dwords per staged row
    __shared__ __attribute__((aligned(16))) uint16_t s_band[NS + NM][QLW];    // staged rows: 0..NS-1 source y0-2.., NS.. mask y0-1..
    __shared__ uint2    s_tri[NS][QLW];                                       // [r][i] = samples i..i+2 of source row r
    __shared__ uint64_t s_pk[NM][8];                                          // bit i: mask row m holds the peak value at column i
    __shared__ uint16_t s_list[R * QW];                                       // the listed pixels, (row << 8) | column
    __shared__ __attribute__((aligned(16))) uint16_t s_out[R][QW];
    __shared__ int s_lim[33];
    __shared__ int s_count;
    FIELD16(P);
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    const int x0 = blockIdx.x * QW, y0 = blockIdx.y * R;
    if (y0 >= height || x0 >= pitch) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int peak = k.peak;
    if (maskless)
    {
        // no mask sample in the plane: the reference's fill with the peak value is all there is (:371)
        const uint32_t pk2 = (uint32_t)peak | ((uint32_t)peak << 16);
        for (int i = tid; i < R * (QW / 2); i += QW)
        {
            const int j = i / (QW / 2), c2 = i - j * (QW / 2), y = y0 + j, xb = x0 + 2 * c2;
            if (y < height && xb < pitch) *reinterpret_cast<uint32_t *>(Q.c + (size_t)y * pitch + xb) = pad_pair16(pk2, xb, width, padv);
        }
        return;
    }
    if (tid == 0) s_count = 0;
    if (tid < 33) s_lim[tid] = k.limlut[tid];
    if (tid >= 64 && tid < 64 + NM * 3) s_pk[(tid - 64) / 3][5 + (tid - 64) % 3] = 0;   // words past the staged columns
    uint32_t *band = reinterpret_cast<uint32_t *>(&s_band[0][0]);
    {
        // flat addressing (out-of-row columns pick up the neighbouring rows' samples, as the reference's pointer arithmetic
        // does); rows past height + 1 serve no pixel and are not touched
        const uint16_t *sb = Q.b + x0 - QHALO, *mb = Q.a + x0 - QHALO;
        for (int i = tid; i < (NS + NM) * RWD; i += QW)
        {
            const int r = i / RWD, c2 = i - r * RWD;
            const uint16_t *src = r < NS ? sb + (ptrdiff_t)min(y0 - 2 + r, height + 1) * pitch
                                         : mb + (ptrdiff_t)min(y0 - 1 + (r - NS), height) * pitch;
            band[i] = reinterpret_cast<const uint32_t *>(src)[c2];
        }
    }
    __syncthreads();
    // tables (columns 0 .. QLW-3) and peak bitmaps (columns 0 .. QLW-1): a first round for all, the rest for the first wave
    for (int rnd = 0; rnd < 2; rnd++)
    {
        if (rnd == 1 && tid >= 64) break;                            // wave-uniform
        const int i = tid + rnd * QW;
        if (i < QLW - 2)
        {
            const int q = i >> 1;
            const uint32_t sh = (uint32_t)(i & 1) * 2u;
#pragma unroll
            for (int r = 0; r < NS; r++)
            {
                const uint32_t d0 = band[r * RWD + q], d1 = band[r * RWD + q + 1];
                s_tri[r][i] = make_uint2(__builtin_amdgcn_alignbyte(d1, d0, sh), (d1 >> (8u * sh)) & 0xffffu);
            }
        }
#pragma unroll
        for (int m = 0; m < NM; m++)
        {
            const uint64_t w = __ballot(i < QLW && s_band[NS + m][i < QLW ? i : 0] == peak);   // the wave's 64 consecutive columns = one word
            if (lane == 0 && i < QLW) s_pk[m][i >> 6] = w;
        }
    }
    // the pixels that pass the edge test (:392-393), a row of the block per round, listed in column order; the rest of the
    // output is the peak value (the reference's memset)
#pragma unroll
    for (int j = 0; j < R; j++)
    {
        const int y = y0 + j, x = x0 + tid;
        const uint16_t *m = &s_band[NS + j + 1][tid + QHALO];
        const bool act = y >= 1 && y < height - 1 && x >= 1 && x < width - 1 && m[0] == peak && (m[-1] == peak || m[1] == peak);
        s_out[j][tid] = (uint16_t)peak;
        const uint64_t bal = __ballot(act);
        int base = 0;
        if (lane == 0 && bal) base = atomicAdd(&s_count, __popcll(bal));
        base = __builtin_amdgcn_readfirstlane(base);
        if (act) s_list[base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = (uint16_t)((j << 8) | tid);
    }
    __syncthreads();
    const int count = s_count;
    const int nt13 = (uint16_t)((nt << k.shift) * 13), nt19 = (uint16_t)((nt << k.shift) * 19);
    const int maxdt = pl == 0 ? maxd : (maxd >> 1);
    const int len = 2 * maxdt + 1;
    const uint64_t lenmask = (1ull << len) - 1ull;
    const bool edge = y0 <= 1 || y0 + R - 1 >= height - 2;
    if (R >= 4 && !edge && count >= dense_min)         // block-uniform
    {
        constexpr int RD = R >= 4 ? R : 1;
        const int lx = tid, px = x0 + lx, b = lx + QHALO - 1;
        const int startu = max(-px + 1, -maxdt), stopu = min(width - 2 - px, maxdt);
        uint64_t range = 0;
        if (stopu >= startu)
        {
            const int nb = stopu - startu + 1;
            range = (nb >= 64 ? ~0ull : ((1ull << nb) - 1ull)) << (startu + maxdt);
        }
        uint64_t win[RD + 2];
#pragma unroll
        for (int m = 0; m < RD + 2; m++) win[m] = calc_dir_window16(s_pk[m], b - maxdt, lenmask);
        uint32_t up[RD], dn[RD], act = 0, inactive = 0;
#pragma unroll
        for (int j = 0; j < RD; j++)
        {
            const uint16_t *m = &s_band[NS + j + 1][tid + QHALO];
            const bool a = px >= 1 && px < width - 1 && m[0] == peak && (m[-1] == peak || m[1] == peak);      // :392-393
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
            const uint2 *tr = &s_tri[0][b];
            if (__all(inactive == 0)) calc_dir_dense16<RD, false>(tr, maxdt, up, dn, nt13, nt19, ka, kb, kc, kd, ke);
            else                      calc_dir_dense16<RD, true>(tr, maxdt, up, dn, nt13, nt19, ka, kb, kc, kd, ke);
#pragma unroll
            for (int j = 0; j < RD; j++)
                if ((act >> j) & 1u)
                    s_out[j][lx] = (uint16_t)calc_dir_vote16((int)((ka[j] & 511u) / 6u), (int)((kb[j] & 511u) >> 2), (int)((kc[j] & 511u) / 6u),
                                                             (int)((kd[j] & 511u) >> 2), (int)((ke[j] & 511u) >> 2), s_lim, k.neutral, k.shift);
        }
        __syncthreads();
    }
    else if (count)                                   // block-uniform
    {
        for (int p = tid; p < count; p += QW)
        {
            const uint32_t id = s_list[p];
            const int j = (int)(id >> 8), lx = (int)(id & 255u), y = y0 + j, px = x0 + lx, b = lx + QHALO - 1;
            // The steps this pixel takes, as a bit set (bit jj: u = jj - maxdt): inside its range, and - unless on the first /
            // last row - with a mask peak among the three samples above at +u and below at -u (:395-399).  any3 bit t of a
            // row's window: a peak among columns b-maxdt+t .. +2; above it is used as it lies, below reversed.
            const int startu = max(-px + 1, -maxdt), stopu = min(width - 2 - px, maxdt);
            auto any3 = [&](const uint64_t *bits) {
                const int start = b - maxdt, wq = start >> 6, sh = start & 63;
                const uint64_t lo = bits[wq], hi = bits[wq + 1];
                const uint64_t w = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
                return (w | (w >> 1) | (w >> 2)) & lenmask;
            };
            uint64_t pass = 0;
            if (stopu >= startu)
            {
                const int nb = stopu - startu + 1;
                pass = (nb >= 64 ? ~0ull : ((1ull << nb) - 1ull)) << (startu + maxdt);
                if (y != 1)          pass &= any3(s_pk[j]);
                if (y != height - 2) pass &= __brevll(any3(s_pk[j + 2])) >> (64 - len);
            }
            const uint2 *tr = &s_tri[j][b];
            const int out = edge ? calc_dir_search16<true>(tr, pass, maxdt, y == 1, y == height - 2, nt13, nt19, s_lim, k.neutral, k.shift)
                                 : calc_dir_search16<false>(tr, pass, maxdt, false, false, nt13, nt19, s_lim, k.neutral, k.shift);
            s_out[j][lx] = (uint16_t)out;
        }
        __syncthreads();
    }
    // output rows as dwords (two samples); the whole pitch is written, as the one-thread form does
#pragma unroll
    for (int i = tid; i < R * (QW / 2); i += QW)
    {
        const int j = i / (QW / 2), c2 = i - j * (QW / 2), y = y0 + j, xb = x0 + 2 * c2;
        if (y < height && xb < pitch)
            *reinterpret_cast<uint32_t *>(Q.c + (size_t)y * pitch + xb) = pad_pair16(reinterpret_cast<const uint32_t *>(&s_out[j][0])[c2], xb, width, padv);
    }
}

// filter_dir_map (:649-709) / expand_dir_map (:722-773) and, with step 2, the _2x forms (:872-1011).
// a = mask, b = direction map in, c = out.  step 1: rows 1..height-2 looking at rows y+-1 and mask row y;
// step 2: rows y0, y0+2, ... looking at rows y+-2 and mask rows y-1 / y+1.
// (the candidates sit in nine fixed slots - an absent one holds ABSENT16, larger than any sample and farther from any
// midpoint than any vote limit - and a sorting network orders them: no data-dependent loop, no indexed array; the
// midpoint and the vote are sorted_mid16's / vote16's, as in the 8-bit k_dir_map)
constexpr int ABSENT16 = 1 << 20;

__device__ __forceinline__ void cswap16(int &a, int &b)
{
    const int lo = min(a, b), hi = max(a, b);
    a = lo; b = hi;
}

// midpoint of the n present values among 9 slots (n >= 4)
__device__ __forceinline__ int mid9q(int &v0, int &v1, int &v2, int &v3, int &v4, int &v5, int &v6, int &v7, int &v8, int n)
{
    cswap16(v0, v3); cswap16(v1, v7); cswap16(v2, v5); cswap16(v4, v8);
    cswap16(v0, v7); cswap16(v2, v4); cswap16(v3, v8); cswap16(v5, v6);
    cswap16(v0, v2); cswap16(v1, v3); cswap16(v4, v5); cswap16(v7, v8);
    cswap16(v1, v4); cswap16(v3, v6); cswap16(v5, v7);
    cswap16(v0, v1); cswap16(v2, v4); cswap16(v3, v5); cswap16(v6, v8);
    cswap16(v2, v3); cswap16(v4, v5); cswap16(v6, v7);
    cswap16(v1, v2); cswap16(v3, v4); cswap16(v5, v6);
    // n = 4..9: lower middle index (n-1)>>1 = 1,2,2,3,3,4 ; upper n>>1 = 2,2,3,3,4,4
    const int lo = n <= 4 ? v1 : (n <= 6 ? v2 : (n <= 8 ? v3 : v4));
    const int hi = n <= 5 ? v2 : (n <= 7 ? v3 : v4);
    return (n & 1) ? hi : (lo + hi + 1) >> 1;
}

// midpoint of the n present values among 6 slots (n >= 3)
__device__ __forceinline__ int mid6q(int &v0, int &v1, int &v2, int &v3, int &v4, int &v5, int n)
{
    cswap16(v0, v5); cswap16(v1, v3); cswap16(v2, v4);
    cswap16(v1, v2); cswap16(v3, v4);
    cswap16(v0, v3); cswap16(v2, v5);
    cswap16(v0, v1); cswap16(v2, v3); cswap16(v4, v5);
    cswap16(v1, v2); cswap16(v3, v4);
    // n = 3..6: lower middle index 1,1,2,2 ; upper 1,2,2,3
    const int lo = n <= 4 ? v1 : v2;
    const int hi = n <= 3 ? v1 : (n <= 5 ? v2 : v3);
    return (n & 1) ? hi : (lo + hi + 1) >> 1;
}

__device__ __forceinline__ void vote1q(int v, int mid, int lim, int &sum, int &cnt)
{
    const bool in = iabs16(v - mid) <= lim;      // never true for ABSENT16
    cnt += in;
    sum += in ? v : 0;
}

// one sample of filter_dir_map / expand_dir_map (:649-773, _2x :872-1011) whose neighbourhood holds enough directions
__device__ __forceinline__ int dir_map_px16(int u0, int u1, int u2, int c0, int c1, int c2, int n0, int n1, int n2,
                                            bool up_ok, bool dn_ok, int expand, int peak, int neutral, int sh2, const int *limlut)
{
    const bool h0 = up_ok && u0 != peak, h1 = up_ok && u1 != peak, h2 = up_ok && u2 != peak;
    const bool h3 = c0 != peak, h4 = !expand && c1 != peak, h5 = c2 != peak;
    const bool h6 = dn_ok && n0 != peak, h7 = dn_ok && n1 != peak, h8 = dn_ok && n2 != peak;
    const int u = h0 + h1 + h2 + h3 + h4 + h5 + h6 + h7 + h8;
    if (u < (expand ? 5 : 4)) return expand ? c1 : peak;
    int v0 = h0 ? u0 : ABSENT16, v1 = h1 ? u1 : ABSENT16, v2 = h2 ? u2 : ABSENT16;
    int v3 = h3 ? c0 : ABSENT16, v4 = h4 ? c1 : ABSENT16, v5 = h5 ? c2 : ABSENT16;
    int v6 = h6 ? n0 : ABSENT16, v7 = h7 ? n1 : ABSENT16, v8 = h8 ? n2 : ABSENT16;
    const int mid = mid9q(v0, v1, v2, v3, v4, v5, v6, v7, v8, u);
    const int lim = limlut[iabs16(mid - neutral) >> sh2];
    int sum = 0, count = 0;
    vote1q(v0, mid, lim, sum, count); vote1q(v1, mid, lim, sum, count); vote1q(v2, mid, lim, sum, count);
    vote1q(v3, mid, lim, sum, count); vote1q(v4, mid, lim, sum, count); vote1q(v5, mid, lim, sum, count);
    vote1q(v6, mid, lim, sum, count); vote1q(v7, mid, lim, sum, count); vote1q(v8, mid, lim, sum, count);
    const int val = (int)(((float)(sum + mid) / (float)(count + 1)) + 0.5f);
    if (expand) return count >= 5 ? val : c1;
    if (count < 4 || (count < 5 && c1 == peak)) return peak;
    return val;
}

// ---- filter_dir_map / _2x in registers, the vote on PAIRS of samples (eedi2.hip: dir_map_pair, k_dir_map4) ----------------
// With a dense mask nearly every sample of the pass reaches its sort and its vote; two horizontally adjacent samples ride
// in the halves of a dword - which is how 16-bit samples lie in memory, so the nine slots of a pair are three dwords of
// each row as they are or realigned by two bytes, no unpacking.  An absent slot (a peak; a row that does not count)
// holds 0x7fff: above every sample and 0x7000 or more from every midpoint, the cap of the vote's limit.
typedef uint16_t u16x2d __attribute__((ext_vector_type(2)));
typedef int16_t i16x2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2d pkd(uint32_t v) { return __builtin_bit_cast(u16x2d, v); }
__device__ __forceinline__ uint32_t und(u16x2d v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ u16x2d pkd1(uint32_t both) { return pkd(both * 0x00010001u); }
__device__ __forceinline__ u16x2d pkd_lt(u16x2d a, u16x2d b) { return (u16x2d)((u16x2d)(a - b) >> 15); }   // halves below 2^15
__device__ __forceinline__ void cswap2d(u16x2d &a, u16x2d &b)
{
    const u16x2d lo = __builtin_elementwise_min(a, b), hi = __builtin_elementwise_max(a, b);
    a = lo; b = hi;
}
struct Win16 { uint32_t w0, w1, w2, w3; };       // samples x-2 .. x+5 of a row

// the pair of samples at columns K, K + 1 of the thread's four (K = 0 or 2); returns the pass's values for both halves
template <int K>
__device__ __forceinline__ uint32_t dir_map_pair16(const Win16 &wu, const Win16 &wc, const Win16 &wd, int expand, int peak, int neutral, int shift)
{
    u16x2d v[9];
    {
        const Win16 *rows[3] = { &wu, &wc, &wd };
#pragma unroll
        for (int r = 0; r < 3; r++)
        {
            const Win16 &w = *rows[r];
            if (K == 0)
            {
                v[3 * r + 0] = pkd(__builtin_amdgcn_alignbyte(w.w1, w.w0, 2u));       // columns -1, 0
                v[3 * r + 1] = pkd(w.w1);                                             // 0, 1
                v[3 * r + 2] = pkd(__builtin_amdgcn_alignbyte(w.w2, w.w1, 2u));       // 1, 2
            }
            else
            {
                v[3 * r + 0] = pkd(__builtin_amdgcn_alignbyte(w.w2, w.w1, 2u));       // 1, 2
                v[3 * r + 1] = pkd(w.w2);                                             // 2, 3
                v[3 * r + 2] = pkd(__builtin_amdgcn_alignbyte(w.w3, w.w2, 2u));       // 3, 4
            }
        }
    }
    const u16x2d c1 = v[4];
    // (a direction value can lie ABOVE the peak: neutral + (32 << (2 + shift)) is one more than it - a search distance of 32 gets
    // there - so the peak is found by equality)
    const u16x2d peakv = pkd1((uint32_t)peak), lift = pkd1(0x7fffu - (uint32_t)peak);
    uint32_t absent = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
    {
        const u16x2d a = pkd_lt(v[i] ^ peakv, pkd1(1));                 // 1 per half that holds the peak
        absent += und(a);
        v[i] = v[i] + a * lift;
    }
    u16x2d s0 = v[0], s1 = v[1], s2 = v[2], s3 = v[3], s4 = v[4], s5 = v[5], s6 = v[6], s7 = v[7], s8 = v[8];
    cswap2d(s0, s3); cswap2d(s1, s7); cswap2d(s2, s5); cswap2d(s4, s8);
    cswap2d(s0, s7); cswap2d(s2, s4); cswap2d(s3, s8); cswap2d(s5, s6);
    cswap2d(s0, s2); cswap2d(s1, s3); cswap2d(s4, s5); cswap2d(s7, s8);
    cswap2d(s1, s4); cswap2d(s3, s6); cswap2d(s5, s7);
    cswap2d(s0, s1); cswap2d(s2, s4); cswap2d(s3, s5); cswap2d(s6, s8);
    cswap2d(s2, s3); cswap2d(s4, s5); cswap2d(s6, s7);
    cswap2d(s1, s2); cswap2d(s3, s4); cswap2d(s5, s6);
    const u16x2d ab = pkd(absent), one = pkd1(1), zero = pkd1(0);
    const uint32_t m5 = und(zero - pkd_lt(pkd1(3), ab));                // n <= 5 <=> absent >= 4
    const uint32_t m7 = und(zero - pkd_lt(one, ab));                    // n <= 7 <=> absent >= 2
    const uint32_t modd = und((ab & one) - one);                        // n odd <=> absent even
#define PKD_SEL(m, x, y) (((m) & (x)) | (~(m) & (y)))
    const uint32_t hi = PKD_SEL(m5, und(s2), PKD_SEL(m7, und(s3), und(s4)));
    const uint32_t lo = PKD_SEL(m5, und(s1), PKD_SEL(m7, und(s2), und(s3)));
    const u16x2d mid = pkd(PKD_SEL(modd, hi, und((u16x2d)((pkd(lo) + pkd(hi) + one) >> 1))));
#undef PKD_SEL
    // the limit: (limlut[i] << shift) with limlut in closed form (eedi2.hip: limlut2), i = |mid - neutral| >> (2 + shift); the two
    // last entries (a -1 stored as a sample: every present value is in) capped at 0x7000
    const i16x2d t = __builtin_bit_cast(i16x2d, (u16x2d)(mid - pkd1((uint32_t)neutral)));
    const u16x2d ii = __builtin_bit_cast(u16x2d, __builtin_elementwise_max(t, (i16x2d)(-t))) >> (uint16_t)(2 + shift);
    const u16x2d g = pkd_lt(pkd1(7), ii);
    const u16x2d l8 = __builtin_elementwise_min((u16x2d)(((ii - g) >> 1) + pkd1(6)), pkd1(12));
    const u16x2d lim1 = __builtin_elementwise_max((u16x2d)((l8 << (uint16_t)shift) + one), (u16x2d)(pkd_lt(pkd1(30), ii) * pkd1(0x7000)));
    u16x2d sum = zero, cnt = zero;
#pragma unroll
    for (int i = 0; i < 9; i++)
    {
        const u16x2d d = __builtin_elementwise_max(v[i], mid) - __builtin_elementwise_min(v[i], mid);
        const u16x2d in = pkd_lt(d, lim1);
        cnt += in;
        sum += in * v[i];
    }
    const uint32_t sm = und((u16x2d)(sum + mid)), ct = und(cnt);
    uint32_t out = 0;
#pragma unroll
    for (int h = 0; h < 2; h++)
    {
        const int n = 9 - (int)((absent >> (16 * h)) & 0xffffu);
        const int count = n >= 4 ? (int)((ct >> (16 * h)) & 0xffffu) : 0;              // (fewer: the midpoint may be an absent slot)
        const int val = (int)(((float)((sm >> (16 * h)) & 0xffffu) / (float)(count + 1)) + 0.5f);
        const int c = (int)((und(c1) >> (16 * h)) & 0xffffu);
        int res;
        if (expand) res = count >= 5 ? val : c;
        else        res = (count < 4 || (count < 5 && c == peak)) ? peak : val;
        out |= ((uint32_t)res & 0xffffu) << (16 * h);
    }
    return out;
}

// a = mask, b = direction map in, c = out.  Four samples per thread; in the _2x form a thread takes the rows 2r and 2r + 1 -
// the one with the rebuilt rows' parity is worked on, the other copied (eedi2.hip: k_dir_map4)
__global__ __launch_bounds__(256) void q_dir_map4(Q3 P, K16 k, int step, int expand)
{
    FIELD16(P);
    const int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    const int r = blockIdx.y * blockDim.y + threadIdx.y;
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    const int y0 = step == 1 ? 1 : 2 - tff;
    const int y = step == 1 ? r : 2 * r + (y0 & 1);
    if (x >= width) return;
    const int peak = k.peak;
    const uint32_t peak2 = (uint32_t)peak * 0x00010001u;
    auto put = [&](int yy, uint2 v) {
        uint16_t *o = Q.c + (size_t)yy * pitch + x;
        if (x + 3 < width) *reinterpret_cast<uint2 *>(o) = v;
        else
        {
            const uint16_t o4[4] = { (uint16_t)(v.x & 0xffffu), (uint16_t)(v.x >> 16), (uint16_t)(v.y & 0xffffu), (uint16_t)(v.y >> 16) };
            for (int j = 0; j < 4 && x + j < width; j++) o[j] = o4[j];
        }
    };
    if (maskless)
    {
        const int ya = step == 1 ? r : 2 * r, nrows = step == 1 ? 1 : 2;
        for (int i = 0; i < nrows && ya + i < height; i++) put(ya + i, make_uint2(peak2, peak2));
        return;
    }
    const int yc = 2 * r + 1 - (y0 & 1);
    const bool copy = step != 1 && yc < height;
    uint2 vcopy = make_uint2(0u, 0u);
    if (copy) vcopy = *reinterpret_cast<const uint2 *>(Q.b + (size_t)yc * pitch + x);
    if (y >= height) { if (copy) put(yc, vcopy); return; }
    const uint16_t *dc = Q.b + (size_t)y * pitch + x;
    const bool row_ok = step == 1 ? (y >= 1 && y < height - 1) : (y >= y0 && y < height - 1);
    const uint2 own = *reinterpret_cast<const uint2 *>(dc);
    if (!row_ok)
    {
        put(y, own);
        if (copy) put(yc, vcopy);
        return;
    }
    const uint16_t *mk = Q.a + (size_t)y * pitch + x;
    const uint2 m0 = *reinterpret_cast<const uint2 *>(step == 1 ? mk : mk - (ptrdiff_t)pitch);
    const uint2 m1 = step == 1 ? make_uint2(0u, 0u) : *reinterpret_cast<const uint2 *>(mk + pitch);
    const bool up_ok = step == 1 || y > 1, dn_ok = step == 1 || y < height - 2;
    // the samples the pass works on (:658 / :738): inside the row, on the mask, and for expand a peak
    const int mm0[4] = { (int)(m0.x & 0xffffu), (int)(m0.x >> 16), (int)(m0.y & 0xffffu), (int)(m0.y >> 16) };
    const int mm1[4] = { (int)(m1.x & 0xffffu), (int)(m1.x >> 16), (int)(m1.y & 0xffffu), (int)(m1.y >> 16) };
    const int oo[4] = { (int)(own.x & 0xffffu), (int)(own.x >> 16), (int)(own.y & 0xffffu), (int)(own.y >> 16) };
    uint32_t work = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
        const int xx = x + j;
        const bool masked = mm0[j] == peak || (step != 1 && mm1[j] == peak);
        if (xx >= 1 && xx < width - 1 && masked && !(expand && oo[j] != peak)) work |= 1u << j;
    }
    uint2 res = own;
    if (work)
    {
        asm volatile("" ::: "memory");                         // a real branch: waves without such a sample skip the loads and the votes
        auto ldwin = [&](const uint16_t *row) -> Win16 {
            const uint2 c4 = *reinterpret_cast<const uint2 *>(row);
            const uint32_t l = x >= 2 ? *reinterpret_cast<const uint32_t *>(row - 2) : 0u;
            const uint32_t rr = x + 4 < pitch ? *reinterpret_cast<const uint32_t *>(row + 4) : 0u;
            return Win16{ l, c4.x, c4.y, rr };
        };
        const Win16 none = { peak2, peak2, peak2, peak2 };
        const Win16 wc = ldwin(dc);
        const Win16 wu = up_ok ? ldwin(dc - (ptrdiff_t)step * pitch) : none, wd = dn_ok ? ldwin(dc + (ptrdiff_t)step * pitch) : none;
        const uint32_t p01 = dir_map_pair16<0>(wu, wc, wd, expand, peak, k.neutral, k.shift);
        const uint32_t p23 = dir_map_pair16<2>(wu, wc, wd, expand, peak, k.neutral, k.shift);
        const uint32_t s01 = ((work & 1u) ? 0x0000ffffu : 0u) | ((work & 2u) ? 0xffff0000u : 0u);
        const uint32_t s23 = ((work & 4u) ? 0x0000ffffu : 0u) | ((work & 8u) ? 0xffff0000u : 0u);
        res.x = (res.x & ~s01) | (p01 & s01);
        res.y = (res.y & ~s23) | (p23 & s23);
    }
    put(y, res);
    if (copy) put(yc, vcopy);
}

// Two phases, as for 8-bit samples (eedi2.hip: k_dir_map_c).  Phase 1, four samples per thread: which samples reach the
// sort at all (inside the mask, enough usable directions around them) - on real pictures a small minority, spread so
// that nearly every wave holds a few, and a wave pays for the sort if one lane needs it.  They are queued (an LDS list
// per workgroup of 4 rows x 256 samples) and phase 2 gives each queued sample a lane of its own.  a = edge mask,
// b = direction map in, c = out; step 1 (half height) or 2.
// post != 0 (the last expand_dir_map_2x of a field): eedi2_post_process (:1349-1378, q_post) rides along, as in the
// 8-bit engine - it is pointwise in the map this pass has just made (e = the map before the post filters, f = dst2p,
// rebuilt rows only; the rows it averages are of the other parity, which no thread of the pass writes).
__global__ __launch_bounds__(256) void q_dir_map(Q3 P, K16 k, int step, int expand, int post)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_out[4][256];
    __shared__ uint16_t s_list[4 * 256];
    __shared__ int s_count;
    __shared__ int s_lim[33];
    FIELD16(P);
    const int y0 = step == 1 ? 1 : 2 - tff;
    // step 2: a thread row takes the PAIR of rows 2r, 2r + 1 - the one with the rebuilt rows' parity goes through the pass,
    // the other is only copied: fetched now, stored when the workgroup is done (eedi2.hip: k_dir_map_c)
    const int rb = blockIdx.y * 4, r = rb + threadIdx.y;
    const int bx0 = 4 * (blockIdx.x * 64), x = bx0 + 4 * threadIdx.x, y = step == 1 ? r : 2 * r + (y0 & 1);
    const int yb = step == 1 ? rb : 2 * rb + (y0 & 1);                              // row of thread row 0
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    if (bx0 >= width || (step == 1 ? rb : 2 * rb) >= height) return;                 // whole workgroup outside
    if (maskless)
    {
        // no mask sample in the plane: the pass is its bit_blit of a map of peaks (eedi2.hip: k_dir_map4)
        if (x < width)
        {
            const int ya = step == 1 ? r : 2 * r, nrows = step == 1 ? 1 : 2;
            for (int i = 0; i < nrows && ya + i < height; i++)
                for (int j = 0; j < 4 && x + j < width; j++) Q.c[(size_t)(ya + i) * pitch + x + j] = (uint16_t)k.peak;
        }
        return;
    }
    const int tid = threadIdx.y * 64 + threadIdx.x;
    if (tid == 0) s_count = 0;
    if (tid < 33) s_lim[tid] = k.limlut[tid];
    __syncthreads();
    const int peak = k.peak;
    const int yc = 2 * r + 1 - (y0 & 1);
    const bool copy = step != 1 && x < width && yc < height;
    uint2 vcopy = make_uint2(0u, 0u);
    if (copy) vcopy = *reinterpret_cast<const uint2 *>(Q.b + (size_t)yc * pitch + x);
    const bool inside = x < width && y < height;
    const bool row_ok = step == 1 ? (y >= 1 && y < height - 1) : (y >= y0 && y < height - 1);
    const uint16_t *dc = Q.b + (size_t)y * pitch + x;
    if (inside)
    {
        uint2 out = *reinterpret_cast<const uint2 *>(dc);
        if (row_ok)
        {
            const bool up_ok = step == 1 || y > 1, dn_ok = step == 1 || y < height - 2;
            // six samples x-1 .. x+4 of the three rows: usable (non-peak) flags, bit j = sample x-1+j
            auto usable6 = [&](const uint16_t *row) -> uint32_t {
                const uint2 v = *reinterpret_cast<const uint2 *>(row);
                const int l = row[-1], r = row[4];
                return (l != peak ? 1u : 0u) | ((int)(v.x & 0xffffu) != peak ? 2u : 0u) | ((int)(v.x >> 16) != peak ? 4u : 0u) |
                       ((int)(v.y & 0xffffu) != peak ? 8u : 0u) | ((int)(v.y >> 16) != peak ? 16u : 0u) | (r != peak ? 32u : 0u);
            };
            const uint32_t fc = usable6(dc);
            const uint32_t fu = up_ok ? usable6(dc - (ptrdiff_t)step * pitch) : 0u;
            const uint32_t fd = dn_ok ? usable6(dc + (ptrdiff_t)step * pitch) : 0u;
            const uint16_t *mk = Q.a + (size_t)y * pitch + x;
            const uint2 m0 = *reinterpret_cast<const uint2 *>(step == 1 ? mk : mk - (ptrdiff_t)pitch);
            const uint2 m1 = step == 1 ? make_uint2(0u, 0u) : *reinterpret_cast<const uint2 *>(mk + pitch);
            const int mm0[4] = { (int)(m0.x & 0xffffu), (int)(m0.x >> 16), (int)(m0.y & 0xffffu), (int)(m0.y >> 16) };
            const int mm1[4] = { (int)(m1.x & 0xffffu), (int)(m1.x >> 16), (int)(m1.y & 0xffffu), (int)(m1.y >> 16) };
            uint32_t o4[4] = { out.x & 0xffffu, out.x >> 16, out.y & 0xffffu, out.y >> 16 };
            uint32_t sortpx = 0;
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                const int xx = x + j;
                const bool masked = mm0[j] == peak || (step != 1 && mm1[j] == peak);
                const bool centre_usable = (fc >> (j + 1)) & 1u;
                const bool cand = xx >= 1 && xx < width - 1 && masked && !(expand && centre_usable);
                // usable values among the 3 x 3 (expand leaves the centre out, :671)
                const int u = __popc((fu >> j) & 7u) + __popc((fd >> j) & 7u) + __popc((fc >> j) & (expand ? 5u : 7u));
                const bool enough = u >= (expand ? 5 : 4);
                if (cand && !enough && !expand) o4[j] = (uint32_t)peak;               // too few neighbours: peak
                if (cand && enough) sortpx |= 1u << j;
            }
            out = make_uint2(o4[0] | (o4[1] << 16), o4[2] | (o4[3] << 16));
            if (sortpx)
            {
                int at = atomicAdd(&s_count, __popc(sortpx));
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if ((sortpx >> j) & 1u) s_list[at++] = (uint16_t)((threadIdx.y << 8) | (4 * threadIdx.x + j));
            }
        }
        *reinterpret_cast<uint2 *>(&s_out[threadIdx.y][4 * threadIdx.x]) = out;
    }
    __syncthreads();
    const int count = s_count;
    for (int i = tid; i < count; i += 256)
    {
        const int e = s_list[i], ly = e >> 8, lx = e & 255;
        const int yy = yb + step * ly;
        const uint16_t *c = Q.b + (size_t)yy * pitch + bx0 + lx;
        const bool up_ok = step == 1 || yy > 1, dn_ok = step == 1 || yy < height - 2;
        const uint16_t *up = up_ok ? c - (ptrdiff_t)step * pitch : c, *dn = dn_ok ? c + (ptrdiff_t)step * pitch : c;
        s_out[ly][lx] = (uint16_t)dir_map_px16(up[-1], up[0], up[1], c[-1], c[0], c[1], dn[-1], dn[0], dn[1], up_ok, dn_ok, expand,
                                               peak, k.neutral, 2 + k.shift, s_lim);
    }
    __syncthreads();
    if (inside)
    {
        const uint2 v = *reinterpret_cast<const uint2 *>(&s_out[threadIdx.y][4 * threadIdx.x]);
        uint16_t *o = Q.c + (size_t)y * pitch + x;
        if (x + 3 < width) *reinterpret_cast<uint2 *>(o) = v;
        else
        {
            const uint16_t o4[4] = { (uint16_t)(v.x & 0xffffu), (uint16_t)(v.x >> 16), (uint16_t)(v.y & 0xffffu), (uint16_t)(v.y >> 16) };
            for (int j = 0; j < 4 && x + j < width; j++) o[j] = o4[j];
        }
        if (post && row_ok)
        {
            const size_t at = (size_t)fld * P.fstride + (size_t)y * pitch + x;
            const uint2 om4 = *reinterpret_cast<const uint2 *>(P.e[pl] + at);
            uint16_t *d = P.f[pl] + at;
            const uint2 up4 = *reinterpret_cast<const uint2 *>(d - pitch), dn4 = *reinterpret_cast<const uint2 *>(d + pitch);
            const uint2 cur4 = *reinterpret_cast<const uint2 *>(d);
            auto s4 = [](const uint2 &w, int j) -> int { return (int)(((j < 2 ? w.x : w.y) >> (16 * (j & 1))) & 0xffffu); };
            uint32_t out[4];
            bool any = false;
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                const int nm = s4(v, j), om = s4(om4, j);
                const int lim = s_lim[iabs16(nm - k.neutral) >> (2 + k.shift)];
                const bool fix = iabs16(nm - om) > lim && om != k.peak && om != k.neutral;
                out[j] = fix ? (uint32_t)((s4(up4, j) + s4(dn4, j) + 1) >> 1) : (uint32_t)s4(cur4, j);
                any |= fix;
            }
            if (any)
            {
                if (x + 3 < width) *reinterpret_cast<uint2 *>(d) = make_uint2(out[0] | (out[1] << 16), out[2] | (out[3] << 16));
                else for (int j = 0; j < 4 && x + j < width; j++) d[j] = (uint16_t)out[j];
            }
        }
    }
    if (copy)
    {
        uint16_t *o = Q.c + (size_t)yc * pitch + x;
        if (x + 3 < width) *reinterpret_cast<uint2 *>(o) = vcopy;
        else
        {
            const uint16_t o4[4] = { (uint16_t)(vcopy.x & 0xffffu), (uint16_t)(vcopy.x >> 16), (uint16_t)(vcopy.y & 0xffffu), (uint16_t)(vcopy.y >> 16) };
            for (int j = 0; j < 4 && x + j < width; j++) o[j] = o4[j];
        }
    }
}

// filter_dir_map and the expand_dir_map behind it in one launch (eedi2.hip: k_dir_map_fe): a workgroup makes the filtered
// map of its QFE_R rows x 256 samples and of the one-sample ring around them in LDS (q_dir_map4's vote), expands out of
// that (q_dir_map's two phases) and stores the expanded map only.  a = mask, b = map in, c = out (not b).  STEP 1: the half-
// height pair; STEP 2: the _2x pair on the lattice of the rows it rebuilds (row r = plane row 2 r + (y0 & 1), neighbours
// y -+ 2 where they exist, mask rows y - 1 / y + 1), the plane rows between them copied.
#ifndef FE16_ROWS
#define FE16_ROWS 14
#endif
constexpr int QFE_R = FE16_ROWS, QFE_LW = 272;                      // LDS row: the group of 4 left of the tile, 256 samples, the group right of it
// POST (the pair in front of eedi2_post_process, which rides along as in q_dir_map): d = where the filtered map goes as
// well, f = the picture, the old map = the pass's own input (eedi2.hip: k_dir_map_fe<2, true>); padv = what the padding of
// the output's rows gets.
template <int STEP, bool POST>
__global__ __launch_bounds__(256) void q_dir_map_fe(Q3 P, K16 k, int padv)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_f[QFE_R + 2][QFE_LW];      // lattice rows rb - 1 .. rb + QFE_R
    __shared__ __attribute__((aligned(16))) uint16_t s_out[QFE_R][256];
    __shared__ uint16_t s_list[QFE_R * 256];
    __shared__ int s_count;
    __shared__ int s_lim[33];
    static_assert(QFE_R + 2 <= 64, "a lane per ring row");
    FIELD16(P);
    const int y0 = STEP == 1 ? 1 : 2 - tff, par = STEP == 1 ? 0 : (y0 & 1);
    const int rb = blockIdx.y * QFE_R;
    const int bx0 = 256 * blockIdx.x, x = bx0 + 4 * threadIdx.x;
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    if (bx0 >= pitch || STEP * rb >= height) return;
    const int peak = k.peak;
    const uint32_t peak2 = (uint32_t)peak * 0x00010001u;
    auto row_ok = [&](int y) { return STEP == 1 ? (y >= 1 && y < height - 1) : (y >= y0 && y < height - 1); };
    if (maskless)
    {
        if (x < pitch)
            for (int lr = threadIdx.y; lr < QFE_R; lr += 4)
#pragma unroll
                for (int i = 0; i < STEP; i++)
                    if (STEP * (rb + lr) + i < height)
                    {
                        const size_t at = (size_t)(STEP * (rb + lr) + i) * pitch + x;
                        *reinterpret_cast<uint2 *>(Q.c + at) = make_uint2(pad_pair16(peak2, x, width, padv), pad_pair16(peak2, x + 2, width, padv));
                        if (POST)
                            for (int j = 0; j < 4 && x + j < width; j++) (P.d[pl] + (size_t)fld * P.fstride + at)[j] = (uint16_t)peak;
                    }
        return;
    }
    const int tid = threadIdx.y * 64 + threadIdx.x;
    if (tid == 0) s_count = 0;
    if (tid < 33) s_lim[tid] = k.limlut[tid];
    // the rows between the lattice's: fetched now, stored at the end
    constexpr int NCOPY = STEP == 1 ? 1 : (QFE_R + 3) / 4;
    uint2 vcopy[NCOPY];
    if (STEP != 1)
#pragma unroll
        for (int h = 0; h < NCOPY; h++)
        {
            const int lr = threadIdx.y + 4 * h, yc = 2 * (rb + lr) + 1 - par;
            vcopy[h] = (lr < QFE_R && yc < height && x < width) ? *reinterpret_cast<const uint2 *>(Q.b + (size_t)yc * pitch + x) : make_uint2(0u, 0u);
        }
    auto masked4 = [&](int y, int xx) -> uint32_t {                // bit j: sample xx + j is inside the row and under / above a mask sample
        const uint16_t *mk = Q.a + (size_t)y * pitch + xx;
        const uint2 m0 = *reinterpret_cast<const uint2 *>(STEP == 1 ? mk : mk - (ptrdiff_t)pitch);
        const uint2 m1 = STEP == 1 ? make_uint2(0u, 0u) : *reinterpret_cast<const uint2 *>(mk + pitch);
        const int mm0[4] = { (int)(m0.x & 0xffffu), (int)(m0.x >> 16), (int)(m0.y & 0xffffu), (int)(m0.y >> 16) };
        const int mm1[4] = { (int)(m1.x & 0xffffu), (int)(m1.x >> 16), (int)(m1.y & 0xffffu), (int)(m1.y >> 16) };
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (xx + j >= 1 && xx + j < width - 1 && (mm0[j] == peak || (STEP != 1 && mm1[j] == peak))) w |= 1u << j;
        return w;
    };
    // SIDE 0: the thread's four samples; -1 / +1: the ring's group left / right of the tile, of which only the sample next to
    // the tile is ever read (half the vote)
    auto filtered = [&](int r, int xx, auto side) -> uint2 {
        constexpr int SIDE = decltype(side)::value;
        const int y = STEP * r + par;
        if (r < 0 || y >= height || xx < 0 || xx >= width) return make_uint2(0u, 0u);
        const uint16_t *dc = Q.b + (size_t)y * pitch + xx;
        const uint2 own = *reinterpret_cast<const uint2 *>(dc);
        if (!row_ok(y)) return own;
        uint32_t work = masked4(y, xx);
        if (SIDE < 0) work &= 8u;
        if (SIDE > 0) work &= 1u;
        uint2 res = own;
        if (work)
        {
            asm volatile("" ::: "memory");
            auto ldwin = [&](const uint16_t *row) -> Win16 {
                const uint2 c4 = *reinterpret_cast<const uint2 *>(row);
                const uint32_t l = xx >= 2 ? *reinterpret_cast<const uint32_t *>(row - 2) : 0u;
                const uint32_t rr = xx + 4 < pitch ? *reinterpret_cast<const uint32_t *>(row + 4) : 0u;
                return Win16{ l, c4.x, c4.y, rr };
            };
            const bool up_ok = STEP == 1 || y > 1, dn_ok = STEP == 1 || y < height - 2;
            const Win16 none = { peak2, peak2, peak2, peak2 };
            const Win16 wc = ldwin(dc);
            const Win16 wu = up_ok ? ldwin(dc - (ptrdiff_t)STEP * pitch) : none, wd = dn_ok ? ldwin(dc + (ptrdiff_t)STEP * pitch) : none;
            const uint32_t p01 = SIDE < 0 ? 0u : dir_map_pair16<0>(wu, wc, wd, 0, peak, k.neutral, k.shift);
            const uint32_t p23 = SIDE > 0 ? 0u : dir_map_pair16<2>(wu, wc, wd, 0, peak, k.neutral, k.shift);
            const uint32_t s01 = ((work & 1u) ? 0x0000ffffu : 0u) | ((work & 2u) ? 0xffff0000u : 0u);
            const uint32_t s23 = ((work & 4u) ? 0x0000ffffu : 0u) | ((work & 8u) ? 0xffff0000u : 0u);
            res.x = (res.x & ~s01) | (p01 & s01);
            res.y = (res.y & ~s23) | (p23 & s23);
        }
        return res;
    };
    for (int lr = threadIdx.y; lr < QFE_R + 2; lr += 4)
        *reinterpret_cast<uint2 *>(&s_f[lr][4 + 4 * threadIdx.x]) = filtered(rb - 1 + lr, x, std::integral_constant<int, 0>());
    if (threadIdx.y == 0 && (int)threadIdx.x < QFE_R + 2)
        *reinterpret_cast<uint2 *>(&s_f[threadIdx.x][0]) = filtered(rb - 1 + (int)threadIdx.x, bx0 - 4, std::integral_constant<int, -1>());
    if (threadIdx.y == 1 && (int)threadIdx.x < QFE_R + 2)
        *reinterpret_cast<uint2 *>(&s_f[threadIdx.x][260]) = filtered(rb - 1 + (int)threadIdx.x, bx0 + 256, std::integral_constant<int, 1>());
    __syncthreads();
    // expand_dir_map (:722-773) out of s_f: which samples reach the sort (q_dir_map, phase 1)
    for (int lr = threadIdx.y; lr < QFE_R; lr += 4)
    {
        const int y = STEP * (rb + lr) + par;
        if (x >= width || y >= height) continue;
        const uint16_t *row = &s_f[lr + 1][4 + 4 * threadIdx.x];
        const uint2 out = *reinterpret_cast<const uint2 *>(row);
        if (row_ok(y))
        {
            const bool up_ok = STEP == 1 || y > 1, dn_ok = STEP == 1 || y < height - 2;
            // six samples x-1 .. x+4 of the three rows: usable (non-peak) flags, bit j = sample x-1+j
            auto usable6 = [&](const uint16_t *rw) -> uint32_t {
                const uint2 v = *reinterpret_cast<const uint2 *>(rw);
                const int l = rw[-1], r = rw[4];
                return (l != peak ? 1u : 0u) | ((int)(v.x & 0xffffu) != peak ? 2u : 0u) | ((int)(v.x >> 16) != peak ? 4u : 0u) |
                       ((int)(v.y & 0xffffu) != peak ? 8u : 0u) | ((int)(v.y >> 16) != peak ? 16u : 0u) | (r != peak ? 32u : 0u);
            };
            const uint32_t fc = usable6(row), fu = up_ok ? usable6(row - QFE_LW) : 0u, fd = dn_ok ? usable6(row + QFE_LW) : 0u;
            const uint32_t on_mask = masked4(y, x);
            uint32_t sortpx = 0;
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                const bool cand = ((on_mask >> j) & 1u) && !((fc >> (j + 1)) & 1u);                         // expand only fills peak samples
                const int u = __popc((fu >> j) & 7u) + __popc((fd >> j) & 7u) + __popc((fc >> j) & 5u);     // the centre is left out (:671)
                if (cand && u >= 5) sortpx |= 1u << j;
            }
            if (sortpx)
            {
                int at = atomicAdd(&s_count, __popc(sortpx));
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if ((sortpx >> j) & 1u) s_list[at++] = (uint16_t)((lr << 8) | (4 * threadIdx.x + j));
            }
        }
        *reinterpret_cast<uint2 *>(&s_out[lr][4 * threadIdx.x]) = out;
    }
    __syncthreads();
    const int count = s_count;
    for (int i = tid; i < count; i += 256)
    {
        const int e = s_list[i], ly = e >> 8, lx = e & 255;
        const int y = STEP * (rb + ly) + par;
        const bool up_ok = STEP == 1 || y > 1, dn_ok = STEP == 1 || y < height - 2;
        const uint16_t *c = &s_f[ly + 1][4 + lx], *up = up_ok ? c - QFE_LW : c, *dn = dn_ok ? c + QFE_LW : c;
        s_out[ly][lx] = (uint16_t)dir_map_px16(up[-1], up[0], up[1], c[-1], c[0], c[1], dn[-1], dn[0], dn[1], up_ok, dn_ok, 1,
                                               peak, k.neutral, 2 + k.shift, s_lim);
    }
    __syncthreads();
    // the padding of the rows (padv): the peak value where the reference expands into a plane that the fill in front of the
    // passes (calc_directions', mark_directions_2x's) has covered, 0 where into one nothing has
    auto padded = [&](uint2 v) { return make_uint2(pad_pair16(v.x, x, width, padv), pad_pair16(v.y, x + 2, width, padv)); };
    auto put_d = [&](int yy, uint2 v) {                            // the filtered map: the row's samples only, as the pass's bit_blit
        if (x >= width) return;
        uint16_t *o = P.d[pl] + (size_t)fld * P.fstride + (size_t)yy * pitch + x;
        if (x + 3 < width) *reinterpret_cast<uint2 *>(o) = v;
        else
        {
            const uint16_t o4[4] = { (uint16_t)(v.x & 0xffffu), (uint16_t)(v.x >> 16), (uint16_t)(v.y & 0xffffu), (uint16_t)(v.y >> 16) };
            for (int j = 0; j < 4 && x + j < width; j++) o[j] = o4[j];
        }
    };
#pragma unroll
    for (int h = 0; h < (QFE_R + 3) / 4; h++)
    {
        const int lr = threadIdx.y + 4 * h, y = STEP * (rb + lr) + par;
        if (lr >= QFE_R || x >= pitch) continue;
        if (y < height)
        {
            const uint2 v = x < width ? *reinterpret_cast<const uint2 *>(&s_out[lr][4 * threadIdx.x]) : make_uint2(0u, 0u);
            *reinterpret_cast<uint2 *>(Q.c + (size_t)y * pitch + x) = padded(v);
            if (POST) put_d(y, *reinterpret_cast<const uint2 *>(&s_f[lr + 1][4 + 4 * threadIdx.x]));
            if (POST && x < width && row_ok(y))
            {
                const size_t at = (size_t)y * pitch + x;
                const uint2 om4 = *reinterpret_cast<const uint2 *>(Q.b + at);
                uint16_t *d = P.f[pl] + (size_t)fld * P.fstride + at;
                const uint2 up4 = *reinterpret_cast<const uint2 *>(d - pitch), dn4 = *reinterpret_cast<const uint2 *>(d + pitch);
                const uint2 cur4 = *reinterpret_cast<const uint2 *>(d);
                auto s4 = [](const uint2 &w, int j) -> int { return (int)(((j < 2 ? w.x : w.y) >> (16 * (j & 1))) & 0xffffu); };
                uint32_t out[4];
                bool any = false;
#pragma unroll
                for (int j = 0; j < 4; j++)
                {
                    const int nm = s4(v, j), om = s4(om4, j);
                    const int lim = s_lim[iabs16(nm - k.neutral) >> (2 + k.shift)];
                    const bool fix = iabs16(nm - om) > lim && om != k.peak && om != k.neutral;
                    out[j] = fix ? (uint32_t)((s4(up4, j) + s4(dn4, j) + 1) >> 1) : (uint32_t)s4(cur4, j);
                    any |= fix;
                }
                if (any)
                {
                    if (x + 3 < width) *reinterpret_cast<uint2 *>(d) = make_uint2(out[0] | (out[1] << 16), out[2] | (out[3] << 16));
                    else for (int j = 0; j < 4 && x + j < width; j++) d[j] = (uint16_t)out[j];
                }
            }
        }
        if (STEP != 1)
        {
            const int yc = 2 * (rb + lr) + 1 - par;
            if (yc < height)
            {
                *reinterpret_cast<uint2 *>(Q.c + (size_t)yc * pitch + x) = padded(vcopy[h < NCOPY ? h : 0]);
                if (POST) put_d(yc, vcopy[h < NCOPY ? h : 0]);
            }
        }
    }
}

// eedi2_filter_map (:538-635): a = mask, b = direction map in, c = out.  The form of the 8-bit kernel (eedi2.hip:
// k_filter_map): a workgroup stages its 4 + 2 rows of the map (256 samples + 8 either side) in LDS with 8-byte loads, a
// thread owns four samples of its row, and both walks of a sample run in ONE loop whose step test is sign arithmetic:
// lim - |s - ref| is negative when s is off by more than lim, s - peak unless s is the peak, 2 peak - 1 - c - s only
// when both are the peak; the step trips on the sign of (off(s) & notpeak(s)) | (off(c) & notpeak(c)) | bothpeak.
constexpr int QFM_W = 256, QFM_R = 4, QFM_HALO = 8, QFM_LW = QFM_W + 2 * QFM_HALO;

__device__ __forceinline__ int qfm_step(uint32_t s, uint32_t c, uint32_t ref, int lim, int peak)
{
    const int off_s = lim - (int)__builtin_amdgcn_sad_u16(s, ref, 0u), off_c = lim - (int)__builtin_amdgcn_sad_u16(c, ref, 0u);
    return (off_s & ((int)s - peak)) | (off_c & ((int)c - peak)) | (2 * peak - 1 - (int)c - (int)s);
}

__global__ __launch_bounds__(256) void q_filter_map(Q3 P, K16 k)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_d[QFM_R + 2][QFM_LW];
    FIELD16(P);
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    const int bx0 = blockIdx.x * QFM_W, by0 = blockIdx.y * QFM_R;
    if (bx0 >= width || by0 >= height) return;                   // whole workgroup outside
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int peak = k.peak;
    if (maskless)
    {
        // no mask sample in the plane: the pass is its bit_blit of a map of peaks (eedi2.hip: k_dir_map4)
        const int x = bx0 + 4 * threadIdx.x, y = by0 + threadIdx.y;
        if (y < height)
            for (int j = 0; j < 4 && x + j < width; j++) Q.c[(size_t)y * pitch + x + j] = (uint16_t)peak;
        return;
    }
    {
        // 6 rows x 68 groups of four samples for 256 threads: both loads of a thread in flight before the first LDS store
        constexpr int G = QFM_LW / 4, N = (QFM_R + 2) * G;
        static_assert(N <= 2 * 256, "two staged groups per thread");
        uint2 v[2] = { make_uint2(0u, 0u), make_uint2(0u, 0u) }; // outside the plane: never looked at (the walks are clipped to the row)
#pragma unroll
        for (int j = 0; j < 2; j++)
        {
            const int i = tid + 256 * j, r = i / G, c4 = i - r * G;
            const int yy = by0 - 1 + r, col = bx0 - QFM_HALO + 4 * c4;
            if (i < N && yy >= 0 && yy < height && col >= 0 && col < pitch)
                v[j] = *reinterpret_cast<const uint2 *>(Q.b + (size_t)yy * pitch + col);
        }
#pragma unroll
        for (int j = 0; j < 2; j++)
            if (tid + 256 * j < N) reinterpret_cast<uint2 *>(&s_d[0][0])[tid + 256 * j] = v[j];
    }
    const int x = bx0 + 4 * threadIdx.x, y = by0 + threadIdx.y;
    const bool inside = x < width && y < height;
    uint2 m4 = make_uint2(0u, 0u);
    const bool mrow = inside && y >= 1 && y < height - 1;
    if (mrow) m4 = *reinterpret_cast<const uint2 *>(Q.a + (size_t)y * pitch + x);
    __syncthreads();
    if (!inside) return;
    const uint16_t *rc = &s_d[threadIdx.y + 1][QFM_HALO + 4 * threadIdx.x];
    // the four samples and their mask as 64-bit words (sample j in bits 16 j ..): the loop below is not unrolled
    unsigned long long out = *reinterpret_cast<const unsigned long long *>(rc);
    const unsigned long long mw = (unsigned long long)m4.x | ((unsigned long long)m4.y << 32);
#pragma unroll 1
    for (int j = 0; j < 4; j++)
    {
        const int xx = x + j;
        // candidates: mask peak, map not peak, inside the frame of :557-560
        if (!(mrow && xx >= 1 && xx < width - 1 && (int)((mw >> (16 * j)) & 0xffffull) == peak && (int)rc[j] != peak)) continue;
        const uint16_t *dc = rc + j, *dp = dc - QFM_LW, *dn = dc + QFM_LW;
        const uint32_t ref = dc[0];
        int dir = ((int)ref - k.neutral) >> 2;
        const int lim = max(iabs16(dir) * 2, 12 << (2 + k.shift));
        dir >>= 2 + k.shift;
        // the four ranges of :565-620 with neg = min(dir, 0), pos = max(dir, 0): [max(-x, neg), min(w - x - 1, pos)] above,
        // [max(-x, -pos), min(w - x - 1, -neg)] below; the shorter walk repeats its last step
        const int neg = min(dir, 0), pos = max(dir, 0);
        const int tf = max(-xx, neg), tt = min(width - xx - 1, pos), bf = max(-xx, -pos), bt = min(width - xx - 1, -neg);
        const int n = max(tt - tf, bt - bf);
        int any_t = 0, any_b = 0;
        for (int i = 0; i <= n; i++)
        {
            const int jt = min(tf + i, tt), jb = min(bf + i, bt);
            any_t |= qfm_step(dp[jt], dc[jt], ref, lim, peak);
            any_b |= qfm_step(dn[jb], dc[jb], ref, lim, peak);
        }
        if ((any_t & any_b) < 0) out = (out & ~(0xffffull << (16 * j))) | ((unsigned long long)peak << (16 * j));
    }
    uint16_t *o = Q.c + (size_t)y * pitch + x;
    if (x + 3 < width) *reinterpret_cast<uint2 *>(o) = make_uint2((uint32_t)out, (uint32_t)(out >> 32));
    else for (int j = 0; j < 4 && x + j < width; j++) o[j] = (uint16_t)(out >> (16 * j));
}

// eedi2_mark_directions_2x (:787-858) and the three eedi2_upscale_by_2 line doublings in front of it (:98-108,
// decomb_template.c:408-410), as eedi2.hip's k_mark_2x4: a = mskp, b = dstp (the half-height direction map), g = srcp come in
// at half height, d (dst2p), e (tmp2p2), f (msk2p) leave doubled, c = out (tmp2p).  The rows y - 1 / y + 1 of the doubled
// maps that mark_directions reads are rows (y - 1) >> 1 / (y + 1) >> 1 of the half-height ones, so it reads those.
// Four samples per thread and a PAIR of full-height rows per thread row (2r, 2r + 1 - both doubled from half-height row r):
// the row with the rebuilt rows' parity is worked on, the other is the memset's peak.  `height` = full height.  Every load
// goes out ahead of the first store.
__global__ __launch_bounds__(256) void q_mark_2x(Q3 P, K16 k, int padv)
{
    FIELD16(P);
    const int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x), r = blockIdx.y * blockDim.y + threadIdx.y;
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    const int y0 = 2 - tff;
    if (x >= pitch || 2 * r >= height) return;
    const int peak = k.peak;
    const uint2 peak4 = make_uint2((uint32_t)peak | ((uint32_t)peak << 16), (uint32_t)peak | ((uint32_t)peak << 16));
    const int y = 2 * r + (y0 & 1), yc = 2 * r + 1 - (y0 & 1);
    const bool two = 2 * r + 1 < height;
    const bool rebuilt = y < height && !maskless && y >= y0 && y < height - 1;
    const size_t hs = (size_t)r * pitch + x, fs = (size_t)(2 * r) * pitch + x, off = (size_t)fld * P.fstride;
    const uint16_t *Qg = P.g[pl] + off;
    uint16_t *Qd = P.d[pl] + off, *Qe = P.e[pl] + off, *Qf = P.f[pl] + off;
    const uint2 vg = *reinterpret_cast<const uint2 *>(Qg + hs), vb = *reinterpret_cast<const uint2 *>(Q.b + hs),
                va = *reinterpret_cast<const uint2 *>(Q.a + hs);
    uint2 a4 = peak4, b4 = peak4, k0 = make_uint2(0u, 0u), k1 = make_uint2(0u, 0u);
    int al = 0, ar = 0, bl = 0, br = 0;
    if (rebuilt)
    {
        const int ra = (y - 1) >> 1, rb = (y + 1) >> 1;
        const uint16_t *d0 = Q.b + (size_t)ra * pitch + x, *d1 = Q.b + (size_t)rb * pitch + x;
        k0 = ra == r ? va : *reinterpret_cast<const uint2 *>(Q.a + (size_t)ra * pitch + x);
        k1 = rb == r ? va : *reinterpret_cast<const uint2 *>(Q.a + (size_t)rb * pitch + x);
        // samples x-1 .. x+4 of the two direction rows
        a4 = ra == r ? vb : *reinterpret_cast<const uint2 *>(d0);
        b4 = rb == r ? vb : *reinterpret_cast<const uint2 *>(d1);
        al = x > 0 ? (int)d0[-1] : 0; ar = x + 4 < pitch ? (int)d0[4] : 0;
        bl = x > 0 ? (int)d1[-1] : 0; br = x + 4 < pitch ? (int)d1[4] : 0;
    }
    *reinterpret_cast<uint2 *>(Qd + fs) = vg;
    *reinterpret_cast<uint2 *>(Qe + fs) = vb;
    *reinterpret_cast<uint2 *>(Qf + fs) = va;
    if (two)
    {
        *reinterpret_cast<uint2 *>(Qd + fs + pitch) = vg;
        *reinterpret_cast<uint2 *>(Qe + fs + pitch) = vb;
        *reinterpret_cast<uint2 *>(Qf + fs + pitch) = va;
    }
    // padv: what the rows' padding gets (0 when the map goes to the plane the fused dir-map pass reads, whose padding the
    // reference never writes; eedi2.hip: k_mark_2x4)
    auto padded = [&](uint2 v) { return make_uint2(pad_pair16(v.x, x, width, padv), pad_pair16(v.y, x + 2, width, padv)); };
    if (yc < height) *reinterpret_cast<uint2 *>(Q.c + (size_t)yc * pitch + x) = padded(peak4);
    if (y >= height) return;
    uint2 *o = reinterpret_cast<uint2 *>(Q.c + (size_t)y * pitch + x);
    if (!rebuilt) { *o = padded(peak4); return; }                              // (no mask sample in the plane: nothing but the fill, :800)
    const int A[6] = { al, (int)(a4.x & 0xffffu), (int)(a4.x >> 16), (int)(a4.y & 0xffffu), (int)(a4.y >> 16), ar };
    const int B[6] = { bl, (int)(b4.x & 0xffffu), (int)(b4.x >> 16), (int)(b4.y & 0xffffu), (int)(b4.y >> 16), br };
    const int M0[4] = { (int)(k0.x & 0xffffu), (int)(k0.x >> 16), (int)(k0.y & 0xffffu), (int)(k0.y >> 16) };
    const int M1[4] = { (int)(k1.x & 0xffffu), (int)(k1.x >> 16), (int)(k1.y & 0xffffu), (int)(k1.y >> 16) };
    uint32_t out[4] = { (uint32_t)peak, (uint32_t)peak, (uint32_t)peak, (uint32_t)peak };
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
        const int xx = x + j;
        if (xx >= 1 && xx < width - 1 && (M0[j] == peak || M1[j] == peak))
        {
            const int a0 = A[j], a1 = A[j + 1], a2 = A[j + 2], b0 = B[j], b1 = B[j + 1], b2 = B[j + 2];
            const bool h0 = a0 != peak, h1 = a1 != peak, h2 = a2 != peak, h3 = b0 != peak, h4 = b1 != peak, h5 = b2 != peak;
            const int n = h0 + h1 + h2 + h3 + h4 + h5;
            if (n >= 3)
            {
                int v0 = h0 ? a0 : ABSENT16, v1 = h1 ? a1 : ABSENT16, v2 = h2 ? a2 : ABSENT16;
                int v3 = h3 ? b0 : ABSENT16, v4 = h4 ? b1 : ABSENT16, v5 = h5 ? b2 : ABSENT16;
                const int mid = mid6q(v0, v1, v2, v3, v4, v5, n);
                const int lim = k.limlut[iabs16(mid - k.neutral) >> (2 + k.shift)];
                int u = 0;
                if (iabs16(a0 - b0) <= lim || !h0 || !h3) u++;
                if (iabs16(a1 - b1) <= lim || !h1 || !h4) u++;
                if (iabs16(a2 - b0) <= lim || !h2 || !h5) u++;                                   // sic (:835)
                if (u >= 2)
                {
                    int sum = 0, count = 0;
                    vote1q(v0, mid, lim, sum, count); vote1q(v1, mid, lim, sum, count); vote1q(v2, mid, lim, sum, count);
                    vote1q(v3, mid, lim, sum, count); vote1q(v4, mid, lim, sum, count); vote1q(v5, mid, lim, sum, count);
                    const int val = (int)(((float)(sum + mid) / (float)(count + 1)) + 0.5f);
                    if (!(count < n - 2 || count < 2)) out[j] = (uint32_t)val;
                }
            }
        }
    }
    *o = padded(make_uint2(out[0] | (out[1] << 16), out[2] | (out[3] << 16)));
}

// eedi2_fill_gaps_2x (:1025-1132): a = msk2p, b = direction map in, c = out.
// The same pass in the form of the 8-bit kernel (eedi2.hip: k_fill_gaps_b), copy included: a workgroup owns QF_W
// consecutive samples of one row.  Rows the pass does not rebuild are copied.  For a rebuilt row the seven rows involved
// (dc = y, mask y-1 / y+1 / y-3 / y+3, direction y-2 / y+2) are staged in LDS with QF_HALO samples either side (same flat
// addressing), turned into four bit rows by ballots (a walk stops here / the direction is known / the row above, below
// ends the "continues" state), the gap samples of the span are compacted into a list, and each listed sample finds its
// gap's ends and its support with 64-bit scans; only min / max over a supported gap still reads samples.  A walk that
// leaves the staged span takes the sample-by-sample path on memory (rare).  Every sample of a fillable gap computes the
// same (u, v, back, forward, verdict) as its neighbours in the gap, so each thread writes its own sample only.
constexpr int QF_W = 1024, QF_HALO = 64, QF_LW = QF_W + 2 * QF_HALO;

// QF_R rebuilt rows (and the copied rows between them) per workgroup, as k_fill_gaps_b in eedi2.hip: 2 QF_R + 5 staged rows
// serve QF_R rebuilt rows, the rows' own bitmaps are made once, a quarter of the waves.
constexpr int QF_R = 4, QF_ND = QF_R + 2, QF_NM = QF_R + 3, QF_WORDS = QF_LW / 64 + 1;
__global__ __launch_bounds__(256) void q_fill_gaps_b(Q3 P, K16 k)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_d[QF_ND][QF_LW];     // direction rows yb - 2, yb, .. , yb + 2 QF_R
    __shared__ __attribute__((aligned(16))) uint16_t s_m[QF_NM][QF_LW];     // mask rows yb - 3, yb - 1, .. , yb + 2 QF_R + 1
    __shared__ __attribute__((aligned(16))) uint16_t s_out[QF_R][QF_W];
    __shared__ uint16_t s_list[QF_R * QF_W];
    __shared__ int s_count;
    __shared__ uint64_t s_stop[QF_R][QF_WORDS], s_np[QF_R][QF_WORDS], s_bt[QF_R][QF_WORDS], s_bb[QF_R][QF_WORDS];   // one bit per staged column
    FIELD16(P);
    const int y0 = 2 - tff;
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    const int x0 = blockIdx.x * QF_W, tid = threadIdx.x;
    const int ya = (int)blockIdx.y * 2 * QF_R;                     // the workgroup's plane rows: ya .. ya + 2 QF_R - 1
    if (ya >= height || x0 >= width) return;
    const int yb = ya + (y0 & 1);                                  // its rows of the rebuilt parity: yb, yb + 2, ..
    const int peak = k.peak;
    const int x = x0 + 4 * tid;
    if (maskless)
    {
        // no mask sample in the plane, no gap to fill (:1048-1050): the pass is its bit_blit of a map of peaks
        if (x < width)
            for (int i = 0; i < 2 * QF_R && ya + i < height; i++)
                for (int j = 0; j < 4 && x + j < width; j++) Q.c[(size_t)(ya + i) * pitch + x + j] = (uint16_t)peak;
        return;
    }
    auto rebuilt = [&](int y) { return y >= y0 && y < height - 1; };
    // the rows that are only copied (the reference's bit_blit)
    uint2 vcopy[2 * QF_R];
#pragma unroll
    for (int i = 0; i < 2 * QF_R; i++)
    {
        const int y = ya + i;
        vcopy[i] = make_uint2(0u, 0u);
        if (x < width && y < height && !((((y - y0) & 1) == 0) && rebuilt(y)))
            vcopy[i] = *reinterpret_cast<const uint2 *>(Q.b + (size_t)y * pitch + x);
    }
    if (tid == 0) s_count = 0;
    const int lo = x0 - QF_HALO;                                   // column of staged sample 0 (a multiple of 4)
    const int nq = (min(QF_W, (width - x0 + 3) & ~3) + 2 * QF_HALO) / 4;       // groups of four samples, <= 288
    {
        // all loads of a thread in flight before its first LDS store; rows outside the plane are rows no pixel's tests
        // reach (:1076, :1090): they are read from the nearest row inside
        const bool h0 = tid < nq, h1 = tid + 256 < nq;
        uint2 vd[QF_ND][2], vm[QF_NM][2];
#pragma unroll
        for (int r = 0; r < QF_ND; r++)
        {
            const uint2 *row = reinterpret_cast<const uint2 *>(Q.b + (size_t)min(max(yb - 2 + 2 * r, 0), height - 1) * pitch + lo);
            vd[r][0] = h0 ? row[tid] : make_uint2(0u, 0u);
            vd[r][1] = h1 ? row[tid + 256] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int r = 0; r < QF_NM; r++)
        {
            const uint2 *row = reinterpret_cast<const uint2 *>(Q.a + (size_t)min(max(yb - 3 + 2 * r, 0), height - 1) * pitch + lo);
            vm[r][0] = h0 ? row[tid] : make_uint2(0u, 0u);
            vm[r][1] = h1 ? row[tid + 256] : make_uint2(0u, 0u);
        }
        if (h0)
        {
#pragma unroll
            for (int r = 0; r < QF_ND; r++) reinterpret_cast<uint2 *>(s_d[r])[tid] = vd[r][0];
#pragma unroll
            for (int r = 0; r < QF_NM; r++) reinterpret_cast<uint2 *>(s_m[r])[tid] = vm[r][0];
        }
        if (h1)
        {
#pragma unroll
            for (int r = 0; r < QF_ND; r++) reinterpret_cast<uint2 *>(s_d[r])[tid + 256] = vd[r][1];
#pragma unroll
            for (int r = 0; r < QF_NM; r++) reinterpret_cast<uint2 *>(s_m[r])[tid + 256] = vm[r][1];
        }
    }
    __syncthreads();
    // rebuilt row r = yb + 2r: direction rows s_d[r] (y - 2), s_d[r + 1] (y), s_d[r + 2] (y + 2); mask rows s_m[r] (y - 3),
    // s_m[r + 1] (y - 1), s_m[r + 2] (y + 1), s_m[r + 3] (y + 3)
    const unsigned staged = 4u * (unsigned)nq;
    for (int q = 0; q < (QF_LW + 255) / 256; q++)
    {
        const unsigned col = tid + 256 * q, cc = min(col, staged - 1u);
        const bool in = col < staged;
        uint64_t nd[QF_ND], pm[QF_NM];
#pragma unroll
        for (int r = 0; r < QF_ND; r++) nd[r] = __ballot(in & (s_d[r][cc] != peak));       // direction known
#pragma unroll
        for (int r = 0; r < QF_NM; r++) pm[r] = __ballot(in & (s_m[r][cc] == peak));       // on the mask
        const uint64_t inw = __ballot(in);
        if ((tid & 63) == 0 && (col >> 6) < (unsigned)QF_WORDS)
        {
#pragma unroll
            for (int r = 0; r < QF_R; r++)
            {
                const uint64_t mc = pm[r + 1], mn = pm[r + 2];
                s_np[r][col >> 6] = nd[r + 1];
                s_stop[r][col >> 6] = nd[r + 1] | (inw & ~mc & ~mn);
                s_bt[r][col >> 6] = inw & (~nd[r] | (~pm[r] & ~mc));
                s_bb[r][col >> 6] = inw & (~nd[r + 2] | (~mn & ~pm[r + 3]));
            }
        }
    }
#pragma unroll
    for (int r = 0; r < QF_R; r++)
    {
        const int y = yb + 2 * r;
        if (x < width && y < height && rebuilt(y))
        {
            const int c = 4 * tid + QF_HALO;
            *reinterpret_cast<uint2 *>(&s_out[r][4 * tid]) = *reinterpret_cast<const uint2 *>(&s_d[r + 1][c]);
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                const int xx = x + j;
                if (xx >= 1 && xx < width - 1 && s_d[r + 1][c + j] == peak && (s_m[r + 1][c + j] == peak || s_m[r + 2][c + j] == peak))
                    s_list[atomicAdd(&s_count, 1)] = (uint16_t)((r << 12) | (4 * tid + j));
            }
        }
    }
    __syncthreads();
    const int count = s_count;
    const int eight = 8 << k.shift, twenty = 20 << k.shift, five_hundred = 500 << k.shift;
    for (int i = tid; i < count; i += 256)
    {
        const int e = s_list[i], r = e >> 12, lx = e & 0xfff, px = x0 + lx, y = yb + 2 * r;
        const uint16_t *DC = s_d[r + 1], *DP = s_d[r], *DN = s_d[r + 2];
        const uint16_t *MC = s_m[r + 1], *MN = s_m[r + 2], *MP = s_m[r], *MNN = s_m[r + 3];
        const uint16_t *gd = Q.b + (size_t)y * pitch, *gma = Q.a + (ptrdiff_t)(y - 1) * pitch;
        auto rd = [&](const uint16_t *srow, const uint16_t *grow, int col) -> int {
            const unsigned kk = (unsigned)(col - lo);
            return kk < staged ? (int)srow[kk] : (int)grow[col];
        };
        int u = px - 1, back = five_hundred, forward = -five_hundred;
        int v = px + 1;
        int tc = 1, bc = 1, mint = five_hundred, maxt = -twenty, minb = five_hundred, maxb = -twenty;
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
            if (y <= 2 || any_in(s_bt[r])) { tc = 0; mint = maxt = twenty; }
            else for (int j = a0; j <= a1; j++) { const int t = DP[j]; mint = min(mint, t); maxt = max(maxt, t); }
            if (y >= height - 3 || any_in(s_bb[r])) { bc = 0; minb = maxb = twenty; }
            else for (int j = a0; j <= a1; j++) { const int t = DN[j]; minb = min(minb, t); maxb = max(maxb, t); }
        }
        else
        {
            const uint16_t *gmn = gma + 2 * (ptrdiff_t)pitch;
            while (u)
            {
                const int d = rd(DC, gd, u);
                if (d != peak) { back = d; break; }
                if (rd(MC, gma, u) != peak && rd(MN, gmn, u) != peak) break;
                u--;
            }
            while (v < width)
            {
                const int d = rd(DC, gd, v);
                if (d != peak) { forward = d; break; }
                if (rd(MC, gma, v) != peak && rd(MN, gmn, v) != peak) break;
                v++;
            }
            const uint16_t *gdp = gd - 2 * (ptrdiff_t)pitch, *gdn = gd + 2 * (ptrdiff_t)pitch;
            const uint16_t *gmp = gma - 2 * (ptrdiff_t)pitch, *gmnn = gmn + 2 * (ptrdiff_t)pitch;
            for (int j = u; j <= v; j++)
            {
                if (tc)
                {
                    int t;
                    if (y <= 2 || (t = rd(DP, gdp, j)) == peak || (rd(MP, gmp, j) != peak && rd(MC, gma, j) != peak)) { tc = 0; mint = maxt = twenty; }
                    else { mint = min(mint, t); maxt = max(maxt, t); }
                }
                if (bc)
                {
                    int t;
                    if (y >= height - 3 || (t = rd(DN, gdn, j)) == peak || (rd(MN, gmn, j) != peak && rd(MNN, gmnn, j) != peak)) { bc = 0; minb = maxb = twenty; }
                    else { minb = min(minb, t); maxb = max(maxb, t); }
                }
            }
        }
        if (maxt == -twenty) maxt = mint = twenty;
        if (maxb == -twenty) maxb = minb = twenty;
        const int far = max(iabs16(forward - k.neutral), iabs16(back - k.neutral));
        const int thresh = max(max(far >> 2, eight), max(iabs16(mint - maxt), iabs16(minb - maxb)));
        const int flim = min(far >> (2 + k.shift), 6);
        if (iabs16(forward - back) <= thresh && (v - u - 1 <= flim || tc || bc))
        {
            const double step = (double)(forward - back) / (double)(v - u);
            const int j = px - u - 1;
            s_out[r][lx] = (uint16_t)(back + (int)(j * step + 0.5));
        }
    }
    __syncthreads();
    if (x < width)
    {
#pragma unroll
        for (int i = 0; i < 2 * QF_R; i++)
        {
            const int y = ya + i;
            if (y >= height) break;
            const bool work = (((y - y0) & 1) == 0) && rebuilt(y);
            const uint2 v = work ? *reinterpret_cast<const uint2 *>(&s_out[(i - (y0 & 1)) >> 1][4 * tid]) : vcopy[i];
            uint16_t *o = Q.c + (size_t)y * pitch + x;
            if (x + 3 < width) *reinterpret_cast<uint2 *>(o) = v;
            else
            {
                const uint16_t o4[4] = { (uint16_t)(v.x & 0xffffu), (uint16_t)(v.x >> 16), (uint16_t)(v.y & 0xffffu), (uint16_t)(v.y >> 16) };
                for (int j = 0; j < 4 && x + j < width; j++) o[j] = o4[j];
            }
        }
    }
}

// plain copy of the visible width (eedi2_bit_blit :46-68): a = in, c = out; eight samples per thread
__global__ void q_blit(Q3 P)
{
    FIELD16(P);
    const int x = 8 * (blockIdx.x * blockDim.x + threadIdx.x), y = blockIdx.y * blockDim.y + threadIdx.y;
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    if (x >= width || y >= height) return;
    const uint16_t *in = Q.a + (size_t)y * pitch + x;
    uint16_t *out = Q.c + (size_t)y * pitch + x;
    if (x + 7 < width) *reinterpret_cast<uint4 *>(out) = *reinterpret_cast<const uint4 *>(in);
    else for (int i = 0; x + i < width; i++) out[i] = in[i];
}

// interpolate_lattice in two launches, as for 8-bit samples (eedi2.hip: k_lattice_cand / k_lattice_resolve).  Of all a
// pixel's tests only one looks at the direction value just written at x-1 (the left-hand half of :1194); everything else
// - including the whole "outcome B" the pixel takes when that test fails - reads values no pixel of the pass changes.
// q_lattice_cand packs into 64 bits:
//   [15:0] valA = vertical average (outcome A)   [31:16] valB   [47:32] newB = outcome-B direction value
//   bit 48 = "always A" (direction == peak)      bit 49 = right-hand test |d[x] - d[x+1]| > lim
// q_lattice_resolve16 (one workgroup per row) resolves which outcome each pixel takes: every pixel is a 2-state map of
// its left neighbour's outcome, composed by a prefix scan, and writes the row.
// The candidates are made the way the 8-bit kernel makes them (eedi2.hip: k_lattice_cand_q): a workgroup takes 1024
// samples of a rebuilt row with the five rows involved staged in LDS; every thread packs the words of its four samples as
// if they were "always A" and queues the ones that carry a direction (roughly one in ten on real pictures, but nearly
// every wave holds some); the queue then gets one lane per sample, stage by stage - the variance / edge tests, the search
// around the sample's direction, the short search - each stage either finishing the word or handing the sample on.
constexpr int LQ16_W = 1024, LQ16_HALO = 40, LQ16_LW = LQ16_W + 2 * LQ16_HALO;   // |u| <= 34, +-1 for the triples
constexpr unsigned long long LAT16_MORE = 1ull << 63;

struct Lat16
{
    int peak, neutral, sh, sh2, three, nine, nt4, nt7, nt8, nt7_unshifted;
    const int *lim;                    // limlut, in LDS
};

__device__ __forceinline__ unsigned long long lat16_word(int valB, int newB) { return ((unsigned long long)(uint16_t)valB << 16) | ((unsigned long long)(uint16_t)newB << 32); }

// the search around the sample's direction (:1242-1290)
__device__ __forceinline__ unsigned long long lat16_stage_b(const uint16_t *top, const uint16_t *bot, const uint16_t *ot, const uint16_t *ob,
                                                            const uint16_t *dm, int x, int width, unsigned long long base, const Lat16 &L)
{
    const int here = dm[x], peak = L.peak;
    const int lim = L.lim[iabs16(here - L.neutral) >> L.sh2];
    int dir = (here - L.neutral + (1 << (L.sh2 - 1))) >> L.sh2;
    int val = (int)(base & 0xffffu);
    const int startu = (dir - 2 < 0) ? max(-x + 1, max(dir - 2, -width + 2 + x)) : min(x - 1, min(dir - 2, width - 2 - x));
    const int stopu = (dir + 2 < 0) ? max(-x + 1, max(dir + 2, -width + 2 + x)) : min(x - 1, min(dir + 2, width - 2 - x));
    int mn = L.nt8;
    auto near = [&](const uint16_t *row, int i) { return row[i] != peak && iabs16((int)row[i] - here) <= lim; };
    for (int u = startu; u <= stopu; u++)
    {
        const int diff = sad3w(top, x, bot, x - u) + sad3w(bot, x, top, x + u);
        if (!(diff < mn && (near(ot, x - 1 + u) || near(ot, x + u) || near(ot, x + 1 + u)) &&
              (near(ob, x - 1 - u) || near(ob, x - u) || near(ob, x + 1 - u))))
            continue;
        const int h0 = u >> 1, h1 = (u + 1) >> 1;
        const int diff2 = sad3w(top, x + h0, bot, x - h0);
        const int o0 = ot[x + h0], o1 = ot[x + h1], q0 = ob[x - h0], q1 = ob[x - h1];
        if (!(diff2 < L.nt4 && (((iabs16(o0 - q0) <= lim || iabs16(o0 - q1) <= lim) && o0 != peak) ||
                                ((iabs16(o1 - q0) <= lim || iabs16(o1 - q1) <= lim) && o1 != peak))))
            continue;
        if ((iabs16(here - o0) <= lim || iabs16(here - o1) <= lim) && (iabs16(here - q0) <= lim || iabs16(here - q1) <= lim))
        {
            val = ((int)top[x + h0] + (int)top[x + h1] + (int)bot[x - h0] + (int)bot[x - h1] + 2) >> 2;
            mn = diff;
            dir = u;
        }
    }
    if (mn != L.nt8) return base | lat16_word(val, L.neutral + (dir << L.sh2));
    return base | LAT16_MORE;
}

// ---- the forms of eedi2.hip's k_lattice_cand_q for 16-bit samples: the tests out of registers, the search out of
// windows of eight samples with its steps side by side (see there for the why; here a window is four dwords, a step's
// three samples a pair and a single through v_sad_u16, and two samples ride in a packed operation as they lie)
typedef uint16_t u16x2q __attribute__((ext_vector_type(2)));
typedef int16_t i16x2q __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2q pkq(uint32_t v) { return __builtin_bit_cast(u16x2q, v); }
__device__ __forceinline__ uint32_t unq(u16x2q v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ u16x2q pkq1(uint32_t both) { return pkq(both * 0x00010001u); }
__device__ __forceinline__ u16x2q pkq_lt(u16x2q a, u16x2q b) { return (u16x2q)((u16x2q)(a - b) >> 15); }   // halves below 2^15

// the variance and the edge test on the fixed 2 x 5 neighbourhood (:1213-1240), on values (T0 .. T4 / B0 .. B4: the samples
// x - 2 .. x + 2 of the rows above / below)
__device__ __forceinline__ unsigned long long lat16_stage_a_vals(int T0, int T1, int T2, int T3, int T4, int B0, int B1, int B2, int B3, int B4,
                                                                 int lim, bool inner, unsigned long long base, const Lat16 &L)
{
    const int avg = (int)(base & 0xffffu);
    if (lim < L.nine)
    {
        const int sh = L.sh;
        const int sum = (T1 + T2 + T3 + B1 + B2 + B3) >> sh;
        const int t1 = T1 >> sh, t2 = T2 >> sh, t3 = T3 >> sh, b1 = B1 >> sh, b2 = B2 >> sh, b3 = B3 >> sh;
        const int sumsq = t1 * t1 + t2 * t2 + t3 * t3 + b1 * b1 + b2 * b2 + b3 * b3;
        if (6 * sumsq - sum * sum < 576) return base | lat16_word(avg, L.peak);
    }
    if (inner)
    {
        const int three = L.three;
        if ((T2 < max(T0, T1) - three && T2 < max(T4, T3) - three && B2 < max(B0, B1) - three && B2 < max(B4, B3) - three) ||
            (T2 > min(T0, T1) + three && T2 > min(T4, T3) + three && B2 > min(B0, B1) + three && B2 > min(B4, B3) + three))
            return base | lat16_word(avg, L.neutral);
    }
    return base | LAT16_MORE;
}

struct Lat16Rows { const uint32_t *top, *bot, *ot, *ob; int org; };      // the staged rows as dwords; org: local index of sample 0

// samples row[li] .. row[li + 7] (any alignment) as four dwords
__device__ __forceinline__ void lat16_window8(const uint32_t *row, int li, uint32_t (&w)[4])
{
    const uint32_t *p = row + (li >> 1);
    const uint32_t r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3], r4 = p[4];
    const uint32_t sh = ((uint32_t)li & 1u) * 2u;
    w[0] = __builtin_amdgcn_alignbyte(r1, r0, sh); w[1] = __builtin_amdgcn_alignbyte(r2, r1, sh);
    w[2] = __builtin_amdgcn_alignbyte(r3, r2, sh); w[3] = __builtin_amdgcn_alignbyte(r4, r3, sh);
}
// samples row[li] .. row[li + 3] as two dwords
__device__ __forceinline__ void lat16_window4(const uint32_t *row, int li, uint32_t (&w)[2])
{
    const uint32_t *p = row + (li >> 1);
    const uint32_t r0 = p[0], r1 = p[1], r2 = p[2];
    const uint32_t sh = ((uint32_t)li & 1u) * 2u;
    w[0] = __builtin_amdgcn_alignbyte(r1, r0, sh); w[1] = __builtin_amdgcn_alignbyte(r2, r1, sh);
}
// the three samples G, G + 1, G + 2 of a window: the pair as it lies (or realigned), the single zero-extended
template <int G, int N> __device__ __forceinline__ uint32_t lat16_pair(const uint32_t (&w)[N])
{
    if constexpr (G & 1) return __builtin_amdgcn_alignbyte(w[(G + 1) / 2], w[G / 2], 2u);
    else return w[G / 2];
}
template <int G, int N> __device__ __forceinline__ uint32_t lat16_one(const uint32_t (&w)[N])      // sample G
{
    if constexpr (G & 1) return w[G / 2] >> 16;
    else return w[G / 2] & 0xffffu;
}
#define LAT16_SAD3(wa, GA, wb, GB) \
    __builtin_amdgcn_sad_u16(lat16_pair<GA>(wa), lat16_pair<GB>(wb), __builtin_amdgcn_sad_u16(lat16_one<(GA) + 2>(wa), lat16_one<(GB) + 2>(wb), 0u))
// which of a window's eight samples are a direction (not the peak) within lim of d: sample 2i at bit i, 2i + 1 at bit 16 + i
__device__ __forceinline__ uint32_t lat16_near8(const uint32_t (&w)[4], u16x2q d2, u16x2q lim1, u16x2q peak2)
{
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const u16x2q h = pkq(w[i]);
        const i16x2q t = __builtin_bit_cast(i16x2q, (u16x2q)(h - d2));
        const u16x2q a = __builtin_bit_cast(u16x2q, __builtin_elementwise_max(t, (i16x2q)(-t)));
        m |= unq(pkq_lt(a, lim1) & (pkq_lt(h ^ peak2, pkq1(1)) ^ pkq1(1))) << i;      // (not the peak: by equality - a value can lie above it)
    }
    return m;
}
constexpr uint32_t lat16_bit(int b) { return (b & 1) ? 1u << (16 + (b >> 1)) : 1u << (b >> 1); }
constexpr uint32_t lat16_bits3(int g) { return lat16_bit(g) | lat16_bit(g + 1) | lat16_bit(g + 2); }

__device__ __forceinline__ unsigned long long lat16_stage_b_win(const Lat16Rows R, const uint16_t *top, const uint16_t *bot, const uint16_t *ot,
                                                                const uint16_t *ob, const uint16_t *dm, int x, int width,
                                                                unsigned long long base, const Lat16 &L)
{
    const int here = dm[x];
    const int lim = L.lim[iabs16(here - L.neutral) >> L.sh2];
    const int dir = (here - L.neutral + (1 << (L.sh2 - 1))) >> L.sh2;
    const int startu = (dir - 2 < 0) ? max(-x + 1, max(dir - 2, -width + 2 + x)) : min(x - 1, min(dir - 2, width - 2 - x));
    const int stopu = (dir + 2 < 0) ? max(-x + 1, max(dir + 2, -width + 2 + x)) : min(x - 1, min(dir + 2, width - 2 - x));
    if (startu != dir - 2 || stopu != dir + 2) return lat16_stage_b(top, bot, ot, ob, dm, x, width, base, L);   // clamped at a row end
    const int de = dir & ~1, par = dir & 1, hb = (de >> 1) - 1;     // step i: u = de - 2 + i, u >> 1 = hb + (i >> 1), (u + 1) >> 1 = hb + ((i + 1) >> 1)
    const int lx = R.org + x;
    uint32_t bw[4], qw[4], tw[4], ow[4], t2[4], b2[4], o2[2], q2[2], tcw[2], bcw[2];
    lat16_window8(R.bot, lx - de - 4, bw);                   // x - u - 1 at sample 5 - i
    lat16_window8(R.ob, lx - de - 4, qw);
    lat16_window8(R.top, lx + de - 3, tw);                   // x + u - 1 at sample i
    lat16_window8(R.ot, lx + de - 3, ow);
    lat16_window8(R.top, lx + hb - 1, t2);                   // x + (u >> 1) - 1 at sample i >> 1
    lat16_window8(R.bot, lx - hb - 4, b2);                   // x - (u >> 1) - 1 at sample 3 - (i >> 1)
    lat16_window4(R.ot, lx + hb, o2);                        // x + (u >> 1) at sample i >> 1
    lat16_window4(R.ob, lx - hb - 3, q2);                    // x - (u >> 1) at sample 3 - (i >> 1)
    lat16_window4(R.top, lx - 1, tcw);
    lat16_window4(R.bot, lx - 1, bcw);
    const u16x2q d2 = pkq1((uint32_t)here), lim1 = pkq1((uint32_t)min(lim + 1, 0x7fff)), peak2 = pkq1((uint32_t)L.peak);
    const uint32_t near_t = lat16_near8(ow, d2, lim1, peak2), near_b = lat16_near8(qw, d2, lim1, peak2);
    const uint32_t ulim = (uint32_t)lim, ulim2 = 2u * ulim;
    auto within = [&](uint32_t a, uint32_t b) { return a - b + ulim <= ulim2; };              // |a - b| <= lim
    const uint32_t o[4] = { o2[0] & 0xffffu, o2[0] >> 16, o2[1] & 0xffffu, o2[1] >> 16 };
    const uint32_t q[4] = { q2[0] & 0xffffu, q2[0] >> 16, q2[1] & 0xffffu, q2[1] >> 16 };
    const uint32_t uh = (uint32_t)here, peak = (uint32_t)L.peak;
    const bool od[4] = { within(o[0], uh), within(o[1], uh), within(o[2], uh), within(o[3], uh) };
    const bool qd[4] = { within(q[0], uh), within(q[1], uh), within(q[2], uh), within(q[3], uh) };
    const int diff2[3] = { (int)LAT16_SAD3(t2, 0, b2, 3), (int)LAT16_SAD3(t2, 1, b2, 2), (int)LAT16_SAD3(t2, 2, b2, 1) };
    uint32_t best = 0x7fffffffu;
#define LAT16_STEP(i)                                                                                                 \
    {                                                                                                                 \
        constexpr int A = (i) >> 1, B = ((i) + 1) >> 1;                                                               \
        const int diff = (int)(LAT16_SAD3(tcw, 0, bw, 5 - (i)) + LAT16_SAD3(bcw, 0, tw, (i)));                        \
        const bool oq = A == B ? (within(o[A], q[3 - A]) && o[A] != peak)                                             \
                               : (((within(o[A], q[3 - A]) || within(o[A], q[3 - B])) && o[A] != peak) ||             \
                                  ((within(o[B], q[3 - A]) || within(o[B], q[3 - B])) && o[B] != peak));              \
        const bool ok = ((i) == 0 ? par == 0 : (i) == 5 ? par != 0 : true) && diff < L.nt8 &&                         \
                        (near_t & lat16_bits3(i)) && (near_b & lat16_bits3(5 - (i))) && diff2[A] < L.nt4 && oq &&     \
                        (od[A] || od[B]) && (qd[3 - A] || qd[3 - B]);                                                 \
        best = min(best, ok ? ((uint32_t)diff << 3) | (uint32_t)(i) : 0x7fffffffu);                                   \
    }
    LAT16_STEP(0) LAT16_STEP(1) LAT16_STEP(2) LAT16_STEP(3) LAT16_STEP(4) LAT16_STEP(5)
#undef LAT16_STEP
    if (best == 0x7fffffffu) return base | LAT16_MORE;
    const int u = de - 2 + (int)(best & 7u);
    const int h0 = u >> 1, h1 = (u + 1) >> 1;
    const int val = ((int)top[x + h0] + (int)top[x + h1] + (int)bot[x - h0] + (int)bot[x - h1] + 2) >> 2;
    return base | lat16_word(val, L.neutral + (u << L.sh2));
}

// the short search for samples the first one left without a match (:1292-1318)
__device__ __forceinline__ unsigned long long lat16_stage_c(const uint16_t *top, const uint16_t *bot, const uint16_t *dm, int x, int width, int pl,
                                                            unsigned long long base, const Lat16 &L)
{
    int dir = ((int)dm[x] - L.neutral + (1 << (L.sh2 - 1))) >> L.sh2;
    int val = (int)(base & 0xffffu);
    const int lo = min((int)top[x], (int)bot[x]), hi = max((int)top[x], (int)bot[x]);
    const int dd = pl == 0 ? 4 : 2;
    const int su = max(-x + 1, -dd), eu = min(width - 2 - x, dd);
    int mn = L.nt7;
    for (int u = su; u <= eu; u++)
    {
        const int h0 = u >> 1, h1 = (u + 1) >> 1;
        const int p1 = (int)top[x + h0] + (int)top[x + h1];
        const int p2 = (int)bot[x - h0] + (int)bot[x - h1];
        const int diff = sad3w(top, x, bot, x - u) + sad3w(bot, x, top, x + u) + iabs16(p1 - p2);
        if (diff < mn)
        {
            const int valt = (p1 + p2 + 2) >> 2;
            if (valt >= lo && valt <= hi) { val = valt; mn = diff; dir = u; }
        }
    }
    const int newB = (mn == L.nt7_unshifted) ? L.neutral : (int)(uint16_t)(L.neutral + (dir << L.sh2));      // unshifted 7*nt (:1324)
    return base | lat16_word(val, newB);
}

__global__ __launch_bounds__(256) void q_lattice_cand(Q3 P, K16 k, int nt, unsigned long long *__restrict__ cand,
                                                      int cand_pitch, int cand_plane_stride)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_rows[5][LQ16_LW];
    __shared__ __attribute__((aligned(16))) unsigned long long s_cand[LQ16_W];
    __shared__ uint16_t s_list[3][LQ16_W];                        // one queue per stage
    __shared__ int s_count[3];
    __shared__ int s_lim[33];
    FIELD16(P);
    const int field = tff;
    cand += (size_t)fld * (P.fstride / 4);                         // the candidates live in the field's slot (64-bit words)
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    const int x0 = blockIdx.x * LQ16_W, t = threadIdx.x;
    const int ri = blockIdx.y;
    const int nrows = (height - (2 - field)) / 2;
    if (x0 >= width || ri >= nrows) return;
    if (maskless) return;                                          // no direction anywhere: q_lattice_resolve16 averages by itself
    const int y = (2 - field) + 2 * ri;
    if (t < 3) s_count[t] = 0;
    if (t < 33) s_lim[t] = k.limlut[t];
    {
        const uint16_t *g[5] = { Q.b + (size_t)(y - 1) * pitch, Q.b + (size_t)(y + 1) * pitch,
                                 Q.c + (size_t)(y - 1) * pitch, Q.c + (size_t)(y + 1) * pitch,
                                 Q.a + (size_t)y * pitch };
        // only as far right as the row's samples (+ halo) reach; groups of four samples
        const int need4 = (min(LQ16_W, (width - x0 + 3) & ~3) + 2 * LQ16_HALO) / 4;
        // every load of a thread in flight before its first LDS store, one branch per column (eedi2.hip: k_lattice_cand_q)
        static_assert(LQ16_LW / 4 <= 2 * 256, "two staged groups per row and thread");
        uint2 v[5][2] = {};
        const bool h0 = t < need4, h1 = t + 256 < need4;
        if (h0)
        {
#pragma unroll
            for (int r = 0; r < 5; r++) v[r][0] = reinterpret_cast<const uint2 *>(g[r] + x0 - LQ16_HALO)[t];
        }
        if (h1)
        {
#pragma unroll
            for (int r = 0; r < 5; r++) v[r][1] = reinterpret_cast<const uint2 *>(g[r] + x0 - LQ16_HALO)[t + 256];
        }
        if (h0)
        {
#pragma unroll
            for (int r = 0; r < 5; r++) reinterpret_cast<uint2 *>(s_rows[r])[t] = v[r][0];
        }
        if (h1)
        {
#pragma unroll
            for (int r = 0; r < 5; r++) reinterpret_cast<uint2 *>(s_rows[r])[t + 256] = v[r][1];
        }
    }
    __syncthreads();
    Lat16 L;
    L.peak = k.peak; L.neutral = k.neutral; L.sh = k.shift; L.sh2 = 2 + k.shift;
    L.three = 3 << k.shift; L.nine = 9 << k.shift;
    L.nt4 = (uint16_t)((nt << k.shift) * 4); L.nt7 = (uint16_t)((nt << k.shift) * 7); L.nt8 = (uint16_t)((nt << k.shift) * 8);
    L.nt7_unshifted = 7 * nt;
    L.lim = s_lim;
    const uint16_t *top = s_rows[0] + LQ16_HALO - x0, *bot = s_rows[1] + LQ16_HALO - x0;
    const uint16_t *ot = s_rows[2] + LQ16_HALO - x0, *ob = s_rows[3] + LQ16_HALO - x0;
    const uint16_t *dm = s_rows[4] + LQ16_HALO - x0;
    {
        // every sample's word, the variance and the edge test included, four samples per thread out of the rows' dwords
        const int x = x0 + 4 * t;
        if (x < width)
        {
            const uint32_t *T = reinterpret_cast<const uint32_t *>(s_rows[0]) + LQ16_HALO / 2 + 2 * t;   // T[0]: samples x, x + 1
            const uint32_t *B = reinterpret_cast<const uint32_t *>(s_rows[1]) + LQ16_HALO / 2 + 2 * t;
            const uint32_t *D = reinterpret_cast<const uint32_t *>(s_rows[4]) + LQ16_HALO / 2 + 2 * t;
            const uint32_t tw4[4] = { T[-1], T[0], T[1], T[2] }, bw4[4] = { B[-1], B[0], B[1], B[2] };
            const uint32_t dw3[3] = { D[0], D[1], D[2] };
            int tv[8], bv[8], dv[5];
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                tv[2 * i] = (int)(tw4[i] & 0xffffu); tv[2 * i + 1] = (int)(tw4[i] >> 16);
                bv[2 * i] = (int)(bw4[i] & 0xffffu); bv[2 * i + 1] = (int)(bw4[i] >> 16);
            }
            dv[0] = (int)(dw3[0] & 0xffffu); dv[1] = (int)(dw3[0] >> 16); dv[2] = (int)(dw3[1] & 0xffffu); dv[3] = (int)(dw3[1] >> 16);
            dv[4] = (int)(dw3[2] & 0xffffu);
            unsigned long long w[4], base[4];
            int lim[4];
            uint32_t queue = 0, searching = 0;
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                const int d = dv[j], dr = dv[j + 1];
                const int avg = (tv[j + 2] + bv[j + 2] + 1) >> 1;
                lim[j] = s_lim[iabs16(d - L.neutral) >> L.sh2];
                const unsigned long long right = iabs16(d - dr) > lim[j] ? 1ull << 49 : 0ull;
                if (d != L.peak && x + j < width) searching |= 1u << j;
                base[j] = (unsigned long long)(uint16_t)avg | right;
                w[j] = base[j] | lat16_word(avg, L.neutral) | (1ull << 48);                  // the word of a sample without a direction
            }
            if (searching)
            {
#pragma unroll
                for (int j = 0; j < 4; j++)
                {
                    const bool inner = x + j > 1 && x + j < width - 2;
                    const unsigned long long ws = lat16_stage_a_vals(tv[j], tv[j + 1], tv[j + 2], tv[j + 3], tv[j + 4],
                                                                     bv[j], bv[j + 1], bv[j + 2], bv[j + 3], bv[j + 4], lim[j], inner, base[j], L);
                    if ((searching >> j) & 1u)
                    {
                        w[j] = ws & ~LAT16_MORE;
                        if (ws & LAT16_MORE) queue |= 1u << j;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) s_cand[4 * t + j] = w[j];
            if (queue)
            {
                int at = atomicAdd(&s_count[1], __popc(queue));
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if ((queue >> j) & 1u) s_list[1][at++] = (uint16_t)(4 * t + j);
            }
        }
    }
    __syncthreads();
    {
        const Lat16Rows R = { reinterpret_cast<const uint32_t *>(s_rows[0]), reinterpret_cast<const uint32_t *>(s_rows[1]),
                              reinterpret_cast<const uint32_t *>(s_rows[2]), reinterpret_cast<const uint32_t *>(s_rows[3]), LQ16_HALO - x0 };
        for (int i = t, n = s_count[1]; i < n; i += 256)
        {
            const int lx = s_list[1][i];
            const unsigned long long w = lat16_stage_b_win(R, top, bot, ot, ob, dm, x0 + lx, width, s_cand[lx], L);
            if (w & LAT16_MORE) s_list[2][atomicAdd(&s_count[2], 1)] = (uint16_t)lx;
            else s_cand[lx] = w;
        }
    }
    __syncthreads();
    for (int i = t, n = s_count[2]; i < n; i += 256)
    {
        const int lx = s_list[2][i];
        s_cand[lx] = lat16_stage_c(top, bot, dm, x0 + lx, width, pl, s_cand[lx], L);
    }
    __syncthreads();
    {
        const int x = x0 + 4 * t;
        unsigned long long *o = cand + (size_t)pl * cand_plane_stride + (size_t)ri * cand_pitch + x;
        if (x + 3 < width && (((uintptr_t)o) & 15) == 0)
        {
            reinterpret_cast<uint4 *>(o)[0] = *reinterpret_cast<const uint4 *>(&s_cand[4 * t]);
            reinterpret_cast<uint4 *>(o)[1] = *reinterpret_cast<const uint4 *>(&s_cand[4 * t + 2]);
        }
        else for (int j = 0; j < 4 && x + j < width; j++) o[j] = s_cand[4 * t + j];
    }
}

// Four samples per thread and pass over the row, as k_lattice_resolve of the 8-bit engine (eedi2.hip): the candidates of a
// thread come in as two 16-byte loads, the direction row as 8 bytes, a thread composes the 2-state maps of its four samples
// before the wave scan, both rows leave as 8 bytes, limlut sits in LDS.
constexpr int LR16_T = 256, LR16_PX = 4 * LR16_T;

__device__ __forceinline__ unsigned lr16_compose(unsigned later, unsigned earlier)     // later o earlier (earlier applies first)
{
    return ((later >> (earlier & 1u)) & 1u) | (((later >> ((earlier >> 1) & 1u)) & 1u) << 1);
}

__global__ __launch_bounds__(LR16_T) void q_lattice_resolve16(Q3 P, K16 k, const unsigned long long *__restrict__ cand,
                                                              int cand_pitch, int cand_plane_stride)
{
    __shared__ uint8_t s_wmap[LR16_T / 64];        // composed map of each wave
    __shared__ uint8_t s_win[LR16_T / 64];         // resolved state entering each wave
    __shared__ int s_lim[33];
    __shared__ int s_carry;                        // outcome of the last pixel of the previous pass
    FIELD16(P);
    const int field = tff;
    cand += (size_t)fld * (P.fstride / 4);
    const int pitch = P.pitch[pl], width = P.width[pl], height = P.height[pl];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nrows = (height - 1 - (2 - field) + 1) / 2;
    uint16_t *dst = Q.b;
    if ((int)blockIdx.y >= nrows)
    {
        if ((int)blockIdx.y == nrows)                              // the one-row blit (:1162-1179)
            for (int xx = t; xx < width; xx += LR16_T)
            {
                if (field == 1) dst[(size_t)(height - 1) * pitch + xx] = dst[(size_t)(height - 2) * pitch + xx];
                else            dst[xx] = dst[pitch + xx];
            }
        return;
    }
    const int y = (2 - field) + 2 * blockIdx.y;
    uint16_t *mid = dst + (size_t)y * pitch;
    uint16_t *dm = Q.a + (size_t)y * pitch;
    if (maskless)
    {
        // no mask sample in the plane: every direction is a peak, every sample of the row the rounded mean of the samples
        // above and below it (:1192-1197), the direction row stays as it is
        const uint16_t *top = mid - pitch, *bot = mid + pitch;
        for (int xx = t; xx < width; xx += LR16_T) mid[xx] = (uint16_t)(((int)top[xx] + (int)bot[xx] + 1) >> 1);
        return;
    }
    const unsigned long long *cr = cand + (size_t)pl * cand_plane_stride + (size_t)blockIdx.y * cand_pitch;
    const bool cr16 = ((reinterpret_cast<uintptr_t>(cr)) & 15u) == 0;                   // block-uniform
    const bool row8 = ((reinterpret_cast<uintptr_t>(mid) | reinterpret_cast<uintptr_t>(dm)) & 7u) == 0;
    const int sh2 = 2 + k.shift;
    const int before_row = dm[-1];                                 // stands at dm[x-1] for x == 0; never written by this pass
    if (t == 0) s_carry = 0;
    if (t < 33) s_lim[t] = k.limlut[t];
    __syncthreads();
    for (int x0 = 0; x0 < width; x0 += LR16_PX)
    {
        const int x = x0 + 4 * t;
        const int nlive = min(max(width - x, 0), 4);             // samples of this thread inside the row
        unsigned long long c[4] = { 0ull, 0ull, 0ull, 0ull };
        int d[4] = { 0, 0, 0, 0 };
        if (nlive == 4 && cr16)
        {
            const uint4 v0 = reinterpret_cast<const uint4 *>(cr + x)[0], v1 = reinterpret_cast<const uint4 *>(cr + x)[1];
            c[0] = (unsigned long long)v0.x | ((unsigned long long)v0.y << 32); c[1] = (unsigned long long)v0.z | ((unsigned long long)v0.w << 32);
            c[2] = (unsigned long long)v1.x | ((unsigned long long)v1.y << 32); c[3] = (unsigned long long)v1.z | ((unsigned long long)v1.w << 32);
        }
        else
            for (int j = 0; j < nlive; j++) c[j] = cr[x + j];
        if (nlive == 4 && row8)
        {
            const uint2 v = *reinterpret_cast<const uint2 *>(dm + x);
            d[0] = (int)(v.x & 0xffffu); d[1] = (int)(v.x >> 16); d[2] = (int)(v.y & 0xffffu); d[3] = (int)(v.y >> 16);
        }
        else
            for (int j = 0; j < nlive; j++) d[j] = dm[x + j];
        // what the last sample of the thread to the left can leave behind (pa: outcome A, pb: outcome B); the first lane of a
        // wave reads that sample's candidate word
        int pa = __shfl_up(((c[3] >> 48) & 1ull) ? k.peak : k.neutral, 1, 64), pb = __shfl_up((int)((c[3] >> 32) & 0xffffull), 1, 64);
        if (lane == 0 && nlive)
        {
            if (x == 0) { pa = before_row; pb = before_row; }
            else
            {
                const unsigned long long cl = cr[x - 1];
                pa = ((cl >> 48) & 1ull) ? k.peak : k.neutral;
                pb = (int)((cl >> 32) & 0xffffull);
            }
        }
        unsigned m[4], pm[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            const unsigned long long cj = c[j];
            const int lim = s_lim[min(iabs16(d[j] - k.neutral) >> sh2, 32)];
            const bool always_a = (cj >> 48) & 1ull, right = (cj >> 49) & 1ull;
            if (j >= nlive || always_a) m[j] = 0u;
            else
            {
                const unsigned oa = (right && iabs16(d[j] - pa) > lim) ? 0u : 1u;
                const unsigned ob2 = (right && iabs16(d[j] - pb) > lim) ? 0u : 1u;
                m[j] = oa | (ob2 << 1);
            }
            pm[j] = j == 0 ? m[0] : lr16_compose(m[j], pm[j - 1]);
            pa = always_a ? k.peak : k.neutral;
            pb = (int)((cj >> 32) & 0xffffull);
        }
        unsigned tm = pm[3];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1)
        {
            const unsigned e = __shfl_up(tm, off, 64);
            if (lane >= off) tm = lr16_compose(tm, e);
        }
        if (lane == 63) s_wmap[wave] = (uint8_t)tm;
        unsigned before = __shfl_up(tm, 1, 64);
        if (lane == 0) before = 2u;                                  // the identity map
        __syncthreads();
        if (t == 0)
        {
            unsigned state = (unsigned)s_carry;                      // outcome of sample x0 - 1 (irrelevant for x0 == 0)
            for (int w = 0; w < LR16_T / 64; w++)
            {
                s_win[w] = (uint8_t)state;
                state = (s_wmap[w] >> state) & 1u;
            }
        }
        __syncthreads();
        const unsigned sin = (before >> s_win[wave]) & 1u;
        if (nlive)
        {
            uint32_t vm[4], vd[4];
            unsigned last = 0u;
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                const unsigned outcome = (pm[j] >> sin) & 1u;
                const unsigned long long cj = c[j];
                vm[j] = (uint32_t)(outcome ? (cj >> 16) & 0xffffull : cj & 0xffffull);
                vd[j] = outcome ? (uint32_t)((cj >> 32) & 0xffffull) : (uint32_t)(((cj >> 48) & 1ull) ? k.peak : k.neutral);
                if (j == nlive - 1) last = outcome;
            }
            if (nlive == 4 && row8)
            {
                *reinterpret_cast<uint2 *>(mid + x) = make_uint2(vm[0] | (vm[1] << 16), vm[2] | (vm[3] << 16));
                *reinterpret_cast<uint2 *>(dm + x) = make_uint2(vd[0] | (vd[1] << 16), vd[2] | (vd[3] << 16));
            }
            else
                for (int j = 0; j < nlive; j++) { mid[x + j] = (uint16_t)vm[j]; dm[x + j] = (uint16_t)vd[j]; }
            if (x + nlive == min(x0 + LR16_PX, width)) s_carry = (int)last;
        }
        __syncthreads();
    }
}

// ---- post-processing 2/3 (:1391-1904), as in eedi2.hip but on uint16 samples -----------------------
struct Corner16
{
    uint16_t *src, *tmp;
    int      *c[3], *t[3];
    int       pitch, width, height;
};

__device__ __forceinline__ int fold16(int centre, int d, int n, int &hi)
{
    int lo = centre - d;
    hi = centre + d;
    if (lo < 0) lo = hi;
    if (hi >= n) hi = lo;
    return lo;
}

template <bool VERT>
__global__ void q_blur1(Corner16 A)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= A.width || y >= A.height) return;
    const uint16_t *in = VERT ? A.tmp : A.src;
    uint16_t *out = VERT ? A.src : A.tmp;
    const int W[4] = { 26152, 15862, 3539, 291 };
    int acc = (int)in[(size_t)y * A.pitch + x] * W[0] + 32768;
#pragma unroll
    for (int d = 1; d <= 3; d++)
    {
        int hi;
        const int lo = fold16(VERT ? y : x, d, VERT ? A.height : A.width, hi);
        const size_t il = VERT ? (size_t)lo * A.pitch + x : (size_t)y * A.pitch + lo;
        const size_t ih = VERT ? (size_t)hi * A.pitch + x : (size_t)y * A.pitch + hi;
        acc += ((int)in[il] + (int)in[ih]) * W[d];
    }
    out[(size_t)y * A.pitch + x] = (uint16_t)(acc >> 16);
}

__global__ void q_derivatives(Corner16 A, int shift)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= A.width || y >= A.height) return;
    const uint16_t *s = A.src + (size_t)y * A.pitch;
    const uint16_t *up = A.src + (size_t)max(y - 1, 0) * A.pitch, *dn = A.src + (size_t)min(y + 1, A.height - 1) * A.pitch;
    const int ix = ((int)s[min(x + 1, A.width - 1)] - (int)s[max(x - 1, 0)]) >> shift;
    const int iy = ((int)up[x] - (int)dn[x]) >> shift;
    const size_t at = (size_t)y * A.pitch + x;
    A.c[0][at] = (ix * ix) >> 1;
    A.c[1][at] = (iy * iy) >> 1;
    A.c[2][at] = (ix * iy) >> 1;
}

template <bool VERT>
__global__ void q_blur_sqrt2(Corner16 A)
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
        int lo = fold16(VERT ? y : x, d, VERT ? A.height : A.width, hi);
        if (!VERT && d == 3 && x == A.width - 2) lo = hi = x + 3;                  // :1589
        const size_t il = VERT ? (size_t)lo * A.pitch + x : (size_t)y * A.pitch + lo;
        const size_t ih = VERT ? (size_t)hi * A.pitch + x : (size_t)y * A.pitch + hi;
        acc += (in[il] + in[ih]) * W[d];
    }
    out[(size_t)y * A.pitch + x] = acc >> (VERT ? 18 : 16);
}

__global__ void q_post_corner(Corner16 A, const uint16_t *msk, uint16_t *dst, int field, int height, int peak, int neutral)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y * blockDim.y + threadIdx.y;
    const int y = 8 - field + 2 * r;
    if (x < 4 || x >= A.width - 4 || y >= height - 7) return;
    const size_t at = (size_t)y * A.pitch + x;
    const int m = msk[at];
    if (m == peak || m == neutral) return;
    bool hit = false;
#pragma unroll
    for (int q = 0; q < 2; q++)
    {
        const size_t i = (size_t)(3 + r + q) * A.pitch + x;
        const int a = A.c[0][i], b = A.c[1][i], c = A.c[2][i];
        const double s = (double)(a + b);
        const double resp = (double)(a * b - c * c) - 0.09 * s * s;
        hit |= (int)resp > 775;
    }
    if (hit) dst[at] = (uint16_t)(((int)dst[at - A.pitch] + (int)dst[at + A.pitch] + 1) >> 1);
}

} // namespace

// ------------------------------------------------------------------- engine
// The field batching of the 8-bit engine (eedi2.hip): a slot per field (its nine scratch frames + lattice candidates),
// fields queued with add_field() and run by launch(): the field extraction for all of them in one launch, the five mask
// passes field after field (each field's edge mask reads the finished mask of the field before, eedi2_template.c:132),
// every later pass once with blockIdx.z = 3 * field + plane.
Eedi2Engine16::Eedi2Engine16(hbhip_ctx *ctx, const PicGeometry &geo, const Eedi2Params &p, int capacity)
    : EediEngineBase(ctx, geo, p, capacity, "decomb EEDI2 (16-bit)")
{
}

int Eedi2Engine16::init()
{
    if (geo_.bps != 2 || geo_.depth < 9 || geo_.depth > 16) return HBHIP_ERR_UNSUPPORTED;
    const int rc = init_slots({ 2 * GUARD16, sizeof(unsigned long long), QM_W, QM_H, QM_OY });     // (guard in bytes)
    if (rc != HBHIP_OK) return rc;
    HBHIP_CHECK(ctx_, hipStreamSynchronize(ctx_->stream));
    return HBHIP_OK;
}

// the per-depth constants the kernels take (eedi2_init_limlut :23-33)
static K16 make_k16(int depth)
{
    K16 k;
    k.shift = depth - 8;
    k.peak = (1 << depth) - 1;
    k.neutral = 1 << (depth - 1);
    static const int base[33] = { 6, 6, 7, 7, 8, 8, 9, 9, 9, 10, 10, 11, 11, 12, 12, 12, 12, 12, 12, 12,
                                  12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, -1, -1 };        // eedi2.c:21-25
    for (int i = 0; i < 33; i++) k.limlut[i] = (uint16_t)((uint16_t)base[i] << k.shift);
    return k;
}

// The field extraction (decomb_template.c:455-473) and the five mask passes (:390-397) of fields f0 .. f0 + n - 1 of the
// batch on st: one kernel, one launch (EediEngineBase::launch in eedi2.hip has the parts and the streams).
int Eedi2Engine16::enqueue_mask(int f0, int n, hbhip_ctx *lc, hipStream_t st, uint32_t *epoch_out)
{
    const EediFrame srcp = at_slot(half_[0], start_ + f0), mskp = at_slot(half_[1], start_ + f0),
                    old = at_slot(half_[1], f0 ? start_ + f0 - 1 : last_slot_);
    const K16 k = make_k16(geo_.depth);
    Q3 P;
    memset(&P, 0, sizeof(P));
    P.fstride = slot_bytes_ / 2;
    P.tffbits = tffbits_ >> f0;
    MaskSrc16 S;
    memset(&S, 0, sizeof(S));
    for (int f = 0; f < n; f++) for (int c = 0; c < 3; c++) S.frame[f][c] = (const uint16_t *)src_frame_[f0 + f][c];
    for (int c = 0; c < 3; c++)
    {
        P.pitch[c] = srcp.stride[c] / 2; P.width[c] = srcp.width[c]; P.height[c] = srcp.height[c];
        P.a[c] = (uint16_t *)srcp.plane[c]; P.b[c] = (uint16_t *)old.plane[c]; P.c[c] = (uint16_t *)mskp.plane[c];
        S.sp[c] = src_pitch_[c] / 2;
    }
    const int mth = par_.magnitude_threshold * 10, vth = par_.laplacian_threshold * 81, lth = par_.variance_threshold;   // sic: swapped (decomb_template.c:390)
    const unsigned gx = (srcp.width[0] + QM_W - 1) / QM_W, gy = (srcp.height[0] + QM_H - 1) / QM_H;
    uint32_t epoch = 0;
    { const int erc = next_epoch(lc, &epoch); if (erc != HBHIP_OK) return erc; }
    *epoch_out = epoch;
    uint32_t *pflags = plane_flags_ + 3 * f0;
    if (n == 1)
        HBHIP_LAUNCH_ON(lc, st, "eedi2_16_mask_passes", q_mask_fused, dim3(gx, gy, 3), dim3(QM_T), 0, P, S, k, 0, 0, mth, vth, lth,
                        par_.erosion_threshold, par_.dilation_threshold, pflags, epoch);
    else
    {
        // one launch, field-major: a field's chain tiles, then its upper tiles (see Eedi2Engine::enqueue_mask)
        MaskChain C = eedi_mask_chain_tiles(srcp, QM_W, QM_H, QM_OY);
        C.flags = chain_flags_;
        C.pflags = pflags;
        C.epoch = epoch;
        C.group = C.ntiles + C.nupper;
        C.has = chain_has_ + (size_t)f0 * C.group;
        guard_.bind(C);
        HBHIP_LAUNCH_ON(lc, st, "eedi2_16_mask_passes", q_mask_chain, dim3((unsigned)(C.group * n)), dim3(QM_T), 0, P, S, k, C, mth, vth, lth,
                        par_.erosion_threshold, par_.dilation_threshold);
        HBHIP_LAUNCH_ON(lc, st, "eedi2_16_mask_repair", q_mask_chain_repair, dim3(3u * (unsigned)n), dim3(QM_T), 0, P, S, k, C, n, mth, vth, lth,
                        par_.erosion_threshold, par_.dilation_threshold);
    }
    HBHIP_CHECK(lc, hipGetLastError());
    return HBHIP_OK;
}

// the passes behind the mask for fields f0 .. f0 + n - 1 of the batch, on st
int Eedi2Engine16::enqueue_passes(int f0, int n, hbhip_ctx *lc, hipStream_t st, uint32_t epoch)
{
    const EediFrame srcp = at_slot(half_[0], start_ + f0), mskp = at_slot(half_[1], start_ + f0), tmpp = at_slot(half_[2], start_ + f0),
                    dstp = at_slot(half_[3], start_ + f0);
    const EediFrame dst2p = at_slot(full_[0], start_ + f0), tmp2p2 = at_slot(full_[1], start_ + f0), msk2p = at_slot(full_[2], start_ + f0),
                    tmp2p = at_slot(full_[3], start_ + f0), dst2mp = at_slot(full_[4], start_ + f0);
    unsigned long long *cand = reinterpret_cast<unsigned long long *>(cand_raw_) + (size_t)(start_ + f0) * (slot_bytes_ / sizeof(unsigned long long));
    const K16 k = make_k16(geo_.depth);

    const dim3 blk(64, 4);
    const unsigned gz = 3u * (unsigned)n;
    auto geom = [&](Q3 &P, const EediFrame &f) {
        for (int c = 0; c < 3; c++) { P.pitch[c] = f.stride[c] / 2; P.width[c] = f.width[c]; P.height[c] = f.height[c]; }
    };
    auto bind = [&](uint16_t *(&slot)[3], const EediFrame &f) { for (int c = 0; c < 3; c++) slot[c] = (uint16_t *)f.plane[c]; };
    auto grid = [&](const EediFrame &f, bool whole_pitch, unsigned z) {
        const int w = whole_pitch ? f.stride[0] / 2 : f.width[0];
        return dim3((w + 63) / 64, (f.height[0] + 3) / 4, z);
    };
    auto grid4 = [&](const EediFrame &f, unsigned z) {                          // four samples per thread, blocks of 64 x 4 threads
        return dim3(hbhip_grid_x((f.width[0] + 255) / 256), (f.height[0] + 3) / 4, z);
    };
    auto grid4p = [&](const EediFrame &f, unsigned z) {                         // the same with a thread row per PAIR of rows (step 2)
        return dim3(hbhip_grid_x((f.width[0] + 255) / 256), ((f.height[0] + 1) / 2 + 3) / 4, z);
    };
    Q3 P;
    memset(&P, 0, sizeof(P));
    P.fstride = slot_bytes_ / 2;
    P.tffbits = tffbits_ >> f0;

    // half-height passes (decomb_template.c:398-404), all fields per launch from here on
    geom(P, srcp);
    P.pflags = plane_flags_ + 3 * f0;
    P.pepoch = epoch;
    // filter_dir_map and expand_dir_map as one launch (q_dir_map_fe; eedi2.hip: Eedi2Engine::enqueue_passes): calc_directions
    // then writes dstp, and leaves the padding of its rows alone, so that the fused pass leaves the expanded map - and the
    // padding calc_directions' fill gives it - in tmpp, where the reference has them
    const bool fused = par_.maximum_search_distance <= QHALO - 2 && hbhip_dev_int("HBHIP_EEDI2_FUSE_DIRMAP", 1) != 0;
    bind(P.a, mskp); bind(P.b, srcp); bind(P.c, fused ? dstp : tmpp);
    if (par_.maximum_search_distance <= QHALO - 2)
        // 256 columns x 4 rows per block, the mostly listed blocks in the dense form (eedi2.hip: Eedi2Engine::enqueue_passes);
        // its keys hold sums of 12-bit samples at most
        HBHIP_LAUNCH_ON(lc, st, "eedi2_16_calc_directions", q_calc_dir_rows<4>, dim3(hbhip_grid_x((srcp.stride[0] / 2 + QW - 1) / QW), (srcp.height[0] + 3) / 4, gz),
                     dim3(QW), 0, P, k, par_.maximum_search_distance, par_.noise_threshold, k.peak < (1 << 12) ? QW * 4 / 2 : 1 << 30,
                     fused ? 0 : k.peak);
    else
        HBHIP_LAUNCH_ON(lc, st, "eedi2_16_calc_directions", q_calc_dir, grid(srcp, true, gz), blk, 0, P, k, par_.maximum_search_distance, par_.noise_threshold);
    if (fused)
    {
        bind(P.a, mskp); bind(P.b, dstp); bind(P.c, tmpp);
        HBHIP_LAUNCH_ON(lc, st, "eedi2_16_filter_expand_dir_map", (q_dir_map_fe<1, false>),
                        dim3(hbhip_grid_x((srcp.stride[0] / 2 + 255) / 256), (srcp.height[0] + QFE_R - 1) / QFE_R, gz), blk, 0, P, k, k.peak);
    }
    else
    {
        bind(P.a, mskp); bind(P.b, tmpp); bind(P.c, dstp);
        HBHIP_LAUNCH_ON(lc, st, "eedi2_16_filter_dir_map", q_dir_map4, grid4(srcp, gz), blk, 0, P, k, 1, 0);
        bind(P.a, mskp); bind(P.b, dstp); bind(P.c, tmpp);
        HBHIP_LAUNCH_ON(lc, st, "eedi2_16_expand_dir_map", q_dir_map, grid4(srcp, gz), blk, 0, P, k, 1, 1, 0);
    }
    bind(P.a, mskp); bind(P.b, tmpp); bind(P.c, dstp);
    HBHIP_LAUNCH_ON(lc, st, "eedi2_16_filter_map", q_filter_map, grid4(srcp, gz), blk, 0, P, k);
    // the three line doublings + mark_directions_2x in one launch (full-height geometry)
    geom(P, dst2p);
    // the pair of _2x dir-map passes behind it as one launch too (q_dir_map_fe<2>): the marked map then goes to dst2mp; and the
    // pair in front of post_process with tmp2p and tmp2p2 in each other's places from here on, so that it reads the map where
    // the reference's copy of it would go (no blit) and writes the new one where the reference has it (eedi2.hip:
    // Eedi2Engine::enqueue_passes has the whole argument)
    const bool fused2 = hbhip_dev_int("HBHIP_EEDI2_FUSE_DIRMAP_2X", 1) != 0;
    const bool post1 = par_.post_processing == 1 || par_.post_processing == 3;
    const bool swapped = fused2 && post1 && hbhip_dev_int("HBHIP_EEDI2_FUSE_DIRMAP_POST", 1) != 0;
    const EediFrame &map2 = swapped ? tmp2p2 : tmp2p, &omsk2 = swapped ? tmp2p : tmp2p2;
    bind(P.a, mskp); bind(P.b, dstp); bind(P.g, srcp); bind(P.c, fused2 ? dst2mp : map2); bind(P.d, dst2p); bind(P.e, omsk2); bind(P.f, msk2p);
    HBHIP_LAUNCH_ON(lc, st, "eedi2_16_mark_directions_2x", q_mark_2x,                                    // a thread row per pair of rows, four samples per thread
                 dim3(hbhip_grid_x((dst2p.stride[0] / 2 + 255) / 256), ((dst2p.height[0] + 1) / 2 + 3) / 4, gz), blk, 0, P, k, fused2 ? 0 : k.peak);
    const dim3 fe2_grid(hbhip_grid_x((dst2p.stride[0] / 2 + 255) / 256), ((dst2p.height[0] + 1) / 2 + QFE_R - 1) / QFE_R, gz);
    if (fused2)
    {
        bind(P.a, msk2p); bind(P.b, dst2mp); bind(P.c, map2);
        HBHIP_LAUNCH_ON(lc, st, "eedi2_16_filter_expand_dir_map_2x", (q_dir_map_fe<2, false>), fe2_grid, blk, 0, P, k, swapped ? 0 : k.peak);
    }
    else
    {
        bind(P.a, msk2p); bind(P.b, tmp2p); bind(P.c, dst2mp);
        HBHIP_LAUNCH_ON(lc, st, "eedi2_16_filter_dir_map_2x", q_dir_map4, grid4p(dst2p, gz), blk, 0, P, k, 2, 0);
        bind(P.a, msk2p); bind(P.b, dst2mp); bind(P.c, tmp2p);
        HBHIP_LAUNCH_ON(lc, st, "eedi2_16_expand_dir_map_2x", q_dir_map, grid4p(dst2p, gz), blk, 0, P, k, 2, 1, 0);
    }
    for (int pass = 0; pass < 2; pass++)
    {
        const EediFrame &in = pass ? dst2mp : map2, &out = pass ? map2 : dst2mp;
        bind(P.a, msk2p); bind(P.b, in); bind(P.c, out);
        HBHIP_LAUNCH_ON(lc, st, "eedi2_16_fill_gaps_2x", q_fill_gaps_b, dim3(hbhip_grid_x((dst2p.width[0] + QF_W - 1) / QF_W), (dst2p.height[0] + 2 * QF_R - 1) / (2 * QF_R), gz), dim3(256), 0, P, k);
    }
    bind(P.a, map2); bind(P.b, dst2p); bind(P.c, omsk2);
    {
        const int nrows = (dst2p.height[0] - 1) / 2;                // rows y0, y0 + 2, ... < height - 1 for either parity (even heights)
        HBHIP_LAUNCH_ON(lc, st, "eedi2_16_lattice_candidates", q_lattice_cand, dim3(hbhip_grid_x((dst2p.width[0] + LQ16_W - 1) / LQ16_W), nrows, gz), dim3(256), 0, P, k,
                     par_.noise_threshold, cand, cand_pitch_, cand_plane_stride_);
        HBHIP_LAUNCH_ON(lc, st, "eedi2_16_lattice_resolve", q_lattice_resolve16, dim3(1, nrows + 1, gz), dim3(LR16_T), 0, P, k,
                     (const unsigned long long *)cand, cand_pitch_, cand_plane_stride_);
    }
    if (swapped)
    {
        // filter_dir_map_2x, expand_dir_map_2x and post_process in one launch: tmp2p2 -> tmp2p, the filtered map to dst2mp,
        // the corrections to dst2p
        bind(P.a, msk2p); bind(P.b, tmp2p2); bind(P.c, tmp2p); bind(P.d, dst2mp); bind(P.f, dst2p);
        HBHIP_LAUNCH_ON(lc, st, "eedi2_16_filter_expand_dir_map_2x_post", (q_dir_map_fe<2, true>), fe2_grid, blk, 0, P, k, k.peak);
    }
    else if (post1)
    {
        // (the copy as a store of the filter behind it, as the 8-bit engine has it, cost that launch what the blit costs: 44 us)
        bind(P.a, tmp2p); bind(P.c, tmp2p2);
        HBHIP_LAUNCH_ON(lc, st, "eedi2_16_blit", q_blit, dim3(hbhip_grid_x((dst2p.width[0] + 511) / 512), (dst2p.height[0] + 3) / 4, gz), blk, 0, P);   // eedi2_bit_blit(tmp2p -> tmp2p2)
        bind(P.a, msk2p); bind(P.b, tmp2p); bind(P.c, dst2mp);
        HBHIP_LAUNCH_ON(lc, st, "eedi2_16_filter_dir_map_2x", q_dir_map4, grid4p(dst2p, gz), blk, 0, P, k, 2, 0);
        // + eedi2_post_process (new map = what the pass writes, old map tmp2p2, picture dst2p): folded into the pass
        bind(P.a, msk2p); bind(P.b, dst2mp); bind(P.c, tmp2p); bind(P.e, tmp2p2); bind(P.f, dst2p);
        HBHIP_LAUNCH_ON(lc, st, "eedi2_16_expand_dir_map_2x", q_dir_map, grid4p(dst2p, gz), blk, 0, P, k, 2, 1, 1);
    }
    if (par_.post_processing == 2 || par_.post_processing == 3)
    {
        // field after field, plane after plane: the derivative arrays carry values along (eedi2.hip, CornerArgs)
        for (int f = 0; f < n; f++)
        {
            const int tff = (int)((tffbits_ >> (f0 + f)) & 1u);
            const size_t foff = (size_t)f * slot_bytes_;
            for (int c = 0; c < 3; c++)
            {
                Corner16 A;
                A.src = (uint16_t *)(srcp.plane[c] + foff); A.tmp = (uint16_t *)(tmpp.plane[c] + foff);
                for (int i = 0; i < 3; i++) { A.c[i] = deriv_[i]; A.t[i] = deriv_tmp_[i]; }
                A.pitch = srcp.stride[c] / 2; A.width = srcp.width[c]; A.height = srcp.height[c];
                const dim3 g1((A.width + 63) / 64, (A.height + 3) / 4, 1), g3(g1.x, g1.y, 3);
                HBHIP_LAUNCH_ON(lc, st, "eedi2_16_gaussian_blur1_h", q_blur1<false>, g1, blk, 0, A);
                HBHIP_LAUNCH_ON(lc, st, "eedi2_16_gaussian_blur1_v", q_blur1<true>, g1, blk, 0, A);
                HBHIP_LAUNCH_ON(lc, st, "eedi2_16_calc_derivatives", q_derivatives, g1, blk, 0, A, k.shift);
                HBHIP_LAUNCH_ON(lc, st, "eedi2_16_gaussian_blur_sqrt2_h", q_blur_sqrt2<false>, g3, blk, 0, A);
                HBHIP_LAUNCH_ON(lc, st, "eedi2_16_gaussian_blur_sqrt2_v", q_blur_sqrt2<true>, g3, blk, 0, A);
                const int rows = (dst2p.height[c] - 7 - (8 - tff) + 1) / 2;
                if (rows > 0)
                    HBHIP_LAUNCH_ON(lc, st, "eedi2_16_post_process_corner", q_post_corner, dim3((A.width + 63) / 64, (rows + 3) / 4, 1), blk, 0, A,
                                 (const uint16_t *)(tmp2p2.plane[c] + foff), (uint16_t *)(dst2p.plane[c] + foff), tff, dst2p.height[c], k.peak, k.neutral);
            }
        }
    }
    HBHIP_CHECK(lc, hipGetLastError());
    return HBHIP_OK;
}
