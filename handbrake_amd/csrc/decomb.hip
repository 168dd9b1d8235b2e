// decomb.hip — decomb (yadif / blend / cubic / EEDI2-guided) for gfx950, 8 / 10 / 12-bit samples.
//
//   decomb_plane4_kernel  replaces yadif_decomb_filter_work + yadif_filter_line,
//                         cubic_interpolate_line, blend_filter_line and the row
//                         copies of filter (both instantiations of the template)
//                         (libhb/templates/decomb_template.c:43-107, 279-361, 579-898)
//   DecombFilter          replaces store_ref / process_frame / hb_decomb_work frame
//                         logic               (libhb/decomb.c:195-200, 495-612)
//
// Four samples per thread, the planes of up to 16 frames per launch.  Rows of the parity being
// rebuilt are interpolated, the others are copied from the current frame; every
// decision (vertical-edge rows, the x margin of the spatial search, first/last
// row stride mirroring, C-style truncating /40) follows the reference so the
// output is bit-exact.  HBM-bound: reads prev/cur/next (+EEDI2 guess), writes one
// frame = 4 (5) bytes per pixel algorithmic; vertical neighbours come from L2.
#include "hbhip_internal.h"
#include "eedi2_engine.h"
#include <algorithm>
#include <vector>

namespace {

enum { M_YADIF = 1, M_BLEND = 2, M_CUBIC = 4, M_EEDI2 = 8, M_BOB = 16, M_SELECTIVE = 32 };
enum { PIC_TFF = 0x0008, PIC_PROGRESSIVE = 0x0010 };

struct DecombPlane
{
    const uint8_t *prev, *cur, *next, *guess;
    uint8_t       *dst;
    int            pitch, guess_pitch, dst_pitch, w, h;
};

struct DecombArgs
{
    DecombPlane pl[3];
    int mode;            // resolved per-frame mode (decomb_template.c:820-833)
    int parity;          // rows with (y & 1) == (parity ? 0 : 1) are rebuilt
    int field_parity;    // parity ^ tff, the `parity` argument of yadif_filter_line
};

// the reference's crop table (init_crop_table, :23-41): clamp to [0, max_value]
__device__ __forceinline__ int cropv(int v, int maxv) { return v < 0 ? 0 : v > maxv ? maxv : v; }
__device__ __forceinline__ int cubic4(int y0, int y1, int y2, int y3, int maxv)
{
    return cropv((y0 * -3 + y1 * 23 + y2 * 23 + y3 * -3) / 40, maxv);      // :43-48, C division
}
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }

// ---- decomb_plane_kernel with four samples per thread and several frames per launch (every depth) --------------
// One plane of one 1080p frame is ~2 MB: a launch per frame is mostly dispatch latency, and a thread per byte spends
// its time on load instructions.  Here a thread owns one aligned dword of its row (the rows it needs come in as
// dwords or 12-byte windows x-4 .. x+7, the arithmetic is the scalar kernel's, per byte), and grid.z runs over the
// planes of up to DB_FRAMES frames that share the geometry (the frames of a chain batch).
constexpr int DB_FRAMES = 16;

struct DecombFrame
{
    const uint8_t *prev[3], *cur[3], *next[3], *guess[3];
    uint8_t       *dst[3];
    int mode, parity, field_parity, pad;
};

struct DecombBatch
{
    DecombFrame f[DB_FRAMES];
    int pitch[3], guess_pitch[3], dst_pitch[3], w[3], h[3];
    int n = 0;
    int ff = 0;          // 0: decomb's yadif; 1 / 2: FFmpeg's (the Deinterlace filter), 2 = its nospatial modes - see the kernel
};

// four adjacent samples of a row as they lie in memory: a dword of bytes, or two dwords of 16-bit samples
template <typename PIX> struct Px4;
template <> struct Px4<uint8_t>
{
    typedef uint32_t T;
    static __device__ __forceinline__ T zero() { return 0u; }
    static __device__ __forceinline__ int get(T v, int k) { return (int)((v >> (8 * k)) & 0xffu); }
    static __device__ __forceinline__ T pack(const int (&o)[4]) { return ((uint32_t)o[0] & 0xffu) | (((uint32_t)o[1] & 0xffu) << 8) | (((uint32_t)o[2] & 0xffu) << 16) | ((uint32_t)o[3] << 24); }
};
template <> struct Px4<uint16_t>
{
    typedef uint2 T;
    static __device__ __forceinline__ T zero() { return make_uint2(0u, 0u); }
    static __device__ __forceinline__ int get(T v, int k) { return (int)(((k < 2 ? v.x : v.y) >> (16 * (k & 1))) & 0xffffu); }
    static __device__ __forceinline__ T pack(const int (&o)[4]) { return make_uint2(((uint32_t)o[0] & 0xffffu) | ((uint32_t)o[1] << 16), ((uint32_t)o[2] & 0xffffu) | ((uint32_t)o[3] << 16)); }
};
template <typename PIX> struct W12 { typename Px4<PIX>::T w0, w1, w2; };       // samples x-4 .. x+7 of a row

template <typename PIX>
__device__ __forceinline__ int w12b(const W12<PIX> &w, int i)     // sample at column x+i, i in [-4, 7] (constant after unrolling)
{
    const int k = i + 4;
    return Px4<PIX>::get(k < 4 ? w.w0 : (k < 8 ? w.w1 : w.w2), k & 3);
}

// the units either side are only read where they exist inside the row (pitch in samples)
template <typename PIX>
__device__ __forceinline__ W12<PIX> ldw12(const uint8_t *row, int x, int pitch)
{
    typedef typename Px4<PIX>::T T;
    const T *p = reinterpret_cast<const T *>(row + (size_t)x * sizeof(PIX));
    return W12<PIX>{ x >= 4 ? p[-1] : Px4<PIX>::zero(), p[0], x + 4 < pitch ? p[1] : Px4<PIX>::zero() };
}

template <typename PIX>
__global__ __launch_bounds__(256) void decomb_plane4_kernel(DecombBatch B, int maxv)
{
    typedef Px4<PIX> X4;
    typedef typename X4::T T;
    typedef W12<PIX> WIN;
    const int pl = blockIdx.z % 3;
    const DecombFrame &F = B.f[blockIdx.z / 3];
    const int w = B.w[pl], h = B.h[pl], st = B.pitch[pl];             // st: bytes between rows
    const int pitch_s = st / (int)sizeof(PIX);
    const int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    const uint8_t *cb = F.cur[pl];
    const uint8_t *crow = cb + (size_t)y * st;
    uint8_t *orow = F.dst[pl] + (size_t)y * B.dst_pitch[pl];
    const int mode = F.mode;
    auto store = [&](T v) {
        PIX *o = reinterpret_cast<PIX *>(orow) + x;
        if (x + 3 < w) *reinterpret_cast<T *>(o) = v;
        else for (int k = 0; k < 4 && x + k < w; k++) o[k] = (PIX)X4::get(v, k);
    };
    auto dw = [&](const uint8_t *row) { return *reinterpret_cast<const T *>(row + (size_t)x * sizeof(PIX)); };
    auto dwb = [](T v, int k) { return X4::get(v, k); };

    if (mode == 0) { store(dw(crow)); return; }                                    // pass-through (:892-897)
    if ((mode & M_EEDI2) && !(mode & M_YADIF))                                    // EEDI2 only (:855-875)
    {
        store(dw(F.guess[pl] + (size_t)y * B.guess_pitch[pl]));
        return;
    }
    if ((y & 1) != (F.parity ? 0 : 1)) { store(dw(crow)); return; }               // kept field (:795-807)

    int o4[4] = { 0, 0, 0, 0 };
    if (mode == M_BLEND)                                                           // :300-361
    {
        int u1, u2, d1, d2;
        if (y > 1 && y < h - 2) { u1 = -st; u2 = -2 * st; d1 = st; d2 = 2 * st; }
        else if (y == 0)        { u1 = u2 = 0; d1 = st; d2 = 2 * st; }
        else if (y == 1)        { u1 = u2 = -st; d1 = st; d2 = 2 * st; }
        else if (y == h - 2)    { u1 = -st; u2 = -2 * st; d1 = d2 = st; }
        else                    { u1 = -st; u2 = -2 * st; d1 = d2 = 0; }
        const T a2 = dw(crow + u2), a1 = dw(crow + u1), c0 = dw(crow), b1 = dw(crow + d1), b2 = dw(crow + d2);
#pragma unroll
        for (int k = 0; k < 4; k++)
            o4[k] = cropv((-dwb(a2, k) + 2 * dwb(a1, k) + 6 * dwb(c0, k) + 2 * dwb(b1, k) - dwb(b2, k)) >> 3, maxv);
        store(X4::pack(o4));
        return;
    }
    if (mode == M_CUBIC)                                                           // :50-107
    {
        // which rows stand in for p0..p3 depends on y only
        int o0, o1, o2, o3;
        if (y >= 3)                { o0 = -3 * st; o1 = -st; }
        else if (y == 2 || y == 1) { o0 = o1 = -st; }
        else                       { o0 = o1 = st; }
        if (y <= h - 4)                    { o2 = st; o3 = 3 * st; }
        else if (y == h - 3 || y == h - 2) { o2 = o3 = st; }
        else                               { o2 = o3 = -st; }
        const T p0 = dw(crow + o0), p1 = dw(crow + o1), p2 = dw(crow + o2), p3 = dw(crow + o3);
#pragma unroll
        for (int k = 0; k < 4; k++) o4[k] = cubic4(dwb(p0, k), dwb(p1, k), dwb(p2, k), dwb(p3, k), maxv);
        store(X4::pack(o4));
        return;
    }
    if (!(mode & M_YADIF)) return;                                                 // untouched, as the reference leaves it

    // ---- yadif_filter_line (:579-712)
    const uint8_t *prow = F.prev[pl] + (size_t)y * st, *nrow = F.next[pl] + (size_t)y * st;
    const uint8_t *p2row = F.field_parity ? prow : crow, *n2row = F.field_parity ? crow : nrow;
    const int sp = y ? -st : st;
    const int sn = y + 1 < h ? st : -st;
    // the rows next to the top / bottom edge skip the test against rows y +- 2: decomb's three rows either side
    // (decomb_template.c:600-603), vf_yadif.c's one (filter_slice: `mode = 2` for y == 1 and y + 2 == h, or every row in
    // the nospatial modes) - it then mirrors where decomb does not get to (2 * sp / 2 * sn below)
    const bool vertical_edge = B.ff ? (B.ff == 2 || y == 1 || y + 2 == h) : (y < 3) || (y > h - 4);
    const bool use_cubic = (mode & M_CUBIC) && !vertical_edge;
    const int margin = (mode & M_CUBIC) ? 3 : 2;
    const bool spatial = !(mode & M_EEDI2);

    const T p20 = dw(p2row), n20 = dw(n2row);
    const T ppu = dw(prow + sp), ppd = dw(prow + sn), pnu = dw(nrow + sp), pnd = dw(nrow + sn);
    T p2a = X4::zero(), p2b = X4::zero(), n2a = X4::zero(), n2b = X4::zero();         // rows y-2 / y+2
    if (!vertical_edge) { p2a = dw(p2row + 2 * sp); p2b = dw(p2row + 2 * sn); n2a = dw(n2row + 2 * sp); n2b = dw(n2row + 2 * sn); }
    T g = X4::zero();
    if (!spatial) g = dw(F.guess[pl] + (size_t)y * B.guess_pitch[pl]);
    WIN cu, cd, cu3 = { X4::zero(), X4::zero(), X4::zero() }, cd3 = cu3;             // rows y+sp, y+sn, y-3, y+3 of cur
    if (spatial)
    {
        cu = ldw12<PIX>(crow + sp, x, pitch_s); cd = ldw12<PIX>(crow + sn, x, pitch_s);
        if (use_cubic) { cu3 = ldw12<PIX>(crow - 3 * st, x, pitch_s); cd3 = ldw12<PIX>(crow + 3 * st, x, pitch_s); }
    }
    else
    {
        cu = WIN{ X4::zero(), dw(crow + sp), X4::zero() }; cd = WIN{ X4::zero(), dw(crow + sn), X4::zero() };
    }
    // 8-bit samples: the diagonal scores - three absolute differences between the row above and the row below, for the
    // five slopes of four samples - are v_sad_u8 of three-byte windows cut out of the rows' twelve bytes (v_perm_b32, the
    // fourth byte zero): 16 + 20 instructions for all of a thread's scores instead of ~17 for each of the twenty
    uint32_t CU[9] = {}, CD[9] = {};                                            // window starting at byte s of cu / cd, s = 1 .. 8
    if constexpr (sizeof(PIX) == 1)
    {
        if (spatial)
        {
#pragma unroll
            for (int s = 1; s <= 8; s++)
            {
                const int q = s <= 5 ? s : s - 4;
                const uint32_t sel = (uint32_t)q | ((uint32_t)(q + 1) << 8) | ((uint32_t)(q + 2) << 16) | 0x0c000000u;
                CU[s] = s <= 5 ? __builtin_amdgcn_perm((uint32_t)cu.w1, (uint32_t)cu.w0, sel) : __builtin_amdgcn_perm((uint32_t)cu.w2, (uint32_t)cu.w1, sel);
                CD[s] = s <= 5 ? __builtin_amdgcn_perm((uint32_t)cd.w1, (uint32_t)cd.w0, sel) : __builtin_amdgcn_perm((uint32_t)cd.w2, (uint32_t)cd.w1, sel);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        const int cc = w12b(cu, k), e = w12b(cd, k);
        const int P2 = dwb(p20, k), N2 = dwb(n20, k);
        const int d = (P2 + N2) >> 1;
        const int td0 = abs(P2 - N2);
        const int td1 = (abs(dwb(ppu, k) - cc) + abs(dwb(ppd, k) - e)) >> 1;
        const int td2 = (abs(dwb(pnu, k) - cc) + abs(dwb(pnd, k) - e)) >> 1;
        int diff = max3i(td0 >> 1, td1, td2);
        int pred;
        if (!spatial) pred = dwb(g, k);
        else
        {
            // c[sp + i] = cu(k + i), c[sn + i] = cd(k + i); with the cubic predictor sp = -st, sn = st (no vertical edge)
            pred = use_cubic ? cubic4(w12b(cu3, k), cc, e, w12b(cd3, k), maxv) : (cc + e) >> 1;
            const int xx = x + k;
            if (xx > margin && xx < w - (margin + 1))
            {
                auto slope = [&](int j) -> int {                                // samples k-1+j .. k+1+j above against k-1-j .. k+1-j below
                    if constexpr (sizeof(PIX) == 1)
                        return (int)__builtin_amdgcn_sad_u8(CU[3 + k + j], CD[3 + k - j], 0u);
                    else
                        return abs(w12b(cu, k - 1 + j) - w12b(cd, k - 1 - j)) + abs(w12b(cu, k + j) - w12b(cd, k - j)) +
                               abs(w12b(cu, k + 1 + j) - w12b(cd, k + 1 - j));
                };
                int best = slope(0) - 1;
                auto check = [&](int j) -> bool {
                    const int score = slope(j);
                    if (score >= best) return false;
                    best = score;
                    if (use_cubic)
                    {
                        // :541-570
                        if (j == -1)      pred = cubic4(w12b(cu3, k - 3), w12b(cu, k - 1), w12b(cd, k + 1), w12b(cd3, k + 3), maxv);
                        else if (j == -2) pred = cubic4((w12b(cu3, k - 4) + w12b(cu, k - 4)) / 2, w12b(cu, k - 2), w12b(cd, k + 2),
                                                        (w12b(cd3, k + 4) + w12b(cd, k + 4)) / 2, maxv);
                        else if (j == 1)  pred = cubic4(w12b(cu3, k + 3), w12b(cu, k + 1), w12b(cd, k - 1), w12b(cd3, k - 3), maxv);
                        else              pred = cubic4((w12b(cu3, k + 4) + w12b(cu, k + 4)) / 2, w12b(cu, k + 2), w12b(cd, k - 2),
                                                        (w12b(cd3, k - 4) + w12b(cd, k - 4)) / 2, maxv);
                    }
                    else pred = (w12b(cu, k + j) + w12b(cd, k - j)) >> 1;
                    return true;
                };
                if (check(-1)) check(-2);
                if (check(1)) check(2);
            }
        }
        if (!vertical_edge)
        {
            const int b = (dwb(p2a, k) + dwb(n2a, k)) >> 1;
            const int f = (dwb(p2b, k) + dwb(n2b, k)) >> 1;
            const int mx = max3i(d - e, d - cc, min(b - cc, f - e));
            const int mn = min3i(d - e, d - cc, max(b - cc, f - e));
            diff = max3i(diff, mn, -mx);
        }
        if (pred > d + diff)      pred = d + diff;
        else if (pred < d - diff) pred = d - diff;
        o4[k] = pred;
    }
    store(X4::pack(o4));
}

// ------------------------------------------------------------------- host side
// ------------------------------------------------------------------------------------------
// FFmpeg's yadif, which is what the reference's "Deinterlace" filter is (libhb/deinterlace.c:43-143
// only builds `yadif=mode=send_frame|send_field[_nospatial]:deint=...:parity=...` for libavfilter;
// the arithmetic is vf_yadif.c's - not in the reference tree, parity unpinned, restated in
// oracle/decomb_oracle.c:orc_yadif_ff_plane).  Same frame ring as decomb (first frame filtered
// against itself as `prev`, last one against itself as `next`), same field order.  Per rebuilt
// sample: temporal prediction d = (prev2 + next2) / 2, bounded to +-diff around it, where diff is
// the largest of the three temporal differences and - unless `nospatial` or next to the top /
// bottom edge - is widened by the vertical-neighbour check over rows y +- 2; the spatial
// prediction is the average of the rows above / below, replaced by a diagonal average when one of
// the +-1 (then +-2) diagonals matches better - only for 3 <= x < w - 3 (vf_yadif.c: filter_edges).
// That is decomb's yadif without the cubic predictor and with another rule for the rows next to the edge:
// it runs as decomb_plane4_kernel with DecombBatch::ff set (four samples per thread, the frames of a batch
// in one launch) since round 5; the one-sample-per-thread kernel this comment used to head is gone.

// FFmpeg's bwdif as the reference's "Bwdif" filter configures it (deinterlace.c:46 -> vf_bwdif.c; parity
// unpinned, restated in oracle/decomb_oracle.c:orc_bwdif_plane, which lists where it follows the in-tree
// Metal port platform/macosx/shaders/bwdif_vt.metal and where the C filter differs from that port).
// One thread per sample.  field_end = yadif->current_field == YADIF_FIELD_END: the intra (spatial only)
// filter of the first field of a stream and of the last field of a bob stream.  df = bytes per sample:
// vf_bwdif.c's mirror tests at the top / bottom rows compare against it, (y + df) < h, y > df - 1, ...
// Four adjacent samples per thread: every tap is a whole row above / below the sample in one of the three frames, so a
// thread's eighteen tap rows come in as one dword (two for 16-bit samples) each and the arithmetic is the scalar
// filter's per sample.
template <typename PIX>
__global__ __launch_bounds__(256) void bwdif_kernel(DecombArgs a, int field_end, int clip_max)
{
    typedef Px4<PIX> X4;
    typedef typename X4::T T;
    const DecombPlane &P = a.pl[blockIdx.z];
    const int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= P.w || y >= P.h) return;
    const int st = P.pitch, df = (int)sizeof(PIX), h = P.h;              // st: bytes between rows
    const uint8_t *cur = P.cur + (size_t)y * st;
    uint8_t *orow = P.dst + (size_t)y * P.dst_pitch;
    auto dw = [&](const uint8_t *row) { return *reinterpret_cast<const T *>(row + (size_t)x * sizeof(PIX)); };
    auto store = [&](T v) {
        PIX *o = reinterpret_cast<PIX *>(orow) + x;
        if (x + 3 < P.w) *reinterpret_cast<T *>(o) = v;
        else for (int k = 0; k < 4 && x + k < P.w; k++) o[k] = (PIX)X4::get(v, k);
    };
    if (a.mode == 0 || !((y ^ a.parity) & 1))              // pass-through frame, or a row of the kept field
    {
        store(dw(cur));
        return;
    }
    constexpr int lf0 = 4309, lf1 = 213, hf0 = 5570, hf1 = 3801, hf2 = 1016, sp0 = 5077, sp1 = 981;
    int o4[4];
    if (field_end)
    {
        const int prefs = (y + df) < h ? st : -st, mrefs = y > (df - 1) ? -st : st;
        const int prefs3 = (y + 3 * df) < h ? 3 * st : -st, mrefs3 = y > (3 * df - 1) ? -3 * st : st;
        const T c1 = dw(cur + mrefs), e1 = dw(cur + prefs), c3 = dw(cur + mrefs3), e3 = dw(cur + prefs3);
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const int v = (sp0 * (X4::get(c1, k) + X4::get(e1, k)) - sp1 * (X4::get(c3, k) + X4::get(e3, k))) >> 13;
            o4[k] = min(max(v, 0), clip_max);
        }
        store(X4::pack(o4));
        return;
    }
    const uint8_t *prev = P.prev + (size_t)y * st, *next = P.next + (size_t)y * st;
    const uint8_t *prev2 = a.field_parity ? prev : cur;
    const uint8_t *next2 = a.field_parity ? cur : next;
    const bool edge = (y < 4) || ((y + 5) > h);
    const int prefs = edge ? ((y + df) < h ? st : -st) : st, mrefs = edge ? (y > (df - 1) ? -st : st) : -st;
    const bool spat = edge ? !((y < 2) || ((y + 3) > h)) : true;
    const T wc = dw(cur + mrefs), we = dw(cur + prefs), wp0 = dw(prev2), wn0 = dw(next2);
    const T wpm = dw(prev + mrefs), wpp = dw(prev + prefs), wnm = dw(next + mrefs), wnp = dw(next + prefs);
    T wp2a = X4::zero(), wn2a = X4::zero(), wp2b = X4::zero(), wn2b = X4::zero();
    if (spat) { wp2a = dw(prev2 - 2 * st); wn2a = dw(next2 - 2 * st); wp2b = dw(prev2 + 2 * st); wn2b = dw(next2 + 2 * st); }
    T wc3a = X4::zero(), wc3b = X4::zero(), wp4a = X4::zero(), wn4a = X4::zero(), wp4b = X4::zero(), wn4b = X4::zero();
    if (!edge)
    {
        wc3a = dw(cur - 3 * st); wc3b = dw(cur + 3 * st);
        wp4a = dw(prev2 - 4 * st); wn4a = dw(next2 - 4 * st); wp4b = dw(prev2 + 4 * st); wn4b = dw(next2 + 4 * st);
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        const int c = X4::get(wc, k), e = X4::get(we, k);
        const int p0 = X4::get(wp0, k), n0 = X4::get(wn0, k);
        const int d = (p0 + n0) >> 1;
        const int td0 = abs(p0 - n0);
        const int td1 = (abs(X4::get(wpm, k) - c) + abs(X4::get(wpp, k) - e)) >> 1;
        const int td2 = (abs(X4::get(wnm, k) - c) + abs(X4::get(wnp, k) - e)) >> 1;
        int diff = max(max(td0 >> 1, td1), td2);
        if (!diff) { o4[k] = d; continue; }
        const int s2a = X4::get(wp2a, k) + X4::get(wn2a, k), s2b = X4::get(wp2b, k) + X4::get(wn2b, k);
        if (spat)
        {
            const int b = (s2a >> 1) - c;
            const int f = (s2b >> 1) - e;
            const int dc = d - c, de = d - e;
            const int mx = max(max(de, dc), min(b, f));
            const int mn = min(min(de, dc), max(b, f));
            diff = max(max(diff, mn), -mx);
        }
        int interpol;
        if (edge)
            interpol = (c + e) >> 1;
        else
        {
            const int c3 = X4::get(wc3a, k) + X4::get(wc3b, k);
            if (abs(c - e) > td0)
                interpol = (((hf0 * (p0 + n0) - hf1 * (s2a + s2b)
                              + hf2 * (X4::get(wp4a, k) + X4::get(wn4a, k) + X4::get(wp4b, k) + X4::get(wn4b, k))) >> 2)
                            + lf0 * (c + e) - lf1 * c3) >> 13;
            else
                interpol = (sp0 * (c + e) - sp1 * c3) >> 13;
        }
        if (interpol > d + diff) interpol = d + diff;
        else if (interpol < d - diff) interpol = d - diff;
        o4[k] = min(max(interpol, 0), clip_max);
    }
    store(X4::pack(o4));
}

class DecombFilter : public hbhip_filter
{
public:
    DecombFilter(hbhip_ctx *c, const hbhip_decomb_params &p) : hbhip_filter(c), par(p) {}
    ~DecombFilter() override
    {
        for (DevPicture *p : late_unref) hbhip_pic_release(p, ctx);
        late_unref.clear();
        delete eedi;
    }

    int setup(int width, int height, int depth, int lcw, int lch)
    {
        in_geo.set(width, height, depth, lcw, lch);
        out_geo = in_geo;
        pool.configure(ctx, in_geo);
        if (ff_yadif) return HBHIP_OK;
        // the mode a HEAVY frame is filtered with (decomb_template.c:828-831); reject the
        // combinations for which the reference leaves the rebuilt rows unwritten (:741-789)
        const int hm = par.mode & ~M_SELECTIVE;
        if (!(hm == 0 || hm == M_BLEND || hm == M_CUBIC || (hm & M_YADIF) || (hm & M_EEDI2)))
            return HBHIP_ERR_UNSUPPORTED;
        if (par.mode & M_EEDI2)
        {
            if (par.post_processing < 0 || par.post_processing > 3) return HBHIP_ERR_UNSUPPORTED;
            Eedi2Params ep = { par.magnitude_threshold, par.variance_threshold, par.laplacian_threshold,
                               par.dilation_threshold, par.erosion_threshold, par.noise_threshold,
                               par.maximum_search_distance, par.post_processing };
            // EEDI2 runs depend on each other only through the edge mask (the first of ~15 kernels), and every kernel
            // of a run is too short to fill the GPU.  So the fields of a chain batch (both fields of every frame with
            // bob) are queued and go through each pass together: one launch per pass for up to `fields` fields
            // (EediEngineBase), followed by the blends that consume the guesses.  Driven a frame at a time (work() of the
            // plugin), a frame's fields still share their launches.
            // fields per batch (1..32): two parts of 16 (EediEngineBase::launch).  A field's slot holds 4 half-height and 5
            // full-height scratch frames plus the lattice candidates - about 11 bytes per frame pixel at 8 bits (34 MB at
            // 1080p), 22 at 10 / 12 bits - and the engine keeps fields + 1 slots: 1.1 / 2.3 GB at 1080p for 32 fields, four
            // times that at 2160p.  Up to 1080p a batch is 32 fields; above, 16 (a 2160p field already fills the GPU four
            // times over, the second part adds nothing there).  Should the slab not fit beside what else lives on the GPU
            // the batch is halved until it does.
            int fields = hbhip_dev_int("HBHIP_EEDI2_FIELDS", (long long)in_geo.width * in_geo.height <= 1920LL * 1088 ? 32 : 16);
            for (;;)
            {
                if (in_geo.bps != 1) eedi = new (std::nothrow) Eedi2Engine16(ctx, in_geo, ep, fields);   // 10 / 12-bit samples: eedi2_16.hip
                else                 eedi = new (std::nothrow) Eedi2Engine(ctx, in_geo, ep, fields);
                if (!eedi) return HBHIP_ERR_NOMEM;
                const int rc = eedi->init();
                if (rc == HBHIP_OK) break;
                delete eedi;
                eedi = nullptr;
                if (rc != HBHIP_ERR_HIP || fields <= 2) return rc;
                (void)hipGetLastError();                                  // hipErrorOutOfMemory of the slab: try half
                fields /= 2;
            }
        }
        return HBHIP_OK;
    }

    DevPicture *acquire_input() override
    {
        DevPicture *p = pool.acquire();
        if (p) { p->refs = 0; p->flags = next_flags; p->combed = next_combed; }
        return p;
    }
    int use_frames() override { pool.use_frames(true); frames_mode = true; return HBHIP_OK; }
    void adopt_input(DevPicture *p) override { p->refs = 0; p->flags = next_flags; p->combed = next_combed; }

    int submit(DevPicture *pic) override
    {
        // hb_decomb_work (decomb.c:573-612)
        if (!ready)
        {
            store_ref(pic);
            store_ref(pic);
            ready = true;
            return HBHIP_OK;                   // HB_FILTER_DELAY
        }
        store_ref(pic);
        int rc = process_frame();
        if (rc != HBHIP_OK || !deferred) { const int rc2 = flush_batch(); if (rc == HBHIP_OK) rc = rc2; }
        return rc;
    }

    int flush() override
    {
        int rc = HBHIP_OK;
        if (ref[2] != nullptr && !flushed)
        {
            store_ref(ref[2]);                 // duplicate the last frame (decomb.c:584-589)
            flushed = true;
            if (ff_bwdif) bw_field = BW_BACK_END;          // ff_yadif_request_frame at EOF
            rc = process_frame();
        }
        const int rc2 = flush_batch();
        return rc != HBHIP_OK ? rc : rc2;
    }

    // fused chain: the whole batch is submitted before anything downstream looks at the outputs
    void defer_launches(bool on) override { deferred = on; if (!on) (void)flush_batch(); }
    int  kick() override { return flush_batch(); }

    int pending() override { return (int)outq.size(); }
    DevPicture *pop_output() override
    {
        if (outq.empty()) return nullptr;
        // a frame that is handed out must have been launched: whoever pulls without a kick gets one here
        if (!gathered.empty() || (eedi && eedi->queued() > 0))
            (void)flush_batch();
        DevPicture *p = outq.front();
        outq.pop_front();
        return p;
    }
    void recycle_output(DevPicture *p) override { pool.release(p); }

    int next_flags = 0, next_combed = 0;
    EediEngineBase *engine() { return eedi; }

private:
    void unref(DevPicture *p)
    {
        if (!p || --p->refs != 0) return;
        // a queued EEDI2 field or a blend gathered for the batch launch may still read it: hand it back when the
        // batch is out
        if (!gathered.empty() || (eedi && eedi->queued() > 0))
            late_unref.push_back(p);
        else hbhip_pic_release(p, ctx);                    // possibly another filter's picture (fused chain)
    }
    // launch what has been gathered: the queued EEDI2 fields, then the blends (which read their guesses)
    int flush_batch()
    {
        int rc = eedi ? eedi->launch(ctx) : HBHIP_OK;
        const int rc2 = launch_gathered(ctx);
        for (DevPicture *p : late_unref) hbhip_pic_release(p, ctx);
        late_unref.clear();
        return rc != HBHIP_OK ? rc : rc2;
    }
    void store_ref(DevPicture *p)              // decomb.c:195-200
    {
        unref(ref[0]);
        ref[0] = ref[1];
        ref[1] = ref[2];
        ref[2] = p;
        p->refs++;
    }

    // slot: the EEDI2 slot holding this frame's guess (8-bit EEDI2 modes)
    int launch(DevPicture *dst, int mode, int parity, int tff, int slot = -1)
    {
        DecombArgs a;
        for (int c = 0; c < 3; c++)
        {
            DecombPlane &P = a.pl[c];
            P.prev = ref[0]->plane[c]; P.cur = ref[1]->plane[c]; P.next = ref[2]->plane[c];
            P.guess = nullptr; P.guess_pitch = 0;
            if ((mode & M_EEDI2) && eedi && slot >= 0)
            {
                const EediFrame g = eedi->result(slot);
                P.guess = g.plane[c];
                P.guess_pitch = g.stride[c];
            }
            P.dst = dst->plane[c];
            P.pitch = ref[1]->pitch[c]; P.dst_pitch = dst->pitch[c];
            P.w = in_geo.pw[c]; P.h = in_geo.ph[c];
        }
        a.mode = mode; a.parity = parity; a.field_parity = parity ^ tff;
        dim3 block(64, 4), grid((in_geo.pw[0] + 63) / 64, (in_geo.ph[0] + 3) / 4, 3);
        const int maxv = (1 << in_geo.depth) - 1;
        if (ff_bwdif)
        {
            // yadif->current_field (yadif_common.c, vf_bwdif.c:filter): BACK_END becomes END at the second
            // field; END selects the intra filter and is consumed by the first field that is filtered
            if (bw_second && bw_field == BW_BACK_END) bw_field = BW_END;
            const int field_end = mode != 0 && bw_field == BW_END;
            const dim3 grid4(hbhip_grid_x((in_geo.pw[0] + 255) / 256), (in_geo.ph[0] + 3) / 4, 3);      // four samples per thread
            if (in_geo.bps == 2) HBHIP_LAUNCH(ctx, "bwdif", bwdif_kernel<uint16_t>, grid4, block, 0, a, field_end, maxv);
            else                 HBHIP_LAUNCH(ctx, "bwdif", bwdif_kernel<uint8_t>, grid4, block, 0, a, field_end, maxv);
            if (mode != 0 && bw_field == BW_END) bw_field = BW_NORMAL;
            HBHIP_CHECK(ctx, hipGetLastError());
            return HBHIP_OK;
        }
        geo.ff = ff_yadif ? (ff_nospatial ? 2 : 1) : 0;        // FFmpeg's yadif: the same kernel, its rule for the edge rows
        // Four samples per thread (every depth).  The frames are gathered and go out together (launch_gathered, DB_FRAMES per
        // launch): at the end of the call, or - inside a chain batch - when the batch is complete or the EEDI2 engine
        // is full (a frame whose guess is still queued there cannot be launched before the engine).
        DecombFrame F;
        for (int c = 0; c < 3; c++)
        {
            const DecombPlane &P = a.pl[c];
            F.prev[c] = P.prev; F.cur[c] = P.cur; F.next[c] = P.next; F.guess[c] = P.guess; F.dst[c] = P.dst;
            geo.pitch[c] = P.pitch; geo.dst_pitch[c] = P.dst_pitch; geo.w[c] = P.w; geo.h[c] = P.h;
            if (P.guess) geo.guess_pitch[c] = P.guess_pitch;        // the same for every frame that has a guess
        }
        F.mode = a.mode; F.parity = a.parity; F.field_parity = a.field_parity; F.pad = 0;
        gathered.push_back(F);
        return HBHIP_OK;
    }

    int launch_gathered(hbhip_ctx *lc)
    {
        for (size_t i0 = 0; i0 < gathered.size(); i0 += DB_FRAMES)
        {
            DecombBatch B = geo;
            B.n = (int)std::min<size_t>(DB_FRAMES, gathered.size() - i0);
            for (int k = 0; k < B.n; k++) B.f[k] = gathered[i0 + k];
            const dim3 block(64, 4), grid(hbhip_grid_x((B.w[0] + 255) / 256), (B.h[0] + 3) / 4, 3 * B.n);
            const int maxv = (1 << in_geo.depth) - 1;
            if (ff_yadif)
            {
                if (in_geo.bps == 2) HBHIP_LAUNCH(lc, "yadif", decomb_plane4_kernel<uint16_t>, grid, block, 0, B, maxv);
                else                 HBHIP_LAUNCH(lc, "yadif", decomb_plane4_kernel<uint8_t>, grid, block, 0, B, maxv);
            }
            else
            {
                if (in_geo.bps == 2) HBHIP_LAUNCH(lc, "decomb_plane", decomb_plane4_kernel<uint16_t>, grid, block, 0, B, maxv);
                else                 HBHIP_LAUNCH(lc, "decomb_plane", decomb_plane4_kernel<uint8_t>, grid, block, 0, B, maxv);
            }
        }
        const bool any = !gathered.empty();
        gathered.clear();
        if (any) HBHIP_CHECK(lc, hipGetLastError());
        return HBHIP_OK;
    }

    int process_frame()                        // decomb.c:495-571 + filter_8 mode choice
    {
        DevPicture *cur = ref[1];
        const bool selective = par.mode & M_SELECTIVE;
        if (selective && cur->combed == 0)
        {
            DevPicture *o = pool.acquire();
            if (!o) return HBHIP_ERR_NOMEM;
            o->tag = cur->tag << 1; o->aux = 0;
            bw_second = false;
            int rc = launch(o, 0, 0, 0);       // plain copy of ref[1]
            if (rc != HBHIP_OK) return rc;
            outq.push_back(o);
            return HBHIP_OK;
        }
        int tff;
        if (par.parity < 0)
            // yadif_common.c:return_frame looks at the frame's interlaced flag, which HandBrake sets
            // from s.combed (hbffmpeg.c:107-114); decomb looks at PIC_FLAG_PROGRESSIVE_FRAME (decomb.c:519-528)
            tff = ff_yadif ? (cur->combed ? !!(cur->flags & PIC_TFF) : 1)
                           : (((cur->flags & PIC_PROGRESSIVE) == 0) ? !!(cur->flags & PIC_TFF) : 1);
        else
            tff = (par.parity & 1) ^ 1;

        const int is_combed = selective ? cur->combed : 2;
        int mode = 0;
        if ((par.mode & M_BLEND) && is_combed == 1) mode = M_BLEND;
        else if (is_combed != 0)                    mode = par.mode & ~M_SELECTIVE;

        const int nframes = (par.mode & M_BOB) ? 2 : 1;
        for (int frame = 0; frame < nframes; frame++)
        {
            const int parity = frame ^ tff ^ 1;
            int slot = -1;
            if ((mode & M_EEDI2) && eedi)
            {
                if (eedi->queued() == eedi->capacity())
                {
                    int rc = flush_batch();
                    if (rc != HBHIP_OK) return rc;
                }
                slot = eedi->add_field(cur, !parity);                            // pv->tff = !parity (decomb.c:542)
                if (slot < 0) return HBHIP_ERR_ARG;
            }
            DevPicture *o = pool.acquire();
            if (!o) return HBHIP_ERR_NOMEM;
            o->tag = (cur->tag << 1) | frame; o->aux = frame;
            bw_second = frame == 1;
            int rc = launch(o, mode, parity, tff, slot);
            if (rc != HBHIP_OK) return rc;
            outq.push_back(o);
        }
        return HBHIP_OK;
    }

public:
    bool ff_yadif = false;      // FFmpeg's yadif instead of decomb's line filters (hbhip_yadif_create)
    bool ff_bwdif = false;      // FFmpeg's bwdif (hbhip_bwdif_create); implies ff_yadif's frame / flag handling
    enum { BW_NORMAL = 0, BW_END = 1, BW_BACK_END = 2 };
    int  bw_field = BW_END;     // yadif->current_field: END from the first frame on until a field is filtered
    bool bw_second = false;
    int  ff_nospatial = 0;      // send_frame_nospatial / send_field_nospatial
private:
    hbhip_decomb_params par;
    PicturePool pool;
    DevPicture *ref[3] = {nullptr, nullptr, nullptr};
    std::deque<DevPicture *> outq;
    EediEngineBase *eedi = nullptr;        // EEDI2 (Eedi2Engine at 8 bits, Eedi2Engine16 at 10 / 12): fields queued here run together (flush_batch)
    std::vector<DevPicture *> late_unref;  // input pictures whose last reference went while something gathered could still read them
    bool deferred = false;
    std::vector<DecombFrame> gathered;     // blends waiting for their launch (launch_gathered)
    DecombBatch geo;                       // the pitches / sizes they share
    bool ready = false, flushed = false;
};

} // namespace

extern "C" int hbhip_decomb_create(hbhip_ctx *ctx, const hbhip_decomb_params *p, int width, int height,
                                   int depth, int log2_chroma_w, int log2_chroma_h, hbhip_filter **out)
{
    if (!ctx || !p || !out) return HBHIP_ERR_ARG;
    *out = nullptr;
    if (depth != 8 && depth != 10 && depth != 12) return HBHIP_ERR_UNSUPPORTED;
    if (width < 8 || height < 8) return HBHIP_ERR_UNSUPPORTED;
    (void)hipSetDevice(ctx->device);
    DecombFilter *f = new (std::nothrow) DecombFilter(ctx, *p);
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

extern "C" int hbhip_yadif_create(hbhip_ctx *ctx, int spatial_check, int bob, int selective, int parity,
                                  int width, int height, int depth, int log2_chroma_w, int log2_chroma_h,
                                  hbhip_filter **out)
{
    if (!ctx || !out) return HBHIP_ERR_ARG;
    *out = nullptr;
    if (depth != 8 && depth != 10 && depth != 12) return HBHIP_ERR_UNSUPPORTED;
    if (width < 8 || height < 8) return HBHIP_ERR_UNSUPPORTED;
    (void)hipSetDevice(ctx->device);
    hbhip_decomb_params p;
    memset(&p, 0, sizeof(p));
    // the frame ring, field order, bob pairing and combed-only selection are decomb's (M_YADIF keeps
    // process_frame on its deinterlacing branch; the kernel is vf_yadif's)
    p.mode = M_YADIF | (bob ? M_BOB : 0) | (selective ? M_SELECTIVE : 0);
    p.parity = parity;
    DecombFilter *f = new (std::nothrow) DecombFilter(ctx, p);
    if (!f) return HBHIP_ERR_NOMEM;
    f->ff_yadif = true;
    f->ff_nospatial = !spatial_check;
    int rc = f->setup(width, height, depth, log2_chroma_w, log2_chroma_h);
    if (rc != HBHIP_OK)
    {
        delete f;
        return rc;
    }
    *out = f;
    return HBHIP_OK;
}

extern "C" int hbhip_bwdif_create(hbhip_ctx *ctx, int bob, int selective, int parity,
                                  int width, int height, int depth, int log2_chroma_w, int log2_chroma_h,
                                  hbhip_filter **out)
{
    // same frame ring / parity / selection as the yadif object; only the line filter and its
    // current_field state differ
    int rc = hbhip_yadif_create(ctx, 1, bob, selective, parity, width, height, depth, log2_chroma_w, log2_chroma_h, out);
    if (rc != HBHIP_OK) return rc;
    static_cast<DecombFilter *>(*out)->ff_bwdif = true;
    return HBHIP_OK;
}

extern "C" int hbhip_decomb_push(hbhip_filter *f, const hbhip_host_frame *in, int64_t tag, int pic_flags, int combed)
{
    DecombFilter *d = dynamic_cast<DecombFilter *>(f);
    if (!d) return HBHIP_ERR_ARG;
    d->next_flags = pic_flags;
    d->next_combed = combed;
    return hbhip_filter_push(f, in, tag);
}

extern "C" int hbhip_decomb_push_dev(hbhip_filter *f, const hbhip_dev_frame *in, int64_t tag, int pic_flags, int combed)
{
    DecombFilter *d = dynamic_cast<DecombFilter *>(f);
    if (!d) return HBHIP_ERR_ARG;
    d->next_flags = pic_flags;
    d->next_combed = combed;
    return hbhip_filter_push_dev(f, in, tag);
}

extern "C" int hbhip_decomb_push_frame(hbhip_filter *f, hbhip_frame *fr, int64_t tag, int pic_flags, int combed)
{
    DecombFilter *d = dynamic_cast<DecombFilter *>(f);
    if (!d) return HBHIP_ERR_ARG;
    d->next_flags = pic_flags;
    d->next_combed = combed;
    return hbhip_filter_push_frame(f, fr, tag);
}

// Test hook: download one plane of one EEDI2 scratch frame (0..3 = eedi_half[],
// 4..8 = eedi_full[], decomb.c:64-74) so every pass can be pinned against the oracle.
extern "C" int hbhip_decomb_debug_eedi_plane(hbhip_filter *f, int buffer, int plane, uint8_t *dst, int dst_stride,
                                             int *stride, int *height)
{
    DecombFilter *d = dynamic_cast<DecombFilter *>(f);
    if (!d || !d->engine() || buffer < 0 || buffer > 8 || plane < 0 || plane > 2) return HBHIP_ERR_ARG;
    const int slot = d->engine()->last_slot();                                                // the latest run's scratch
    const EediFrame fr = buffer < 4 ? d->engine()->half(buffer, slot) : d->engine()->full(buffer - 4, slot);
    if (stride) *stride = fr.stride[plane];
    if (height) *height = fr.height[plane];
    if (dst == nullptr) return HBHIP_OK;
    if (dst_stride < fr.stride[plane]) return HBHIP_ERR_ARG;
    hbhip_ctx *ctx = f->ctx;
    HBHIP_CHECK(ctx, hipMemcpy2DAsync(dst, dst_stride, fr.plane[plane], fr.stride[plane], fr.stride[plane],
                                      fr.height[plane], hipMemcpyDeviceToHost, ctx->stream));
    HBHIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return HBHIP_OK;
}
