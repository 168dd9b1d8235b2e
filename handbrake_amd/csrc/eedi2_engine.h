// eedi2_engine.h — internal interface between the decomb filter (decomb.hip) and
// the EEDI2 pass pipeline (eedi2.hip).  Not part of the ABI.
#pragma once

#include "hbhip_internal.h"

#include <cstdlib>
#include <cstring>
#include <vector>

struct Eedi2Params
{
    int magnitude_threshold, variance_threshold, laplacian_threshold;
    int dilation_threshold, erosion_threshold, noise_threshold;
    int maximum_search_distance, post_processing;
};

// A 3-plane frame laid out byte-for-byte like hb_frame_buffer_init() lays it out
// (libhb/fifo.c:820-881): plane p at base + sum of stride*height of the planes
// before it, stride = width rounded up to 64.  EEDI2 reads a few bytes outside
// rows / planes (eedi2_template.c:395-447, 1194-1195), so the layout — not just
// the pixels — is part of its behaviour; `guard` zero bytes surround the frame.
struct EediFrame
{
    uint8_t *alloc = nullptr;     // owner of the memory (16-bit engine; the 8-bit engine's frames live in its slab)
    uint8_t *base = nullptr;
    uint8_t *plane[3] = {nullptr, nullptr, nullptr};
    int      stride[3] = {0, 0, 0};
    int      width[3] = {0, 0, 0};
    int      height[3] = {0, 0, 0};
    size_t   bytes = 0;
};

// The lower mask tiles of all fields of a batch in ONE launch: a tile of field f waits for the (up to nine) lower tiles
// of field f - 1 whose rows its LDS frame reads, through one flag per tile in device memory.  A flag holds the number
// of the launch that completed the tile (`epoch`, so nothing is cleared between launches).  Workgroups are numbered
// field-major and dispatched in that order, so a waiting workgroup only ever waits for one that is already resident
// or done; the wait is bounded all the same.  When it runs out the tile does NOT abort anything: it raises the launch's
// error word, goes on with the mask as it finds it (atomic loads: a defined value) and publishes itself like every
// other tile, so the launch always ends; a repair pass queued behind the launch - its workgroup 0 - (eedi_chain_repair_tile
// loop: k_mask_chain_repair / q_mask_chain_repair) looks at the word and, only if it is up, recomputes the lower tiles
// of the launch's fields serially in field order - the per-field form of the same arithmetic, no waits - and counts the
// event for the host, which logs it once (MaskChainGuard).  Two workgroups on different XCDs do not share
// an L2: the chain's mask words and flags therefore move as agent-scope relaxed atomics (sc1 loads and write-through
// stores, which are coherent across the XCDs), ordered by "all my stores have completed" (an explicit s_waitcnt
// vmcnt(0) in every wave) before the flag is written.  Agent-scope FENCES do the same job for plain accesses but write back / invalidate the whole
// L2 each time: measured, 6 600 of them per launch made the chain 3.4 ms slower than the per-field launches it replaces.
struct MaskChain
{
    uint32_t *flags;          // [field][tile]
    uint32_t  epoch;
    int tx[3], ty0[3], tyn[3], base[3];   // per plane: tiles per row, first lower tile row, lower tile rows, first tile number
    int ntiles;               // lower tiles of one field, all planes
    int ubase[3], nupper;     // the upper tiles of a field (rows 0 .. ty0 - 1), numbered plane by plane behind the lower ones
    int group;                // workgroups per field of a launch: ntiles, or ntiles + nupper when the upper tiles ride along
    uint32_t *pflags;         // [field][plane]: set to `epoch` by any tile that leaves a mask pixel set - a plane whose flag is
                              // not the epoch afterwards has an empty mask, and every later pass only copies it (8-bit engine)
    uint32_t *has;            // [field][tile of the group]: set to `epoch` by a tile that leaves a mask pixel set; the pass behind the
                              // launch (k_mask_chain_repair) folds them into pflags.  Null: every tile looks at pflags itself
    uint32_t *err;            // raised by a tile whose wait ran out (MaskChainGuard::err)
    uint32_t *fallbacks;      // host-visible count of repaired launches (MaskChainGuard::count_dev)
    int       spin_limit;     // polls before a wait gives up
};

// Per engine: the chain's error word (device), the count of repaired launches (mapped host memory the repair pass bumps
// with a system-scope atomic) and the host's bookkeeping.  poll() is called at every launch(): no synchronisation, it only
// reads the host word.
struct MaskChainGuard
{
    uint32_t *err = nullptr;              // device
    uint32_t *count_host = nullptr;       // hipHostMalloc (mapped)
    uint32_t *count_dev = nullptr;        // its device address
    uint32_t  seen = 0;
    int  init(hbhip_ctx *ctx);
    void destroy();
    void poll(const char *who);
    void bind(MaskChain &C) const;
};
// polls a chain wait makes before it gives up (about a second at the default).  hbhip_debug_mask_chain (ABI, tests) sets it.
int  eedi_chain_spin_limit();
void eedi_chain_note_fallbacks(uint32_t n);

// the lower tiles of one field, numbered plane by plane; a tile row is "upper" (no row of its LDS frame reaches the half
// of the mask that is kept from the previous field) while by * tile_h + tile_h + oy <= height / 2
static inline MaskChain eedi_mask_chain_tiles(const EediFrame &srcp, int tile_w, int tile_h, int oy)
{
    MaskChain C;
    memset(&C, 0, sizeof(C));
    for (int c = 0; c < 3; c++)
    {
        const int tys = (srcp.height[c] + tile_h - 1) / tile_h;
        C.tx[c] = (srcp.width[c] + tile_w - 1) / tile_w;
        while (C.ty0[c] < tys && C.ty0[c] * tile_h + tile_h + oy <= srcp.height[c] / 2) C.ty0[c]++;
        C.tyn[c] = tys - C.ty0[c];
        C.base[c] = C.ntiles;
        C.ntiles += C.tx[c] * C.tyn[c];
        C.ubase[c] = C.nupper;
        C.nupper += C.tx[c] * C.ty0[c];
    }
    C.group = C.ntiles;
    return C;
}

// blockIdx.x = field * C.group + tile -> field, plane, tile column and row; returns whether the tile is a lower one (a
// link of the chain).  With C.group = ntiles + nupper the upper tiles of a field - which no other tile waits for and which
// wait for none - sit behind its lower ones in the dispatch order: they keep the CUs busy while the next field's lower
// tiles wait for this field's.
// Workgroups go to the eight XCDs round robin in dispatch order (workgroup i to XCD i mod 8), each XCD with an L2 of its
// own: numbered row by row, the tiles that share a halo - left / right neighbours, and the rows above / below, 15 tiles
// away in a 1080p luma plane - would sit on different XCDs and fetch each other's halo from HBM (the counters had the pass
// at 1.96 x its algorithmic bytes).  So the position p of a workgroup among the n lower tiles of its field is turned into
// the tile number by XCD: the positions that land on XCD k take a contiguous run of tiles.  A permutation inside a field,
// so a tile still only waits for tiles of the field before it, all dispatched earlier.  MEASURED AND NOT ADOPTED
// (profiles/r5w_mask_chain_xcd_order.json): 43 % fewer bytes read, and 138 -> 155 us per launch - the launch is a chain
// of tile latencies, and a run of neighbouring tiles on one XCD serialises what row-by-row numbering spreads over eight.
#ifndef EEDI_XCD_ORDER
#define EEDI_XCD_ORDER 0
#endif
__device__ __forceinline__ int eedi_xcd_order(int p, int n, int first)       // first = blockIdx.x of the field's position 0
{
#if EEDI_XCD_ORDER
    const int s = first & 7, k = (p + s) & 7;
    int run = 0;                                                               // tiles on the XCDs before k
    for (int q = 0; q < k; q++)
    {
        const int f0 = (q - s) & 7;                                            // first position of the field that lands on XCD q
        run += f0 < n ? (n - f0 + 7) >> 3 : 0;
    }
    return run + ((p - ((k - s) & 7)) >> 3);
#else
    (void)n; (void)first;
    return p;
#endif
}

// (rotating the tile columns of the chroma planes - 8 tiles per row at 1080p, so each column sits on one XCD - the way
// hbhip_grid_x() does for the other passes made this launch 3 % slower, 138 -> 143 us: it is a chain of tile latencies)
__device__ __forceinline__ bool eedi_chain_tile(const MaskChain &C, int &fld, int &pl, int &bx, int &by)
{
    fld = (int)blockIdx.x / C.group;
    int tile = (int)blockIdx.x - fld * C.group;
    if (tile < C.ntiles)
    {
        tile = eedi_xcd_order(tile, C.ntiles, fld * C.group);
        pl = tile >= C.base[2] ? 2 : tile >= C.base[1] ? 1 : 0;
        tile -= C.base[pl];
        const int ry = tile / C.tx[pl];
        bx = tile - ry * C.tx[pl];
        by = C.ty0[pl] + ry;
        return true;
    }
    tile -= C.ntiles;
    pl = tile >= C.ubase[2] ? 2 : tile >= C.ubase[1] ? 1 : 0;
    tile -= C.ubase[pl];
    by = tile / C.tx[pl];
    bx = tile - by * C.tx[pl];
    return false;
}

// threads 0 .. 8 of the workgroup: the flag of the previous field's tile at (bx + t % 3 - 1, by + t / 3 - 1), if there is one
__device__ __forceinline__ const uint32_t *eedi_chain_flag(const MaskChain &C, int fld, int pl, int bx, int by)
{
    const int t = threadIdx.x;
    if (t >= 9) return nullptr;
    const int nx = bx + t % 3 - 1, ny = by + t / 3 - 1;
    if (nx < 0 || nx >= C.tx[pl] || ny < C.ty0[pl] || ny >= C.ty0[pl] + C.tyn[pl]) return nullptr;
    return C.flags + (size_t)(fld - 1) * C.ntiles + C.base[pl] + (ny - C.ty0[pl]) * C.tx[pl] + nx;
}

// wait for those tiles.  `seen`: what a look at the flag returned that the caller issued in front of its own loads (the
// flag's round trip then rides with theirs: most of the time the tile of the field before is long done)
__device__ __forceinline__ void eedi_chain_wait(const MaskChain &C, const uint32_t *flag, uint32_t seen)
{
    if (flag && seen != C.epoch)
    {
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != C.epoch)
        {
            if (++spins > C.spin_limit)
            {
                // the dispatch order this rests on did not hold (or a test says so): no abort - the repair pass
                // behind the launch redoes the lower tiles in order
                __hip_atomic_store(C.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();                                           // (a workgroup fence: the mask loads that follow stay below)
}
__device__ __forceinline__ void eedi_chain_wait(const MaskChain &C, int fld, int pl, int bx, int by)
{
    eedi_chain_wait(C, eedi_chain_flag(C, fld, pl, bx, by), C.epoch + 1u);
}

// after the tile's mask stores (agent-scope atomics): publish the tile.  Returns the workgroup's OR of `pred` (the barrier
// the publication needs anyway carries it)
__device__ __forceinline__ bool eedi_chain_signal(const MaskChain &C, int fld, int pl, int bx, int by, bool pred = false)
{
    // each wave: its stores have completed (a workgroup-scope release fence does not wait for them - waves of a
    // workgroup share their L1 - and without the wait the flag overtakes mask words still in flight: seen as a handful
    // of wrong mask samples in one run out of a few)
#if defined(HBHIP_DEV) && defined(HBHIP_CHAIN_RELEASE)
    // the form VERDICT r4 asked to have measured (never shipped): no explicit wait, the flag as an agent-scope RELEASE
    // store by one thread behind the barrier.  DESIGN 4.11 has the numbers - and why it is not equivalent: the release
    // orders thread 0's wave only; the other waves' mask stores are ordered before it by a workgroup barrier, which on
    // gfx950 does not wait for stores in flight to the L2.
    const bool any = __syncthreads_or(pred);
    if (threadIdx.x == 0)
        __hip_atomic_store(C.flags + (size_t)fld * C.ntiles + C.base[pl] + (by - C.ty0[pl]) * C.tx[pl] + bx, C.epoch,
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return any;
#else
    __builtin_amdgcn_s_waitcnt(0x0f70);                        // vmcnt(0), gfx9 encoding
    const bool any = __syncthreads_or(pred);
    if (threadIdx.x == 0)
        __hip_atomic_store(C.flags + (size_t)fld * C.ntiles + C.base[pl] + (by - C.ty0[pl]) * C.tx[pl] + bx, C.epoch,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return any;
#endif
}

// the tile's end: it left a mask sample set (`has`, workgroup-uniform).  With C.has the tile stores a word of its own and
// is gone (its place among the launch's words: field-major, a field's lower tiles by number, then its upper ones - whatever
// order the workgroups were given the tiles in) - the pass behind the launch folds the words into the plane flags
// (eedi_chain_fold_has); without, one thread looks at the plane flag and raises it if it is not up yet.
// (The look costs the workgroup a round trip at its end: 122 -> 140 us per launch when it came in.  Measured and worse:
// the same load at the tile's start, where it sits in front of the tile's own loads - 350 us; no look at all but a store
// into one of four words per plane - 257 us: thousands of tiles storing to the same few words serialise in the L2.)
__device__ __forceinline__ void eedi_chain_note_has(const MaskChain &C, int fld, int pl, int bx, int by, bool has)
{
    if (!has || threadIdx.x != 0) return;
    if (C.has)
    {
        const int tile = by >= C.ty0[pl] ? C.base[pl] + (by - C.ty0[pl]) * C.tx[pl] + bx : C.ntiles + C.ubase[pl] + by * C.tx[pl] + bx;
        __hip_atomic_store(C.has + (size_t)fld * C.group + tile, C.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    else if (__hip_atomic_load(C.pflags + 3 * fld + pl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != C.epoch)
        __hip_atomic_store(C.pflags + 3 * fld + pl, C.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// workgroup b of the pass behind a chain launch: field b / 3, plane b % 3 - the plane flag out of its tiles' words
__device__ __forceinline__ void eedi_chain_fold_has(const MaskChain &C, int nthreads)
{
    if (!C.has) return;
    const int fld = (int)blockIdx.x / 3, pl = (int)blockIdx.x - 3 * fld;
    const uint32_t *h = C.has + (size_t)fld * C.group;
    const int nl = C.tx[pl] * C.tyn[pl], nu = C.tx[pl] * C.ty0[pl];
    bool any = false;
    for (int i = threadIdx.x; i < nl + nu; i += nthreads)
        any |= h[i < nl ? C.base[pl] + i : C.ntiles + C.ubase[pl] + (i - nl)] == C.epoch;
    if (__syncthreads_or(any) && threadIdx.x == 0) C.pflags[3 * fld + pl] = C.epoch;
}

// the repair pass's loop head: lower tile number `tile` of a field -> plane, column, row (eedi_chain_tile without blockIdx)
__device__ __forceinline__ void eedi_chain_lower_tile(const MaskChain &C, int tile, int &pl, int &bx, int &by)
{
    pl = tile >= C.base[2] ? 2 : tile >= C.base[1] ? 1 : 0;
    tile -= C.base[pl];
    const int ry = tile / C.tx[pl];
    bx = tile - ry * C.tx[pl];
    by = C.ty0[pl] + ry;
}

// HBHIP_EEDI2_FORK=0: the passes of a whole batch on the caller's stream, one launch per pass (profiling runs: a launch then
// covers all fields of the batch, as the counters' bookkeeping and the kernel-timer pass of bench.py assume)
static inline bool eedi_fork_enabled()
{
    static const bool on = [] { const char *e = getenv("HBHIP_EEDI2_FORK"); return !(e && *e == '0'); }();
    return on;
}

// What the two engines (8-bit samples: eedi2.hip, 10 / 12-bit samples: eedi2_16.hip) share - everything on the host side
// that does not name a kernel.  Fields are queued with add_field() and run by launch(): the mask passes field after field
// (the edge mask is the one piece of state a run takes from the one before it: the lower half of MSKPF keeps the previous
// run's mask, eedi2_template.c:132), every pass behind them once for all queued fields.  A field's scratch frames live in
// its slot; result(slot) stays valid until the slot is reused, i.e. for `capacity` more fields.
//  * slots: GUARD, then the nine scratch frames (4 half-height: SRCPF, MSKPF, TMPPF, DSTPF, decomb.c:64-68; 5 full-height:
//    DST2PF, TMP2PF2, MSK2PF, TMP2PF, DST2MPF, :69-74) with a GUARD behind each (zeroed once, never written: the passes'
//    reads outside rows and planes land there), then the lattice candidates; capacity + 1 slots, so that a batch never
//    writes the slot whose mask its first field reads;
//  * launch(): a batch goes out in parts of at most EEDI_PART fields - one mask launch per part (a chain of that many links),
//    the passes behind it, forked over side streams where that is allowed (the comment in EediEngineBase::launch);
//  * the numbering of mask launches (`epoch`: chain flags and plane flags are never cleared between launches).
// A derived engine says how its slots are laid out (EediLayout) and queues its kernels (enqueue_mask / enqueue_passes).
constexpr int EEDI_MAX_BATCH = 32;
constexpr int EEDI_PART = 16;
struct EediLayout
{
    size_t guard;                     // bytes in front of / behind every scratch frame
    size_t cand_elem;                 // bytes per lattice candidate word
    int    tile_w, tile_h, tile_oy;   // the mask kernel's tile (eedi_mask_chain_tiles)
};
class EediEngineBase
{
public:
    virtual ~EediEngineBase();
    virtual int init() = 0;                        // allocate capacity + 1 slots (zeroed once)
    int  capacity() const { return cap_; }
    int  queued() const { return n_; }
    // eedi2_planer (decomb_template.c:455-473): field extraction + the pass sequence of eedi2_interpolate_plane for
    // the 3 planes; `tff` is pv->tff.  Returns the field's slot (< 0: queue full / picture not dword aligned); `cur`
    // must stay alive until launch() has been called.
    int  add_field(const DevPicture *cur, int tff);
    int  launch(hbhip_ctx *lc);                    // run the queued fields on lc's stream
    int  last_slot() const { return last_slot_; }
    EediFrame result(int slot) const { return at_slot(full_[0], slot); }   // eedi_full[DST2PF]
    EediFrame half(int i, int slot) const { return at_slot(half_[i], slot); }
    EediFrame full(int i, int slot) const { return at_slot(full_[i], slot); }

protected:
    EediEngineBase(hbhip_ctx *ctx, const PicGeometry &geo, const Eedi2Params &p, int capacity, const char *who);
    int  init_slots(const EediLayout &L);          // the checks both engines make, slots, flags, side streams, derivative arrays
    size_t place_frame(EediFrame &f, int width, int height, size_t at) const;
    EediFrame at_slot(const EediFrame &f, int slot) const;
    int  next_epoch(hbhip_ctx *lc, uint32_t *epoch);
    virtual bool may_fork() const { return true; } // beyond what launch() itself rules out
    virtual bool mask_ahead() const { return false; }   // the next part's mask on a stream of its own (launch())
    // the five mask passes (+ the field extraction) of fields f0 .. f0 + n - 1 of the batch on st; *epoch: the launch's number
    virtual int enqueue_mask(int f0, int n, hbhip_ctx *lc, hipStream_t st, uint32_t *epoch) = 0;
    // everything after them for fields f0 .. f0 + n - 1 (of one mask launch: `epoch`), on st
    virtual int enqueue_passes(int f0, int n, hbhip_ctx *lc, hipStream_t st, uint32_t epoch) = 0;

    hbhip_ctx  *ctx_;
    PicGeometry geo_;
    Eedi2Params par_;
    const char *who_;
    int         cap_ = 1, n_ = 0, start_ = 0, last_slot_ = 0;
    uint32_t    tffbits_ = 0;
    const uint8_t *src_frame_[EEDI_MAX_BATCH][3];  // the queued fields' frames
    int         src_pitch_[3] = {0, 0, 0};
    uint8_t    *slab_ = nullptr;
    size_t      slot_bytes_ = 0;
    EediFrame   half_[4];    // slot 0's SRCPF, MSKPF, TMPPF, DSTPF          (decomb.c:64-68)
    EediFrame   full_[5];    // slot 0's DST2PF, TMP2PF2, MSK2PF, TMP2PF, DST2MPF (decomb.c:69-74)
    uint8_t    *cand_raw_ = nullptr;    // slot 0's interpolate_lattice candidates (cand_pitch_ words a row, per plane cand_plane_stride_)
    int         cand_pitch_ = 0, cand_plane_stride_ = 0;
    int         chain_ntiles_ = 0, chain_group_ = 0;   // mask chain: lower tiles / all tiles of one field
    uint32_t   *chain_flags_ = nullptr; //             one completion flag per lower tile and field of a batch
    uint32_t   *chain_has_ = nullptr;   //             one word per tile and field: the tile left a mask sample set (MaskChain::has)
    uint32_t    chain_epoch_ = 0;       //             the number of the last mask launch (never 0: 0 is "no launch" in the flag arrays)
    MaskChainGuard guard_;              //             what happens when a wait of the chain runs out
    uint32_t   *plane_flags_ = nullptr; // [field of the batch][plane]: == the mask launch's number when the plane's new mask has a sample set
    static constexpr int MAX_SIDE = 3;
    int         nside_ = 0;
    hipStream_t ahead_ = nullptr;       // mask_ahead(): the next part's mask launch runs here
    hipStream_t side_[MAX_SIDE] = {};   // the later groups of a part's fields run their passes here, beside the first group's
    hipEvent_t  ev_fork_ = nullptr, ev_mask_ = nullptr, ev_join_[MAX_SIDE] = {};
    int        *deriv_[3] = {nullptr, nullptr, nullptr};       // post-processing 2/3: cx2, cy2, cxy (decomb.c:398-403)
    int        *deriv_tmp_[3] = {nullptr, nullptr, nullptr};   //                      tmpc, one per array
};

// EEDI2 on 8-bit samples (eedi2.hip).
class Eedi2Engine : public EediEngineBase
{
public:
    Eedi2Engine(hbhip_ctx *ctx, const PicGeometry &geo, const Eedi2Params &p, int capacity);
    ~Eedi2Engine() override;
    int  init() override;

private:
    bool may_fork() const override;
    int enqueue_mask(int f0, int n, hbhip_ctx *lc, hipStream_t st, uint32_t *epoch) override;
    int enqueue_passes(int f0, int n, hbhip_ctx *lc, hipStream_t st, uint32_t epoch) override;
    uint32_t   *work_list_ = nullptr;   // calc_directions fallback: compacted edge pixels
    int        *work_count_ = nullptr;
};

// EEDI2 on 10 / 12-bit samples (eedi2_16.hip): the same engine on uint16 samples.  EediFrame strides are in BYTES, the
// planes hold uint16 samples in the layout hb_frame_buffer_init gives a 16-bit frame.
class Eedi2Engine16 : public EediEngineBase
{
public:
    Eedi2Engine16(hbhip_ctx *ctx, const PicGeometry &geo, const Eedi2Params &p, int capacity);
    int  init() override;

private:
    bool mask_ahead() const override { return true; }
    int enqueue_mask(int f0, int n, hbhip_ctx *lc, hipStream_t st, uint32_t *epoch) override;
    int enqueue_passes(int f0, int n, hbhip_ctx *lc, hipStream_t st, uint32_t epoch) override;
};
