// eedi2_engine.h — internal interface between the decomb filter (decomb.hip) and
// the EEDI2 pass pipeline (eedi2.hip).  Not part of the ABI.
#pragma once

#include "hbhip_internal.h"

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
    uint8_t *alloc = nullptr;
    uint8_t *base = nullptr;
    uint8_t *plane[3] = {nullptr, nullptr, nullptr};
    int      stride[3] = {0, 0, 0};
    int      width[3] = {0, 0, 0};
    int      height[3] = {0, 0, 0};
    size_t   bytes = 0;
};

// The edge mask is the one piece of EEDI2 state that runs depend on (the lower half of MSKPF keeps
// the previous run's mask, eedi2_template.c:132).  When consecutive runs go to different engines
// (different HIP streams), they all work on one set of mask buffers, in run order: a lone engine
// alternates between two, an engine of a ring always writes its own (mask[i] = its MSKPF frame) and
// reads the one the previous run wrote.
constexpr int EEDI_MAX_RING = 8;
struct EediMaskShare
{
    EediFrame  mask[EEDI_MAX_RING];
    int        sel = 0;               // which one holds the current mask
    hipEvent_t ev_mask = nullptr;     // recorded behind every mask kernel: the next run's mask kernel waits for it
    bool       ev_valid = false;
};

class Eedi2Engine
{
public:
    // main: the filter's context when this engine executes on a context (stream) of its own;
    // share + ring_index: the ring's mask state and this engine's place in it (see EediMaskShare)
    Eedi2Engine(hbhip_ctx *ctx, const PicGeometry &geo, const Eedi2Params &p,
                hbhip_ctx *main = nullptr, EediMaskShare *share = nullptr, int ring_index = -1);
    ~Eedi2Engine();
    int  init();                                   // allocate the 9 scratch frames (zeroed once)
    // eedi2_planer (decomb_template.c:455-473): field extraction + the pass
    // sequence of eedi2_interpolate_plane for the 3 planes; `tff` is pv->tff.
    // wait_for: an event of the main stream behind which `cur` is complete and this engine's previous
    // result has been consumed (side engines only; the main-stream engine is ordered by its stream)
    int  run(const DevPicture *cur, int tff, hipEvent_t wait_for = nullptr);
    int  mark_done();                              // side engine: whatever was launched on its stream so far belongs to the run
    int  join();                                   // side engine: make the main stream wait for the last run
    hbhip_ctx *stream_ctx() { return ctx_; }       // the context (stream) the engine launches on
    EediMaskShare *share() { return share_; }
    const EediFrame &result() const { return full_[0]; }   // eedi_full[DST2PF]
    // MSKPF alternates between two buffers (the fused mask kernel reads the previous field's mask
    // while it writes the new one): index 1 always names the current one
    const EediFrame &half(int i) const { return i == 1 ? share_->mask[share_->sel] : half_[i]; }
    const EediFrame &full(int i) const { return full_[i]; }

private:
    int alloc_frame(EediFrame &f, int width, int height);
    // the five mask passes (+ the field extraction when `frame` is given); sel = mask buffer to write, old = the previous run's
    int enqueue_mask(int sel, int old, const DevPicture *frame, int start_line);
    int enqueue_passes(int tff, int sel, hbhip_ctx *lc);   // everything after them, launched on lc's stream
    hipGraphExec_t graph_[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // captured pass sequence per (field parity, mask buffer)
    EediMaskShare  own_share_;                     // the two MSKPF buffers (mask[0] is half_[1]) when not shared
    EediMaskShare *share_ = nullptr;
    int         ring_index_ = -1;                  // >= 0: engine of a ring, writes share_->mask[ring_index_]
    hbhip_ctx  *main_ = nullptr;                   // != ctx_ for a side engine
    hbhip_ctx  *cap_ctx_ = nullptr;                // private stream the pass sequence is captured on
    hipEvent_t  ev_done_ = nullptr;
    bool        use_graph_ = true;
    hbhip_ctx  *ctx_;
    PicGeometry geo_;
    Eedi2Params par_;
    EediFrame   half_[4];    // SRCPF, MSKPF, TMPPF, DSTPF          (decomb.c:64-68)
    EediFrame   full_[5];    // DST2PF, TMP2PF2, MSK2PF, TMP2PF, DST2MPF (decomb.c:69-74)
    uint32_t   *work_list_ = nullptr;   // calc_directions: compacted edge pixels
    int        *work_count_ = nullptr;
    uint32_t   *cand_ = nullptr;        // interpolate_lattice: per-pixel candidate outcomes
    int         cand_pitch_ = 0, cand_plane_stride_ = 0;
    int        *deriv_[3] = {nullptr, nullptr, nullptr};       // post-processing 2/3: cx2, cy2, cxy (decomb.c:398-403)
    int        *deriv_tmp_[3] = {nullptr, nullptr, nullptr};   //                      tmpc, one per array
};

// EEDI2 on 10 / 12-bit samples (eedi2_16.hip): same role as Eedi2Engine, first correct form (one thread
// per sample and pass, serial lattice rows, no graphs / pairing).  EediFrame strides are in BYTES, the
// planes hold uint16 samples in the layout hb_frame_buffer_init gives a 16-bit frame.
class Eedi2Engine16
{
public:
    Eedi2Engine16(hbhip_ctx *ctx, const PicGeometry &geo, const Eedi2Params &p);
    ~Eedi2Engine16();
    int  init();
    int  run(const DevPicture *cur, int tff);
    const EediFrame &result() const { return full_[0]; }
    const EediFrame &half(int i) const { return half_[i]; }
    const EediFrame &full(int i) const { return full_[i]; }

private:
    int alloc_frame(EediFrame &f, int width, int height);
    // the field extraction (do_fill) and / or the pass sequence behind it (do_rest), launched on lc's stream
    int enqueue(const DevPicture *cur, int tff, hbhip_ctx *lc, bool do_fill, bool do_rest);
    hipGraphExec_t graph_[2] = {nullptr, nullptr};   // captured pass sequence per field parity
    hbhip_ctx  *cap_ctx_ = nullptr;                  // private stream the pass sequence is captured on
    bool        use_graph_ = true;
    hbhip_ctx  *ctx_;
    PicGeometry geo_;
    Eedi2Params par_;
    EediFrame   half_[4];    // SRCPF, MSKPF, TMPPF, DSTPF
    EediFrame   full_[5];    // DST2PF, TMP2PF2, MSK2PF, TMP2PF, DST2MPF
    unsigned long long *cand_ = nullptr;   // interpolate_lattice: per-pixel candidate outcomes
    int         cand_pitch_ = 0, cand_plane_stride_ = 0;
    int        *deriv_[3] = {nullptr, nullptr, nullptr};
    int        *deriv_tmp_[3] = {nullptr, nullptr, nullptr};
};
