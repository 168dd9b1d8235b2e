// eedi2_engine.h — internal interface between the decomb filter (decomb.hip) and
// the EEDI2 pass pipeline (eedi2.hip).  Not part of the ABI.
#pragma once

#include "hbhip_internal.h"

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

// EEDI2 on 8-bit samples (eedi2.hip).  Fields are queued with add_field() and run by launch(): the mask passes field
// after field (the edge mask is the one piece of state a run takes from the one before it: the lower half of MSKPF
// keeps the previous run's mask, eedi2_template.c:132), every pass behind them once for all queued fields.  A field's
// scratch frames live in its slot; result(slot) stays valid until the slot is reused, i.e. for `capacity` more fields.
constexpr int EEDI_MAX_BATCH = 32;
class Eedi2Engine
{
public:
    Eedi2Engine(hbhip_ctx *ctx, const PicGeometry &geo, const Eedi2Params &p, int capacity);
    ~Eedi2Engine();
    int  init();                                   // allocate capacity + 1 slots (zeroed once)
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

private:
    size_t place_frame(EediFrame &f, int width, int height, size_t at);
    EediFrame at_slot(const EediFrame &f, int slot) const;
    int enqueue_mask(int n, hbhip_ctx *lc);                 // the five mask passes (+ the field extraction)
    int enqueue_passes(int f0, int n, hbhip_ctx *lc);       // everything after them, fields f0 .. f0 + n - 1 of the batch
    hbhip_ctx  *ctx_;
    PicGeometry geo_;
    Eedi2Params par_;
    int         cap_ = 1, n_ = 0, start_ = 0, last_slot_ = 0;
    uint32_t    tffbits_ = 0;
    const uint8_t *src_frame_[EEDI_MAX_BATCH][3];  // the queued fields' frames
    int         src_pitch_[3] = {0, 0, 0};
    uint8_t    *slab_ = nullptr;
    size_t      slot_bytes_ = 0;
    EediFrame   half_[4];    // slot 0's SRCPF, MSKPF, TMPPF, DSTPF          (decomb.c:64-68)
    EediFrame   full_[5];    // slot 0's DST2PF, TMP2PF2, MSK2PF, TMP2PF, DST2MPF (decomb.c:69-74)
    uint32_t   *chain_flags_ = nullptr; // mask chain: one completion flag per lower tile and field of a batch
    uint32_t    chain_epoch_ = 0;       //             the number of the last chain launch
    uint32_t   *work_list_ = nullptr;   // calc_directions fallback: compacted edge pixels
    int        *work_count_ = nullptr;
    uint32_t   *cand_ = nullptr;        // slot 0's interpolate_lattice candidates
    int         cand_pitch_ = 0, cand_plane_stride_ = 0;
    int        *deriv_[3] = {nullptr, nullptr, nullptr};       // post-processing 2/3: cx2, cy2, cxy (decomb.c:398-403)
    int        *deriv_tmp_[3] = {nullptr, nullptr, nullptr};   //                      tmpc, one per array
};

// EEDI2 on 10 / 12-bit samples (eedi2_16.hip): the same engine (slots, add_field / launch) on uint16 samples, one
// thread per sample and pass; the five mask passes are separate launches per field.  EediFrame strides are in BYTES,
// the planes hold uint16 samples in the layout hb_frame_buffer_init gives a 16-bit frame.
class Eedi2Engine16
{
public:
    Eedi2Engine16(hbhip_ctx *ctx, const PicGeometry &geo, const Eedi2Params &p, int capacity);
    ~Eedi2Engine16();
    int  init();
    int  capacity() const { return cap_; }
    int  queued() const { return n_; }
    int  add_field(const DevPicture *cur, int tff);   // returns the field's slot
    int  launch(hbhip_ctx *lc);
    int  last_slot() const { return last_slot_; }
    EediFrame result(int slot) const { return at_slot(full_[0], slot); }
    EediFrame half(int i, int slot) const { return at_slot(half_[i], slot); }
    EediFrame full(int i, int slot) const { return at_slot(full_[i], slot); }

private:
    size_t place_frame(EediFrame &f, int width, int height, size_t at);
    EediFrame at_slot(const EediFrame &f, int slot) const;
    int enqueue(int n, hbhip_ctx *lc);
    hbhip_ctx  *ctx_;
    PicGeometry geo_;
    Eedi2Params par_;
    int         cap_ = 1, n_ = 0, start_ = 0, last_slot_ = 0;
    uint32_t    tffbits_ = 0;
    const uint8_t *src_frame_[EEDI_MAX_BATCH][3];
    int         src_pitch_[3] = {0, 0, 0};
    uint8_t    *slab_ = nullptr;
    size_t      slot_bytes_ = 0;
    EediFrame   half_[4];    // slot 0's SRCPF, MSKPF, TMPPF, DSTPF
    EediFrame   full_[5];    // slot 0's DST2PF, TMP2PF2, MSK2PF, TMP2PF, DST2MPF
    unsigned long long *cand_ = nullptr;   // slot 0's interpolate_lattice candidates
    int         cand_pitch_ = 0, cand_plane_stride_ = 0;
    int        *deriv_[3] = {nullptr, nullptr, nullptr};
    int        *deriv_tmp_[3] = {nullptr, nullptr, nullptr};
};
