// eedi2.hip — EEDI2 pass pipeline (placeholder until the passes land).
#include "eedi2_engine.h"

Eedi2Engine::Eedi2Engine(hbhip_ctx *ctx, const PicGeometry &geo, const Eedi2Params &p)
    : ctx_(ctx), geo_(geo), par_(p) {}
Eedi2Engine::~Eedi2Engine() {}
int Eedi2Engine::init() { return HBHIP_ERR_UNSUPPORTED; }
int Eedi2Engine::run(const DevPicture *, int) { return HBHIP_ERR_UNSUPPORTED; }
int Eedi2Engine::alloc_frame(EediFrame &, int, int) { return HBHIP_ERR_UNSUPPORTED; }
