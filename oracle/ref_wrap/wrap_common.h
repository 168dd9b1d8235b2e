/* Pre-define the reference's include guards so its own umbrella headers are
 * skipped and our libhb-compatible subset (include/hbhip_libhb.h) is used in
 * their place.  The reference .c files are compiled where they lie under
 * $(REF)/libhb (found through -I) and are never copied into this repo. */
#ifndef HBREF_WRAP_COMMON_H
#define HBREF_WRAP_COMMON_H
#include "hbhip_libhb.h"
#define HANDBRAKE_HANDBRAKE_H
#define HANDBRAKE_FFMPEG_H
#define HANDBRAKE_PORTS_H
#define HANDBRAKE_COMMON_H
#define HANDBRAKE_INTERNAL_H
#define HBREF_EXPORT __attribute__((visibility("default")))
#endif
