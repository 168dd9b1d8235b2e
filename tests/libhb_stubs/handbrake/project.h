/* stand-in for the generated handbrake/project.h (make/configure.py writes it): just enough for a syntax check */
#ifndef HB_PROJECT_STUB_H
#define HB_PROJECT_STUB_H
#define HB_PROJECT_TITLE "HandBrake"
#define HB_PROJECT_NAME "HandBrake"
#define HB_PROJECT_VERSION "0.0.0"
#define HB_PROJECT_BUILD 0
#define HB_PROJECT_FEATURE_QSV 0
#define HB_PROJECT_FEATURE_VCE 0
#define HB_PROJECT_FEATURE_NVENC 0
#define HB_PROJECT_FEATURE_NVDEC 0
#define HB_PROJECT_FEATURE_MF 0
#define HB_PROJECT_FEATURE_X265 0
#define HB_PROJECT_FEATURE_FDK_AAC 0
#define HB_PROJECT_FEATURE_FFMPEG_AAC 0
#define HB_PROJECT_FEATURE_GTK 0
#define HB_PROJECT_FEATURE_LIBDOVI 0
#define HB_PROJECT_FEATURE_NUMA 0
#endif
