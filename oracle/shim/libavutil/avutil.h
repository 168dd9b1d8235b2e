/* stub: everything vfr.c needs of libavutil is in include/hbhip_libhb.h */
