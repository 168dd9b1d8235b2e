/* libav_stub.h - the few libav* / jansson declarations libhb's public headers mention, as opaque stand-ins.
 * TEST INFRASTRUCTURE: lets tests/test_boundary_cpu.py compile the HIP drop-ins with -DHBHIP_IN_LIBHB against the
 * reference's REAL handbrake/handbrake.h + internal.h (syntax and type checking only, nothing is linked): libav and
 * jansson are not in this image.  Layouts are arbitrary - only names and kinds matter for that check. */
#ifndef LIBAV_STUB_H
#define LIBAV_STUB_H
#include <stdint.h>
#include <stddef.h>
typedef struct AVRational { int num, den; } AVRational;
enum AVPixelFormat { AV_PIX_FMT_NONE = -1, AV_PIX_FMT_YUV420P = 0, AV_PIX_FMT_YUV422P = 4, AV_PIX_FMT_YUV444P = 5,
                     AV_PIX_FMT_YUV420P10 = 62, AV_PIX_FMT_YUV422P10 = 64, AV_PIX_FMT_YUV444P10 = 68, AV_PIX_FMT_YUV420P12 = 123,
                     AV_PIX_FMT_YUV422P12 = 127, AV_PIX_FMT_YUV444P12 = 131, AV_PIX_FMT_NV12 = 23, AV_PIX_FMT_P010 = 158,
                     AV_PIX_FMT_QSV = 114, AV_PIX_FMT_CUDA = 117, AV_PIX_FMT_D3D11 = 172, AV_PIX_FMT_VIDEOTOOLBOX = 160,
                     AV_PIX_FMT_AMF_SURFACE = 250 };
enum AVSampleFormat { AV_SAMPLE_FMT_NONE = -1 };
enum AVFrameSideDataType { AV_FRAME_DATA_PANSCAN };
enum AVPacketSideDataType { AV_PKT_DATA_PALETTE };
enum AVChromaLocation { AVCHROMA_LOC_UNSPECIFIED = 0, AVCHROMA_LOC_LEFT = 1 };
enum AVColorRange { AVCOL_RANGE_UNSPECIFIED = 0, AVCOL_RANGE_MPEG = 1, AVCOL_RANGE_JPEG = 2 };
typedef struct AVComponentDescriptor { int plane, step, offset, shift, depth; } AVComponentDescriptor;
typedef struct AVPixFmtDescriptor { const char *name; uint8_t nb_components, log2_chroma_w, log2_chroma_h; uint64_t flags;
                                    AVComponentDescriptor comp[4]; } AVPixFmtDescriptor;
const AVPixFmtDescriptor *av_pix_fmt_desc_get(enum AVPixelFormat pix_fmt);
int av_image_get_linesize(enum AVPixelFormat pix_fmt, int width, int plane);
const char *av_get_pix_fmt_name(enum AVPixelFormat pix_fmt);
enum AVPixelFormat av_get_pix_fmt(const char *name);
int av_pix_fmt_count_planes(enum AVPixelFormat pix_fmt);
#define FFMIN(a, b) ((a) > (b) ? (b) : (a))      /* libavutil/macros.h */
#define FFMAX(a, b) ((a) > (b) ? (a) : (b))
typedef struct AVFrame AVFrame;
typedef struct AVCodecContext AVCodecContext;
typedef struct AVCodec AVCodec;
typedef struct AVDictionary AVDictionary;
typedef struct AVBufferRef AVBufferRef;
typedef struct AVPacket AVPacket;
typedef struct AVFrameSideData AVFrameSideData;
typedef struct AVChannelLayout { int order, nb_channels; uint64_t mask; void *opaque; } AVChannelLayout;
typedef struct AVStereo3D { int type, flags; } AVStereo3D;
typedef struct AVSphericalMapping { int projection; } AVSphericalMapping;
typedef struct AVMasteringDisplayMetadata { AVRational display_primaries[3][2], white_point[2], min_luminance, max_luminance;
                                            int has_primaries, has_luminance; } AVMasteringDisplayMetadata;
typedef struct AVContentLightMetadata { unsigned MaxCLL, MaxFALL; } AVContentLightMetadata;
typedef struct AVAmbientViewingEnvironment { AVRational ambient_illuminance, ambient_light_x, ambient_light_y; } AVAmbientViewingEnvironment;
typedef struct AVDOVIDecoderConfigurationRecord { uint8_t dv_version_major, dv_version_minor, dv_profile, dv_level, rpu_present_flag,
                                                  el_present_flag, bl_present_flag, dv_bl_signal_compatibility_id; } AVDOVIDecoderConfigurationRecord;
struct SwsContext;
struct SwrContext;
#define AV_NOPTS_VALUE ((int64_t)UINT64_C(0x8000000000000000))
#define AV_NUM_DATA_POINTERS 8
void *av_malloc(size_t size);
void  av_free(void *ptr);
#endif
