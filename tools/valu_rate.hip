// valu_rate.hip — issue rate of the vector-ALU instruction classes the filter kernels are made of,
// measured on the GPU they run on (VERDICT r01 "pin the VALU peak").
//
// For each class a kernel runs `iters` x 64 instructions on 8 independent registers (dependency
// distance 8) from k = 1..8 waves per SIMD; nothing touches memory inside the timed loop.
// Reported per (class, k):
//   cyc_per_inst  = mean per-wave s_memtime delta / (k * instructions per wave)
//                   -> cycles one SIMD needs per wave64 instruction (2.0 = SIMD-32 full rate, 4.0 = half)
//   ginst_s_chip  = wave-instructions / wall time (HIP events), whole chip
// bench.py derives the VALU roofline of the NLMeans kernel from the committed result
// (profiles/r02_valu_rate.json), weighting the classes by the kernel's static instruction mix.
//
// build: hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define R8(OP) OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7")
#define R64(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP)

// 32-bit register classes: "r = r op b"
#define OP_ADD(r)     "v_add_u32 " r ", " r ", %8\n"
#define OP_SUB(r)     "v_sub_u32 " r ", " r ", %8\n"
#define OP_AND(r)     "v_and_b32 " r ", " r ", %8\n"
#define OP_MIN(r)     "v_min_u32 " r ", " r ", %8\n"
#define OP_LSHL(r)    "v_lshlrev_b32 " r ", 1, " r "\n"
#define OP_MAD24(r)   "v_mad_i32_i24 " r ", " r ", %8, " r "\n"
#define OP_MADU24(r)  "v_mad_u32_u24 " r ", " r ", %8, " r "\n"
#define OP_MUL24(r)   "v_mul_u32_u24 " r ", " r ", %8\n"
#define OP_MULLO(r)   "v_mul_lo_u32 " r ", " r ", %8\n"
#define OP_ADD3(r)    "v_add3_u32 " r ", " r ", %8, %8\n"
#define OP_SDWA(r)    "v_sub_u32_sdwa " r ", " r ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n"
#define OP_DPP(r)     "v_add_u32_dpp " r ", " r ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_DPPROW(r)  "v_add_u32_dpp " r ", " r ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_ALIGNB(r)  "v_alignbyte_b32 " r ", " r ", %8, 1\n"
#define OP_SAD(r)     "v_sad_u8 " r ", " r ", %8, " r "\n"
#define OP_DOT4(r)    "v_dot4_u32_u8 " r ", " r ", %8, " r "\n"
#define OP_PERM(r)    "v_perm_b32 " r ", " r ", %8, %8\n"
#define OP_BFE(r)     "v_bfe_u32 " r ", " r ", 8, 8\n"
#define OP_CVTUB(r)   "v_cvt_f32_ubyte0 " r ", " r "\n"
#define OP_CVTFU(r)   "v_cvt_f32_u32 " r ", " r "\n"
#define OP_CVTUF(r)   "v_cvt_u32_f32 " r ", " r "\n"
#define OP_MULF(r)    "v_mul_f32 " r ", " r ", %8\n"
#define OP_ADDF(r)    "v_add_f32 " r ", " r ", %8\n"
#define OP_FMAF(r)    "v_fma_f32 " r ", " r ", %8, " r "\n"
#define OP_RCPF(r)    "v_rcp_f32 " r ", " r "\n"
#define OP_CNDMASK(r) "v_cndmask_b32 " r ", " r ", %8, vcc\n"
#define OP_CMP(r)     "v_cmp_lt_u32 vcc, " r ", %8\n"
#define OP_CNDS(r)    "v_cndmask_b32_e64 " r ", " r ", %8, s[10:11]\n"
#define OP_CMPS(r)    "v_cmp_lt_u32_e64 s[10:11], " r ", %8\n"
#define OP_CMPCND(r)  "v_cmp_lt_u32 vcc, " r ", %8\nv_cndmask_b32 " r ", " r ", %8, vcc\n"
#define OP_MINMAX(r)  "v_min_u32 " r ", " r ", %8\nv_max_u32 " r ", " r ", %8\n"
#define OP_MINF(r)    "v_min_f32 " r ", " r ", %8\n"
#define OP_MAXF(r)    "v_max_f32 " r ", " r ", %8\n"
#define OP_SUBF(r)    "v_sub_f32 " r ", " r ", %8\n"
#define OP_MAC(r)     "v_fmac_f32 " r ", " r ", %8\n"
#define OP_MED3(r)    "v_med3_i32 " r ", " r ", %8, %8\n"
#define OP_MAXU(r)    "v_max_u32 " r ", " r ", %8\n"
#define OP_MINI(r)    "v_min_i32 " r ", " r ", %8\n"
#define OP_OR(r)      "v_or_b32 " r ", " r ", %8\n"
#define OP_XOR(r)     "v_xor_b32 " r ", " r ", %8\n"
#define OP_MOV(r)     "v_mov_b32 " r ", %8\n"
#define OP_MOVDPP(r)  "v_mov_b32_dpp " r ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_LSHR(r)    "v_lshrrev_b32 " r ", 1, " r "\n"
#define OP_ASHR(r)    "v_ashrrev_i32 " r ", 1, " r "\n"
#define OP_ADDCO(r)   "v_add_co_u32 " r ", vcc, " r ", %8\n"
#define OP_SUBREV(r)  "v_subrev_u32 " r ", " r ", %8\n"
#define OP_LSHLADD(r) "v_lshl_add_u32 " r ", " r ", 1, %8\n"
#define OP_ADDLSHL(r) "v_add_lshl_u32 " r ", " r ", %8, 1\n"
#define OP_ANDOR(r)   "v_and_or_b32 " r ", " r ", %8, %8\n"
#define OP_BFI(r)     "v_bfi_b32 " r ", " r ", %8, %8\n"
#define OP_MULI24(r)  "v_mul_i32_i24 " r ", " r ", %8\n"
#define OP_MADI16(r)  "v_mad_i32_i16 " r ", " r ", %8, " r "\n"
#define OP_MADU16(r)  "v_mad_u32_u16 " r ", " r ", %8, " r "\n"
#define OP_DOT2(r)    "v_dot2_u32_u16 " r ", " r ", %8, " r "\n"
#define OP_DOT4I(r)   "v_dot4_i32_i8 " r ", " r ", %8, " r "\n"
#define OP_PKADD16(r) "v_pk_add_u16 " r ", " r ", %8\n"
#define OP_PKSUB16(r) "v_pk_sub_i16 " r ", " r ", %8\n"
#define OP_PKMUL16(r) "v_pk_mul_lo_u16 " r ", " r ", %8\n"
#define OP_PKMAD16(r) "v_pk_mad_u16 " r ", " r ", %8, " r "\n"
#define OP_PKMIN16(r) "v_pk_min_u16 " r ", " r ", %8\n"
#define OP_CVTUB2(r)  "v_cvt_f32_ubyte2 " r ", " r "\n"
#define OP_CVTPKU8(r) "v_cvt_pk_u8_f32 " r ", " r ", 1, %8\n"
#define OP_CVTI32(r)  "v_cvt_i32_f32 " r ", " r "\n"
#define OP_SADU16(r)  "v_sad_u16 " r ", " r ", %8, " r "\n"
#define OP_MSAD(r)    "v_msad_u8 " r ", " r ", %8, " r "\n"
#define OP_SDWAADD(r) "v_add_u32_sdwa " r ", " r ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define OP_SDWAMUL(r) "v_mul_u32_u24_sdwa " r ", " r ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n"
// f16 / mixed-precision classes (round 6: can the NLMeans SSD leave the half-rate integer classes?)
#define OP_PKADDH(r)  "v_pk_add_f16 " r ", " r ", %8\n"
#define OP_PKMULH(r)  "v_pk_mul_f16 " r ", " r ", %8\n"
#define OP_PKFMAH(r)  "v_pk_fma_f16 " r ", " r ", %8, " r "\n"
#define OP_DOT2F(r)   "v_dot2_f32_f16 " r ", " r ", %8, " r "\n"
#define OP_DOT2CF(r)  "v_dot2c_f32_f16 " r ", " r ", %8\n"
#define OP_FMAMIX(r)  "v_fma_mix_f32 " r ", " r ", %8, " r " op_sel_hi:[1,1,0]\n"
#define OP_FMAMIXLO(r) "v_fma_mixlo_f16 " r ", " r ", %8, " r "\n"
#define OP_CVTFH(r)   "v_cvt_f32_f16 " r ", " r "\n"
#define OP_CVTHF(r)   "v_cvt_f16_f32 " r ", " r "\n"
#define OP_CVTPKRTZ(r) "v_cvt_pkrtz_f16_f32 " r ", " r ", %8\n"
#define OP_ADDH(r)    "v_add_f16 " r ", " r ", %8\n"
#define OP_FMACH(r)   "v_mac_f16 " r ", " r ", %8\n"
#define OP_DOT2I16(r) "v_dot2_i32_i16 " r ", " r ", %8, " r "\n"
#define OP_DOT2CI16(r) "v_dot2c_i32_i16 " r ", " r ", %8\n"
#define OP_MULHI(r)   "v_mul_hi_u32 " r ", " r ", %8\n"
#define OP_ADDF_DPP(r) "v_add_f32_dpp " r ", " r ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_FMA_NEG(r) "v_fma_f32 " r ", " r ", -%8, " r "\n"
#define OP_MULF_SDWA(r) "v_mul_f32_sdwa " r ", " r ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"
#define OP_CVTUB_SDWA(r) "v_cvt_f32_u32_sdwa " r ", " r " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
#define OP_FLOORF(r)  "v_floor_f32 " r ", " r "\n"
#define OP_MAD_U64(r) "v_mad_u32_u24 " r ", " r ", %8, %8\n"
#define OP_PKADDI16(r) "v_pk_add_i16 " r ", " r ", %8\n"
#define OP_PKMAXH(r)  "v_pk_max_f16 " r ", " r ", %8\n"
#define OP_SUBF_E64(r) "v_sub_f32_e64 " r ", " r ", |%8|\n"
#define OP_LDEXP(r)   "v_ldexp_f32 " r ", " r ", 2\n"
// 64-bit register classes
#define OP_PKMUL(r)   "v_pk_mul_f32 " r ", " r ", %8\n"
#define OP_PKADD(r)   "v_pk_add_f32 " r ", " r ", %8\n"
#define OP_PKFMA(r)   "v_pk_fma_f32 " r ", " r ", %8, " r "\n"
#define OP_ADDF64(r)  "v_add_f64 " r ", " r ", %8\n"
#define OP_LSHLADD64(r) "v_lshl_add_u64 " r ", " r ", 0, %8\n"

template <typename T>
struct Init;
template <>
struct Init<uint32_t> { static __device__ uint32_t make(uint32_t s) { return s * 2654435761u + 12345u; } };
typedef float float2v __attribute__((ext_vector_type(2)));
template <>
struct Init<float2v> { static __device__ float2v make(uint32_t s) { float2v v; v.x = 1.0f + (s & 7) * 1e-3f; v.y = 1.0f - (s & 3) * 1e-3f; return v; } };
template <>
struct Init<double> { static __device__ double make(uint32_t s) { return 1.0 + (s & 15) * 1e-6; } };
template <>
struct Init<uint64_t> { static __device__ uint64_t make(uint32_t s) { return (uint64_t)s * 0x9E3779B97F4A7C15ull; } };

#define DEFINE_KERNEL(NAME, TYPE, OP)                                                                   \
    __global__ void __launch_bounds__(256) NAME(uint32_t *sink, uint64_t *cycles, int iters)              \
    {                                                                                                       \
        uint32_t s = threadIdx.x + blockIdx.x * 977u;                                                       \
        TYPE a0 = Init<TYPE>::make(s), a1 = Init<TYPE>::make(s + 1), a2 = Init<TYPE>::make(s + 2),      \
             a3 = Init<TYPE>::make(s + 3), a4 = Init<TYPE>::make(s + 4), a5 = Init<TYPE>::make(s + 5),  \
             a6 = Init<TYPE>::make(s + 6), a7 = Init<TYPE>::make(s + 7), b = Init<TYPE>::make(s + 8);   \
        __syncthreads();                                                                                    \
        uint64_t t0 = __builtin_readcyclecounter();                                                         \
        for (int i = 0; i < iters; i++)                                                                     \
            asm volatile(R64(OP)                                                                            \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b)                                                                           \
                         : "vcc", "s10", "s11");                                                                          \
        uint64_t t1 = __builtin_readcyclecounter();                                                         \
        TYPE r = a0;                                                                                        \
        uint32_t acc = 0;                                                                                   \
        const TYPE all[8] = {a0, a1, a2, a3, a4, a5, a6, a7};                                               \
        for (int k = 0; k < 8; k++)                                                                         \
        {                                                                                                   \
            r = all[k];                                                                                     \
            uint32_t w[sizeof(TYPE) / 4];                                                                   \
            __builtin_memcpy(w, &r, sizeof(TYPE));                                                          \
            for (unsigned j = 0; j < sizeof(TYPE) / 4; j++) acc ^= w[j];                                    \
        }                                                                                                   \
        if (acc == 0x12345678u) sink[0] = acc;     /* keeps the chain alive, practically never stores */   \
        if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                \
    }

DEFINE_KERNEL(k_add, uint32_t, OP_ADD)
DEFINE_KERNEL(k_sub, uint32_t, OP_SUB)
DEFINE_KERNEL(k_and, uint32_t, OP_AND)
DEFINE_KERNEL(k_min, uint32_t, OP_MIN)
DEFINE_KERNEL(k_lshl, uint32_t, OP_LSHL)
DEFINE_KERNEL(k_mad24, uint32_t, OP_MAD24)
DEFINE_KERNEL(k_madu24, uint32_t, OP_MADU24)
DEFINE_KERNEL(k_mul24, uint32_t, OP_MUL24)
DEFINE_KERNEL(k_mullo, uint32_t, OP_MULLO)
DEFINE_KERNEL(k_add3, uint32_t, OP_ADD3)
DEFINE_KERNEL(k_sdwa, uint32_t, OP_SDWA)
DEFINE_KERNEL(k_dpp_wave, uint32_t, OP_DPP)
DEFINE_KERNEL(k_dpp_row, uint32_t, OP_DPPROW)
DEFINE_KERNEL(k_alignbyte, uint32_t, OP_ALIGNB)
DEFINE_KERNEL(k_sad_u8, uint32_t, OP_SAD)
DEFINE_KERNEL(k_dot4_u8, uint32_t, OP_DOT4)
DEFINE_KERNEL(k_perm, uint32_t, OP_PERM)
DEFINE_KERNEL(k_bfe, uint32_t, OP_BFE)
DEFINE_KERNEL(k_cvt_f32_ubyte, uint32_t, OP_CVTUB)
DEFINE_KERNEL(k_cvt_f32_u32, uint32_t, OP_CVTFU)
DEFINE_KERNEL(k_cvt_u32_f32, uint32_t, OP_CVTUF)
DEFINE_KERNEL(k_mul_f32, uint32_t, OP_MULF)
DEFINE_KERNEL(k_add_f32, uint32_t, OP_ADDF)
DEFINE_KERNEL(k_fma_f32, uint32_t, OP_FMAF)
DEFINE_KERNEL(k_rcp_f32, uint32_t, OP_RCPF)
DEFINE_KERNEL(k_cndmask, uint32_t, OP_CNDMASK)
DEFINE_KERNEL(k_cmp, uint32_t, OP_CMP)
DEFINE_KERNEL(k_cnd_sgpr, uint32_t, OP_CNDS)
DEFINE_KERNEL(k_cmp_sgpr, uint32_t, OP_CMPS)
DEFINE_KERNEL(k_cmp_cnd_pair, uint32_t, OP_CMPCND)
DEFINE_KERNEL(k_min_max_pair, uint32_t, OP_MINMAX)
DEFINE_KERNEL(k_min_f32, uint32_t, OP_MINF)
DEFINE_KERNEL(k_max_f32, uint32_t, OP_MAXF)
DEFINE_KERNEL(k_sub_f32, uint32_t, OP_SUBF)
DEFINE_KERNEL(k_fmac_f32, uint32_t, OP_MAC)
DEFINE_KERNEL(k_med3, uint32_t, OP_MED3)
DEFINE_KERNEL(k_max_u32, uint32_t, OP_MAXU)
DEFINE_KERNEL(k_min_i32, uint32_t, OP_MINI)
DEFINE_KERNEL(k_or, uint32_t, OP_OR)
DEFINE_KERNEL(k_xor, uint32_t, OP_XOR)
DEFINE_KERNEL(k_mov, uint32_t, OP_MOV)
DEFINE_KERNEL(k_mov_dpp, uint32_t, OP_MOVDPP)
DEFINE_KERNEL(k_lshr, uint32_t, OP_LSHR)
DEFINE_KERNEL(k_ashr, uint32_t, OP_ASHR)
DEFINE_KERNEL(k_add_co, uint32_t, OP_ADDCO)
DEFINE_KERNEL(k_subrev, uint32_t, OP_SUBREV)
DEFINE_KERNEL(k_lshl_add, uint32_t, OP_LSHLADD)
DEFINE_KERNEL(k_add_lshl, uint32_t, OP_ADDLSHL)
DEFINE_KERNEL(k_and_or, uint32_t, OP_ANDOR)
DEFINE_KERNEL(k_bfi, uint32_t, OP_BFI)
DEFINE_KERNEL(k_mul_i24, uint32_t, OP_MULI24)
DEFINE_KERNEL(k_mad_i16, uint32_t, OP_MADI16)
DEFINE_KERNEL(k_mad_u16, uint32_t, OP_MADU16)
DEFINE_KERNEL(k_dot2, uint32_t, OP_DOT2)
DEFINE_KERNEL(k_dot4i, uint32_t, OP_DOT4I)
DEFINE_KERNEL(k_pk_add16, uint32_t, OP_PKADD16)
DEFINE_KERNEL(k_pk_sub16, uint32_t, OP_PKSUB16)
DEFINE_KERNEL(k_pk_mul16, uint32_t, OP_PKMUL16)
DEFINE_KERNEL(k_pk_mad16, uint32_t, OP_PKMAD16)
DEFINE_KERNEL(k_pk_min16, uint32_t, OP_PKMIN16)
DEFINE_KERNEL(k_cvt_ub2, uint32_t, OP_CVTUB2)
DEFINE_KERNEL(k_cvt_pk_u8, uint32_t, OP_CVTPKU8)
DEFINE_KERNEL(k_cvt_i32, uint32_t, OP_CVTI32)
DEFINE_KERNEL(k_sad_u16, uint32_t, OP_SADU16)
DEFINE_KERNEL(k_msad, uint32_t, OP_MSAD)
DEFINE_KERNEL(k_sdwa_add_word, uint32_t, OP_SDWAADD)
DEFINE_KERNEL(k_sdwa_mul_byte, uint32_t, OP_SDWAMUL)
DEFINE_KERNEL(k_pk_add_f16, uint32_t, OP_PKADDH)
DEFINE_KERNEL(k_pk_mul_f16, uint32_t, OP_PKMULH)
DEFINE_KERNEL(k_pk_fma_f16, uint32_t, OP_PKFMAH)
DEFINE_KERNEL(k_dot2_f32_f16, uint32_t, OP_DOT2F)
DEFINE_KERNEL(k_dot2c_f32_f16, uint32_t, OP_DOT2CF)
DEFINE_KERNEL(k_fma_mix, uint32_t, OP_FMAMIX)
DEFINE_KERNEL(k_fma_mixlo, uint32_t, OP_FMAMIXLO)
DEFINE_KERNEL(k_cvt_f32_f16, uint32_t, OP_CVTFH)
DEFINE_KERNEL(k_cvt_f16_f32, uint32_t, OP_CVTHF)
DEFINE_KERNEL(k_cvt_pkrtz, uint32_t, OP_CVTPKRTZ)
DEFINE_KERNEL(k_add_f16, uint32_t, OP_ADDH)
DEFINE_KERNEL(k_fmac_f16, uint32_t, OP_FMACH)
DEFINE_KERNEL(k_dot2_i16, uint32_t, OP_DOT2I16)
DEFINE_KERNEL(k_dot2c_i16, uint32_t, OP_DOT2CI16)
DEFINE_KERNEL(k_mul_hi, uint32_t, OP_MULHI)
DEFINE_KERNEL(k_add_f32_dpp, uint32_t, OP_ADDF_DPP)
DEFINE_KERNEL(k_fma_neg, uint32_t, OP_FMA_NEG)
DEFINE_KERNEL(k_mul_f32_sdwa, uint32_t, OP_MULF_SDWA)
DEFINE_KERNEL(k_cvt_ub_sdwa, uint32_t, OP_CVTUB_SDWA)
DEFINE_KERNEL(k_floor_f32, uint32_t, OP_FLOORF)
DEFINE_KERNEL(k_pk_add_i16, uint32_t, OP_PKADDI16)
DEFINE_KERNEL(k_pk_max_f16, uint32_t, OP_PKMAXH)
DEFINE_KERNEL(k_sub_f32_abs, uint32_t, OP_SUBF_E64)
DEFINE_KERNEL(k_ldexp, uint32_t, OP_LDEXP)
DEFINE_KERNEL(k_pk_mul_f32, float2v, OP_PKMUL)
DEFINE_KERNEL(k_pk_add_f32, float2v, OP_PKADD)
DEFINE_KERNEL(k_pk_fma_f32, float2v, OP_PKFMA)
DEFINE_KERNEL(k_add_f64, double, OP_ADDF64)
DEFINE_KERNEL(k_lshl_add_u64, uint64_t, OP_LSHLADD64)

typedef void (*kern_t)(uint32_t *, uint64_t *, int);
struct Class { const char *name; kern_t fn; };

int main(int argc, char **argv)
{
    const char *out_path = argc > 1 ? argv[1] : "valu_rate.json";
    const Class classes[] = {
        {"v_add_u32", k_add}, {"v_sub_u32", k_sub}, {"v_and_b32", k_and}, {"v_min_u32", k_min},
        {"v_lshlrev_b32", k_lshl}, {"v_mad_i32_i24", k_mad24}, {"v_mad_u32_u24", k_madu24},
        {"v_mul_u32_u24", k_mul24}, {"v_mul_lo_u32", k_mullo}, {"v_add3_u32", k_add3},
        {"v_sub_u32_sdwa", k_sdwa}, {"v_add_u32_dpp wave_shr", k_dpp_wave}, {"v_add_u32_dpp row_shr", k_dpp_row},
        {"v_alignbyte_b32", k_alignbyte}, {"v_sad_u8", k_sad_u8}, {"v_dot4_u32_u8", k_dot4_u8},
        {"v_perm_b32", k_perm}, {"v_bfe_u32", k_bfe},
        {"v_cvt_f32_ubyte0", k_cvt_f32_ubyte}, {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_cvt_u32_f32", k_cvt_u32_f32},
        {"v_mul_f32", k_mul_f32}, {"v_add_f32", k_add_f32}, {"v_fma_f32", k_fma_f32}, {"v_rcp_f32", k_rcp_f32},
        {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_u32", k_cmp},
        {"v_cndmask_b32_e64 sgpr mask", k_cnd_sgpr}, {"v_cmp_lt_u32_e64 sgpr dst", k_cmp_sgpr},
        {"v_cmp+v_cndmask pair (2 insts)", k_cmp_cnd_pair}, {"v_min+v_max pair (2 insts)", k_min_max_pair},
        {"v_min_f32", k_min_f32}, {"v_max_f32", k_max_f32}, {"v_sub_f32", k_sub_f32}, {"v_fmac_f32", k_fmac_f32},
        {"v_med3_i32", k_med3}, {"v_max_u32", k_max_u32}, {"v_min_i32", k_min_i32}, {"v_or_b32", k_or}, {"v_xor_b32", k_xor},
        {"v_mov_b32", k_mov}, {"v_mov_b32_dpp row_shr", k_mov_dpp}, {"v_lshrrev_b32", k_lshr}, {"v_ashrrev_i32", k_ashr},
        {"v_add_co_u32", k_add_co}, {"v_subrev_u32", k_subrev}, {"v_lshl_add_u32", k_lshl_add}, {"v_add_lshl_u32", k_add_lshl},
        {"v_and_or_b32", k_and_or}, {"v_bfi_b32", k_bfi}, {"v_mul_i32_i24", k_mul_i24}, {"v_mad_i32_i16", k_mad_i16},
        {"v_mad_u32_u16", k_mad_u16}, {"v_dot2_u32_u16", k_dot2}, {"v_dot4_i32_i8", k_dot4i},
        {"v_pk_add_u16", k_pk_add16}, {"v_pk_sub_i16", k_pk_sub16}, {"v_pk_mul_lo_u16", k_pk_mul16},
        {"v_pk_mad_u16", k_pk_mad16}, {"v_pk_min_u16", k_pk_min16}, {"v_cvt_f32_ubyte2", k_cvt_ub2},
        {"v_cvt_pk_u8_f32", k_cvt_pk_u8}, {"v_cvt_i32_f32", k_cvt_i32}, {"v_sad_u16", k_sad_u16}, {"v_msad_u8", k_msad},
        {"v_add_u32_sdwa word", k_sdwa_add_word}, {"v_mul_u32_u24_sdwa byte", k_sdwa_mul_byte},
        {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_add_f32", k_pk_add_f32}, {"v_pk_fma_f32", k_pk_fma_f32},
        {"v_add_f64", k_add_f64}, {"v_lshl_add_u64", k_lshl_add_u64},
        {"v_pk_add_f16", k_pk_add_f16}, {"v_pk_mul_f16", k_pk_mul_f16}, {"v_pk_fma_f16", k_pk_fma_f16},
        {"v_dot2_f32_f16", k_dot2_f32_f16}, {"v_dot2c_f32_f16", k_dot2c_f32_f16}, {"v_fma_mix_f32", k_fma_mix},
        {"v_fma_mixlo_f16", k_fma_mixlo}, {"v_cvt_f32_f16", k_cvt_f32_f16}, {"v_cvt_f16_f32", k_cvt_f16_f32},
        {"v_cvt_pkrtz_f16_f32", k_cvt_pkrtz}, {"v_add_f16", k_add_f16}, {"v_fmac_f16", k_fmac_f16},
        {"v_dot2_i32_i16", k_dot2_i16}, {"v_dot2c_i32_i16", k_dot2c_i16}, {"v_mul_hi_u32", k_mul_hi},
        {"v_add_f32_dpp wave_shr", k_add_f32_dpp}, {"v_fma_f32 neg src", k_fma_neg}, {"v_mul_f32_sdwa", k_mul_f32_sdwa},
        {"v_cvt_f32_u32_sdwa byte", k_cvt_ub_sdwa}, {"v_floor_f32", k_floor_f32}, {"v_pk_add_i16", k_pk_add_i16},
        {"v_pk_max_f16", k_pk_max_f16}, {"v_sub_f32_e64 abs", k_sub_f32_abs}, {"v_ldexp_f32", k_ldexp},
    };
    const char *only = argc > 2 ? argv[2] : NULL;          // run only the classes whose name contains this
    const int nclasses = sizeof(classes) / sizeof(classes[0]);
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int iters = 2000;                               // x 64 instructions = 128 000 per wave
    uint32_t *sink;
    uint64_t *cycles;
    CHECK(hipMalloc(&sink, 256));
    CHECK(hipMalloc(&cycles, sizeof(uint64_t) * cus * 8 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<uint64_t> host(cus * 8 * 4);
    std::string json = "{\n  \"device\": \"" + std::string(prop.name) + "\", \"gcn_arch\": \"" + prop.gcnArchName +
                       "\", \"cus\": " + std::to_string(cus) + ", \"clock_mhz_max\": " + std::to_string(prop.clockRate / 1000) +
                       ", \"instructions_per_wave\": " + std::to_string(iters * 64) + ",\n  \"classes\": {\n";
    bool first = true;
    for (int c = 0; c < nclasses; c++)
    {
        if (only != NULL && strstr(classes[c].name, only) == NULL && c < nclasses - 24) continue;   // (the round-6 classes always)
        json += std::string(first ? "" : ",\n") + "    \"" + classes[c].name + "\": {";
        first = false;
        for (int k = 1; k <= 8; k *= 2)
        {
            const int blocks = cus * k;                  // k 256-thread blocks per CU = k waves per SIMD
            for (int rep = 0; rep < 2; rep++)            // first = warm-up
            {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(classes[c].fn, dim3(blocks), dim3(256), 0, 0, sink, cycles, iters);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
            }
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(host.data(), cycles, sizeof(uint64_t) * blocks * 4, hipMemcpyDeviceToHost));
            double sum = 0;
            for (int i = 0; i < blocks * 4; i++) sum += (double)host[i];
            const double mean_cyc = sum / (blocks * 4);
            const double n_wave = (double)iters * 64;
            const double cyc_per_inst = mean_cyc / (k * n_wave);
            const double ginst = (double)blocks * 4 * n_wave / (ms * 1e-3) / 1e9;
            char buf[256];
            snprintf(buf, sizeof(buf), "%s\"k%d\": {\"cyc_per_inst\": %.3f, \"ginst_s_chip\": %.1f, \"wall_ms\": %.3f, \"counter_mhz\": %.0f}",
                     k == 1 ? "" : ", ", k, cyc_per_inst, ginst, ms, mean_cyc / (ms * 1e3));
            json += buf;
            printf("%-26s k=%d  %.3f cyc/inst/SIMD  %.1f Ginst/s chip  (%.3f ms)\n", classes[c].name, k, cyc_per_inst, ginst, ms);
        }
        json += "}";
    }
    json += "\n  }\n}\n";
    FILE *f = fopen(out_path, "w");
    if (f) { fputs(json.c_str(), f); fclose(f); }
    return 0;
}
