// salu_rate.hip — issue rate of the SCALAR instructions the short EEDI2 passes are full of, measured on the GPU they run
// on.  Several of those passes execute as many scalar as vector instructions (profiles/r3d_chain_pmc_summary.json:
// k_filter_map 53 M SALU against 29 M VALU per launch, k_fill_gaps_b 51 M / 27 M, k_lattice_cand_q 84 M / 83 M): a CU has
// ONE scalar unit behind its four SIMDs (MI355X_MICROARCH.md), so for them the scalar issue rate is a ceiling of its own,
// next to the vector one of tools/valu_rate.hip.
//
// Same method as valu_rate.hip: per class a kernel runs `iters` x 64 instructions on 8 independent SGPRs from k = 1..8
// waves per SIMD; nothing touches memory in the timed loop.  Reported per (class, k):
//   cyc_per_inst_cu = mean per-wave cycle-counter delta / (4 k x instructions per wave)  -> cycles ONE CU needs per
//                     scalar instruction (4 k waves share its scalar unit)
//   ginst_s_chip    = wave-instructions / wall time (HIP events), whole chip
// and the same for mixed streams (a vector and a scalar instruction alternating; a taken / not-taken branch per
// instruction), which say whether the two pipes really issue side by side.
//
// build: hipcc --offload-arch=gfx950 -O2 tools/salu_rate.hip -o tools/salu_rate ; run: tools/salu_rate out.json
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define R8(OP) OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7")
#define R64(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP)

// scalar classes: r = r op %8 (SGPRs)
#define S_ADD(r)     "s_add_u32 " r ", " r ", %8\n"
#define S_AND(r)     "s_and_b32 " r ", " r ", %8\n"
#define S_LSHL(r)    "s_lshl_b32 " r ", " r ", 1\n"
#define S_MUL(r)     "s_mul_i32 " r ", " r ", %8\n"
#define S_MULHI(r)   "s_mul_hi_u32 " r ", " r ", %8\n"
#define S_BCNT(r)    "s_bcnt1_i32_b32 " r ", " r "\n"
#define S_FF1(r)     "s_ff1_i32_b32 " r ", " r "\n"
#define S_CMP(r)     "s_cmp_lt_u32 " r ", %8\n"
#define S_CSEL(r)    "s_cselect_b32 " r ", " r ", %8\n"
#define S_CMPSEL(r)  "s_cmp_lt_u32 " r ", %8\ns_cselect_b32 " r ", " r ", %8\n"
#define S_MOV(r)     "s_mov_b32 " r ", %8\n"
#define S_AND64(r)   "s_and_b64 s[10:11], s[10:11], exec\n"                    /* 64-bit mask arithmetic as around branches */
#define S_SAVEEXEC(r) "s_and_saveexec_b64 s[10:11], exec\n"                     /* exec &= exec: unchanged, the cost is the point */
#define S_NOP(r)     "s_nop 0\n"
#define S_BR_NT(r)   "s_cmp_eq_u32 " r ", " r "\ns_cbranch_scc0 1f\n1:\n"      /* compare + branch NOT taken */
#define S_BR_T(r)    "s_branch 1f\n1:\n"                                         /* branch taken (to the next instruction) */
#define S_BR_EXECZ(r) "s_cbranch_execz 1f\n1:\n"                                 /* the skip branch behind every saveexec: not taken */
// mixed streams
// (operands of the mixed kernels: %0-%7 the scalar chains, %8 / %9 vector in-out, %10 scalar in, %11 vector in)
#define M_VS(r)      "v_add_u32 %8, %8, %11\ns_add_u32 " r ", " r ", %10\n"    /* vector + scalar alternating */
#define M_VVS(r)     "v_add_u32 %8, %8, %11\nv_and_b32 %9, %9, %11\ns_add_u32 " r ", " r ", %10\n"
#define M_VCMPSAVE(r) "v_cmp_lt_u32 vcc, %8, %11\ns_and_b64 s[10:11], vcc, exec\n"   /* what a divergent `if` costs before its body */

#define DEFINE_KERNEL(NAME, OP)                                                                                       \
    __global__ void __launch_bounds__(256) NAME(uint32_t *sink, uint64_t *cycles, int iters)                            \
    {                                                                                                                     \
        uint32_t s = __builtin_amdgcn_readfirstlane(blockIdx.x * 977u + 12345u);                                        \
        uint32_t a0 = s, a1 = s + 1, a2 = s + 2, a3 = s + 3, a4 = s + 4, a5 = s + 5, a6 = s + 6, a7 = s + 7, b = s | 3; \
        __syncthreads();                                                                                                  \
        uint64_t t0 = __builtin_readcyclecounter();                                                                       \
        for (int i = 0; i < iters; i++)                                                                                   \
            asm volatile(R64(OP)                                                                                          \
                         : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3), "+s"(a4), "+s"(a5), "+s"(a6), "+s"(a7)               \
                         : "s"(b)                                                                                         \
                         : "scc", "vcc", "s10", "s11");                                                                   \
        uint64_t t1 = __builtin_readcyclecounter();                                                                       \
        const uint32_t acc = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                            \
        if (acc == 0x12345678u) sink[0] = acc;                                                                            \
        if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                              \
    }

// the mixed streams write their vector operands (%8, %9): in / out operands of their own, never the thread index
#define DEFINE_MIXED(NAME, OP)                                                                                        \
    __global__ void __launch_bounds__(256) NAME(uint32_t *sink, uint64_t *cycles, int iters)                            \
    {                                                                                                                     \
        uint32_t s = __builtin_amdgcn_readfirstlane(blockIdx.x * 977u + 12345u);                                        \
        uint32_t a0 = s, a1 = s + 1, a2 = s + 2, a3 = s + 3, a4 = s + 4, a5 = s + 5, a6 = s + 6, a7 = s + 7, b = s | 3; \
        uint32_t w0 = threadIdx.x * 5 + 1, w1 = threadIdx.x * 3 + 1, w2 = threadIdx.x + 7;                              \
        __syncthreads();                                                                                                  \
        uint64_t t0 = __builtin_readcyclecounter();                                                                       \
        for (int i = 0; i < iters; i++)                                                                                   \
            asm volatile(R64(OP)                                                                                          \
                         : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3), "+s"(a4), "+s"(a5), "+s"(a6), "+s"(a7),              \
                           "+v"(w0), "+v"(w2)                                                                             \
                         : "s"(b), "v"(w1)                                                                                \
                         : "scc", "vcc", "s10", "s11");                                                                   \
        uint64_t t1 = __builtin_readcyclecounter();                                                                       \
        const uint32_t acc = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ w0 ^ w2;                                            \
        if (acc == 0x12345678u) sink[0] = acc;                                                                            \
        if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                              \
    }

DEFINE_KERNEL(k_s_add, S_ADD)
DEFINE_KERNEL(k_s_and, S_AND)
DEFINE_KERNEL(k_s_lshl, S_LSHL)
DEFINE_KERNEL(k_s_mul, S_MUL)
DEFINE_KERNEL(k_s_mulhi, S_MULHI)
DEFINE_KERNEL(k_s_bcnt, S_BCNT)
DEFINE_KERNEL(k_s_ff1, S_FF1)
DEFINE_KERNEL(k_s_cmp, S_CMP)
DEFINE_KERNEL(k_s_csel, S_CSEL)
DEFINE_KERNEL(k_s_cmpsel, S_CMPSEL)
DEFINE_KERNEL(k_s_mov, S_MOV)
DEFINE_KERNEL(k_s_and64, S_AND64)
DEFINE_KERNEL(k_s_saveexec, S_SAVEEXEC)
DEFINE_KERNEL(k_s_nop, S_NOP)
DEFINE_KERNEL(k_s_br_nt, S_BR_NT)
DEFINE_KERNEL(k_s_br_t, S_BR_T)
DEFINE_KERNEL(k_s_br_execz, S_BR_EXECZ)
DEFINE_MIXED(k_m_vs, M_VS)
DEFINE_MIXED(k_m_vvs, M_VVS)
DEFINE_MIXED(k_m_vcmpsave, M_VCMPSAVE)

typedef void (*kern_t)(uint32_t *, uint64_t *, int);
struct Class { const char *name; kern_t fn; int per_slot; };      // per_slot: instructions one OP expands to

int main(int argc, char **argv)
{
    const char *out_path = argc > 1 ? argv[1] : "salu_rate.json";
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const Class classes[] = {
        {"s_add_u32", k_s_add, 1}, {"s_and_b32", k_s_and, 1}, {"s_lshl_b32", k_s_lshl, 1}, {"s_mul_i32", k_s_mul, 1},
        {"s_mul_hi_u32", k_s_mulhi, 1}, {"s_bcnt1_i32_b32", k_s_bcnt, 1}, {"s_ff1_i32_b32", k_s_ff1, 1},
        {"s_cmp_lt_u32", k_s_cmp, 1}, {"s_cselect_b32", k_s_csel, 1}, {"s_cmp+s_cselect pair (2 insts)", k_s_cmpsel, 2},
        {"s_mov_b32", k_s_mov, 1}, {"s_and_b64 with exec", k_s_and64, 1}, {"s_and_saveexec_b64", k_s_saveexec, 1},
        {"s_nop 0", k_s_nop, 1}, {"s_cmp+s_cbranch not taken (2 insts)", k_s_br_nt, 2}, {"s_branch taken", k_s_br_t, 1},
        {"s_cbranch_execz not taken", k_s_br_execz, 1},
        {"v_add_u32 + s_add_u32 alternating (2 insts)", k_m_vs, 2}, {"2 VALU + 1 SALU (3 insts)", k_m_vvs, 3},
        {"v_cmp + s_and_b64 (2 insts)", k_m_vcmpsave, 2},
    };
    const int nclasses = sizeof(classes) / sizeof(classes[0]);
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int iters = 1000;
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
                       ",\n  \"classes\": {\n";
    for (int c = 0; c < nclasses; c++)
    {
        json += std::string("    \"") + classes[c].name + "\": {";
        for (int k = 1; k <= 8; k *= 2)
        {
            const int blocks = cus * k;                  // k 256-thread blocks per CU = k waves per SIMD, 4 k per CU
            for (int rep = 0; rep < 2; rep++)
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
            const double n_wave = (double)iters * 64 * classes[c].per_slot;
            const double cyc_per_inst_cu = mean_cyc / (4.0 * k * n_wave);
            const double ginst = (double)blocks * 4 * n_wave / (ms * 1e-3) / 1e9;
            char buf[256];
            snprintf(buf, sizeof(buf), "%s\"k%d\": {\"cyc_per_inst_cu\": %.3f, \"ginst_s_chip\": %.1f, \"wall_ms\": %.3f, \"counter_mhz\": %.0f}",
                     k == 1 ? "" : ", ", k, cyc_per_inst_cu, ginst, ms, mean_cyc / (ms * 1e3));
            json += buf;
            printf("%-46s k=%d  %.3f cyc/inst/CU  %.1f Ginst/s chip  (%.3f ms)\n", classes[c].name, k, cyc_per_inst_cu, ginst, ms);
        }
        json += c + 1 < nclasses ? "},\n" : "}\n";
    }
    json += "  }\n}\n";
    FILE *f = fopen(out_path, "w");
    if (f) { fputs(json.c_str(), f); fclose(f); }
    return 0;
}
