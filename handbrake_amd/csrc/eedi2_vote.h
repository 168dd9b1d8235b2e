// eedi2_vote.h - the rounded average of the EEDI2 direction votes, shared by eedi2.hip and tools/vote_avg_check.hip (which runs
// every (a, b) the kernels can produce through it on the GPU; tests/test_eedi2_gpu.py::test_vote_avg_every_case; the float identity on the host: tests/test_eedi2_identities_cpu.py).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

// (int)((float)a / (float)b + 0.5f) of the votes (a = sum + mid <= 2559, b = count + 1 <= 10) without the IEEE division:
// the float expression equals floor((2a + b) / 2b) there (every case checked on the host), and that quotient is the
// truncated product with v_rcp_f32's reciprocal, one too small at most (a quotient that is not an integer is at least
// 1 / 20 away from one; an exact one can come out a hair low): one compare puts it right.
__device__ __forceinline__ int vote_avg(int a, int b)
{
    const uint32_t n = 2u * (uint32_t)a + (uint32_t)b, m = 2u * (uint32_t)b;
    uint32_t q = (uint32_t)((float)n * __builtin_amdgcn_rcpf((float)m));
    q += (n - q * m >= m) ? 1u : 0u;
    return (int)q;
}

