// vote_avg_check.hip - every (a, b) the EEDI2 dir-map kernels hand to vote_avg (a = sum + mid <= 2559, b = count + 1 = 1..10;
// handbrake_amd/csrc/eedi2_vote.h) through the function ON THE GPU (its v_rcp_f32 is the hardware's), against the float
// expression of the reference (eedi2_template.c:703, :767, :850) and against the integer floor both are claimed to equal.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -Ihandbrake_amd/csrc tools/vote_avg_check.hip -o tools/vote_avg_check
#include "eedi2_vote.h"
#include <cstdio>
#include <vector>
constexpr int NA = 2560, NB = 10;
__global__ void k(int *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NA * NB) return;
    out[i] = vote_avg(i / NB, 1 + i % NB);
}
int main()
{
    int *d;
    if (hipMalloc(&d, sizeof(int) * NA * NB) != hipSuccess) { printf("no device\n"); return 2; }
    hipLaunchKernelGGL(k, dim3((NA * NB + 255) / 256), dim3(256), 0, 0, d);
    std::vector<int> h(NA * NB);
    if (hipMemcpy(h.data(), d, sizeof(int) * NA * NB, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 2; }
    int bad_float = 0, bad_floor = 0;
    for (int i = 0; i < NA * NB; i++)
    {
        const int a = i / NB, b = 1 + i % NB;
        volatile float q = (float)a / (float)b;              // the reference's expression, operation by operation
        volatile float r = q + 0.5f;
        bad_float += h[i] != (int)r;
        bad_floor += h[i] != (2 * a + b) / (2 * b);
    }
    printf("vote_avg: %d cases, %d differ from the float expression, %d from floor((2a+b)/2b)\n", NA * NB, bad_float, bad_floor);
    return bad_float || bad_floor;
}
