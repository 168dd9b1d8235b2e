#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__global__ void k(const uint8_t *in, uint32_t *out, uint32_t *out2, int n)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[1024 + 16];
    for (int i = threadIdx.x; i < 1024 + 16; i += blockDim.x) s[i] = in[i];
    __syncthreads();
    const int t = threadIdx.x;
    uint32_t v;
    const uint32_t addr = (uint32_t)(uintptr_t)(s + t);   // LDS byte address
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[t] = v;
    uint32_t w;
    memcpy(&w, s + t, 4);
    out2[t] = w;
}
int main()
{
    uint8_t h[1040]; for (int i = 0; i < 1040; i++) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t *d; uint32_t *o, *o2; hipMalloc(&d, 1040); hipMalloc(&o, 4096); hipMalloc(&o2, 4096);
    hipMemcpy(d, h, 1040, hipMemcpyHostToDevice);
    k<<<1, 256>>>(d, o, o2, 0);
    uint32_t r[256], r2[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, 1024, hipMemcpyDeviceToHost);
    int bad = 0, bad2 = 0;
    for (int t = 0; t < 256; t++) { uint32_t e; memcpy(&e, h + t, 4); bad += r[t] != e; bad2 += r2[t] != e; }
    printf("unaligned ds_read_b32: %d mismatches of 256 (memcpy form %d)\n", bad, bad2);
    return 0;
}
