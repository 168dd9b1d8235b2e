// probe: does ds_read_b32 return the right bytes at unaligned LDS addresses on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k(unsigned *out)
{
    __shared__ unsigned char s[256];
    for (int i = threadIdx.x; i < 256; i += 64) s[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    unsigned v;
    memcpy(&v, s + threadIdx.x + 1, 4);                 // compiler's choice
    unsigned w;
    const unsigned addr = (unsigned)(size_t)(s + threadIdx.x + 1);
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(addr));
    out[threadIdx.x] = v;
    out[64 + threadIdx.x] = w;
}
int main()
{
    unsigned *d, h[128];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    k<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad_c = 0, bad_a = 0;
    for (int t = 0; t < 64; t++)
    {
        unsigned want = 0;
        for (int b = 0; b < 4; b++) want |= (unsigned)(unsigned char)((t + 1 + b) * 7 + 3) << (8 * b);
        bad_c += h[t] != want;
        bad_a += h[64 + t] != want;
    }
    printf("memcpy path mismatches %d, raw unaligned ds_read_b32 mismatches %d (lane1: got %08x)\n", bad_c, bad_a, h[65]);
    return 0;
}
