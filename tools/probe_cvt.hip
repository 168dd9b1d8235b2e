// probe: rounding behaviour of v_cvt_pk_u8_f32 on gfx950 (is it truncation or round-to-nearest?)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *in, unsigned *out, int n)
{
    int i = threadIdx.x;
    if (i >= n) return;
    unsigned r;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(r) : "v"(in[i]));
    out[i] = r;
}
int main()
{
    const float h[] = {0.f, 0.49f, 0.5f, 0.51f, 0.99f, 1.0f, 1.5f, 2.5f, 3.5f, 126.99f, 127.0f, 127.5f, 254.7f, 255.0f, 255.5f, 256.f, 300.f, 1e9f, -0.5f, -1.f, -3.7f};
    const int n = sizeof(h) / sizeof(h[0]);
    float *d; unsigned *o; unsigned ho[64];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess || hipMalloc(&o, 256) != hipSuccess) return 1;
    (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, n);
    (void)hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("%g -> %u\n", h[i], ho[i]);
    return 0;
}
