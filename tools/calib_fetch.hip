// calib_fetch.hip — calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access
// widths this library uses (MI355X_MICROARCH.md "HBM": only the 16 B/lane streaming read is
// calibrated there — "calibrate on a known byte count in your own access pattern").
// Each kernel touches a known number of bytes of a buffer larger than the 256 MiB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o /tmp/calib_fetch
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o c --output-format csv -- /tmp/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void k_read4(const uint32_t *__restrict__ src, uint32_t *sink, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v = i < n ? src[i] : 0;
    if (v == 0x12345679u) sink[0] = v;
}

__global__ void k_read16(const uint4 *__restrict__ src, uint32_t *sink, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 v = i < n ? src[i] : make_uint4(0, 0, 0, 0);
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345679u) sink[0] = v.x;
}

// The NLMeans tile load: a 128x64 output tile reads a (128+16)x(64+12) window as 36 dwords
// per row, 256 threads striding over the 76*36 dwords, pitch 2048 (nlmeans.hip load_tile).
__global__ void k_tile4(const uint32_t *__restrict__ src, uint32_t *sink, int pitch_dw, int rows)
{
    const int x0 = blockIdx.x * 32 - 2, y0 = blockIdx.y * 64 - 6;
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < 76 * 36; i += 256)
    {
        int r = y0 + i / 36, c = x0 + i % 36;
        r = r < 0 ? 0 : (r >= rows ? rows - 1 : r);
        c = c < 0 ? 0 : (c >= 480 ? 479 : c);
        acc ^= src[(size_t)r * pitch_dw + c];
    }
    if (acc == 0x12345679u) sink[0] = acc;
}

__global__ void k_write4(uint32_t *dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (uint32_t)i;
}

__global__ void k_write1(uint8_t *dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (uint8_t)i;
}

__global__ void k_read1(const uint8_t *__restrict__ src, uint32_t *sink, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v = i < n ? src[i] : 0;
    if (v == 0xfeu) sink[0] = v;
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    uint8_t *buf; uint32_t *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, bytes);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++)
    {
        size_t n4 = bytes / 4, n16 = bytes / 16;
        k_read4<<<dim3((unsigned)(n4 / 256)), 256>>>((const uint32_t *)buf, sink, n4);
        k_read16<<<dim3((unsigned)(n16 / 256)), 256>>>((const uint4 *)buf, sink, n16);
        const int rows = (int)(bytes / 2048);
        k_tile4<<<dim3(15, rows / 64), 256>>>((const uint32_t *)buf, sink, 512, rows);
        k_write4<<<dim3((unsigned)(n4 / 256)), 256>>>((uint32_t *)buf, n4);
        const size_t n1 = bytes / 4;       // 256 MiB of byte accesses
        k_read1<<<dim3((unsigned)(n1 / 256)), 256>>>(buf, sink, n1);
        k_write1<<<dim3((unsigned)(n1 / 256)), 256>>>(buf, n1);
        (void)hipDeviceSynchronize();
    }
    printf("expected_bytes k_read4=%zu k_read16=%zu k_tile4_unique=%zu k_write4=%zu k_read1=%zu k_write1=%zu\n",
           bytes, bytes, (size_t)(bytes / 2048) * 1920, bytes, bytes / 4, bytes / 4);
    return 0;
}
