// Does a producer -> consumer hand-over through the 256 MB Infinity Cache (MALL) beat HBM?
// Kernel A writes a buffer of S bytes, kernel B reads it right after; also a plain re-read of
// the same S bytes.  hipcc -O3 --offload-arch=gfx950 tools/mall_probe.hip -o /tmp/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void wr(float4 *out, long long n4, float v)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        out[i] = make_float4(v, v, v, v);
}
__global__ void rd(const float4 *__restrict__ in, float *sink, long long n4)
{
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 v = in[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 1234.5f) sink[0] = s;
}
__global__ void cp(const float4 *__restrict__ in, float4 *__restrict__ out, long long n4)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        out[i] = in[i];
}

int main()
{
    const long long total = 8LL << 30;               // walk 8 GiB in chunks of S
    float4 *a, *b; float *sink;
    CK(hipMalloc(&a, total)); CK(hipMalloc(&b, total)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 0, total)); CK(hipMemset(b, 0, total));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long sizes[] = {16LL << 20, 32LL << 20, 64LL << 20, 128LL << 20, 192LL << 20, 256LL << 20, 512LL << 20, 2048LL << 20};
    for (long long S : sizes) {
        const long long n4 = S / 16, chunks = total / S;
        const int grid = 256 * 8;
        // (1) write chunk then read it back, chunk after chunk
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            for (long long c = 0; c < chunks; c++) {
                wr<<<grid, 256>>>(a + c * n4, n4, 1.f);
                rd<<<grid, 256>>>(a + c * n4, sink, n4);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms1; CK(hipEventElapsedTime(&ms1, e0, e1));
        // (2) copy a -> b chunk, then read b chunk and a chunk again (three consumers of a resident chunk)
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            for (long long c = 0; c < chunks; c++) {
                cp<<<grid, 256>>>(a + c * n4, b + c * n4, n4);
                rd<<<grid, 256>>>(b + c * n4, sink, n4);
                rd<<<grid, 256>>>(b + c * n4, sink, n4);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms2; CK(hipEventElapsedTime(&ms2, e0, e1));
        printf("chunk %5lld MiB: write+read %.2f ms = %.0f GB/s of touched bytes | copy+2 reads %.2f ms = %.0f GB/s\n",
               S >> 20, ms1, 2.0 * total / ms1 / 1e6, ms2, 4.0 * total / ms2 / 1e6);
    }
    return 0;
}
