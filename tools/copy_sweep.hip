// Experiment (not part of the library): what read+write streaming rate does this MI355X give a plain
// float4 copy, and with which launch shape?  MI355X_MICROARCH.md quotes 6.29 TB/s for "float4 copy";
// round 1's tools/copy_bench.hip (grid-stride, 2048..16384 blocks, and per-wave segments) stayed at
// 5.0-5.7 TB/s.  This sweep covers the shapes that file did not: one float4 per thread without a loop,
// K float4 per thread (loads first, then stores), block sizes 256/512/1024, several buffer sizes, and a
// destination shifted against the source by a few KiB (same-channel read/write collisions).
//   hipcc --offload-arch=gfx950 -O3 tools/copy_sweep.hip -o /tmp/copy_sweep && /tmp/copy_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// one float4 per thread, no loop
__global__ void copy_one(const float4 *__restrict__ in, float4 *__restrict__ out, long long n4)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) out[i] = in[i];
}

// K float4 per thread, block-contiguous chunks: all loads, then all stores
template <int K>
__global__ void copy_k(const float4 *__restrict__ in, float4 *__restrict__ out, long long n4)
{
    const long long base = (long long)blockIdx.x * blockDim.x * K + threadIdx.x;
    float4 v[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const long long i = base + (long long)k * blockDim.x;
        if (i < n4) v[k] = in[i];
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        const long long i = base + (long long)k * blockDim.x;
        if (i < n4) out[i] = v[k];
    }
}

// grid-stride, K float4 in flight per thread
template <int K>
__global__ void copy_stride_k(const float4 *__restrict__ in, float4 *__restrict__ out, long long n4)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (K - 1) * stride < n4; i += K * stride) {
        float4 v[K];
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = in[i + k * stride];
#pragma unroll
        for (int k = 0; k < K; k++) out[i + k * stride] = v[k];
    }
    for (; i < n4; i += stride) out[i] = in[i];
}

// read only / write only, K per thread
template <int K>
__global__ void read_k(const float4 *__restrict__ in, float *__restrict__ out, long long n4)
{
    const long long base = (long long)blockIdx.x * blockDim.x * K + threadIdx.x;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const long long i = base + (long long)k * blockDim.x;
        if (i < n4) { const float4 v = in[i]; s += v.x + v.y + v.z + v.w; }
    }
    if (s == 12345.678f) out[0] = s;
}
template <int K>
__global__ void write_k(float4 *__restrict__ out, long long n4)
{
    const long long base = (long long)blockIdx.x * blockDim.x * K + threadIdx.x;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const long long i = base + (long long)k * blockDim.x;
        if (i < n4) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    }
}

template <typename F> float timeit(F f, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    std::vector<float> t;
    for (int i = 0; i < reps; i++) {
        CK(hipEventRecord(a));
        f();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main()
{
    const long long nmax = 64LL * 57600000LL;       // configs[2]: 64 ch x 600 s x 96 kHz floats
    char *bin, *bout;
    const size_t pad = 1 << 20;
    CK(hipMalloc(&bin, nmax * 4 + pad)); CK(hipMalloc(&bout, nmax * 4 + pad));
    CK(hipMemset(bin, 1, nmax * 4 + pad)); CK(hipMemset(bout, 0, nmax * 4 + pad));
    printf("buffers: in %p out %p\n", (void *)bin, (void *)bout);
    for (long long n : {nmax / 16, nmax / 4, nmax}) {
        const long long n4 = n / 4;
        const double gb = 2.0 * n * 4 / 1e9;
        printf("---- %.2f GB per array\n", n * 4 / 1e9);
        for (size_t shift : {(size_t)0, (size_t)4096, (size_t)65536 + 2048}) {
            const float4 *in = (const float4 *)bin;
            float4 *out = (float4 *)(bout + shift);
            for (int bs : {256, 512, 1024}) {
                const long long nb = (n4 + bs - 1) / bs;
                float ms = timeit([&] { hipLaunchKernelGGL(copy_one, dim3((unsigned)nb), dim3(bs), 0, 0, in, out, n4); }, 7);
                printf("copy_one   shift %6zu block %4d            : %.3f ms %.0f GB/s\n", shift, bs, ms, gb / ms * 1e3);
            }
            if (shift != 0) continue;
#define RUN_K(K, bs)                                                                                              \
    {                                                                                                             \
        const long long nb = (n4 + (long long)bs * K - 1) / ((long long)bs * K);                                  \
        float ms = timeit([&] { hipLaunchKernelGGL(copy_k<K>, dim3((unsigned)nb), dim3(bs), 0, 0, in, out, n4); }, 7); \
        printf("copy_k     K %2d block %4d                   : %.3f ms %.0f GB/s\n", K, bs, ms, gb / ms * 1e3);  \
    }
            RUN_K(2, 256) RUN_K(4, 256) RUN_K(8, 256) RUN_K(2, 512) RUN_K(4, 512) RUN_K(4, 1024) RUN_K(8, 64)
#undef RUN_K
#define RUN_S(K, blocks)                                                                                          \
    {                                                                                                             \
        float ms = timeit([&] { hipLaunchKernelGGL(copy_stride_k<K>, dim3(blocks), dim3(256), 0, 0, in, out, n4); }, 7); \
        printf("copy_stride K %2d blocks %6d x256            : %.3f ms %.0f GB/s\n", K, blocks, ms, gb / ms * 1e3); \
    }
            RUN_S(1, 2048) RUN_S(1, 8192) RUN_S(4, 2048) RUN_S(4, 4096) RUN_S(4, 8192) RUN_S(8, 2048) RUN_S(2, 16384)
#undef RUN_S
            {
                const long long nb = (n4 + 256 * 4 - 1) / (256 * 4);
                float ms = timeit([&] { hipLaunchKernelGGL(read_k<4>, dim3((unsigned)nb), dim3(256), 0, 0, in, (float *)out, n4); }, 7);
                printf("read_k     K  4 block  256                   : %.3f ms %.0f GB/s (read only)\n", ms, gb / 2 / ms * 1e3);
                ms = timeit([&] { hipLaunchKernelGGL(write_k<4>, dim3((unsigned)nb), dim3(256), 0, 0, out, n4); }, 7);
                printf("write_k    K  4 block  256                   : %.3f ms %.0f GB/s (write only)\n", ms, gb / 2 / ms * 1e3);
            }
        }
        {
            float ms = timeit([&] { CK(hipMemcpyAsync(bout, bin, n * 4, hipMemcpyDeviceToDevice, 0)); }, 7);
            printf("hipMemcpy D2D                                : %.3f ms %.0f GB/s\n", ms, gb / ms * 1e3);
        }
    }
    return 0;
}
