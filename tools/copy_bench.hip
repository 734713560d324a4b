// Experiment (not part of the library): HBM ceilings for the access patterns of the
// hot-path kernels.  hipcc --offload-arch=gfx950 -O3 tools/copy_bench.hip -o /tmp/copy_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// (a) classic grid-stride float4 copy
__global__ void copy_grid(const float4 *__restrict__ in, float4 *__restrict__ out, long long n4)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        out[i] = in[i];
}

// (b) one 64-lane wave per contiguous segment, 8 x 1 KB loads then 8 x 1 KB stores per tile
__global__ __launch_bounds__(64) void copy_segments(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                    long long seg4, long long n4)
{
    long long base = (long long)blockIdx.x * seg4;
    long long end = base + seg4 < n4 ? base + seg4 : n4;
    for (long long t = base; t < end; t += 512) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = in[t + 64 * k + threadIdx.x];
#pragma unroll
        for (int k = 0; k < 8; k++) out[t + 64 * k + threadIdx.x] = v[k];
    }
}

// (c) like (b) but waves of a workgroup (256 threads) take interleaved tiles of one segment
__global__ __launch_bounds__(256) void copy_segments_wg(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                        long long seg4, long long n4)
{
    long long base = (long long)blockIdx.x * seg4;
    long long end = base + seg4 < n4 ? base + seg4 : n4;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (long long t = base + 512 * wave; t < end; t += 2048) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = in[t + 64 * k + lane];
#pragma unroll
        for (int k = 0; k < 8; k++) out[t + 64 * k + lane] = v[k];
    }
}

// (b') like (b) with non-temporal loads and/or stores (streaming data is touched once)
template <bool NTL, bool NTS>
__global__ __launch_bounds__(64) void copy_segments_nt(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                       long long seg4, long long n4)
{
    long long base = (long long)blockIdx.x * seg4;
    long long end = base + seg4 < n4 ? base + seg4 : n4;
    for (long long t = base; t < end; t += 512) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float4 *p = in + t + 64 * k + threadIdx.x;
            if (NTL) {
                v[k].x = __builtin_nontemporal_load(&p->x); v[k].y = __builtin_nontemporal_load(&p->y);
                v[k].z = __builtin_nontemporal_load(&p->z); v[k].w = __builtin_nontemporal_load(&p->w);
            } else {
                v[k] = *p;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float4 *q = out + t + 64 * k + threadIdx.x;
            if (NTS) {
                __builtin_nontemporal_store(v[k].x, &q->x); __builtin_nontemporal_store(v[k].y, &q->y);
                __builtin_nontemporal_store(v[k].z, &q->z); __builtin_nontemporal_store(v[k].w, &q->w);
            } else {
                *q = v[k];
            }
        }
    }
}

// (b'') like (b) but walking the segment from its end to its start (the envelope's backward sweep)
template <bool BACKWARD>
__global__ __launch_bounds__(64) void copy_segments_dir(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                        long long seg4, long long n4)
{
    const long long base = (long long)blockIdx.x * seg4;
    const long long end = base + seg4 < n4 ? base + seg4 : n4;
    const long long tiles = (end - base) / 512;
    float4 v[8];
    for (long long i = 0; i < tiles; i++) {
        const long long t = base + (BACKWARD ? tiles - 1 - i : i) * 512;
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = in[t + 64 * k + threadIdx.x];
#pragma unroll
        for (int k = 0; k < 8; k++) out[t + 64 * k + threadIdx.x] = v[k];
    }
}

// (d) read-only: each wave sums its segment (read ceiling)
__global__ __launch_bounds__(64) void read_segments(const float4 *__restrict__ in, float *__restrict__ out,
                                                    long long seg4, long long n4)
{
    long long base = (long long)blockIdx.x * seg4;
    long long end = base + seg4 < n4 ? base + seg4 : n4;
    float s = 0.f;
    for (long long t = base; t < end; t += 512) {
#pragma unroll
        for (int k = 0; k < 8; k++) { float4 v = in[t + 64 * k + threadIdx.x]; s += v.x + v.y + v.z + v.w; }
    }
    if (s == 12345.678f) out[blockIdx.x] = s;
}

template <typename F> float timeit(F f, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const long long n = 64LL * 57600000LL;       // configs[2]: 64 ch x 600 s x 96 kHz floats
    const long long n4 = n / 4;
    float4 *in, *out;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4));
    CK(hipMemset(in, 1, n * 4)); CK(hipMemset(out, 0, n * 4));
    const double gb = 2.0 * n * 4 / 1e9;
    for (int blocks : {2048, 4096, 8192, 16384}) {
        float ms = timeit([&] { hipLaunchKernelGGL(copy_grid, dim3(blocks), dim3(256), 0, 0, in, out, n4); }, 5);
        printf("copy_grid       blocks %6d x256: %.3f ms %.0f GB/s\n", blocks, ms, gb / ms * 1e3);
    }
    for (int waves : {2048, 4096, 5120, 8192, 16384}) {
        long long seg4 = (n4 / waves + 511) / 512 * 512;
        int nb = (int)((n4 + seg4 - 1) / seg4);
        float ms = timeit([&] { hipLaunchKernelGGL(copy_segments, dim3(nb), dim3(64), 0, 0, in, out, seg4, n4); }, 5);
        printf("copy_segments   waves %6d: %.3f ms %.0f GB/s\n", nb, ms, gb / ms * 1e3);
    }
    {
        long long seg4 = (n4 / 4096 + 511) / 512 * 512;
        int nb = (int)((n4 + seg4 - 1) / seg4);
        float ms = timeit([&] { hipLaunchKernelGGL((copy_segments_nt<false, true>), dim3(nb), dim3(64), 0, 0, in, out, seg4, n4); }, 5);
        printf("copy_segments nt stores        : %.3f ms %.0f GB/s\n", ms, gb / ms * 1e3);
        ms = timeit([&] { hipLaunchKernelGGL((copy_segments_nt<true, false>), dim3(nb), dim3(64), 0, 0, in, out, seg4, n4); }, 5);
        printf("copy_segments nt loads         : %.3f ms %.0f GB/s\n", ms, gb / ms * 1e3);
        ms = timeit([&] { hipLaunchKernelGGL((copy_segments_nt<true, true>), dim3(nb), dim3(64), 0, 0, in, out, seg4, n4); }, 5);
        printf("copy_segments nt loads + stores: %.3f ms %.0f GB/s\n", ms, gb / ms * 1e3);
    }
    for (int waves : {2048, 4096}) {
        long long seg4 = (n4 / waves + 511) / 512 * 512;
        int nb = (int)((n4 + seg4 - 1) / seg4);
        float ms = timeit([&] { hipLaunchKernelGGL((copy_segments_dir<true>), dim3(nb), dim3(64), 0, 0, in, out, seg4, n4); }, 5);
        printf("copy_segments backward waves %5d: %.3f ms %.0f GB/s\n", nb, ms, gb / ms * 1e3);
    }
    for (int wgs : {1024, 2048, 4096}) {
        long long seg4 = (n4 / wgs + 2047) / 2048 * 2048;
        int nb = (int)((n4 + seg4 - 1) / seg4);
        float ms = timeit([&] { hipLaunchKernelGGL(copy_segments_wg, dim3(nb), dim3(256), 0, 0, in, out, seg4, n4); }, 5);
        printf("copy_segments_wg wgs %6d: %.3f ms %.0f GB/s\n", nb, ms, gb / ms * 1e3);
    }
    for (int waves : {4096, 8192}) {
        long long seg4 = (n4 / waves + 511) / 512 * 512;
        int nb = (int)((n4 + seg4 - 1) / seg4);
        float ms = timeit([&] { hipLaunchKernelGGL(read_segments, dim3(nb), dim3(64), 0, 0, in, (float *)out, seg4, n4); }, 5);
        printf("read_segments   waves %6d: %.3f ms %.0f GB/s (read only)\n", nb, ms, gb / 2 / ms * 1e3);
    }
    return 0;
}
