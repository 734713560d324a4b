// Experiment (not part of the library): does the NUMBER OF CONCURRENT STREAMS set the rate of a read+write
// sweep with persistent waves?  The IIR sweeps walk 4096 segments at once (one wave each, 8 KB tiles, next
// tile prefetched); a plain float4 copy with one short-lived wave per KB reaches 6.2 TB/s on the same box
// (tools/copy_sweep.hip), the IIR sweeps' memory pattern alone 5.6-5.8.  Here 4096 persistent waves (16 per
// CU) share S streams: the W/S waves of a group take the tiles of their stream round-robin, so S = 4096 is
// the IIR sweeps' pattern and S = 1 a chip-wide contiguous front.  Modes: copy, read only, write only; the
// next tile's loads are issued before the current tile's stores (as the sweeps do).
//   hipcc --offload-arch=gfx950 -O3 tools/stream_pattern.hip -o /tmp/stream_pattern && /tmp/stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MODE 0 copy, 1 read, 2 write.  tile = TK KB per wave and step.
template <int MODE, int TK>
__global__ __launch_bounds__(64) void sweep(const float4 *__restrict__ in, float4 *__restrict__ out, long long n4,
                                            int streams, int waves_per_stream, long long *ends = nullptr)
{
    const long long w0 = ends ? wall_clock64() : 0;
    const int lane = threadIdx.x;
    const int w = blockIdx.x;
    const int s = w / waves_per_stream, j = w % waves_per_stream;
    const long long per_stream = n4 / streams;                       // float4 per stream
    const long long tile4 = 64LL * TK;                                // float4 per tile
    const long long tiles = per_stream / tile4;
    const float4 *src = in + (long long)s * per_stream;
    float4 *dst = out + (long long)s * per_stream;
    float4 cur[TK], nxt[TK];
    float acc = 0.f;
    long long t = j;
    if (MODE != 2 && t < tiles) {
#pragma unroll
        for (int k = 0; k < TK; k++) cur[k] = src[t * tile4 + 64 * k + lane];
    }
    for (; t < tiles; t += waves_per_stream) {
        const long long tn = t + waves_per_stream;
        if (MODE != 2 && tn < tiles) {
#pragma unroll
            for (int k = 0; k < TK; k++) nxt[k] = src[tn * tile4 + 64 * k + lane];
        }
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < TK; k++) dst[t * tile4 + 64 * k + lane] = cur[k];
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < TK; k++) acc += cur[k].x + cur[k].y + cur[k].z + cur[k].w;
        } else {
#pragma unroll
            for (int k = 0; k < TK; k++) dst[t * tile4 + 64 * k + lane] = make_float4(1.f, 2.f, 3.f, (float)t);
        }
        if (MODE != 2) {
#pragma unroll
            for (int k = 0; k < TK; k++) cur[k] = nxt[k];
        }
    }
    if (acc == 12345.678f) out[0].x = acc;
    if (ends && threadIdx.x == 0) {                 // when did this wave start and end (100 MHz ticks), and where did it run
        ends[3 * blockIdx.x] = w0;
        ends[3 * blockIdx.x + 1] = wall_clock64();
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        ends[3 * blockIdx.x + 2] = hw;
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
    const long long n = 64LL * 57600000LL;       // configs[2]: 64 ch x 600 s x 96 kHz floats
    const long long n4 = n / 4;
    float4 *in, *out;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4));
    CK(hipMemset(in, 1, n * 4)); CK(hipMemset(out, 0, n * 4));
    const int W = 4096;
    for (int streams : {4096, 1024, 256, 64, 8, 1}) {
        const int wps = W / streams;
        const double gb = (double)(n4 / streams / 512 * 512) * streams * 16 / 1e9;
        float c = timeit([&] { hipLaunchKernelGGL((sweep<0, 8>), dim3(W), dim3(64), 0, 0, in, out, n4, streams, wps); }, 5);
        float r = timeit([&] { hipLaunchKernelGGL((sweep<1, 8>), dim3(W), dim3(64), 0, 0, in, out, n4, streams, wps); }, 5);
        float wr = timeit([&] { hipLaunchKernelGGL((sweep<2, 8>), dim3(W), dim3(64), 0, 0, in, out, n4, streams, wps); }, 5);
        printf("streams %5d (8 KB tiles): copy %.3f ms %5.0f GB/s | read %.3f ms %5.0f GB/s | write %.3f ms %5.0f GB/s\n",
               streams, c, 2 * gb / c * 1e3, r, gb / r * 1e3, wr, gb / wr * 1e3);
    }
    for (int streams : {4096, 1}) {
        const int wps = W / streams;
        const double gb = (double)(n4 / streams / 128 * 128) * streams * 16 / 1e9;
        float c = timeit([&] { hipLaunchKernelGGL((sweep<0, 2>), dim3(W), dim3(64), 0, 0, in, out, n4, streams, wps); }, 5);
        printf("streams %5d (2 KB tiles): copy %.3f ms %5.0f GB/s\n", streams, c, 2 * gb / c * 1e3);
    }
    {   // do the 4096 waves of the sweeps' pattern end together?  (a tail with few waves left cannot fill the HBM pipes)
        long long *ends; CK(hipMalloc(&ends, W * 24));
        hipLaunchKernelGGL((sweep<0, 8>), dim3(W), dim3(64), 0, 0, in, out, n4, W, 1, ends);
        hipLaunchKernelGGL((sweep<0, 8>), dim3(W), dim3(64), 0, 0, in, out, n4, W, 1, ends);
        CK(hipDeviceSynchronize());
        std::vector<long long> h(3 * W);
        CK(hipMemcpy(h.data(), ends, W * 24, hipMemcpyDeviceToHost));
        long long t0 = h[0];
        for (int i = 0; i < W; i++) t0 = std::min(t0, h[3 * i]);
        std::vector<double> e(W);
        double slot_sum[16] = {0}; int slot_n[16] = {0};
        for (int i = 0; i < W; i++) {
            e[i] = (h[3 * i + 1] - t0) * 1e-5;       // ms
            const int slot = (int)(h[3 * i + 2] & 15); // wave slot within the SIMD
            slot_sum[slot] += e[i]; slot_n[slot]++;
        }
        std::vector<double> sorted = e;
        std::sort(sorted.begin(), sorted.end());
        printf("copy, 4096 streams: waves end at min %.3f  10%% %.3f  median %.3f  90%% %.3f  max %.3f ms\n", sorted[0], sorted[W / 10],
               sorted[W / 2], sorted[W * 9 / 10], sorted[W - 1]);
        printf("mean end by wave slot of the SIMD:");
        for (int k = 0; k < 16; k++) if (slot_n[k]) printf("  %d: %.3f (%d)", k, slot_sum[k] / slot_n[k], slot_n[k]);
        printf("\n");
    }
    for (int wv : {8192, 2048}) {
        const double gb = (double)(n4 / wv / 512 * 512) * wv * 16 / 1e9;
        float c = timeit([&] { hipLaunchKernelGGL((sweep<0, 8>), dim3(wv), dim3(64), 0, 0, in, out, n4, wv, 1); }, 5);
        printf("waves %5d, one stream each (8 KB tiles): copy %.3f ms %5.0f GB/s\n", wv, c, 2 * gb / c * 1e3);
    }
    return 0;
}
