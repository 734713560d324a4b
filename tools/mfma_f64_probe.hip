// Experiment (not part of the library; VERDICT round 2, Next #2b): could the phase-1 dot products of the fused forward
// sweep (128 fp64 multiply-adds per lane and tile, f = sum_j G[j] x[j]) leave the VALU for the matrix pipe?
// Measures, with the clock the chip actually holds (clock64 of the last wave of a workgroup), 4 waves per SIMD
// (one 1024-thread workgroup per CU, pinned by its LDS):
//   * the issue cost of v_mfma_f64_4x4x4_4b_f64 (256 MACs: 4 blocks of 4 x 4 x 4 -- the shape without waste for a
//     [64 rows x 32 samples] . [32 x 4 states] product) and of v_mfma_f64_16x16x4_f64 (1024 MACs) next to v_fma_f64
//     (64 MACs per wave instruction);
//   * MIXED: half of the waves of every SIMD issue MFMAs, the other half v_fma_f64 -- does the VALU stream keep
//     its rate while the matrix pipe works (co-issue), and what clock does the chip hold then?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

// MODE 0: every wave v_fma_f64; 1: every wave mfma 4x4x4; 2: every wave mfma 16x16x4;
// 3: waves 0,2 of a SIMD mfma 4x4x4, waves 1,3 v_fma_f64 (wave w sits on SIMD w % 4: waves w/4 even -> mfma)
// 4: like 3 with 16x16x4; 5: every wave alternates 4 v_fma_f64 with 1 mfma 4x4x4 (the ratio of phase 1 : rest would be ~1:6)
template <int MODE>
__global__ __launch_bounds__(1024) void probe(double *out, long long *clk, int iters, double seed)
{
    __shared__ float lds[24 * 1024];             // 96 KB: one workgroup per CU
    lds[threadIdx.x] = (float)seed;
    const int wave = threadIdx.x >> 6;
    double d[8], a = seed + threadIdx.x, b = seed - 0.5;
    v4d acc[4];
    for (int i = 0; i < 8; i++) d[i] = seed + i;
    for (int i = 0; i < 4; i++) acc[i] = (v4d){seed, seed, seed, seed};
    double m1[8];
    for (int i = 0; i < 8; i++) m1[i] = seed * i;
    const double md = 0.999, cd = 0.001;
    const bool mf_wave = ((wave >> 2) & 1) == 0;
    __syncthreads();
    const long long c0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (MODE == 0 || ((MODE == 3 || MODE == 4) && !mf_wave)) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(md), "v"(cd));
            } else if (MODE == 1 || (MODE == 3 && mf_wave)) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(m1[i]) : "v"(a), "v"(b));
            } else if (MODE == 2 || (MODE == 4 && mf_wave)) {
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            } else if (MODE == 5) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(md), "v"(cd));
                asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(m1[r & 7]) : "v"(a), "v"(b));
                asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(m1[(r + 4) & 7]) : "v"(a), "v"(b));
            }
        }
    }
    const long long c1 = clock64();
    double s = 0;
    for (int i = 0; i < 8; i++) s += d[i] + m1[i];
    for (int i = 0; i < 4; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 16 + wave] = c1 - c0;
}

template <int MODE> void run(const char *name, double *out, long long *clk, double per_iter_valu, double per_iter_mfma_macs)
{
    const int iters = 1500, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 1024>>>(out, clk, 10, 1.0);
    hipEventRecord(e0);
    probe<MODE><<<blocks, 1024>>>(out, clk, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static long long h[256 * 16];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double last = 0, mf = 0, va = 0;
    for (int i = 0; i < blocks; i++) {
        long long mx = 0, a = 0, b = 0;
        for (int w = 0; w < 16; w++) {
            mx = h[16 * i + w] > mx ? h[16 * i + w] : mx;
            if (((w >> 2) & 1) == 0) a = h[16 * i + w] > a ? h[16 * i + w] : a; else b = h[16 * i + w] > b ? h[16 * i + w] : b;
        }
        last += (double)mx / blocks; mf += (double)a / blocks; va += (double)b / blocks;
    }
    // per SIMD: 4 waves; per iteration a wave issues per_iter_valu VALU instructions and/or MFMAs worth per_iter_mfma_macs MACs
    printf("%-46s %7.3f ms  clock %4.0f MHz  cycles of the launch %9.0f", name, ms, last / (ms * 1e-3) / 1e6, last);
    if (per_iter_valu > 0) printf("  | %5.2f cycles per v_fma_f64 and SIMD (its waves done after %9.0f)", (MODE == 3 || MODE == 4 ? va : last) / ((double)iters * per_iter_valu), MODE == 3 || MODE == 4 ? va : last);
    if (per_iter_mfma_macs > 0) printf("  | %5.1f MACs per cycle and SIMD on the matrix pipe (its waves done after %9.0f)", (double)iters * per_iter_mfma_macs / (MODE == 3 || MODE == 4 ? mf : last), MODE == 3 || MODE == 4 ? mf : last);
    printf("\n");
}

int main()
{
    double *out; long long *clk;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&clk, 256 * 16 * 8);
    // per SIMD and iteration: 4 waves x 128 v_fma_f64 | 4 waves x 128 mfma 4x4x4 (256 MACs) | 4 waves x 64 mfma 16x16x4 (1024 MACs)
    run<0>("every wave: v_fma_f64", out, clk, 4 * 128.0, 0);
    run<1>("every wave: v_mfma_f64_4x4x4_4b", out, clk, 0, 4 * 128.0 * 256);
    run<2>("every wave: v_mfma_f64_16x16x4", out, clk, 0, 4 * 64.0 * 1024);
    run<3>("half the waves mfma 4x4x4, half v_fma_f64", out, clk, 2 * 128.0, 2 * 128.0 * 256);
    run<4>("half the waves mfma 16x16x4, half v_fma_f64", out, clk, 2 * 128.0, 2 * 64.0 * 1024);
    run<5>("every wave: 8 v_fma_f64 then 2 mfma 4x4x4", out, clk, 4 * 128.0, 4 * 32.0 * 256);
    return 0;
}
