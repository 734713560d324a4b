// Experiment (not part of the library): which kind of work makes this MI355X lower its engine clock?
// The fused forward sweep runs at 1.72 GHz although the chip's peak is 2.4 GHz; at that
// clock it is bound by VALU issue.  Each variant below keeps every CU busy with 16 waves for a few
// milliseconds and reports shader clocks / 100 MHz ticks of wave 0 (clock64 vs wall_clock64):
//   fma64   dependent-free v_fma_f64 streams        fma32   the same in v_fma_f32
//   pk32    v_pk_fma_f32                            lds     ds_read_b128 / ds_write_b128 round trips
//   copy    16-byte loads and stores over 2 x 7 GB  mix     copy + fma64 in the chain's proportion
//   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Stamp { long long clk, wall; };

template <int MODE>
__global__ __launch_bounds__(64) void probe(const float4 *__restrict__ in, float4 *__restrict__ out, long long seg4,
                                            int iters, Stamp *st, double seed)
{
    __shared__ float4 lds[512];
    const int lane = threadIdx.x;
    const long long c0 = clock64(), w0 = wall_clock64();
    double a0 = seed + lane, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 0.999999, b = 1e-9;
    float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {f0, f1}, p1 = {f2, f3}, p2 = {f4, f5}, p3 = {f6, f7};
    const long long base = (long long)blockIdx.x * seg4;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || MODE == 5) {           // fp64 FMAs: 8 independent chains x 16
#pragma unroll
            for (int k = 0; k < 16; k++) {
                a0 = fma(a0, m, b); a1 = fma(a1, m, b); a2 = fma(a2, m, b); a3 = fma(a3, m, b);
                a4 = fma(a4, m, b); a5 = fma(a5, m, b); a6 = fma(a6, m, b); a7 = fma(a7, m, b);
            }
        }
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                f0 = fmaf(f0, 0.999999f, 1e-9f); f1 = fmaf(f1, 0.999999f, 1e-9f); f2 = fmaf(f2, 0.999999f, 1e-9f);
                f3 = fmaf(f3, 0.999999f, 1e-9f); f4 = fmaf(f4, 0.999999f, 1e-9f); f5 = fmaf(f5, 0.999999f, 1e-9f);
                f6 = fmaf(f6, 0.999999f, 1e-9f); f7 = fmaf(f7, 0.999999f, 1e-9f);
            }
        }
        if (MODE == 2) {
            const v2f mm = {0.999999f, 0.999998f}, bb = {1e-9f, 2e-9f};
#pragma unroll
            for (int k = 0; k < 32; k++) {
                p0 = __builtin_elementwise_fma(p0, mm, bb); p1 = __builtin_elementwise_fma(p1, mm, bb);
                p2 = __builtin_elementwise_fma(p2, mm, bb); p3 = __builtin_elementwise_fma(p3, mm, bb);
            }
        }
        if (MODE == 3) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float4 v = lds[(lane * 8 + k) & 511];
                v.x += 1.f;
                lds[(lane * 8 + ((k + 3) & 7)) & 511] = v;
            }
        }
        if (MODE == 4 || MODE == 5) {           // 8 KB in, 8 KB out per iteration and wave
            const long long t = base + (long long)it * 512;
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = in[t + 64 * k + lane];
#pragma unroll
            for (int k = 0; k < 8; k++) out[t + 64 * k + lane] = v[k];
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (blockIdx.x == 0 && lane == 0) { st->clk = c1 - c0; st->wall = w1 - w0; }
    const double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + p0.x + p0.y + p1.x +
                     p1.y + p2.x + p2.y + p3.x + p3.y;
    if (s == 12345.678) out[0].x = (float)s;
}

template <int MODE> void run(const char *name, const float4 *in, float4 *out, long long seg4, int iters, Stamp *st)
{
    const int waves = 256 * 16;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(probe<MODE>, dim3(waves), dim3(64), 0, 0, in, out, seg4, iters, st, 1.0);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
    }
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    Stamp h;
    CK(hipMemcpy(&h, st, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-6s %8.3f ms   engine clock %5.0f MHz", name, ms, (double)h.clk / ((double)h.wall / 100e6) / 1e6);
    if (MODE == 4 || MODE == 5) printf("   %.0f GB/s", 2.0 * waves * (double)iters * 8192 / ms / 1e6);
    if (MODE == 0 || MODE == 5) printf("   %.1f G wave-FMA64/s", (double)waves * iters * 128 / ms / 1e6);
    printf("\n");
}

int main()
{
    const long long n = 64LL * 28800000LL;       // 7.4 GB per array
    float4 *in, *out;
    Stamp *st;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&st, sizeof(Stamp)));
    CK(hipMemset(in, 1, n * 4)); CK(hipMemset(out, 0, n * 4));
    const long long seg4 = n / 4 / (256 * 16) / 512 * 512;
    const int copy_iters = (int)(seg4 / 512);
    for (int round = 0; round < 2; round++) {
        run<0>("fma64", in, out, seg4, 4000, st);
        run<1>("fma32", in, out, seg4, 4000, st);
        run<2>("pk32", in, out, seg4, 4000, st);
        run<3>("lds", in, out, seg4, 20000, st);
        run<4>("copy", in, out, seg4, copy_iters, st);
        run<5>("mix", in, out, seg4, copy_iters, st);
    }
    return 0;
}
