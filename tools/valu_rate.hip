// VALU issue-rate probe for gfx950: wave64 v_fma_f32 vs v_pk_fma_f32 vs v_fma_f64 vs v_pk_add_f32,
// 8 independent accumulators per lane, enough waves to fill every SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float seed)
{
    float a[8]; v2f p[8]; double d[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + i; p[i] = (v2f){seed + i, seed - i}; d[i] = seed + i; }
    const float m = 0.999f, c = 0.001f;
    const v2f m2 = {m, m}, c2 = {c, c};
    const double md = 0.999, cd = 0.001;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
                if (MODE == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(md), "v"(cd));
                if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
                if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "+v"(p[i]) : "v"(m2));
                if (MODE == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char *name, float *out)
{
    const int iters = 2000, blocks = 256 * 8;      // 8 blocks of 4 waves per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(out, 10, 1.f);
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256>>>(out, iters, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * 4 * iters * 128;          // wave instructions
    double per_simd_per_s = insts / 1024 / (ms * 1e-3);
    printf("%-12s %.3f ms  %.2f G wave-inst/s per SIMD (2.4 GHz / x = %.2f cycles per inst)\n", name, ms,
           per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
}

int main()
{
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", out); run<1>("v_pk_fma_f32", out); run<2>("v_fma_f64", out);
    run<3>("v_pk_add_f32", out); run<4>("v_pk_mul_f32", out); run<5>("v_add_f32", out);
    return 0;
}
