// Experiment (not part of the library): issue cost, in engine cycles per wave64 instruction and SIMD, of the
// instructions the IIR and FFT roles are made of -- measured with the clock the chip actually runs at
// (clock64 deltas of a wave that lives through the whole run), 4 waves per SIMD like the chain's kernels,
// 8 independent accumulators per lane.  tools/valu_rate.hip (round 1) divided wall time by an assumed
// 2.4 GHz and only knew VOP3 forms.
//   hipcc -O3 --offload-arch=gfx950 tools/inst_cost.hip -o /tmp/inst_cost && /tmp/inst_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ __launch_bounds__(256) void probe(float *out, long long *clk, int iters, float seed, const double *sc)
{
    __shared__ float lds[256];
    float a[8]; v2f p[8]; double d[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; p[i] = (v2f){seed + i, seed - i}; d[i] = seed + i + threadIdx.x; }
    const float m = 0.999f, c = 0.001f;
    const v2f m2 = {m, m}, c2 = {c, c};
    const double md = 0.999, cd = 0.001;
    const double smd = sc[0];                       // wave-uniform -> SGPR pair
    int idx = (threadIdx.x & 63) * 4;
    lds[threadIdx.x] = seed;
    __syncthreads();
    const long long c0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(md), "v"(cd));
                if (MODE == 1) asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(d[i]) : "s"(smd), "v"(cd));
                if (MODE == 2) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[i]) : "v"(md), "v"(cd));
                if (MODE == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(md));
                if (MODE == 4) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
                if (MODE == 5) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
                if (MODE == 6) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
                if (MODE == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (MODE == 8) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (MODE == 9) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
                if (MODE == 10) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
                if (MODE == 11) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
                if (MODE == 12) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 13) asm volatile("v_mul_f32 %0, |%0|, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 14) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (MODE == 15) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 7]));
                if (MODE == 16) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : );
                if (MODE == 17) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(7)" : "+v"(a[i]) : "v"(idx));
                if (MODE == 18) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 19) asm volatile("v_fma_f64 %0, %1, |%0|, %2" : "+v"(d[i]) : "s"(smd), "v"(cd));
                if (MODE == 20) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(p[i]) : "v"(c2));
                if (MODE == 21) asm volatile("v_cvt_f64_f32 %0, |%1|" : "=v"(d[i]) : "v"(a[i]));
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const long long c1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

template <int MODE> void run(const char *name, float *out, long long *clk, const double *sc)
{
    const int iters = 2000, blocks = 256 * 4;      // 4 blocks of 4 waves per CU: 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(out, clk, 10, 1.f, sc);
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256>>>(out, clk, iters, 1.f, sc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static long long h[1024];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < blocks; i++) mean += (double)h[i] / blocks;
    // a SIMD hosts 4 waves, each issuing iters*128 of the instruction
    printf("%-28s %7.3f ms  clock %4.0f MHz  %5.2f cycles per wave-instruction and SIMD\n", name, ms,
           mean / (ms * 1e-3) / 1e6, mean / ((double)iters * 128 * 4));
}

int main()
{
    float *out; long long *clk; double *sc;
    hipMalloc(&out, 256 * 4 * 256 * 4); hipMalloc(&clk, 1024 * 8); hipMalloc(&sc, 8);
    const double one = 0.999;
    hipMemcpy(sc, &one, 8, hipMemcpyHostToDevice);
    run<0>("v_fma_f64 v,v,v", out, clk, sc); run<1>("v_fma_f64 s,v,v", out, clk, sc); run<2>("v_fmac_f64", out, clk, sc);
    run<3>("v_mul_f64", out, clk, sc); run<4>("v_add_f64", out, clk, sc);
    run<5>("v_cvt_f64_f32", out, clk, sc); run<6>("v_cvt_f32_f64", out, clk, sc); run<21>("v_cvt_f64_f32 |x|", out, clk, sc);
    run<19>("v_fma_f64 s,|v|,v", out, clk, sc);
    run<7>("v_fma_f32 (VOP3)", out, clk, sc); run<8>("v_fmac_f32 (VOP2)", out, clk, sc);
    run<9>("v_pk_fma_f32", out, clk, sc); run<10>("v_pk_add_f32", out, clk, sc); run<20>("v_pk_add_f32 op_sel/neg", out, clk, sc);
    run<11>("v_pk_mul_f32", out, clk, sc);
    run<12>("v_add_f32", out, clk, sc); run<13>("v_mul_f32 |x|", out, clk, sc); run<18>("v_max_f32", out, clk, sc);
    run<14>("v_mov_b32 dpp wave_shr:1", out, clk, sc); run<15>("v_permlane32_swap", out, clk, sc);
    run<16>("v_cndmask_b32", out, clk, sc); run<17>("ds_bpermute_b32", out, clk, sc);
    return 0;
}
