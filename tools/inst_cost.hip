// Experiment (not part of the library): issue cost, in engine cycles per wave64 instruction and SIMD, of the
// instructions the IIR and FFT roles are made of -- measured with the clock the chip actually runs at
// (clock64 deltas of a wave that lives through the whole run), 4 waves per SIMD like the chain's kernels
// (one 1024-thread workgroup per CU, pinned there by its LDS),
// 8 independent accumulators per lane.  tools/valu_rate.hip (round 1) divided wall time by an assumed
// 2.4 GHz and only knew VOP3 forms.
//   hipcc -O3 --offload-arch=gfx950 tools/inst_cost.hip -o /tmp/inst_cost && /tmp/inst_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ __launch_bounds__(1024) void probe(float *out, long long *clk, int iters, float seed, const double *sc)
{
    __shared__ float lds[24 * 1024];             // 96 KB: one workgroup of 16 waves per CU, 4 waves on every SIMD
    lds[threadIdx.x + 1024] = seed;
    float a[8]; v2f p[8]; double d[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; p[i] = (v2f){seed + i, seed - i}; d[i] = seed + i + threadIdx.x; }
    const float m = 0.999f, c = 0.001f;
    const v2f m2 = {m, m}, c2 = {c, c};
    const double md = 0.999, cd = 0.001;
    const double smd = sc[0];                       // wave-uniform -> SGPR pair
    int idx = (threadIdx.x & 63) * 4;
    unsigned long long msk = 0x5555555555555555ull, cm[2] = {0, 0};
    unsigned sl[2] = {0, 0};
    lds[threadIdx.x & 255] = seed;
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
                if (MODE == 22) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (MODE == 23) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 24) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 25) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 26) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "s"(msk));
                if (MODE == 27) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(cm[i & 1]) : "v"(a[i]), "v"(c));
                if (MODE == 28) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 29) asm volatile("v_max_f32 %0, 0, %0" : "+v"(a[i]));
                if (MODE == 30) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(d[(i + 1) & 7]));
                if (MODE == 31) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(c) : "vcc");
                if (MODE == 32) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 33) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(sl[i & 1]) : "v"(a[i]));
                if (MODE == 34) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(a[i]));
                if (MODE == 35) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (MODE == 36) asm volatile("v_fma_f64 %0, %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(d[i]) : "v"(md), "v"(cd));
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const long long c1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + (float)d[i];
    s += (float)(cm[0] + cm[1]) + (float)(sl[0] + sl[1]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 16 + (threadIdx.x >> 6)] = c1 - c0;
}

template <int MODE> void run(const char *name, float *out, long long *clk, const double *sc)
{
    const int iters = 2000, blocks = 256;          // one 1024-thread workgroup per CU (LDS-bound): 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 1024>>>(out, clk, 10, 1.f, sc);
    hipEventRecord(e0);
    probe<MODE><<<blocks, 1024>>>(out, clk, iters, 1.f, sc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // the issue arbiter serves the oldest wave of a SIMD first: the waves of the low slots end early, only the
    // LAST wave of a workgroup has seen all 4 x iters x 128 instructions of its SIMD go by
    static long long h[256 * 16];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0, first = 0;
    for (int i = 0; i < blocks; i++) {
        long long mx = 0, mn = h[16 * i];
        for (int w = 0; w < 16; w++) { mx = h[16 * i + w] > mx ? h[16 * i + w] : mx; mn = h[16 * i + w] < mn ? h[16 * i + w] : mn; }
        mean += (double)mx / blocks; first += (double)mn / blocks;
    }
    // a SIMD hosts 4 waves, each issuing iters*128 of the instruction
    printf("%-28s %7.3f ms  clock %4.0f MHz  %5.2f cycles per wave-instruction and SIMD  (first wave done after %2.0f %%)\n", name, ms,
           mean / (ms * 1e-3) / 1e6, mean / ((double)iters * 128 * 4), 100 * first / mean);
}

int main()
{
    float *out; long long *clk; double *sc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 256 * 16 * 8); hipMalloc(&sc, 8);
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
    run<16>("v_cndmask_b32 (vcc)", out, clk, sc); run<26>("v_cndmask_b32 (sgpr mask)", out, clk, sc);
    run<22>("v_mov_b32", out, clk, sc); run<30>("v_mov_b64", out, clk, sc); run<35>("v_mov_b32 dpp row_shr:1", out, clk, sc);
    run<23>("v_add_u32", out, clk, sc); run<24>("v_lshl_add_u32", out, clk, sc); run<25>("v_and_b32", out, clk, sc);
    run<34>("v_bfe_u32", out, clk, sc); run<31>("v_add_co_u32", out, clk, sc);
    run<27>("v_cmp_lt_f32 -> sgpr", out, clk, sc); run<28>("v_mul_f32", out, clk, sc); run<32>("v_sub_f32", out, clk, sc);
    run<29>("v_max_f32 0, x", out, clk, sc); run<33>("v_readlane_b32", out, clk, sc);
    run<17>("ds_bpermute_b32", out, clk, sc);
    return 0;
}
