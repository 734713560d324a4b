// Experiment (not part of the library): what does a wave instruction COST IN ENERGY on this part?  The fused forward
// sweep runs at the package power cap (1.31-1.33 kW of 1.4 kW), so its time is its energy divided by
// the cap.  This program keeps one instruction class running on every SIMD (4 waves per SIMD, one 1024-thread
// workgroup per CU) for a few seconds while tools/energy_probe.sh samples rocm-smi; it prints the instruction rate,
// the script adds the power: joules per wave instruction = (P - P_idle_loop) / rate.
//   hipcc -O3 --offload-arch=gfx950 tools/energy_probe.hip -o /tmp/energy_probe && /tmp/energy_probe MODE SECONDS
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void probe(float *out, const float *src, float *dst, long long n, int iters, float seed, const double *sc)
{
    __shared__ float lds[24 * 1024];             // 96 KB: one workgroup of 16 waves per CU
    for (int i = threadIdx.x; i < 24 * 1024; i += 1024) lds[i] = seed;
    float a[8]; v2f p[8]; double d[8]; v4f q[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; p[i] = (v2f){seed + i, seed - i}; d[i] = seed + i + threadIdx.x; q[i] = (v4f){seed, seed, seed, seed}; }
    const float m = 0.999f, c = 0.001f;
    const v2f m2 = {m, m}, c2 = {c, c};
    const double md = 0.999, cd = 0.001;
    const double smd = sc[0];
    const int la = (threadIdx.x & 1023) * 16;            // ds_*_b128 address (bytes), conflict-free
    int idx = (threadIdx.x & 63) * 4;
    const long long gid = (long long)blockIdx.x * 1024 + threadIdx.x;
    __syncthreads();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) asm volatile("s_sleep 8");
                if (MODE == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(md), "v"(cd));
                if (MODE == 2) asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(d[i]) : "s"(smd), "v"(cd));
                if (MODE == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
                if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
                if (MODE == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (MODE == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : );
                if (MODE == 7) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (MODE == 8) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(3)" : "=v"(q[i]) : "v"(la));
                if (MODE == 9) asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(3)" : : "v"(la), "v"(q[i]) : "memory");
                if (MODE == 10) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(7)" : "+v"(a[i]) : "v"(idx));
                if (MODE == 11) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(md));
                if (MODE == 12) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
                if (MODE == 13) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
                if (MODE == 14) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(md), "v"(cd));
            }
        if (MODE >= 20) {                        // HBM stream: one float4 in, one out per thread and iteration
            const long long k = (gid + (long long)it * gridDim.x * 1024) % n;
            const v4f *sp = reinterpret_cast<const v4f *>(src) + k;
            v4f *dp = reinterpret_cast<v4f *>(dst) + k;
            if (MODE == 20) *dp = *sp;
            if (MODE == 21) __builtin_nontemporal_store(__builtin_nontemporal_load(sp), dp);
            if (MODE == 22) { const v4f t = *sp; q[it & 7] += t; }                     // read only
            if (MODE == 23) *dp = q[it & 7];                                            // write only
            if (MODE == 24) { const v4f t = __builtin_nontemporal_load(sp); q[it & 7] += t; }
            if (MODE == 25) __builtin_nontemporal_store(q[it & 7], dp);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + (float)d[i] + q[i].x;
    out[gid % (256 * 1024)] = s;
}

template <int MODE> double run(float *out, const float *src, float *dst, long long n, const double *sc, double seconds)
{
    const int iters = 4000, blocks = MODE >= 20 ? 2048 : 256;
    probe<MODE><<<blocks, 1024>>>(out, src, dst, n, 10, 1.f, sc);
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    long long launches = 0;
    double el = 0;
    do {
        for (int k = 0; k < 4; k++) probe<MODE><<<blocks, 1024>>>(out, src, dst, n, iters, 1.f, sc);
        (void)hipDeviceSynchronize();
        launches += 4;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < seconds);
    const double winst = (double)launches * blocks * 16 * iters * 128;    // wave instructions of the probed class
    if (MODE >= 20) printf("mode %d: %.3f s, %.1f GB/s\n", MODE, el, (double)launches * blocks * 1024 * iters * ((MODE == 20 || MODE == 21) ? 32 : 16) / el / 1e9);
    else printf("mode %d: %.3f s, %.3f G wave-instructions per second\n", MODE, el, winst / el / 1e9);
    return el;
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 1;
    const double seconds = argc > 2 ? atof(argv[2]) : 4.0;
    float *out, *src, *dst; double *sc;
    const long long n = 1LL << 28;                     // float4 elements: 4 GiB each way
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&sc, 8);
    (void)hipMalloc(&src, n * 16); (void)hipMalloc(&dst, n * 16);
    (void)hipMemset(src, 0, n * 16);
    const double one = 0.999;
    (void)hipMemcpy(sc, &one, 8, hipMemcpyHostToDevice);
#define CASE(M) case M: run<M>(out, src, dst, n, sc, seconds); break;
    switch (mode) {
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(20) CASE(21) CASE(22) CASE(23) CASE(24) CASE(25)
    default: printf("unknown mode\n");
    }
    return 0;
}
