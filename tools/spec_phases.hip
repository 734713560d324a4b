// Experiment (not part of the library): cost of the phases of the 2048-point spectrogram frame
// kernel without any HBM traffic.  Built on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/spec_phases.hip audian_amd/csrc/ctx.hip -o /tmp/spec_phases
#include "../audian_amd/csrc/spectrogram.hip"

namespace {

template <int PH>
__global__ __launch_bounds__(256, 3) void phases_kernel(const float *__restrict__ tables, float *__restrict__ sink,
                                                         int frames_per_wave)
{
    constexpr int NFFT = 2048, LPF = 64, R1 = 16, R2 = 16, R3 = 4, M = 1024, PPL = 16, MP = M + M / 16;
    constexpr int TW2 = (R2 - 1) * R1, TW3 = R1 * R2, TWN = M / 2 + 1, NTAB = TW2 + TW3 + TWN + M;
    __shared__ float2 smem[NTAB + 4 * MP];
    const float2 *tw2 = smem, *tw3 = smem + TW2, *twn = smem + TW2 + TW3, *win = smem + TW2 + TW3 + TWN;
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    float2 *fb = smem + NTAB + wave * MP;
    const float2 *src = reinterpret_cast<const float2 *>(tables);
    for (int i = tid; i < NTAB; i += 256) smem[i] = src[i];
    __syncthreads();
    const int partner = (64 - l) & 63;
    float acc = 0.f;
    for (int it = 0; it < frames_per_wave; it++) {
        float2 v[PPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < PPL; i++) {
            v[i] = make_float2((float)(l + i + it) * 1e-3f, (float)(l - i) * 1e-3f);
            asm volatile("" : "+v"(v[i].x), "+v"(v[i].y));
            s += v[i].x + v[i].y;
        }
        if (PH >= 1) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
            const float mean = s * (1.0f / NFFT);
#pragma unroll
            for (int t = 0; t < R1; t++) {
                const float2 w = win[l + t * 64];
                v[t] = make_float2((v[t].x - mean) * w.x, (v[t].y - mean) * w.y);
            }
        }
        if (PH >= 2) stockham_stage<R1, 1, M, LPF, false, true>(v, fb, tw2, l);
        if (PH >= 3) stockham_stage<R2, R1, M, LPF, true, true>(v, fb, tw2, l);
        if (PH >= 4) stockham_stage<R3, R1 * R2, M, LPF, true, false, true>(v, fb, tw3, l);
        if (PH >= 5) {
            constexpr int NB3 = PPL / R3;
#pragma unroll
            for (int m = 0; m < PPL / 2; m++) {
                const int k = l + LPF * m;
                const float2 zk = v[(m % NB3) * R3 + m / NB3];
                const int mp = PPL - 1 - m;
                const float2 zsrc = v[(mp % NB3) * R3 + mp / NB3];
                float2 zm;
                zm.x = __shfl(zsrc.x, partner, 64);
                zm.y = __shfl(zsrc.y, partner, 64);
                const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
                const float2 od2 = make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
                const float2 t = cmul(od2, twn[k]);
                const float2 a = cadd(e, t), b = csub(e, t);
                acc += (a.x * a.x + a.y * a.y) + (b.x * b.x + b.y * b.y);
            }
        } else {
#pragma unroll
            for (int i = 0; i < PPL; i++) acc += v[i].x + v[i].y;
        }
    }
    if (acc == 123.456f) sink[blockIdx.x] = acc;
}

template <int PH> float run(const float *tables, float *sink, int blocks, int fpw)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(phases_kernel<PH>, dim3(blocks), dim3(256), 0, 0, tables, sink, fpw);
    hipEventRecord(a);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(phases_kernel<PH>, dim3(blocks), dim3(256), 0, 0, tables, sink, fpw);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

}  // namespace

int main()
{
    hipdsp_ctx *ctx;
    if (hipdsp_ctx_create(0, nullptr, &ctx)) { printf("%s\n", hipdsp_last_error()); return 1; }
    const float *tables;
    if (fft_tables(ctx, 2048, 16, 16, 4, &tables)) { printf("%s\n", hipdsp_last_error()); return 1; }
    float *sink; hipMalloc(&sink, 1 << 20);
    const int fpw = 64, frames = 64 * 11250;             // 120 s config
    const int blocks = frames / (4 * fpw);
    const char *names[] = {"synthetic input only", "+ mean, window", "+ stage 1 (dft16, LDS write)", "+ stage 2 (LDS read, twiddle, dft16, write)",
                           "+ stage 3 (LDS read, twiddle powers, 4 x dft4)", "+ split step (bpermute, twiddle, |.|^2)"};
    float ms[6];
    ms[0] = run<0>(tables, sink, blocks, fpw); ms[1] = run<1>(tables, sink, blocks, fpw); ms[2] = run<2>(tables, sink, blocks, fpw);
    ms[3] = run<3>(tables, sink, blocks, fpw); ms[4] = run<4>(tables, sink, blocks, fpw); ms[5] = run<5>(tables, sink, blocks, fpw);
    for (int i = 0; i < 6; i++) printf("%-52s %.3f ms (+%.3f)\n", names[i], ms[i], i ? ms[i] - ms[i - 1] : ms[i]);
    return 0;
}
