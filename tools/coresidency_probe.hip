// What happens to the IIR sweeps when another kernel holds a few CU slots (an RCCL all-gather in
// the multi-GPU bench, say)?  The sweeps launch ONE wave per (channel, segment) and expect all of
// them to be resident at once; if a few cannot start until others finish, the kernel takes two
// rounds.  This probe runs the forward sweep of BASELINE configs[2] through the C ABI next to a
// small spinning kernel and reports its duration for different `sos_waves_per_cu`.
//   hipcc -O3 --offload-arch=gfx950 -Iinclude tools/coresidency_probe.hip -Laudian_amd -lhip_dsp \
//         -Wl,-rpath,$PWD/audian_amd -o /tmp/coresidency_probe && /tmp/coresidency_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "hip_dsp.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define OK(x) do { int rc = (x); if (rc) { printf("%s: %s\n", #x, hipdsp_last_error()); exit(1); } } while (0)

__global__ __launch_bounds__(256) void spin(long long ticks, float *sink)
{
    const long long t0 = wall_clock64();
    float v = threadIdx.x;
    while (wall_clock64() - t0 < ticks) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) sink[0] = v;
}

int main()
{
    const long long C = 64, T = 57600000;
    hipdsp_ctx *ctx;
    OK(hipdsp_ctx_create(0, nullptr, &ctx));
    void *sa, *sb;
    OK(hipdsp_stream_create(ctx, &sa));
    OK(hipdsp_stream_create(ctx, &sb));
    OK(hipdsp_ctx_set_stream(ctx, sa));
    float *x, *yf, *env, *sink;
    OK(hipdsp_malloc(ctx, 4 * C * T, (void **)&x));
    OK(hipdsp_malloc(ctx, 4 * C * T, (void **)&yf));
    OK(hipdsp_malloc(ctx, 4 * C * T, (void **)&env));
    OK(hipdsp_malloc(ctx, 256, (void **)&sink));
    OK(hipdsp_synth(ctx, x, T, C, T, 96000.0, 7, 0, C));
    // butter(2, [300, 3000], 'bandpass', fs=96000) and butter(2, 20, 'lowpass', fs=96000), sos rows
    const double bp[12] = {0.006858271717317, 0.013716543434634, 0.006858271717317, 1.0, -1.757943364339393, 0.786412433954677,
                           1.0, -2.0, 1.0, 1.0, -1.971867074054536, 0.972353756531464};
    const double lp[6] = {4.281316461e-07, 8.562632922e-07, 4.281316461e-07, 1.0, -1.998148849, 0.998150562};
    hipdsp_sosplan *fp, *ep;
    OK(hipdsp_sosplan_create(ctx, &fp));
    OK(hipdsp_sosplan_create(ctx, &ep));
    OK(hipdsp_sosplan_set(ctx, fp, bp, 2));
    OK(hipdsp_sosplan_set(ctx, ep, lp, 1));
    void *e0, *e1;
    OK(hipdsp_event_create(ctx, &e0));
    OK(hipdsp_event_create(ctx, &e1));
    int freq_khz = 0;
    CK(hipDeviceGetAttribute(&freq_khz, hipDeviceAttributeWallClockRate, 0));
    for (int waves : {16, 12, 8}) {
        OK(hipdsp_ctx_set_option(ctx, "sos_waves_per_cu", waves));
        for (int busy_wgs : {0, 16, 64}) {
            for (int phase = 1; phase <= 2; phase++) {
                float best = 1e9f;
                for (int rep = 0; rep < 3; rep++) {
                    OK(hipdsp_sosfilt_envelope(ctx, fp, ep, x, T, yf, T, env, T, C, T, 1, M_PI / 2, 1, 1));
                    OK(hipdsp_ctx_synchronize(ctx));
                    if (busy_wgs) spin<<<busy_wgs, 256, 0, (hipStream_t)sb>>>((long long)freq_khz * 40, sink);   // ~40 ms
                    OK(hipdsp_event_record(ctx, e0));
                    OK(hipdsp_sosfilt_envelope(ctx, fp, ep, x, T, yf, T, env, T, C, T, 1, M_PI / 2, 1, phase));
                    OK(hipdsp_event_record(ctx, e1));
                    float ms;
                    OK(hipdsp_event_elapsed_ms(ctx, e0, e1, &ms));
                    CK(hipDeviceSynchronize());
                    best = ms < best ? ms : best;
                }
                printf("waves/CU %2d, %2d spinning workgroups, %s sweep: %.3f ms\n", waves, busy_wgs,
                       phase == 1 ? "forward " : "backward", best);
            }
        }
    }
    return 0;
}
