// Internal definitions shared by the libhip_dsp translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "../../include/hip_dsp.h"

void hipdsp_set_error(const char *fmt, ...);

struct hipdsp_ctx {
    int device;
    hipStream_t stream;
    int max_segments;      // 0 = auto
    int n_cus;
    void *scratch;         // envelope state checkpoints, four-step FFT work area
    size_t scratch_bytes;
    hipEvent_t mid_event;  // optional: recorded between envelope fwd and bwd
    void *fft_tables[20];  // per log2(nfft): window | TWM | TWN (device), built on first use
    int force_generic_fft; // tests: use the generic radix-2 kernel for every nfft
    void *fft_tables2[20]; // same for the two-stage kernel: tw2 | twn | window
    int sos_waves_per_cu;  // experiments: resident waves per CU the IIR planner aims for
    int sos_waves_min;     // experiments: >= sos_waves_per_cu forces exactly that many waves per CU (sos.hip: plan_segments)
    int chain_pairs;       // most pairs of waves per CU the fused sweeps' planner uses (0 = 8; experiments)
    int sos_debug;         // measurements only (results wrong): ablation bits of the envelope's backward sweep
    int sos_single_wave_wg; // experiments (A/B): the envelope's backward sweep as single-wave workgroups, as in rounds 1-2
    int sos_prefetch;      // experiments: register prefetch of the next tile in the envelope sweeps
    int spec_no_half;      // experiments/tests: do not reuse the overlapped half frame
    int spec_fpw, spec_kernel;   // experiments (tools/), 0 = defaults
    int spec_debug;        // measurements only (results wrong): ablation bits of spec_pack_kernel (spec_pack.h)
    int chain_debug;       // experiments: ablation bits of the fused forward kernel
    int chain_split_frames; // hipdsp_chain_forward writes only the even frames (hipdsp_chain_backward the odd ones)
    int chain_reserve_cus; // CUs hipdsp_chain_forward leaves without a workgroup (room for a co-resident RCCL kernel)
    long long *sos_trace;  // diagnostics: device buffer (9 int64 per WAVE of the envelope's backward sweep = channels x segments
                           // rows, hipdsp_chain... sizes unknown to the library: the tool sizes it from the planned grid)
    long long sos_trace_rows;  // capacity of sos_trace in rows of 9 int64 (option "sos_trace_rows"; set it BEFORE "sos_trace")
    int sos_fair;          // rotating issue priorities in the single-wave sweeps (sos.hip: rotate_issue_priority)
    int sos_split;         // experiments (A/B): the envelope's backward sweep with compute and mover waves (envsplit.hip)
    int sos_no_pin;        // experiments: scalar table loads left to hipcc's just-in-time placement (A/B of CASC_PIN_GROUPS)
    struct hd_pool *pool;  // stream-ordered cache of freed device blocks (ctx.hip)
    // Device-side fault report: four ints in pinned host memory that kernels can write
    // (word 0 = fault code, 1..3 = detail).  A kernel whose bounded wait runs out stores here instead of
    // carrying on silently; hd_device_fault() turns a non-zero word into HIPDSP_ERR_HIP.
    volatile int *fault_host;
    int *fault_dev;        // the same words as the kernels address them
    int graphs_alive;      // hipGraphs captured on this context that have not been destroyed
    // one byte per (channel, time segment) of the last forward sweep: did its band-pass end non-finite? (sos_device.h:
    // FloodArgs).  1 MiB from the start, grown by hd_seg_flags() outside captures only.
    unsigned char *seg_flags;
    size_t seg_flags_cap;
    // The tile grid of the last forward sweep that parked envelope tile states in `scratch` (sos_device.h: GridShift,
    // hd_note_sweep): the backward sweep that consumes them (hipdsp_sosfilt_envelope phase 2) walks the same grid.
    long long sweep_lead, sweep_env0, sweep_frames, sweep_channels;
    int sweep_sections;
};

// the flags of a forward sweep of `units` (channel, segment) pairs
int hd_seg_flags(hipdsp_ctx *ctx, size_t units, unsigned char **out);

// fault codes a kernel may leave in hipdsp_ctx::fault_host[0]
#define HD_FAULT_CHAIN_HANDOVER 1   // chain_fwd_kernel: a wave waited in vain for its partner's LDS flag
#define HD_FAULT_SPLIT_HANDOVER 2   // env_bwd_split_kernel: the same between a compute wave and its mover

// Reports (and clears) a fault word left by a kernel of this context: HIPDSP_ERR_HIP with a message,
// HIPDSP_OK when there is none.  Called after every synchronisation of the context's stream and at
// the top of the calls that launch such kernels.
int hd_device_fault(hipdsp_ctx *ctx);

struct hipdsp_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};

#define HD_CHECK_HIP(expr)                                                        \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) {                                                   \
            hipdsp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                             __FILE__, __LINE__);                                 \
            return HIPDSP_ERR_HIP;                                                \
        }                                                                         \
    } while (0)

#define HD_REQUIRE(cond, ...)                                                     \
    do {                                                                          \
        if (!(cond)) {                                                            \
            hipdsp_set_error(__VA_ARGS__);                                        \
            return HIPDSP_ERR_INVALID;                                            \
        }                                                                         \
    } while (0)

static inline int hd_launch_status(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        hipdsp_set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return HIPDSP_ERR_HIP;
    }
    return HIPDSP_OK;
}

// Do two planar float arrays (channels rows of `pitch` floats, `frames` used) share memory?  The sweeps are cut into
// time segments that run concurrently and re-read their warm-up from the input: an output over the input is a race.
static inline bool hd_planar_overlap(const float *a, long long a_pitch, long long a_frames, const float *b, long long b_pitch,
                                     long long b_frames, long long channels)
{
    if (a == nullptr || b == nullptr || channels <= 0 || a_frames <= 0 || b_frames <= 0) return false;
    const char *a0 = (const char *)a, *a1 = (const char *)(a + (channels - 1) * a_pitch + a_frames);
    const char *b0 = (const char *)b, *b1 = (const char *)(b + (channels - 1) * b_pitch + b_frames);
    return a0 < b1 && b0 < a1;
}
#define HD_NO_OVERLAP(a, ap, af, b, bp, bf, channels, what)                                             \
    HD_REQUIRE(!hd_planar_overlap((a), (ap), (af), (b), (bp), (bf), (channels)), what " must not overlap")

int hipdsp_scratch(hipdsp_ctx *ctx, size_t bytes, void **out);
// the scratch as a forward sweep left it (tile states of the envelope): for the backward sweeps; HIPDSP_ERR_INVALID when
// anything has asked for the scratch since (hipdsp_scratch() forgets the sweep's note)
int hd_scratch_parked(hipdsp_ctx *ctx, size_t bytes, void **out);
// tw2 | tw3 | twn | window of the three-stage PSD kernel, device memory (spectrogram.hip): served for nfft 2048
// (radix 16 x 16 x 4), 1024 (8 x 8 x 8), 512 (8 x 8 x 4) and 256 (8 x 4 x 4) -- the sizes chain_fwd_kernel is built for
int hd_fft_tables(hipdsp_ctx *ctx, int nfft, const float **dev);
