// sos_device.h -- device helpers, launch argument blocks and small host helpers shared by the IIR sweeps (sos.hip) and
// the fused sweeps (chain.hip).  Everything sits in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "common.h"
#include "sos_plan.h"

struct hipdsp_sosplan {
    SosPlanDev *host;      // pinned
    SosPlanDev *dev;
    hipEvent_t uploaded;
    bool valid;
};

// Non-finite samples (NaN, Inf).  In the reference a band-pass that has seen one stays NaN to the end of the slab
// (scipy sosfilt: the state is NaN from then on), every spectrogram frame from there on is NaN, and the envelope of that
// channel is NaN everywhere (sosfiltfilt's backward pass starts from the forward pass's NaN end).  A sweep cut into
// time segments would recover in the next segment (its warm-up starts from zero state), so
//  * every unit (channel, segment) of a forward sweep leaves one byte in the context's `seg_flags`: is the band-pass
//    state it ended with non-finite? -- always written, never accumulated, no reset needed;
//  * flood_channel() (one block per channel: inside env_fix_kernel when there is one, nan_flood_kernel otherwise)
//    finds the first such segment and overwrites what the LATER segments wrote with NaN: the filtered trace and the
//    spectrogram frames (PSD and dB) that reach past that segment's end.  The segment itself is NaN from the bad
//    sample on by the arithmetic alone;
//  * the envelope: the state slot behind the last tile (`n_tiles`) holds the state the channel's forward sweep ended
//    with, and env_fix_kernel puts NaN there if ANY segment ended non-finite; the backward sweeps look at that slot
//    first and fill the channel with NaN instead of sweeping.
// (tests/test_gpu_parity.py::test_non_finite_*, against scipy.)
struct FloodArgs {
    const unsigned char *flags;  // [channels][n_seg]; NULL: nothing to do
    int n_seg;
    long long seg_len;           // input samples per segment
    float *y;                    // filtered trace, sample p of channel c at y[c * y_pitch + p - skip]; may be NULL
    long long y_pitch, T, skip;
    float *psd, *db;             // [channel][frame][F] with psd_pitch floats per channel; may be NULL
    long long psd_pitch, n_valid;
    int F, nfft, hop;
    int frame_off;               // frame k of psd / db is frame k + frame_off of the sweep's grid (GridShift)
};

// defined in sos.hip: hands the true states over between the time segments of a forward sweep (env_fix_kernel)
// (first_seg: the segment the envelope starts in, GridShift -- segments in front of it left no states)
int hd_launch_env_fix(hipdsp_ctx *ctx, const SosPlanDev *edev, int SE, double *ckpt, long long ckpt_pitch, long long channels,
                      int n_seg, long long seg_len, long long n_tiles, const FloodArgs *flood, int first_seg = 0);

// (not in the anonymous namespace: envsplit.hip's launcher takes it across translation units)
struct BwdArgs {
    const float *in;         // the trace the envelope is taken of (before rectification)
    float *out;
    const double *ckpt;
    long long in_pitch, out_pitch, ckpt_pitch;
    long long T, skip;
    long long n_tiles;       // ceil((T + edge) / TILE)
    long long seg_tiles, warm_tiles;
    int n_seg, edge, rectify, clamp;
    double gain;             // as in CkptArgs
    long long units;         // channels * n_seg (the grid is rounded up to whole workgroups)
    long long lead, env0;    // as in CkptArgs: the grid of the forward sweep that left the tile states (skip >= env0)
    long long *trace;        // option "sos_trace": 9 words per wave (start, end in 100 MHz ticks, HW_ID, 6 clock sums)
    long long trace_rows;    // rows of `trace` (option "sos_trace_rows"): waves beyond it do not report
    int debug;               // measurements only, results wrong (option "sos_debug"): 1 = every interior tile is stored into
                             // the channel's first tile (writes stay in L2), 2 = every prefetch reads the first tile
    int fair;                // rotate_issue_priority() per tile (option "sos_fair", default off: no gain measured)
};

namespace {


typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 4-byte aligned float4
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));   // 4-byte aligned float2

struct SeqArgs {
    const float *in;        // channel 0 of the input
    float *out;             // channel 0 of the output
    long long in_pitch, out_pitch;
    long long N;            // frames
    long long seg_len;      // multiple of TILE
    int n_seg;
    long long skip;         // nbefore: the first `skip` outputs are dropped
    // optional initial state of the cascade at sample 0 (scipy: sosfilt(..., zi = sosfilt_zi * x0)):
    // plan zi * zi_scale * zi_ref[channel * zi_ref_pitch]; NULL = zero state
    const float *zi_ref;
    long long zi_ref_pitch;
    double zi_scale;
    // the cascade filters gain * in (hipdsp_envelope_multi: the pi/2 of the rectified trace rides on the first
    // plan's numerator instead of on every sample); launch_scan turns 0 into 1
    double gain;
    long long units;        // channels * n_seg (the grid is rounded up to whole workgroups)
    unsigned char *flags;   // one byte per unit: did it end with a non-finite state? (FloodArgs)
};

__device__ __forceinline__ long long opaque_zero()
{
    int z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    return (long long)z;
}

// Slot (16 bytes) of quarter-row q of row `row` in the 64 x 8 tile image.  The XOR term must make three
// access patterns conflict-free at once (MI355X_MICROARCH.md, LDS): the coalesced view (a 16-byte
// access per lane, 8 consecutive lanes in one row: any XOR does), the row-per-lane ds_read_b128 (16-lane
// groups {0-3,12-15,20-27}..., banks modulo 64 dwords: rows of equal parity must differ in the term,
// which (row & 7) ^ (row >> 3 & 1) does over any 16 rows distinct modulo 16) and the row-per-lane
// ds_write_b128 (8 consecutive lanes, banks modulo 32 dwords: 8 consecutive rows must differ -- the
// round-1 term (row >> 1) & 7 repeated in pairs there, a 2-way conflict on every store of phase 3:
// 11.7 % of the fused kernel's LDS cycles).
__device__ __forceinline__ int lds_slot(int row, int q) { return row * 8 + (q ^ ((row & 7) ^ ((row >> 3) & 1))); }
__device__ __forceinline__ int lds_float_index(int s) { return lds_slot(s >> 5, (s & 31) >> 2) * 4 + (s & 3); }

// samples p..p+3 of a row of `n` frames, zeros past the end
__device__ __forceinline__ float4 load_four(const float *in, long long p, long long n)
{
    if (p + 4 <= n) {
        const f4u t = *reinterpret_cast<const f4u *>(in + p);
        return make_float4(t.x, t.y, t.z, t.w);
    }
    float4 v;
    v.x = p < n ? in[p] : 0.f;
    v.y = p + 1 < n ? in[p + 1] : 0.f;
    v.z = p + 2 < n ? in[p + 2] : 0.f;
    v.w = p + 3 < n ? in[p + 3] : 0.f;
    return v;
}

// the same for a row whose samples in front of `first` do not exist either (zeros there too)
__device__ __forceinline__ float4 load_four_from(const float *in, long long p, long long first, long long n)
{
    if (p >= first) return load_four(in, p, n);
    float4 v;
    v.x = 0.f;
    v.y = (p + 1 >= first && p + 1 < n) ? in[p + 1] : 0.f;
    v.z = (p + 2 >= first && p + 2 < n) ? in[p + 2] : 0.f;
    v.w = (p + 3 >= first && p + 3 < n) ? in[p + 3] : 0.f;
    return v;
}

// store samples p..p+3 restricted to [lo, hi) at out[p - shift]
__device__ __forceinline__ void store_four(float *out, long long p, float4 v, long long lo, long long hi,
                                           long long shift)
{
    if (p >= lo && p + 4 <= hi) {
        f4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
        *reinterpret_cast<f4u *>(out + (p - shift)) = t;
    } else {
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (p + k >= lo && p + k < hi) out[p + k - shift] = e[k];
    }
}

// The plan tables (coefficients, G, M: up to ~3 KB) are wave-uniform and must come through
// scalar loads (s_load -> SGPR operands of v_fma_f64).  Hoisting them out of the tile loop would
// need ~500 SGPRs and spill through v_writelane; a "memory" clobber would demote them to
// per-lane vector loads.  So every use goes through PLAN_OF(): the same pointer plus an opaque,
// always-zero scalar that the compiler must assume changes each time, which pins the s_load
// next to its use.
#define PLAN_OF(ptr) (reinterpret_cast<const SosPlanDev *>(reinterpret_cast<const char *>(ptr) + opaque_zero()))

// The value of the lane below (lane 0: zero): a full-wave shift by one as a DPP move inside the VALU
// (wave_shr:1, bound_ctrl) instead of a trip through the LDS crossbar (ds_bpermute) plus a select.
__device__ __forceinline__ double casc_wave_shr1(double x)
{
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x138, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// Shift by 32 lanes with zero fill: v_permlane32_swap (gfx950) exchanges the upper half of its first
// operand with the lower half of its second; first operand 0, second x -> (0..0, x[0..31]).
__device__ __forceinline__ double casc_wave_shr32(double x)
{
    const long long b = __builtin_bit_cast(long long, x);
    const auto lo = __builtin_amdgcn_permlane32_swap(0, (int)b, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(0, (int)(b >> 32), false, false);
    return __builtin_bit_cast(double, ((long long)hi[0] << 32) | (unsigned int)lo[0]);
}

__device__ __forceinline__ double casc_wave_shl1(double x)
{
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x130, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double casc_wave_shl32(double x)
{
    const long long b = __builtin_bit_cast(long long, x);
    const auto lo = __builtin_amdgcn_permlane32_swap((int)b, 0, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((int)(b >> 32), 0, false, false);
    return __builtin_bit_cast(double, ((long long)hi[1] << 32) | (unsigned int)lo[1]);
}

// a wave-local fence: LDS operations of one wave execute in order, no workgroup barrier is needed between the
// phases of a tile that a single wave walks
#define WAVE_SYNC()                                          \
    do {                                                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                     \
    } while (0)

constexpr int WPB = 4;     // waves per workgroup of the single-wave-per-unit sweeps: one per SIMD of a CU (see env_bwd_kernel)

struct CkptArgs {
    const float *in;
    float *yf;
    double *ckpt;
    long long in_pitch, yf_pitch, ckpt_pitch;
    long long T, seg_len;
    int n_seg, edge, rectify;
    long long units;        // channels * n_seg (the grid is rounded up to whole workgroups; the fused sweep: ChainArgs::units)
    double gain;            // the envelope filters gain * |y|: folded into its cascade (CASC_GAIN), never into the samples
    unsigned char *flags;   // one byte per unit: did its band-pass end with a non-finite state? (FloodArgs; SF > 0)
    // A sweep whose tile grid does not start at the first sample (GridShift below): the kernel walks sample
    // coordinates p' = sample + lead of a trace of T = frames + lead samples whose first `lead` samples do not exist
    // (`in` and `yf` are shifted by -lead on the host; p' < lead is never dereferenced: zero on the way in, masked
    // on the way out -- zero input from zero state is the band-pass's true state at the first real sample).
    long long lead;         // 0 <= lead < TILE: only tile 0 holds such samples
    long long env0;         // p' of the first sample the envelope is taken of (0: the trace's first, the classic case)
};

// Where a forward sweep puts its tile grid.  The fused sweep's spectrogram frames are register windows of its tiles,
// so frame k -- samples spec_first + k hop ... of the filtered buffer -- must start at a multiple of hop in the sweep's
// coordinates p' = sample + lead; the envelope may start at sample env_first (BufferedData.align_buffer trims its
// pre-roll after a scroll, buffereddata.py:75-88): its first tile then holds scipy's left odd extension and, in front
// of that, the extension's first value, for which zi * value is the cascade's steady state -- which needs the whole
// extension and its sources (edge samples either side of p' = env0) inside ONE tile.  Both are a matter of choosing
// lead (< TILE); `step` is what lead may move by (the hop with a spectrogram in the launch, 128 samples without).
struct GridShift {
    long long lead, env0;
    int frame_off;          // frame k of the output is frame k + frame_off of the sweep's grid
};
inline GridShift hd_grid_shift(long long spec_first, long long env_first, int hop, int edge, bool with_env)
{
    GridShift g;
    const long long step = hop > 0 ? hop : 128;
    g.lead = hop > 0 ? (step - spec_first % step) % step : 0;
    if (with_env && env_first + g.lead > 0) {
        for (int tries = 0; tries < 4; tries++) {
            const long long r = (env_first + g.lead) % TILE;
            if (r >= edge && r + edge < TILE) break;
            g.lead += step;
        }
    }
    g.env0 = with_env ? env_first + g.lead : 0;
    g.frame_off = hop > 0 ? (int)((spec_first + g.lead) / step) : 0;
    return g;
}

// 16-byte global load the compiler does not track: the caller counts vmcnt by hand, so that the
// wait for a prefetched tile does not also wait for the stores issued after it.
typedef float v4f __attribute__((ext_vector_type(4)));
// The lane's eight float4 of a tile in LDS (row 8 k + lane / 8, quarter lane % 8: 1 KB of the trace per k), ALL read
// before anything is done with them: left to itself hipcc takes them one at a time through a single register quad --
// ds_read, s_waitcnt lgkmcnt(0), global_store, eight times over: eight LDS latencies in a row per tile and wave
// (round 5, the sweeps' store sequences in tools/isa_block.py).  The empty asm needs all of them in registers at once.
__device__ __forceinline__ void tile_rows_from_lds(const float4 *lds, int lane, v4f (&v)[8])
{
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float4 t = lds[lds_slot(8 * k + (lane >> 3), lane & 7)];
        v[k] = (v4f){t.x, t.y, t.z, t.w};
    }
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
}
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v4f asm_load16(const void *p)
{
    v4f r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ float asm_load4(const float *p)
{
    float r;
    asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}

// Fair shares of a SIMD for persistent waves that do not talk to each other.  The issue arbiters serve the
// OLDEST wave first: of four waves of one SIMD that walk equal segments, the one in slot 0 gets whatever it
// asks for and ends after 55 % of the launch, the one in slot 3 after 95 % (tools/sweep_trace.py), and the
// tail with a quarter of the waves cannot keep the HBM pipes full.  Called once per tile, this gives the
// four slots four DIFFERENT priorities that rotate with the shader clock (the same clock for all waves of
// the SIMD, so the priorities stay distinct): every wave spends a quarter of the time at each level.
__device__ __forceinline__ void rotate_issue_priority(unsigned slot)
{
    const unsigned p = (slot + (unsigned)(__builtin_readcyclecounter() >> 14)) & 3u;
    if (p == 0) __builtin_amdgcn_s_setprio(0);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}
// max(x, 0) as a single instruction (result as fmaxf(x, 0.f))
__device__ __forceinline__ float max_zero(float x)
{
    float r;
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ void set_issue_priority(int p)     // (s_setprio takes an immediate)
{
    if (p == 0) __builtin_amdgcn_s_setprio(0);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}
__device__ __forceinline__ unsigned wave_slot_of_simd()
{
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(hw));
    return hw;
}

// a forward sweep has parked the envelope's tile states in the context scratch: remember its grid for phase 2
inline void hd_note_sweep(hipdsp_ctx *ctx, long long lead, long long env0, long long frames, long long channels, int SE)
{
    ctx->sweep_lead = lead; ctx->sweep_env0 = env0; ctx->sweep_frames = frames; ctx->sweep_channels = channels;
    ctx->sweep_sections = SE;
}

// The first tile of an envelope that starts inside it, at sample q of the tile (see GridShift): the tile holds the
// rectified trace r; writes scipy's left odd extension ext[i] = 2 r(q) - r(q + edge - i), i < edge
// (scipy/signal/_arraytools.py:99-107) into samples q - edge .. q - 1 and ext[0] into everything in front of them,
// and returns ext[0] (wave-uniform).  Requires edge <= q and q + edge < TILE.  float32 like the right extension;
// the forward and the backward sweep call this on the same float32 samples, so both see the same tile.
__device__ __forceinline__ float env_left_fill(float *ldsf, int lane, int q, int edge)
{
    const float r0 = ldsf[lds_float_index(q)];
    const float e0 = 2.f * r0 - ldsf[lds_float_index(q + edge)];
    float ev = 0.f;
    if (lane < edge) ev = 2.f * r0 - ldsf[lds_float_index(q + edge - lane)];
    WAVE_SYNC();
    if (lane < edge) ldsf[lds_float_index(q - edge + lane)] = ev;
    for (int s2 = lane; s2 < q - edge; s2 += 64) ldsf[lds_float_index(s2)] = e0;
    WAVE_SYNC();
    return e0;
}


// wave-uniform: is any of the D state values NaN or infinite?
template <int D>
__device__ __forceinline__ bool state_not_finite(const double (&c)[D])
{
    bool bad = false;
#pragma unroll
    for (int r = 0; r < D; r++) bad = bad || !(fabs(c[r]) <= 1.7976931348623157e308);
    return bad;
}

// see FloodArgs; the whole block takes part (barriers inside), any block size
__device__ void flood_channel(const FloodArgs &f, long long ch)
{
    __shared__ int first_bad;
    if (f.flags == nullptr || f.n_seg <= 1) return;
    if (threadIdx.x == 0) first_bad = 0x7fffffff;
    __syncthreads();
    const unsigned char *fl = f.flags + ch * f.n_seg;
    for (int s2 = threadIdx.x; s2 < f.n_seg - 1; s2 += blockDim.x)
        if (fl[s2]) atomicMin(&first_bad, s2);
    __syncthreads();
    const int s0 = first_bad;
    if (s0 == 0x7fffffff) return;
    const float nan_ = __builtin_nanf("");
    const long long n1 = (long long)(s0 + 1) * f.seg_len;           // first sample of the first segment to overwrite
    if (f.y != nullptr) {
        float *yc = f.y + ch * f.y_pitch;
        for (long long p = (n1 > f.skip ? n1 : f.skip) + threadIdx.x; p < f.T; p += blockDim.x) yc[p - f.skip] = nan_;
    }
    if (f.psd != nullptr || f.db != nullptr) {
        // frames k with k hop + nfft > n1 (those that only touch the end of segment s0 are NaN already)
        long long k0 = (n1 >= f.nfft ? (n1 - f.nfft) / f.hop + 1 : 0) - f.frame_off;
        if (k0 < 0) k0 = 0;
        for (long long i = k0 * f.F + threadIdx.x; i < f.n_valid * f.F; i += blockDim.x) {
            if (f.psd != nullptr) f.psd[ch * f.psd_pitch + i] = nan_;
            if (f.db != nullptr) f.db[ch * f.psd_pitch + i] = nan_;
        }
    }
}

__global__ __launch_bounds__(256) void nan_flood_kernel(FloodArgs f) { flood_channel(f, blockIdx.x); }

__global__ void zero_rows_kernel(float *__restrict__ y, long long y_pitch, long long n, float value)
{
    long long ch = blockIdx.y;
    float *yo = y + ch * y_pitch;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        yo[i] = value;
}

// a forward sweep without an envelope behind it (no env_fix_kernel to ride on): the flood as a launch of its own
int launch_flood(hipdsp_ctx *ctx, const FloodArgs &f, long long channels)
{
    if (f.flags == nullptr || f.n_seg <= 1 || channels == 0) return HIPDSP_OK;
    hipLaunchKernelGGL(nan_flood_kernel, dim3((unsigned)channels), dim3(256), 0, ctx->stream, f);
    return hd_launch_status("nan_flood_kernel");
}

// the single-wave sweeps: up to "sos_waves_per_cu" (16) waves per CU, four SIMDs per CU
// ("sos_waves_min" = w, experiments: force w waves per CU by making every level below it cost the same)
// (`w_cap`: what the kernel's registers allow -- the register hand-over of the backward sweep holds two waves a SIMD)
// (`model`: which sweep -- 4 the backward sweep, -1 sos_scan_kernel, -2 sos_ckpt_kernel: sos_plan.hip, tile_step_cost)
void plan_segments(const hipdsp_ctx *ctx, long long N, long long channels, long long warm,
                   long long *seg_len, int *n_seg, int model, int w_cap = 16)
{
    int w_max = ctx->sos_waves_per_cu > 0 ? ctx->sos_waves_per_cu : 16;
    if (w_max > w_cap) w_max = w_cap;
    hd_plan_segments_occ(ctx->n_cus, w_max, ctx->sos_waves_min >= w_max ? 0 : model, ctx->max_segments, N, channels, warm, seg_len, n_seg);
}

// the fused sweeps: up to 8 pairs of waves per workgroup = CU ("chain_pairs": fewer, experiments)
constexpr int CHAIN_P = 8;
void plan_segments_chain(const hipdsp_ctx *ctx, long long N, long long channels, long long warm,
                         long long *seg_len, int *n_seg)
{
    int cus = ctx->n_cus - ctx->chain_reserve_cus;
    if (cus < 1) cus = 1;                                  // (options set in an order that leaves none: ADVICE round 2)
    hd_plan_segments_occ(cus, ctx->chain_pairs > 0 ? ctx->chain_pairs : CHAIN_P, 0, ctx->max_segments, N, channels, warm, seg_len,
                      n_seg);
}

}  // namespace

// defined in envsplit.hip: the backward sweep with compute and mover waves (context option "sos_split")
int hd_launch_env_bwd_split(hipdsp_ctx *ctx, const SosPlanDev *edev, int SE, const BwdArgs &b);
