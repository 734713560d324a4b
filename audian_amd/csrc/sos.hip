// Block-parallel biquad cascade for gfx950: BufferedFilter.process (sosfilt) and
// BufferedEnvelope.process (rectify + sosfiltfilt) of bendalab/audian
// (src/audian/bufferedfilter.py:31-36, src/audian/bufferedenvelope.py:34-41).
//
// Parallelisation (the recurrence is serial in time, channels alone cannot fill
// 256 CUs):
//   * a channel is cut into `n_seg` time segments, one 64-lane wave per
//     (channel, segment); a segment that does not start at sample 0 first runs
//     `warm` samples ahead of its range from zero state -- `warm` is chosen on the
//     host such that ||A^warm||_inf < 2^-60 for the cascade's state-transition
//     matrix A, i.e. the forgotten history is below float64 rounding;
//   * inside a segment the wave walks tiles of 64 x L samples; each LANE owns L
//     consecutive samples of the tile, so the wave covers it exactly:
//       phase 1  f_i   = sum_j A^(L-1-j) B x_j        (zero-state end state, dot products)
//       scan     P_i   = A^L P_(i-1) + f_i            (Kogge-Stone over lanes, carry folded
//                                                      into lane 0)
//       phase 3  lane i re-runs the DF-II-transposed cascade over its L samples from
//                the exact state P_(i-1) and emits the outputs.
//   * tiles go HBM -> LDS -> registers with 16-byte accesses; the LDS image is XOR
//     swizzled so the row-per-lane and the coalesced views are both conflict-free.
// Coefficients and state are float64, HBM traffic is float32.
#include "common.h"
#include "sos_plan.h"
#include "fft_device.h"
#include <cmath>
#include <vector>

namespace {

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 4-byte aligned float4

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

// a wave-local fence: LDS operations of one wave execute in order, no workgroup barrier is needed between the
// phases of a tile that a single wave walks
#define WAVE_SYNC()                                          \
    do {                                                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                     \
    } while (0)

constexpr int WPB = 4;     // waves per workgroup of the single-wave-per-unit sweeps: one per SIMD of a CU (see env_bwd_kernel)

// ---- sosfilt: BufferedFilter.process ------------------------------------------------------
template <int S>
__global__ __launch_bounds__(64 * WPB) void sos_scan_kernel(const SosPlanDev *__restrict__ P0, SeqArgs a)
{
    constexpr int D = 2 * S;
    __shared__ float4 lds_all[WPB][64 * 8];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 *lds = lds_all[wave];
    const int lane = threadIdx.x & 63;
    const long long unit = (long long)blockIdx.x * WPB + wave;
    if (unit >= a.units) return;                    // (no workgroup barrier anywhere: a wave may leave)
    const int seg = (int)(unit % a.n_seg);
    const long long ch = unit / a.n_seg;
    const float *in = a.in + ch * a.in_pitch;
    float *out = a.out + ch * a.out_pitch;

    const long long lo = (long long)seg * a.seg_len;
    long long hi = lo + a.seg_len;
    if (hi > a.N) hi = a.N;
    long long start = lo - P0->warm;
    if (start < 0) start = 0;                   // zero state at sample 0 is the true state
    const long long olo = lo > a.skip ? lo : a.skip;

    double carry[D];
#pragma unroll
    for (int r = 0; r < D; r++) carry[r] = 0.0;
    if (a.zi_ref != nullptr && start == 0) {
        // the true state at sample 0 (every other segment forgets it within its warm-up)
        const SosPlanDev *P = PLAN_OF(P0);
        const double x0 = a.zi_scale * (double)a.zi_ref[ch * a.zi_ref_pitch];
#pragma unroll
        for (int r = 0; r < D; r++) carry[r] = P->zi[r] * x0;
    }

    for (long long tile = start; tile < hi; tile += TILE) {
        // ---- HBM -> LDS (coalesced 16 B per lane), LDS -> registers (row per lane)
        // (A register prefetch of the next tile with a hand-counted vmcnt was measured and
        // bought nothing: the kernel already runs at the device's read+write copy rate.)
#pragma unroll
        for (int k = 0; k < 8; k++)
            lds[lds_slot(8 * k + (lane >> 3), lane & 7)] = load_four(in, tile + 256 * k + 4 * lane, a.N);
        WAVE_SYNC();

#define CASC_S S
#define CASC_PLAN() PLAN_OF(P0)
#define CASC_CARRY carry
#define CASC_IN(v) (v)
#define CASC_GAIN a.gain
#include "sos_cascade.inc"
#undef CASC_GAIN
#undef CASC_S
#undef CASC_PLAN
#undef CASC_CARRY
#undef CASC_IN
        WAVE_SYNC();
        if (tile + TILE > olo) {      // warm-up tiles produce no output
#pragma unroll
            for (int k = 0; k < 8; k++)
                store_four(out, tile + 256 * k + 4 * lane, lds[lds_slot(8 * k + (lane >> 3), lane & 7)], olo, hi,
                           a.skip);
        }
        WAVE_SYNC();
    }
}

// ---- envelope without a forward scratch: state checkpoints + recomputation ----------------
// sosfiltfilt's forward output is only ever consumed by its own backward pass.  Instead of
// writing it to HBM and reading it back (8 B per sample), the forward sweep keeps only the
// cascade state that ENTERS each 2048-sample tile (2*SE doubles per tile, < 0.01 B/sample),
// which needs phase 1 and the scan but no phase 3; the backward sweep walks the tiles from the
// end, re-runs the forward cascade on a tile from its checkpoint (bit-identical to what the
// forward sweep would have emitted), and filters the result backwards while it is in LDS.
//   forward sweep   sos_ckpt_kernel<SF, SE>: SF > 0 also runs the band-pass and writes the
//                   filtered trace (the batch chain, 8 B/sample); SF == 0 reads the trace to
//                   rectify as it is (BufferedEnvelope.process alone, 4 B/sample)
//   backward sweep  env_bwd_kernel<SE>: 8 B/sample.
// Tiles are aligned in SAMPLE coordinates p in [0, T + edge) for both sweeps (right odd
// extension at p >= T; the left one is `edge` serial steps before tile 0).  The backward sweep
// starts in the middle of the top tile (p = T+edge-1) from zi * w[T+edge-1]: the rest of that
// tile is filled with the same value, for which zi * value is the cascade's steady state.
struct CkptArgs {
    const float *in;
    float *yf;
    double *ckpt;
    long long in_pitch, yf_pitch, ckpt_pitch;
    long long T, seg_len;
    int n_seg, edge, rectify;
    long long units;        // channels * n_seg (the grid is rounded up to whole workgroups; the fused sweep: ChainArgs::units)
    double gain;            // the envelope filters gain * |y|: folded into its cascade (CASC_GAIN), never into the samples
};

// 16-byte global load the compiler does not track: the caller counts vmcnt by hand, so that the
// wait for a prefetched tile does not also wait for the stores issued after it.
typedef float v4f __attribute__((ext_vector_type(4)));
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

template <int SF, int SE, bool PREFETCH>
__global__ __launch_bounds__(64 * WPB) void sos_ckpt_kernel(const SosPlanDev *__restrict__ PF0,
                                                      const SosPlanDev *__restrict__ PE0, CkptArgs a)
{
    constexpr int DF = SF > 0 ? 2 * SF : 1, DE = 2 * SE;
    __shared__ float4 lds_all[WPB][64 * 8];
    __shared__ float rprev_all[WPB][64];   // rectified samples of the previous tile's last two rows
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 *lds = lds_all[wave];
    float *rprev = rprev_all[wave];
    float *ldsf = reinterpret_cast<float *>(lds);
    const int lane = threadIdx.x & 63;
    const long long unit = (long long)blockIdx.x * WPB + wave;
    if (unit >= a.units) return;                    // (no workgroup barrier anywhere: a wave may leave)
    const int seg = (int)(unit % a.n_seg);
    const long long ch = unit / a.n_seg;
    const float *in = a.in + ch * a.in_pitch;
    float *yf = SF > 0 ? a.yf + ch * a.yf_pitch : nullptr;
    double *ckpt = a.ckpt + ch * a.ckpt_pitch;
    const long long T = a.T;
    const int edge = a.edge;

    const long long lo = (long long)seg * a.seg_len;
    long long hi = lo + a.seg_len;
    const bool last_seg = hi >= T;
    if (hi > T) hi = T;
    // The envelope cascade has NO warm-up: a segment behind the first starts it from zero state at its own first
    // tile, its tile states are therefore the ZERO-STATE ones, and the state it ends with goes where the next
    // segment's first (always zero) tile state would go; env_fix_kernel then hands the true states over from
    // segment to segment and corrects the tile states (exact) -- only the band-pass still warms up.
    const long long env_start = lo;
    const bool env_true = seg == 0;
    long long start = env_start;
    if (SF > 0) start -= PF0->warm;
    if (start < 0) start = 0;               // zero state at sample 0 is the filter's true state
    const long long loop_end = last_seg ? T + edge : hi;   // the right extension may need a tile more

    double cf_[DF], ce_[DE];
#pragma unroll
    for (int r = 0; r < DF; r++) cf_[r] = 0.0;
#pragma unroll
    for (int r = 0; r < DE; r++) ce_[r] = 0.0;
    rprev[lane] = 0.f;

    // Prefetch (see env_bwd_kernel for the rules): the next tile is requested as soon as this one
    // is in LDS; with a band-pass in front (SF > 0) the wait sits right behind the 8 vector stores
    // of the filtered tile (`vmcnt(7)`), otherwise at the end of the iteration.  The fetch is
    // unconditional (a tile that cannot be prefetched fetches the highest full tile instead).
    v4f nx[8];
    bool pre = false;
    const long long top_full = (T / TILE - 1) * TILE;            // host guarantees >= 0
    auto fetch = [&](long long t0) {
#pragma unroll
        for (int k = 0; k < 8; k++) nx[k] = asm_load16(in + t0 + 256 * k + 4 * lane);
    };
    if (PREFETCH) {
        pre = start < loop_end && start + TILE <= T;
        fetch(pre ? start : top_full);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    for (long long tile = start; tile < loop_end; tile += TILE) {
        if (PREFETCH && pre) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                asm volatile("" : "+v"(nx[k]));
                lds[lds_slot(8 * k + (lane >> 3), lane & 7)] = make_float4(nx[k].x, nx[k].y, nx[k].z, nx[k].w);
            }
        } else if (PREFETCH) {
            // a tile that reaches past T: untracked loads from clamped addresses, zeros past T
#pragma unroll 1
            for (int k = 0; k < 8; k++) {
                const long long p = tile + 256 * k + 4 * lane;
                v4f t;
                t.x = asm_load4(in + (p < T ? p : T - 1));
                t.y = asm_load4(in + (p + 1 < T ? p + 1 : T - 1));
                t.z = asm_load4(in + (p + 2 < T ? p + 2 : T - 1));
                t.w = asm_load4(in + (p + 3 < T ? p + 3 : T - 1));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(t));
                lds[lds_slot(8 * k + (lane >> 3), lane & 7)] =
                    make_float4(p < T ? t.x : 0.f, p + 1 < T ? t.y : 0.f, p + 2 < T ? t.z : 0.f, p + 3 < T ? t.w : 0.f);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++)
                lds[lds_slot(8 * k + (lane >> 3), lane & 7)] = load_four(in, tile + 256 * k + 4 * lane, T);
        }
        WAVE_SYNC();
        if (PREFETCH) {
            const long long next = tile + TILE;
            pre = next < loop_end && next + TILE <= T;
            fetch(pre ? next : top_full);
        }
        if constexpr (SF > 0) {
#define CASC_S SF
#define CASC_PLAN() PLAN_OF(PF0)
#define CASC_CARRY cf_
#define CASC_IN(v) (v)
#include "sos_cascade.inc"
#undef CASC_S
#undef CASC_PLAN
#undef CASC_CARRY
#undef CASC_IN
            WAVE_SYNC();
            if (tile + TILE > lo && tile < hi) {
                if (tile >= lo && tile + TILE <= hi) {
                    // interior tile: exactly 8 vector stores, then the counted wait
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const float4 v = lds[lds_slot(8 * k + (lane >> 3), lane & 7)];
                        f4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
                        *reinterpret_cast<f4u *>(yf + tile + 256 * k + 4 * lane) = t;
                    }
                    if (PREFETCH) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                } else {
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        store_four(yf, tile + 256 * k + 4 * lane, lds[lds_slot(8 * k + (lane >> 3), lane & 7)], lo, hi, 0);
                    if (PREFETCH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            } else if (PREFETCH) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // warm-up tile: no stores to count
            }
        }
        if (tile < env_start) {                                  // band-pass warm-up only
            if (PREFETCH && SF == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            WAVE_SYNC();
            continue;
        }
        // ---- envelope input in place: r = |y| (exact; the gain rides on the cascade, CASC_GAIN), then the odd
        // extension past T
        if (a.rectify) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                float4 v = lds[lds_slot(lane, q)];
                v = make_float4(fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w));
                lds[lds_slot(lane, q)] = v;
            }
        }
        WAVE_SYNC();
        auto rval = [&](long long j) -> float {          // r(j) for j in this tile or the row before it
            return j >= tile ? ldsf[lds_float_index((int)(j - tile))] : rprev[64 - (int)(tile - j)];
        };
        if (tile + TILE > T) {
            // ext[T + i] = 2 r(T-1) - r(T-2-i), i < edge (scipy odd_ext); zeros beyond
            float pv = 0.f;
            long long pj = -1;
            if (lane < edge) {
                pj = T + lane;
                if (pj >= tile && pj < tile + TILE) pv = 2.f * rval(T - 1) - rval(T - 2 - lane);
            }
            WAVE_SYNC();
            if (lane < edge && pj >= tile && pj < tile + TILE) ldsf[lds_float_index((int)(pj - tile))] = pv;
            WAVE_SYNC();
        }
        if (env_true && tile == 0) {
            // left odd extension: ext[i] = 2 r(0) - r(edge - i), i < edge, from zi * ext[0];
            // wave-uniform serial steps
            const SosPlanDev *P = PLAN_OF(PE0);
            const double r0 = (double)ldsf[lds_float_index(0)];
            const double x0 = a.gain * (2.0 * r0 - (double)ldsf[lds_float_index(edge)]);
#pragma unroll
            for (int r = 0; r < DE; r++) ce_[r] = P->zi[r] * x0;
            for (int i = 0; i < edge; i++) {
                double cur = a.gain * (2.0 * r0 - (double)ldsf[lds_float_index(edge - i)]);
#pragma unroll
                for (int s2 = 0; s2 < SE; s2++) {
                    const double y = fma(P->coef[s2][0], cur, ce_[2 * s2]);
                    ce_[2 * s2] = fma(-P->coef[s2][3], y, fma(P->coef[s2][1], cur, ce_[2 * s2 + 1]));
                    ce_[2 * s2 + 1] = fma(-P->coef[s2][4], y, P->coef[s2][2] * cur);
                    cur = y;
                }
            }
        }
        // (the slot of a later segment's first tile receives the end state of the segment before it, see below)
        if ((tile > lo || env_true) && lane == 0) {
#pragma unroll
            for (int r = 0; r < DE; r++) ckpt[(tile / TILE) * DE + r] = ce_[r];
        }
        const bool last_tile = tile + TILE >= loop_end;
        if (last_tile && last_seg) {
            if (PREFETCH && SF == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            break;
        }
        // keep the last two rows for an extension that reaches back over the tile border
        {
            const float4 keep0 = lds[lds_slot(62 + ((lane >> 3) & 1), lane & 7)];
            WAVE_SYNC();
            if (lane < 16) {
                rprev[4 * lane] = keep0.x; rprev[4 * lane + 1] = keep0.y;
                rprev[4 * lane + 2] = keep0.z; rprev[4 * lane + 3] = keep0.w;
            }
        }
        // ---- envelope forward: phase 1 and the scan only, the state moves on to the next tile
#define CASC_S SE
#define CASC_PLAN() PLAN_OF(PE0)
#define CASC_CARRY ce_
#define CASC_IN(v) (v)
#define CASC_GAIN a.gain
#define CASC_NO_OUTPUT
#include "sos_cascade.inc"
#undef CASC_NO_OUTPUT
#undef CASC_GAIN
#undef CASC_S
#undef CASC_PLAN
#undef CASC_CARRY
#undef CASC_IN
        if (PREFETCH && SF == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (last_tile) {
            // the (zero-state, for seg > 0) state this segment ends with: into the slot of the next segment's first tile
            if (lane == 0) {
#pragma unroll
                for (int r = 0; r < DE; r++) ckpt[(tile / TILE + 1) * DE + r] = ce_[r];
            }
            break;
        }
        WAVE_SYNC();
    }
}

// Exact state hand-over between the time segments of the envelope's forward sweep.  After the sweep
// the slot of segment k's first tile (k >= 1) holds e_(k-1), the state segment k-1 ended with having started from
// zero (segment 0: from the true initial state), and the other slots hold zero-state tile states.  With
// P = A^segment the true state entering segment k is S_k = e_(k-1) + P S_(k-1) = sum_j P^j e_(k-1-j), cut off
// where ||P^j|| < 2^-60 (the plan's warm-up length says where), and the true state entering tile m of segment k is
// its zero-state one + A^(TILE m) S_k -- added for the tiles of the warm-up length only, beyond which it is below
// float64 rounding.  One block per channel; segments are taken from the last to the first in chunks of the block
// size so that every e_k is read before it is overwritten with S_(k+1).
template <int SE>
__global__ __launch_bounds__(256) void env_fix_kernel(const SosPlanDev *__restrict__ P0, double *ckpt_all, long long ckpt_pitch,
                                                      int n_seg, long long seg_tiles, long long n_tiles)
{
    constexpr int D = 2 * SE;
    double *ckpt = ckpt_all + (long long)blockIdx.x * ckpt_pitch;
    double AT[D][D], P[D][D];
#pragma unroll
    for (int r = 0; r < D; r++)
#pragma unroll
        for (int c = 0; c < D; c++) { AT[r][c] = P0->AT[r * D + c]; P[r][c] = r == c ? 1.0 : 0.0; }
    {   // P = AT^seg_tiles by binary exponentiation
        double Q[D][D];
#pragma unroll
        for (int r = 0; r < D; r++)
#pragma unroll
            for (int c = 0; c < D; c++) Q[r][c] = AT[r][c];
        for (long long e = seg_tiles; e > 0; e >>= 1) {
            double t[D][D];
            if (e & 1) {
#pragma unroll
                for (int r = 0; r < D; r++)
#pragma unroll
                    for (int c = 0; c < D; c++) {
                        double acc = 0.0;
#pragma unroll
                        for (int k = 0; k < D; k++) acc = fma(P[r][k], Q[k][c], acc);
                        t[r][c] = acc;
                    }
#pragma unroll
                for (int r = 0; r < D; r++)
#pragma unroll
                    for (int c = 0; c < D; c++) P[r][c] = t[r][c];
            }
#pragma unroll
            for (int r = 0; r < D; r++)
#pragma unroll
                for (int c = 0; c < D; c++) {
                    double acc = 0.0;
#pragma unroll
                    for (int k = 0; k < D; k++) acc = fma(Q[r][k], Q[k][c], acc);
                    t[r][c] = acc;
                }
#pragma unroll
            for (int r = 0; r < D; r++)
#pragma unroll
                for (int c = 0; c < D; c++) Q[r][c] = t[r][c];
        }
    }
    const long long warm_tiles = P0->warm / TILE;                  // ||A^(TILE warm_tiles)|| < 2^-60
    const long long terms = (warm_tiles + seg_tiles - 1) / seg_tiles + 1;
    for (long long k_hi = n_seg - 1; k_hi >= 1; k_hi -= blockDim.x) {
        const long long k = k_hi - threadIdx.x;
        double S[D];
#pragma unroll
        for (int r = 0; r < D; r++) S[r] = 0.0;
        if (k >= 1) {
            // Horner from the oldest term: S = e_(k-J) ; S = P S + e_(k-J+1) ; ... ; + e_(k-1); e_j sits in slot (j+1) seg_tiles
            long long j = k - terms;
            if (j < 0) j = 0;
            for (; j < k; j++) {
                const double *e = ckpt + (j + 1) * seg_tiles * D;
                double t[D];
#pragma unroll
                for (int r = 0; r < D; r++) {
                    double acc = e[r];
#pragma unroll
                    for (int c = 0; c < D; c++) acc = fma(P[r][c], S[c], acc);
                    t[r] = acc;
                }
#pragma unroll
                for (int r = 0; r < D; r++) S[r] = t[r];
            }
        }
        __syncthreads();
        if (k >= 1) {
            const long long t0 = k * seg_tiles;
            long long cnt = (k == n_seg - 1) ? n_tiles - t0 : seg_tiles;     // tile slots of segment k
            if (cnt > warm_tiles + 1) cnt = warm_tiles + 1;
#pragma unroll
            for (int r = 0; r < D; r++) ckpt[t0 * D + r] = S[r];              // (its zero-state tile state is zero)
            for (long long m = 1; m < cnt; m++) {
                double t[D];
#pragma unroll
                for (int r = 0; r < D; r++) {
                    double acc = 0.0;
#pragma unroll
                    for (int c = 0; c < D; c++) acc = fma(AT[r][c], S[c], acc);
                    t[r] = acc;
                }
#pragma unroll
                for (int r = 0; r < D; r++) { S[r] = t[r]; ckpt[(t0 + m) * D + r] += t[r]; }
            }
        }
        __syncthreads();
    }
}

int launch_env_fix(hipdsp_ctx *ctx, const SosPlanDev *edev, int SE, double *ckpt, long long ckpt_pitch, long long channels,
                   int n_seg, long long seg_len, long long n_tiles)
{
    if (n_seg <= 1) return HIPDSP_OK;
    dim3 grid((unsigned)channels), block(256);
    const long long seg_tiles = seg_len / TILE;
    switch (SE) {
    case 1: hipLaunchKernelGGL((env_fix_kernel<1>), grid, block, 0, ctx->stream, edev, ckpt, ckpt_pitch, n_seg, seg_tiles, n_tiles); break;
    case 2: hipLaunchKernelGGL((env_fix_kernel<2>), grid, block, 0, ctx->stream, edev, ckpt, ckpt_pitch, n_seg, seg_tiles, n_tiles); break;
    case 3: hipLaunchKernelGGL((env_fix_kernel<3>), grid, block, 0, ctx->stream, edev, ckpt, ckpt_pitch, n_seg, seg_tiles, n_tiles); break;
    case 4: hipLaunchKernelGGL((env_fix_kernel<4>), grid, block, 0, ctx->stream, edev, ckpt, ckpt_pitch, n_seg, seg_tiles, n_tiles); break;
    default: return HIPDSP_OK;
    }
    return hd_launch_status("env_fix_kernel");
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
    long long *trace;        // option "sos_trace": 9 words per wave (start, end in 100 MHz ticks, HW_ID, 6 clock sums)
    long long trace_rows;    // rows of `trace` (option "sos_trace_rows"): waves beyond it do not report
    int debug;               // measurements only, results wrong (option "sos_debug"): 1 = every interior tile is stored into
                             // the channel's first tile (writes stay in L2), 2 = every prefetch reads the first tile
    int fair;                // rotate_issue_priority() per tile (option "sos_fair", default off: no gain measured)
};

// WPB waves per workgroup, each an independent (channel, segment) unit with an LDS tile of its own and no workgroup
// barrier anywhere.  With WPB = 4 the hardware puts the four waves of a workgroup on the four SIMDs of ONE CU, so every
// SIMD of the chip carries the same number of these persistent waves whatever else the dispatcher has seen before;
// single-wave workgroups (WPB = 1) land wherever the dispatcher's round-robin stands -- a tiny copy kernel in front of
// the launch (the spectrogram tile of the multi-GPU step) left some SIMDs with three waves and others with one:
// 2.96 -> 3.8 ms at 32 channels (profiles/r03_forcedist_*), and probably the "two modes" of round 2.
template <int SE, bool PREFETCH, bool PIN = true, bool TRACE = false, int WPB_ = WPB>
__global__ __launch_bounds__(64 * WPB_) void env_bwd_kernel(const SosPlanDev *__restrict__ P0, BwdArgs a)
{
    // TRACE (diagnostic build, option "sos_trace"): shader clocks per part of an iteration, summed per wave
    long long tr_acc[6] = {0, 0, 0, 0, 0, 0};
    long long tr_last = TRACE ? clock64() : 0;
#define TRACE_AT(i)                                              \
    do {                                                         \
        if (TRACE) {                                             \
            const long long t_ = clock64();                      \
            tr_acc[(i)] += t_ - tr_last;                         \
            tr_last = t_;                                        \
        }                                                        \
    } while (0)
    constexpr int DE = 2 * SE;
    __shared__ float4 lds_all[WPB_][64 * 8];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 *lds = lds_all[wave];
    float *ldsf = reinterpret_cast<float *>(lds);
    const int lane = threadIdx.x & 63;
    const long long unit = (long long)blockIdx.x * WPB_ + wave;
    if (unit >= a.units) return;                    // (no workgroup barrier anywhere: a wave may leave)
    const int seg = (int)(unit % a.n_seg);
    const long long ch = unit / a.n_seg;
    const float *in = a.in + ch * a.in_pitch;
    float *out = a.out + ch * a.out_pitch;
    const double *ckpt = a.ckpt + ch * a.ckpt_pitch;
    const long long T = a.T;
    const int edge = a.edge;
    const long long trace_t0 = a.trace ? wall_clock64() : 0;
    const unsigned slot = wave_slot_of_simd();

    // reversed tile index rt = n_tiles-1 - (p / TILE): the wave owns rt in [rt_lo, rt_hi)
    const long long rt_lo = (long long)seg * a.seg_tiles;
    long long rt_hi = rt_lo + a.seg_tiles;
    if (rt_hi > a.n_tiles) rt_hi = a.n_tiles;
    long long rt_start = rt_lo - a.warm_tiles;
    if (rt_start < 0) rt_start = 0;

    double cb_[DE];
#pragma unroll
    for (int r = 0; r < DE; r++) cb_[r] = 0.0;

    // Prefetch: the next tile (one below) and its checkpoint are requested right after this
    // tile went into LDS, i.e. before this tile's arithmetic and stores.  The loads are inline
    // asm (hipcc does not track them), and the wait for them sits at the END of the iteration
    // behind the 8 vector stores: `s_waitcnt vmcnt(7)` lets those stores stay in flight while
    // every older load has landed.  Iterations without exactly those stores wait vmcnt(0).
    // (The waits sit inside the branches, not behind a flag, so that the ISA check can follow them.)
    // Every load inside the loop is such an asm load: a load hipcc tracks would make it insert
    // its own vmcnt(0), which drains the prefetch as well.
    v4f nx[8], nck[SE];
    bool pre = false;
    auto fetch = [&](long long tidx) {
        const long long tsrc = (a.debug & 2) ? 0 : tidx;
#pragma unroll
        for (int k = 0; k < 8; k++) nx[k] = asm_load16(in + tsrc * TILE + 256 * k + 4 * lane);
#pragma unroll
        for (int i = 0; i < SE; i++) nck[i] = asm_load16(ckpt + tidx * DE + 2 * i);
    };
    // (Tried in round 3 and dropped: the eight loads in four pairs spread over the forward cascade instead of one
    // burst -- 6.02 against 5.98 ms, profiles/r03_bwd_split_ab.log.)
    // The fetch itself is unconditional (a tile that cannot be prefetched fetches the highest full
    // tile instead and drops it): a conditional asm load would make `nx` a phi of two register
    // sets, and the copies hipcc inserts for it read the registers while the loads are in flight.
    const long long top_full = T / TILE - 1;                     // host guarantees >= 0
    auto prefetchable = [&](long long tidx) { return tidx >= 0 && tidx <= top_full && tidx * TILE + TILE > a.skip; };
    if (PREFETCH) {
        const long long t0 = a.n_tiles - 1 - rt_start;
        pre = rt_start < rt_hi && prefetchable(t0);
        fetch(pre ? t0 : top_full);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the first tile has nothing to hide behind
    }

    for (long long rt = rt_start; rt < rt_hi; rt++) {
        const long long tidx = a.n_tiles - 1 - rt;
        const long long tile = tidx * TILE;
        if (tile + TILE <= a.skip) break;          // nothing below `skip` is kept
        if (a.fair) rotate_issue_priority(slot);
        double cfw_[DE];
        // ---- trace tile -> LDS, rectified
        if (PREFETCH && pre) {
            // (the wait for this tile's prefetch sits at the end of the previous iteration, behind
            // the stores it is counted against)
            // (one wave-uniform branch around the eight; |x| is exact -- the gain rides on the forward cascade, CASC_GAIN)
            if (a.rectify) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    asm volatile("" : "+v"(nx[k]));
                    lds[lds_slot(8 * k + (lane >> 3), lane & 7)] =
                        make_float4(fabsf(nx[k].x), fabsf(nx[k].y), fabsf(nx[k].z), fabsf(nx[k].w));
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    asm volatile("" : "+v"(nx[k]));
                    lds[lds_slot(8 * k + (lane >> 3), lane & 7)] = make_float4(nx[k].x, nx[k].y, nx[k].z, nx[k].w);
                }
            }
#pragma unroll
            for (int i = 0; i < SE; i++) {
                asm volatile("" : "+v"(nck[i]));
                const v2d d = __builtin_bit_cast(v2d, nck[i]);
                cfw_[2 * i] = d.x; cfw_[2 * i + 1] = d.y;
            }
            WAVE_SYNC();
        } else if (PREFETCH) {
            // a tile that touches T (the top one or two of a channel).  Every load of this loop is
            // an untracked asm load: hipcc's own vmcnt(0) for a tracked one would also drain the
            // prefetch issued further down.  Clamped addresses are always valid; samples past T
            // become zero.
#pragma unroll 1
            for (int k = 0; k < 8; k++) {
                const long long p = tile + 256 * k + 4 * lane;
                v4f t;
                t.x = asm_load4(in + (p < T ? p : T - 1));
                t.y = asm_load4(in + (p + 1 < T ? p + 1 : T - 1));
                t.z = asm_load4(in + (p + 2 < T ? p + 2 : T - 1));
                t.w = asm_load4(in + (p + 3 < T ? p + 3 : T - 1));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(t));
                float4 v = make_float4(p < T ? t.x : 0.f, p + 1 < T ? t.y : 0.f, p + 2 < T ? t.z : 0.f,
                                       p + 3 < T ? t.w : 0.f);
                if (a.rectify) v = make_float4(fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w));
                lds[lds_slot(8 * k + (lane >> 3), lane & 7)] = v;
            }
            v4f ck[SE];
#pragma unroll
            for (int i = 0; i < SE; i++) ck[i] = asm_load16(ckpt + tidx * DE + 2 * i);
            float ra = asm_load4(in + (T - 1));
            float rb = asm_load4(in + (lane < edge ? T - 2 - lane : 0));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < SE; i++) {
                asm volatile("" : "+v"(ck[i]));
                const v2d d = __builtin_bit_cast(v2d, ck[i]);
                cfw_[2 * i] = d.x; cfw_[2 * i + 1] = d.y;
            }
            asm volatile("" : "+v"(ra));
            asm volatile("" : "+v"(rb));
            WAVE_SYNC();
            // right odd extension ext[T + i] = 2 r(T-1) - r(T-2-i), i < edge
            if (lane < edge) {
                const long long pj = T + lane;
                if (pj >= tile && pj < tile + TILE) {
                    if (a.rectify) { ra = fabsf(ra); rb = fabsf(rb); }
                    ldsf[lds_float_index((int)(pj - tile))] = 2.f * ra - rb;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float4 v = load_four(in, tile + 256 * k + 4 * lane, T);
                if (a.rectify) v = make_float4(fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w));
                lds[lds_slot(8 * k + (lane >> 3), lane & 7)] = v;
            }
#pragma unroll
            for (int r = 0; r < DE; r++) cfw_[r] = ckpt[tidx * DE + r];
        }
        WAVE_SYNC();
        if (!PREFETCH && tile + TILE > T) {
            // right odd extension ext[T + i] = 2 r(T-1) - r(T-2-i), i < edge, straight from HBM
            if (lane < edge) {
                const long long pj = T + lane;
                if (pj >= tile && pj < tile + TILE) {
                    float ra = in[T - 1], rb = in[T - 2 - lane];
                    if (a.rectify) { ra = fabsf(ra); rb = fabsf(rb); }
                    ldsf[lds_float_index((int)(pj - tile))] = 2.f * ra - rb;
                }
            }
            WAVE_SYNC();
        }
        if (PREFETCH) WAVE_SYNC();
        TRACE_AT(0);                               // tile from the prefetch registers into LDS
        if (PREFETCH) {
            pre = rt + 1 < rt_hi && prefetchable(tidx - 1);
            fetch(pre ? tidx - 1 : top_full);
        }
        TRACE_AT(1);                               // prefetch of the next tile issued
        // ---- forward cascade again, from the state that entered this tile
#define CASC_S SE
#define CASC_PLAN() PLAN_OF(P0)
#define CASC_CARRY cfw_
#define CASC_IN(v) (v)
#define CASC_PIN_GROUPS PIN
#define CASC_GAIN a.gain
#include "sos_cascade.inc"
#undef CASC_GAIN
#undef CASC_CARRY
        WAVE_SYNC();
        TRACE_AT(2);                               // forward cascade
        if (rt == 0) {
            // scipy: backward pass starts from zi * y_fwd[-1]; pad the rest of the tile with it
            const int last = (int)(T + edge - 1 - tile);
            const float v0 = ldsf[lds_float_index(last)];
            WAVE_SYNC();
            for (int s2 = last + 1 + lane; s2 < TILE; s2 += 64) ldsf[lds_float_index(s2)] = v0;
            const SosPlanDev *P = PLAN_OF(P0);
#pragma unroll
            for (int r = 0; r < DE; r++) cb_[r] = P->zi[r] * (double)v0;
            WAVE_SYNC();
        }
        // ---- backward cascade over the forward outputs, last sample first
#define CASC_CARRY cb_
#define CASC_REVERSED
#include "sos_cascade.inc"
#undef CASC_REVERSED
#undef CASC_PIN_GROUPS
#undef CASC_S
#undef CASC_PLAN
#undef CASC_CARRY
#undef CASC_IN
        WAVE_SYNC();
        TRACE_AT(3);                               // backward cascade
        if (rt >= rt_lo) {
            if (tile >= a.skip && tile + TILE <= T) {
                // interior tile: exactly 8 vector stores (max(x, 0) as one instruction: fmaxf() costs a second one
                // that only quiets signalling NaNs)
                if (a.clamp) {
                    const long long tdst = (a.debug & 1) ? a.skip : tile;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const float4 v = lds[lds_slot(8 * k + (lane >> 3), lane & 7)];
                        f4u t; t.x = max_zero(v.x); t.y = max_zero(v.y); t.z = max_zero(v.z); t.w = max_zero(v.w);
                        *reinterpret_cast<f4u *>(out + (tdst + 256 * k + 4 * lane - a.skip)) = t;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const float4 v = lds[lds_slot(8 * k + (lane >> 3), lane & 7)];
                        f4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
                        *reinterpret_cast<f4u *>(out + (tile + 256 * k + 4 * lane - a.skip)) = t;
                    }
                }
                TRACE_AT(4);                       // 8 stores issued
                // the prefetch issued above is older than these 8 stores: all but 7 operations done
                // means every load has landed (one less than 8, so a merged store could not make
                // the wait too weak; tools/check_prefetch_isa.py re-checks the ISA)
                if (PREFETCH) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    float4 v = lds[lds_slot(8 * k + (lane >> 3), lane & 7)];
                    if (a.clamp) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    store_four(out, tile + 256 * k + 4 * lane, v, a.skip, T, a.skip);
                }
                if (PREFETCH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else if (PREFETCH) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // warm-up tile: no stores to count
        }
        WAVE_SYNC();
        TRACE_AT(5);                               // wait for the prefetch
    }
#undef TRACE_AT
    if (a.trace && lane == 0 && unit < a.trace_rows) { // tools/sweep_trace.py: do the waves of a SIMD progress alike?
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        long long *tr = a.trace + 9 * unit;
        tr[0] = trace_t0;
        tr[1] = wall_clock64();
        tr[2] = hw;
#pragma unroll
        for (int i = 0; i < 6; i++) tr[3 + i] = tr_acc[i];
    }
}

// ---- building blocks of sosfiltfilt for cascades longer than one plan (hipdsp_envelope_multi) -------
// ext = odd extension of r = |x| (or x) by `edge` samples on both sides (scipy odd_ext,
// scipy/signal/_arraytools.py:99-107), float32 arithmetic like the fused kernels; the gain of the rectified trace
// rides on the first plan's cascade (SeqArgs::gain)
__global__ void odd_ext_kernel(const float *__restrict__ x, long long x_pitch, long long T, int edge, int rectify,
                               float *__restrict__ out, long long out_pitch)
{
    const long long ch = blockIdx.y;
    const float *xc = x + ch * x_pitch;
    float *oc = out + ch * out_pitch;
    const long long N = T + 2LL * edge;
    auto r = [&](long long k) { const float v = xc[k]; return rectify ? fabsf(v) : v; };
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        float v;
        if (i < edge) v = 2.f * r(0) - r(edge - i);
        else if (i < edge + T) v = r(i - edge);
        else v = 2.f * r(T - 1) - r(T - 2 - (i - edge - T));
        oc[i] = v;
    }
}

// y[c][i] = x[c][N - 1 - (first + i)], i < n, optionally clamped at zero: time reversal (and the final trim)
__global__ void flip_kernel(const float *__restrict__ x, long long x_pitch, long long N, long long first, long long n,
                            int clamp, float *__restrict__ y, long long y_pitch)
{
    const long long ch = blockIdx.y;
    const float *xc = x + ch * x_pitch;
    float *yc = y + ch * y_pitch;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = xc[N - 1 - (first + i)];
        if (clamp) v = fmaxf(v, 0.f);
        yc[i] = v;
    }
}

// ref[c] = x[c][0]
__global__ void first_sample_kernel(const float *__restrict__ x, long long x_pitch, long long channels, float *__restrict__ ref)
{
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < channels) ref[c] = x[c * x_pitch];
}

// pass-through / zero fill for the sos-is-None branches
__global__ void copy_skip_kernel(const float *__restrict__ x, long long x_pitch, float *__restrict__ y,
                                 long long y_pitch, long long n, long long skip)
{
    long long ch = blockIdx.y;
    const float *xi = x + ch * x_pitch + skip;
    float *yo = y + ch * y_pitch;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        yo[i] = xi[i];
}

// ---- forward sweep of the batch chain with the spectrogram fused in -------------------------
// The filtered trace is the one array of the chain that is read twice (by the spectrogram and by
// the envelope's backward sweep).  Here the forward sweep hands every finished tile to an FFT wave
// of its own workgroup through LDS, so the spectrogram never reads it from HBM: 12 instead of
// 16 B/sample for band-pass + envelope states + PSD.  nfft == TILE (2048), hop == TILE/2: tile t
// IS frame 2t, and frame 2t-1 is the second half of tile t-1 followed by the first half of tile t.
//
// A workgroup is P IIR waves (exactly sos_ckpt_kernel<SF, SE, true>'s walk over one
// (channel, segment) each) and P FFT waves, FFT wave p serving IIR wave p.  Two hand-overs per tile:
//   H1  the tile holds the band-pass output      -> the FFT wave copies it into registers
//   H2  the copy is done                         -> the IIR wave may rectify the tile in place
// and between H2 and the next H1 the FFT wave computes its (at most) two frames while the IIR
// wave finishes the tile (envelope state sweep) and brings in the next one.  FLAGS: the hand-overs
// are two monotonic counters per pair in LDS (ready / taken, one writer each, bounded polling with
// s_sleep), so a pair never waits for another pair; otherwise they are workgroup barriers, for which
// all waves of the grid walk the same number of iterations (tile = lo - warm + k*TILE; a segment
// without warm-up, or without the extra extension tile, idles through the others) -- the flag
// variant keeps that iteration space.  Within a wave the single-wave kernels' __syncthreads()
// become wave-local fences.
struct ChainArgs {
    CkptArgs c;
    float *psd;               // (channels, frames_out, TILE/2 + 1)
    float *db;                // optional: decibel(psd), same layout
    long long psd_pitch;
    long long n_valid;        // frames that lie inside the trace
    const float *tables;      // tw2 | tw3 | twn | window of the 2048-point PSD kernel (fft_tables)
    float scale;              // 1 / (fs * sum w^2)
    int n_iter;
    long long warm_total;     // band-pass + envelope warm-up samples
    long long units;          // channels * n_seg
    int debug;                // experiments: 1 = FFT waves only copy, 2 = IIR waves skip the cascades
                              // (bit 4, host side: workgroup barriers instead of the pairwise flags;
                              // bit 8: FFT wave 0 of workgroup 0 withholds one hand-over -- fault-path test)
    int *fault;               // hipdsp_ctx::fault_dev: where a wave that gave up waiting says so
    int split;                // frame split: only the even frames are written here, chain_bwd_kernel writes the odd ones
    long long unit_stride;    // 0: unit = block * NP + pair; else unit = pair * unit_stride + block -- fewer units than the
                              // chip has pairs are spread over all CUs, the pairs of a workgroup that get none idle
};

// One frame of NFFT samples per group of LPF lanes (2048: LPF 64, radix 16 x 16 x 4; 1024: 64, 8 x 8 x 8; 512: two
// frames side by side in a wave, LPF 32, 8 x 8 x 4) from its PPL = NFFT / (2 LPF) values per lane (value t: samples
// 2l + 2 LPF t, + 1 of the frame) -> detrend, Hann, half-length complex FFT, split step, PSD.  `keep` masks the
// stores of a lane group whose frame does not exist.  Same arithmetic as spec_fast_kernel<NFFT, LPF, R1, R2, R3, ...>.
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };

template <int NFFT, int LPF, int R1, int R2, int R3, bool DB, class Hook = NoHook>
__device__ __forceinline__ void psd_frame(const v2f *w, float2 *fb, const float2 *tw2, const float2 *tw3,
                                          const float2 *twn, const float2 *win, int lane, float scale, bool keep,
                                          float *__restrict__ o, float *__restrict__ od, Hook hook = Hook())
{
#pragma clang fp contract(fast)
    constexpr int M = NFFT / 2, PPL = M / LPF;
    static_assert(PPL == R1 && R1 * R2 * R3 == M, "one first-stage butterfly per lane");
    const int l = lane % LPF, g0 = (lane / LPF) * LPF;
    float2 v[PPL];
    v2f acc = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < R1; t++) {
        v[t] = make_float2(w[t].x, w[t].y);
        acc += w[t];
    }
    float sum = acc.x + acc.y;
    if (LPF == 64) {
        sum = wave_sum(sum);
    } else {
#pragma unroll
        for (int d = LPF / 2; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    }
    const float mean = sum * (1.0f / (float)NFFT);
    const v2f mean2 = {mean, mean};
#pragma unroll
    for (int t = 0; t < R1; t++) v[t] = as_f2((as_v2f(v[t]) - mean2) * as_v2f(win[l + LPF * t]));
    hook(0);                                   // mean and window
    stockham_stage<R1, 1, M, LPF, false, true>(v, fb, tw2, l);
    hook(1);                                   // first butterflies, values on their way through LDS
    stockham_stage<R2, R1, M, LPF, true, true>(v, fb, tw2, l);
    hook(2);
    stockham_stage<R3, R1 * R2, M, LPF, true, false, true>(v, fb, tw3, l);
    hook(3);
    // v[u*R3 + t] = Z[k], k = l + LPF*m, m = u + NB3*t; partner bin Z[M-k] from lane LPF-l of the same group
    constexpr int NB3 = PPL / R3;
    const int partner = g0 + ((LPF - l) & (LPF - 1));
    float pk_last = 0.f;
    const v2f hscale2 = {0.5f * scale, 0.5f * scale};
#pragma unroll
    for (int m = 0; m < PPL / 2; m++) {
        const int k = l + LPF * m;
        const float2 zk = v[(m % NB3) * R3 + m / NB3];
        const int mp = PPL - 1 - m;
        const float2 zsrc = v[(mp % NB3) * R3 + mp / NB3];
        float2 zm;
        zm.x = __shfl(zsrc.x, partner, 64);
        zm.y = __shfl(zsrc.y, partner, 64);
        if (m > 0) {
            const int m0 = PPL - m;
            const float2 z0 = v[(m0 % NB3) * R3 + m0 / NB3];
            zm = (l == 0) ? z0 : zm;
        }
        const v2f e = pk_add_conj(as_v2f(zk), as_v2f(zm));
        const v2f t = pk_cmul_negi(pk_sub_conj(as_v2f(zk), as_v2f(zm)), as_v2f(twn[k]));
        const v2f re = pk_sumdiff_x(e, t), im = pk_sumdiff_y(e, t);
        const v2f pw = (re * re + im * im) * hscale2;
        float pk = pw.x, pm = pw.y;
        if (m == 0) {
            const float dc0 = zk.x + zk.y, ny = zk.x - zk.y;
            pk = (l == 0) ? dc0 * dc0 * scale : pk;
            pm = (l == 0) ? ny * ny * scale : pm;
        }
        if (LPF == 64 || keep) {
            o[k] = pk;
            o[M - k] = pm;
            if (DB) { od[k] = to_db(pk); od[M - k] = to_db(pm); }
        }
        pk_last = pk;
    }
    {
        constexpr int mh = PPL / 2;
        const float2 z = v[(mh % NB3) * R3 + mh / NB3];
        const float ph = 2.f * scale * (z.x * z.x + z.y * z.y);
        const int kk = (l == 0) ? M / 2 : l + LPF * (PPL / 2 - 1);
        const float pv = (l == 0) ? ph : pk_last;
        if (LPF == 64 || keep) {
            o[kk] = pv;
            if (DB) od[kk] = to_db(pv);
        }
    }
}

// STAMP (diagnostic build, "chain_debug" bit 32; results stay valid): every wave adds up the shader clocks
// it spends in each part of its loop body and leaves the 16 sums in a.db (which then is NOT a dB output).
#define FPT_OK(hop, g) ((2048 / (hop)) % (g) == 0)
// NFFT / HOP: the window lengths whose frames are register windows of a tile -- 2048 or 1024 samples, hops
// that divide the tile and are multiples of 128 samples (one register of the FFT wave's tile copy).
template <int SF, int SE, int NP, bool FLAGS, bool DB, int NFFT = 2048, int HOP = 1024, bool STAMP = false>
__global__ __launch_bounds__(128 * NP, NP / 2) void chain_fwd_kernel(const SosPlanDev *__restrict__ PF0,
                                                                   const SosPlanDev *__restrict__ PE0, ChainArgs a)
{
    // SE == 0: no envelope behind the filter (the reference's default trace set, plugins.py:11-13: filter + spectrogram)
    static_assert(SF > 0 && SE >= 0 && NP % 2 == 0, "band-pass in front; whole waves per SIMD");
    constexpr int DF = 2 * SF, DE = SE > 0 ? 2 * SE : 1;
    static_assert((NFFT == 2048 || NFFT == 1024 || NFFT == 512 || NFFT == 256) && TILE % HOP == 0 && HOP % 128 == 0 &&
                  HOP <= NFFT && NFFT <= TILE, "frames must be register windows of a tile");
    static_assert(NFFT != 256 || HOP == 128, "256-sample frames: the reference's default, 50 % overlap");
    constexpr int M = NFFT / 2, F = M + 1, MP = M + M / 16;
    constexpr int LPF = NFFT >= 1024 ? 64 : (NFFT == 512 ? 32 : 16);  // lanes per frame; G frames side by side in an FFT wave
    constexpr int G = 64 / LPF;
    constexpr int R1 = NFFT == 2048 ? 16 : 8, R2 = NFFT == 256 ? 4 : R1, R3 = NFFT == 1024 ? 8 : 4;
    constexpr int TW2 = (R2 - 1) * R1, TW3 = R1 * R2, TWN = M / 2 + 1, NTAB = TW2 + TW3 + TWN + M;
    constexpr int PPL = NFFT / 128;             // registers (128 samples each) of one frame
    static_assert(FPT_OK(HOP, G), "whole groups of frames per tile");
    constexpr int FPT = TILE / HOP;             // frames that END inside a tile (a multiple of G)
    constexpr int PREV = (NFFT - HOP) / 128;    // registers of the previous tile a frame can reach back into
    __shared__ float4 tiles[NP][64 * 8];
    __shared__ float rprevs[NP][64];
    __shared__ float2 fbs[NP][G * MP];
    __shared__ float2 tab[NTAB];
    // FLAGS: pairwise hand-over instead of the two workgroup barriers -- ready[p] counts the tiles IIR
    // wave p has finished, taken[p] the tiles FFT wave p has copied (monotonic, one writer each)
    __shared__ int ready[NP], taken[NP];
    // Fair shares of a SIMD: the issue arbiter prefers the waves of the lower slots, so that without help
    // pairs 0 .. NP/2-1 finish their units after 78 % of the launch and the other half then runs alone,
    // latency-bound, on a half-empty CU (tools/chain_stamps.py: wave lifetimes 14.8 M against 19.0 M clocks).
    // Every wave publishes the iteration it is in and compares it with the wave of the same role of pair
    // p ^ NP/2 -- the one it shares its SIMD with: whoever is ahead steps down one priority level until the
    // other has caught up ("chain_debug" bit 128 = off).
    __shared__ int prog[2 * NP];
    if (threadIdx.x < NP) { ready[threadIdx.x] = 0; taken[threadIdx.x] = 0; }
    if (threadIdx.x < 2 * NP) prog[threadIdx.x] = 0;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int pair = wave < NP ? wave : wave - NP;
#define CHAIN_FAIR(iter, behind_prio, ahead_prio)                                                        \
    do {                                                                                                 \
        if (FLAGS && !(a.debug & 128)) {                                                                 \
            if (lane == 0) __hip_atomic_store(&prog[wave], (iter), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
            const int other_ = __builtin_amdgcn_readfirstlane(                                           \
                __hip_atomic_load(&prog[wave ^ (NP / 2)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); \
            set_issue_priority(other_ < (iter) ? (ahead_prio) : (behind_prio));                          \
        }                                                                                                \
    } while (0)
    // Bounded polling: a logic error must not hang the GPU.  After 2^23 naps (a third of a second, far
    // beyond anything a partner wave of the same workgroup can be late by) the wave GIVES UP: it
    // reports the fault through the context's fault word (the host turns it into HIPDSP_ERR_HIP at the
    // next synchronisation), raises the workgroup's abort word so that its partner stops waiting too,
    // and walks the rest of its iterations without waiting for anything -- the launch ends quickly
    // and its outputs are declared invalid, instead of being silently wrong.
    // Macros, not lambdas: through a pointer parameter the flags would be accessed with flat
    // instructions, whose wait also drains the prefetch of the IIR role.
    __shared__ int abort_wg;
    if (threadIdx.x == 0) abort_wg = 0;
    bool gave_up = false;
#define CHAIN_WAIT_FOR(arr, want, iter)                                             \
    do {                                                                            \
        if (!gave_up) {                                                             \
            for (int spin_ = 0;; spin_++) {                                         \
                if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&arr[pair], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >= (want)) break; \
                if ((spin_ & 1023) == 1023) {                                       \
                    const bool told = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&abort_wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0; \
                    if (told || spin_ >= (1 << 23) - 1) {                           \
                        gave_up = true;                                             \
                        if (!told && lane == 0) {                                   \
                            __hip_atomic_store(&abort_wg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
                            a.fault[1] = (int)blockIdx.x; a.fault[2] = pair; a.fault[3] = (iter); \
                            __hip_atomic_store(&a.fault[0], HD_FAULT_CHAIN_HANDOVER, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); \
                        }                                                           \
                        break;                                                      \
                    }                                                               \
                }                                                                   \
                __builtin_amdgcn_s_sleep(1);                                        \
            }                                                                       \
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");        \
        }                                                                           \
    } while (0)
#define CHAIN_POST(arr, value)                                                      \
    do {                                                                            \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");            \
        if (lane == 0) __hip_atomic_store(&arr[pair], (value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
    } while (0)
    {
        const float2 *src = reinterpret_cast<const float2 *>(a.tables);
        for (int i = tid; i < NTAB; i += 128 * NP) tab[i] = src[i];
    }
    __syncthreads();

    const long long unit = a.unit_stride ? (long long)pair * a.unit_stride + blockIdx.x : (long long)blockIdx.x * NP + pair;
    const bool unit_ok = unit < a.units;
    const int seg = unit_ok ? (int)(unit % a.c.n_seg) : 0;
    const long long ch = unit_ok ? unit / a.c.n_seg : 0;
    const long long T = a.c.T;
    const int edge = a.c.edge;
    const long long lo = (long long)seg * a.c.seg_len;
    long long hi = lo + a.c.seg_len;
    const bool last_seg = hi >= T;
    if (hi > T) hi = T;
    // no envelope warm-up: zero-state tile states + env_fix_kernel, exactly as in sos_ckpt_kernel
    const long long env_start = lo;
    const bool env_true = seg == 0;
    long long start = env_start - PF0->warm;
    if (start < 0) start = 0;
    long long loop_end = last_seg ? T + edge : hi;
    if (!unit_ok) loop_end = start;                     // a pair without a unit only takes the barriers
    const long long base = lo - a.warm_total;           // tile of iteration 0 (negative: idle iterations)

    // "chain_debug" bit 16: IIR wave 0 of workgroup 0 reports shader clocks and 100 MHz ticks spent in
    // the kernel into the first 16 bytes of the PSD (measurement of the engine clock under this load)
    const long long dbg_c0 = (a.debug & 16) ? clock64() : 0, dbg_w0 = (a.debug & 16) ? wall_clock64() : 0;
    long long st_acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) st_acc[i] = 0;
    long long st_last = STAMP ? clock64() : 0;
#define STAMP_AT(i)                                              \
    do {                                                         \
        if (STAMP) {                                             \
            const long long t_ = clock64();                      \
            st_acc[(i)] += t_ - st_last;                         \
            st_last = t_;                                        \
        }                                                        \
    } while (0)
    if (wave < NP) {
        // ================= IIR role: sos_ckpt_kernel<SF, SE, true> with the barriers added ==========
        float4 *lds = tiles[pair];
        float *ldsf = reinterpret_cast<float *>(lds);
        float *rprev = rprevs[pair];
        const float *in = a.c.in + ch * a.c.in_pitch;
        float *yf = a.c.yf + ch * a.c.yf_pitch;
        double *ckpt = SE > 0 ? a.c.ckpt + ch * a.c.ckpt_pitch : nullptr;
        double cf_[DF], ce_[DE];
#pragma unroll
        for (int r = 0; r < DF; r++) cf_[r] = 0.0;
#pragma unroll
        for (int r = 0; r < DE; r++) ce_[r] = 0.0;
        rprev[lane] = 0.f;
        v4f nx[8];
        bool pre = false;
        int pending = 0;
        const long long top_full = (T / TILE - 1) * TILE;            // host guarantees >= 0
        auto fetch = [&](long long t0) {
#pragma unroll
            for (int k = 0; k < 8; k++) nx[k] = asm_load16(in + t0 + 256 * k + 4 * lane);
        };
        // what iteration k + 1 will read: its tile if that is a full tile of this unit, else a dummy
        auto prefetchable = [&](long long t0) { return t0 >= start && t0 < loop_end && t0 + TILE <= T; };
        pre = prefetchable(base);
        fetch(pre ? base : top_full);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        for (int it = 0; it < a.n_iter; it++) {
            const long long tile = base + (long long)it * TILE;
            const bool active = tile >= start && tile < loop_end;
            if (!(a.debug & 64)) CHAIN_FAIR(it, 1, 0);
            if (FLAGS && pending) {                            // H2 of the previous (quiet) tile
                CHAIN_WAIT_FOR(taken, pending, it);
                pending = 0;
            }
            if (active) {
                if (pre) {
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        asm volatile("" : "+v"(nx[k]));
                        lds[lds_slot(8 * k + (lane >> 3), lane & 7)] = make_float4(nx[k].x, nx[k].y, nx[k].z, nx[k].w);
                    }
                } else {
                    // a tile that reaches past T: untracked loads from clamped addresses, zeros past T
#pragma unroll 1
                    for (int k = 0; k < 8; k++) {
                        const long long p = tile + 256 * k + 4 * lane;
                        v4f t;
                        t.x = asm_load4(in + (p < T ? p : T - 1));
                        t.y = asm_load4(in + (p + 1 < T ? p + 1 : T - 1));
                        t.z = asm_load4(in + (p + 2 < T ? p + 2 : T - 1));
                        t.w = asm_load4(in + (p + 3 < T ? p + 3 : T - 1));
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        asm volatile("" : "+v"(t));
                        lds[lds_slot(8 * k + (lane >> 3), lane & 7)] =
                            make_float4(p < T ? t.x : 0.f, p + 1 < T ? t.y : 0.f, p + 2 < T ? t.z : 0.f, p + 3 < T ? t.w : 0.f);
                    }
                }
            }
            WAVE_SYNC();
            // unconditional, like in sos_ckpt_kernel (a conditional fetch would turn `nx` into a phi
            // whose copies read registers with loads in flight)
            {
                const long long next = tile + TILE;
                pre = prefetchable(next);
                fetch(pre ? next : top_full);
            }
            STAMP_AT(0);                                       // wait for H2 of the last tile, tile -> LDS, fetch issued
            // Phase 1 of the envelope cascade rides on phase 3 of the band-pass: every filtered sample is
            // multiplied into the envelope's G table while it is still a register -- |y| as a source modifier of
            // the float64 multiply-add, the gain once per tile on the sums -- so that a quiet tile needs neither a
            // pass over the tile in LDS nor a conversion or a multiplication per sample.  The value is the
            // band-pass output AFTER its rounding to float32, converted back: the filtered trace in HBM is what the
            // backward sweep recomputes the forward cascade from, so states and recomputation see the same input
            // (ADVICE round 2).  Tiles that are not quiet (odd extension in reach) ignore the result and take the
            // path through LDS.
            double etap[DE];
#pragma unroll
            for (int r = 0; r < DE; r++) etap[r] = 0.0;
            if (active && !(a.debug & 2)) {
                const double tgain = a.c.gain;
                const SosPlanDev *PEt = PLAN_OF(SE > 0 ? PE0 : PF0);
#define CASC_S SF
#define CASC_PLAN() PLAN_OF(PF0)
#define CASC_CARRY cf_
#define CASC_IN(v) (v)
#define CASC_ROLLED_GROUPS
#define CASC_STAMP(n) STAMP_AT(1 + (n))
#define CASC_TAP(j, e, y)                                                               \
    do {                                                                                \
        if constexpr (SE > 0) {                                                         \
            if (((j) & 3) == 0) PEt = PLAN_OF(PE0);                                     \
            const double rd_ = fabs((double)(e));                                        \
            _Pragma("unroll") for (int r_ = 0; r_ < DE; r_++) etap[r_] = fma(PEt->G[(j) * DE + r_], rd_, etap[r_]); \
        }                                                                               \
    } while (0)
#include "sos_cascade.inc"
#undef CASC_TAP
#undef CASC_STAMP
#undef CASC_ROLLED_GROUPS
#undef CASC_S
#undef CASC_PLAN
#undef CASC_CARRY
#undef CASC_IN
#pragma unroll
                for (int r = 0; r < DE; r++) etap[r] *= tgain;
            }
            WAVE_SYNC();
            if (FLAGS) { if (active) CHAIN_POST(ready, it + 1); }
            else __syncthreads();                              // B1: the tile holds the filtered samples
            if (active && tile >= lo && tile + TILE <= hi) {
                // interior tile: exactly 8 vector stores, then the counted wait
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const float4 v = lds[lds_slot(8 * k + (lane >> 3), lane & 7)];
                    f4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
                    *reinterpret_cast<f4u *>(yf + tile + 256 * k + 4 * lane) = t;
                }
                asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                STAMP_AT(4);                                   // H1, 8 stores of the filtered tile, wait for the prefetch
            } else {
                // border tile of the segment, warm-up or idle: whatever is stored, no stores to count
                if (active && tile + TILE > lo && tile < hi) {
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        store_four(yf, tile + 256 * k + 4 * lane, lds[lds_slot(8 * k + (lane >> 3), lane & 7)], lo, hi, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // An interior tile of the envelope sweep (neither it nor the next one touches T, no left
            // extension) is not modified any more: the rectification rides on the cascade's input
            // and, with the flags, H2 is only needed before the NEXT tile goes into LDS.
            // (the last tile of a segment that is not the trace's last advances the state too: its end state is handed
            // to the next segment by env_fix_kernel)
            const bool last_tile = tile + TILE >= loop_end;
            const bool quiet = SE > 0 && active && tile >= env_start && a.c.rectify && tile + 2 * TILE <= T &&
                               !(env_true && tile == 0) && (!last_tile || !last_seg) && !(a.debug & 2);
            if (FLAGS) {
                if (active) {
                    if (quiet || SE == 0) pending = it + 1;
                    else CHAIN_WAIT_FOR(taken, it + 1, it);
                }
            } else {
                __syncthreads();                               // B2: the FFT wave has its copy
            }
            if constexpr (SE > 0) {
            if (quiet) {
                if ((tile > lo || env_true) && lane == 0) {
#pragma unroll
                    for (int r = 0; r < DE; r++) ckpt[(tile / TILE) * DE + r] = ce_[r];
                }
#define CASC_S SE
#define CASC_PLAN() PLAN_OF(PE0)
#define CASC_CARRY ce_
#define CASC_IN(v) (v)
#define CASC_F_IN etap
#define CASC_NO_OUTPUT
#define CASC_ROLLED_GROUPS
#define CASC_STAMP(n) STAMP_AT(5 + (n))
#include "sos_cascade.inc"
#undef CASC_STAMP
#undef CASC_ROLLED_GROUPS
#undef CASC_NO_OUTPUT
#undef CASC_F_IN
#undef CASC_S
#undef CASC_PLAN
#undef CASC_CARRY
#undef CASC_IN
                if (last_tile && lane == 0) {
#pragma unroll
                    for (int r = 0; r < DE; r++) ckpt[(tile / TILE + 1) * DE + r] = ce_[r];
                }
                WAVE_SYNC();
            } else if (active && tile >= env_start && !(a.debug & 2)) {
                // ---- envelope input in place: r = |y| (the gain rides on the cascade), then the odd extension past T
                if (a.c.rectify) {
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        float4 v = lds[lds_slot(lane, q)];
                        v = make_float4(fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w));
                        lds[lds_slot(lane, q)] = v;
                    }
                }
                WAVE_SYNC();
                auto rval = [&](long long j) -> float {
                    return j >= tile ? ldsf[lds_float_index((int)(j - tile))] : rprev[64 - (int)(tile - j)];
                };
                if (tile + TILE > T) {
                    float pv = 0.f;
                    long long pj = -1;
                    if (lane < edge) {
                        pj = T + lane;
                        if (pj >= tile && pj < tile + TILE) pv = 2.f * rval(T - 1) - rval(T - 2 - lane);
                    }
                    WAVE_SYNC();
                    if (lane < edge && pj >= tile && pj < tile + TILE) ldsf[lds_float_index((int)(pj - tile))] = pv;
                    WAVE_SYNC();
                }
                if (env_true && tile == 0) {
                    const SosPlanDev *Pz = PLAN_OF(PE0);
                    const double r0 = (double)ldsf[lds_float_index(0)];
                    const double x0 = a.c.gain * (2.0 * r0 - (double)ldsf[lds_float_index(edge)]);
#pragma unroll
                    for (int r = 0; r < DE; r++) ce_[r] = Pz->zi[r] * x0;
                    for (int i = 0; i < edge; i++) {
                        double cur = a.c.gain * (2.0 * r0 - (double)ldsf[lds_float_index(edge - i)]);
#pragma unroll
                        for (int s2 = 0; s2 < SE; s2++) {
                            const double y = fma(Pz->coef[s2][0], cur, ce_[2 * s2]);
                            ce_[2 * s2] = fma(-Pz->coef[s2][3], y, fma(Pz->coef[s2][1], cur, ce_[2 * s2 + 1]));
                            ce_[2 * s2 + 1] = fma(-Pz->coef[s2][4], y, Pz->coef[s2][2] * cur);
                            cur = y;
                        }
                    }
                }
                if ((tile > lo || env_true) && lane == 0) {
#pragma unroll
                    for (int r = 0; r < DE; r++) ckpt[(tile / TILE) * DE + r] = ce_[r];
                }
                if (!last_tile || !last_seg) {
                    {
                        const float4 keep0 = lds[lds_slot(62 + ((lane >> 3) & 1), lane & 7)];
                        WAVE_SYNC();
                        if (lane < 16) {
                            rprev[4 * lane] = keep0.x; rprev[4 * lane + 1] = keep0.y;
                            rprev[4 * lane + 2] = keep0.z; rprev[4 * lane + 3] = keep0.w;
                        }
                    }
#define CASC_S SE
#define CASC_PLAN() PLAN_OF(PE0)
#define CASC_CARRY ce_
#define CASC_IN(v) (v)
#define CASC_GAIN a.c.gain
#define CASC_NO_OUTPUT
#define CASC_ROLLED_GROUPS
#include "sos_cascade.inc"
#undef CASC_ROLLED_GROUPS
#undef CASC_NO_OUTPUT
#undef CASC_GAIN
#undef CASC_S
#undef CASC_PLAN
#undef CASC_CARRY
#undef CASC_IN
                    if (last_tile && lane == 0) {
#pragma unroll
                        for (int r = 0; r < DE; r++) ckpt[(tile / TILE + 1) * DE + r] = ce_[r];
                    }
                }
                WAVE_SYNC();
            }
            }   // SE > 0
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last (dummy) prefetch
        STAMP_AT(7);
        if ((a.debug & 16) && blockIdx.x == 0 && wave == 0 && lane == 0) {
            long long *dbg = reinterpret_cast<long long *>(a.psd);
            dbg[0] = clock64() - dbg_c0;
            dbg[1] = wall_clock64() - dbg_w0;
        }
    } else {
        // ================= FFT role ===================================================================
        // The FFT wave is the critical path of its pair (tools/chain_stamps.py: 94 % of its clocks inside the two
        // FFTs of a tile, while the IIR wave waits 44 % of its own for the hand-over): it asks for the SIMD's
        // issue slots first, the IIR waves fill the gaps (-2.2 % for the launch; "chain_debug" bit 64 = off)
        if (!(a.debug & 64)) __builtin_amdgcn_s_setprio(3);
        float2 *fb = fbs[pair];
        const float *tlf = reinterpret_cast<const float *>(tiles[pair]);
        const float2 *tw2 = tab, *tw3 = tab + TW2, *twn = tab + TW2 + TW3, *win = tab + TW2 + TW3 + TWN;
        float *oc = a.psd + ch * a.psd_pitch;
        float *dc = DB ? a.db + ch * a.psd_pitch : nullptr;
        // the tile as 16 registers (register j: samples 128 j + 2 lane, + 1) and the last PREV registers of the
        // tile before it: frame m of a tile (the one that ends (m + 1) HOP samples into it) is the window of
        // PPL consecutive registers that starts at register ((m + 1) HOP - NFFT) / 128 of the two
        v2f cur_[16], prv_[PREV > 0 ? PREV : 1];
#pragma unroll
        for (int j = 0; j < 16; j++) cur_[j] = (v2f){0.f, 0.f};
#pragma unroll
        for (int j = 0; j < (PREV > 0 ? PREV : 1); j++) prv_[j] = (v2f){0.f, 0.f};
        // G == 4 (256-sample frames, 16 lanes each): frame m of a tile is the two 128-sample blocks m - 1 and m.
        // Lane group g takes the four frames m = 4 g + q, q < 4, and keeps THEIR five blocks 4 g - 1 .. 4 g + 3 in its
        // own lanes -- block b as four values per lane (sample pairs l + 16 t, t < 4: the first-stage inputs of the
        // 16-lane FFT), bb_[4 j + t] = block 4 g - 1 + j: no cross-lane move anywhere, 24 instead of 16 LDS loads
        // per tile (a group's first block is its neighbour's last); pv_ carries block 15 into the next tile for
        // group 0, whose block -1 it is.
        v2f bb_[G == 4 ? 20 : 1], pv_[4];
#pragma unroll
        for (int j = 0; j < 4; j++) pv_[j] = (v2f){0.f, 0.f};
        bool have_prev = false;
        for (int it = 0; it < a.n_iter; it++) {
            const long long tile = base + (long long)it * TILE;
            const bool active = tile >= start && tile < loop_end;
            // (which role stands above the other no longer matters once equals keep pace: IIR above FFT 10.95 ms,
            // both on the same two levels 10.83 ms, FFT above IIR 10.85 ms in one process)
            if (!(a.debug & 64)) CHAIN_FAIR(it, 3, 2);
            if (FLAGS) { if (active) CHAIN_WAIT_FOR(ready, it + 1, it); }
            else __syncthreads();                              // B1
            STAMP_AT(8);                                       // waited for the IIR wave's tile
            v2f pvn_[4];
            if (active) {
                if constexpr (G == 4) {
                    const int gq = lane >> 4, l16 = lane & 15;
#pragma unroll
                    for (int j = 0; j < 5; j++)
#pragma unroll
                        for (int tt = 0; tt < 4; tt++) {
                            const int b = 4 * gq - 1 + j;              // (-1: the block the last tile left in pv_)
                            const v2f h = *reinterpret_cast<const v2f *>(tlf + lds_float_index(128 * (b < 0 ? 0 : b) + 2 * (l16 + 16 * tt)));
                            bb_[4 * j + tt] = (j == 0 && b < 0) ? pv_[tt] : h;
                        }
#pragma unroll
                    for (int tt = 0; tt < 4; tt++)
                        pvn_[tt] = *reinterpret_cast<const v2f *>(tlf + lds_float_index(128 * 15 + 2 * (l16 + 16 * tt)));
                } else {
#pragma unroll
                    for (int j = 0; j < 16; j++)
                        cur_[j] = *reinterpret_cast<const v2f *>(tlf + lds_float_index(2 * lane + 128 * j));
                }
            }
            if (FLAGS) {                                        // (the release fence waits for the loads)
                // "chain_debug" bit 8 (fault-path test): FFT wave 0 of workgroup 0 withholds the hand-over of
                // its unit's first tile, so that IIR wave 0 runs into the timeout
                const bool withhold = (a.debug & 8) && blockIdx.x == 0 && pair == 0 && tile == start;
                if (active && !withhold) CHAIN_POST(taken, it + 1);
            }
            else __syncthreads();                              // B2
            STAMP_AT(9);                                       // tile copied, hand-over posted
            if (active) {
                const long long t = tile / TILE;
                if (tile >= lo && tile < hi) {                 // the unit that owns the tile writes its frames
                  if (!(a.debug & 1)) {
                   if constexpr (G == 4) {
                    const int gq = lane >> 4;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        // lane group gq: frame m = 4 gq + q of the tile = its blocks q and q + 1
                        const long long f = t * FPT + 4 * gq + q + 1 - NFFT / HOP;
                        const bool keep = (4 * gq + q > 0 || have_prev) && f >= 0 && f < a.n_valid;
                        if (__builtin_amdgcn_ballot_w64(keep) != 0) {               // (wave-uniform: some group has a frame)
                            v2f w[8];
#pragma unroll
                            for (int i = 0; i < 8; i++) w[i] = bb_[4 * q + i];
                            const long long fc = keep ? f : 0;                      // a masked group still needs a legal address
                            psd_frame<NFFT, LPF, R1, R2, R3, DB>(w, fb + gq * MP, tw2, tw3, twn, win, lane, a.scale, keep,
                                                                 oc + fc * (long long)F, dc + fc * (long long)F);
                        }
                    }
                   } else {
#pragma unroll
                    for (int m = 0; m < FPT; m += G) {
                        const int j0 = ((m + 1) * HOP - NFFT) / 128;          // compile-time after unrolling
                        auto reg = [&](int j) -> v2f { return j < 0 ? prv_[(PREV + j) < 0 ? 0 : (PREV + j)] : cur_[j < 0 ? 0 : j]; };
                        if constexpr (G == 1) {
                            const long long f = t * FPT + m + 1 - NFFT / HOP;
                            if ((j0 >= 0 || have_prev) && f >= 0 && f < a.n_valid && !(a.split && (f & 1))) {
                                v2f w[PPL];
#pragma unroll
                                for (int i = 0; i < PPL; i++) w[i] = reg(j0 + i);
                                if (STAMP) {
                                    STAMP_AT(10);                  // (between the frames)
                                    auto hook = [&](int n) { STAMP_AT(11 + n); };
                                    psd_frame<NFFT, LPF, R1, R2, R3, DB>(w, fb, tw2, tw3, twn, win, lane, a.scale, true,
                                                                         oc + f * (long long)F, dc + f * (long long)F, hook);
                                    STAMP_AT(15);                  // split step, PSD, stores
                                } else {
                                    psd_frame<NFFT, LPF, R1, R2, R3, DB>(w, fb, tw2, tw3, twn, win, lane, a.scale, true,
                                                                         oc + f * (long long)F, dc + f * (long long)F);
                                }
                            }
                        } else {
                            // two frames side by side: lanes 0-31 take frame m, lanes 32-63 frame m + 1 (HOP / 128 registers
                            // further on).  Value t of a lane is samples 2l + 64 t of ITS frame, i.e. register t / 2 of that
                            // frame, lower (t even) or upper (t odd) half of the wave: v_permlane32_swap of the two frames'
                            // registers gives {X.lo | Y.lo} and {X.hi | Y.hi} in one instruction per dword.
                            const int gq = lane / LPF;
                            const long long f = t * FPT + m + gq + 1 - NFFT / HOP;
                            const bool ok0 = (j0 >= 0 || have_prev), ok1 = (j0 + HOP / 128 >= 0 || have_prev);
                            const bool keep = (gq == 0 ? ok0 : ok1) && f >= 0 && f < a.n_valid;
                            const long long fa = t * FPT + m + 1 - NFFT / HOP;
                            if ((ok0 && fa >= 0 && fa < a.n_valid) || (ok1 && fa + 1 >= 0 && fa + 1 < a.n_valid)) {
                                v2f w[2 * PPL];
#pragma unroll
                                for (int u = 0; u < PPL; u++) {
                                    const v2f X = reg(j0 + u), Y = reg(j0 + HOP / 128 + u);
                                    const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_int(X.x), __float_as_int(Y.x), false, false);
                                    const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_int(X.y), __float_as_int(Y.y), false, false);
                                    w[2 * u] = (v2f){__int_as_float(rx[0]), __int_as_float(ry[0])};
                                    w[2 * u + 1] = (v2f){__int_as_float(rx[1]), __int_as_float(ry[1])};
                                }
                                const long long fc = keep ? f : 0;            // a masked group still needs a legal address
                                psd_frame<NFFT, LPF, R1, R2, R3, DB>(w, fb + gq * MP, tw2, tw3, twn, win, lane, a.scale, keep,
                                                                     oc + fc * (long long)F, dc + fc * (long long)F);
                            }
                        }
                    }
                   }
                  } else if (t == -12345) oc[lane] = cur_[0].x + cur_[9].y + prv_[0].x + bb_[0].x;
                }
                if constexpr (G == 4) {
#pragma unroll
                    for (int j = 0; j < 4; j++) pv_[j] = pvn_[j];
                } else {
#pragma unroll
                    for (int j = 0; j < PREV; j++) prv_[j] = cur_[16 - PREV + j];
                }
                have_prev = true;
            }
            STAMP_AT(10);                                      // the tile's (at most) two frames
        }
    }
    if (STAMP && lane == 0) {
        long long *dst = reinterpret_cast<long long *>(a.db) + ((long long)blockIdx.x * 2 * NP + wave) * 16;
#pragma unroll
        for (int i = 0; i < 16; i++) dst[i] = st_acc[i];
    }
#undef STAMP_AT
}

// ---- backward sweep of the batch chain with HALF of the spectrogram fused in ------------------------------
// The forward sweep is bound by VALU issue (its FFT waves are the critical path of every pair), the backward
// sweep by memory with its VALU half idle.  With "frame split" the forward sweep writes only the EVEN frames
// 2t (= tile t itself) and this kernel the ODD ones 2t+1 (second half of tile t + first half of tile t+1, which it
// walked one iteration earlier): both launches then carry one cascade pair and one FFT per tile and 10 bytes per
// sample (4 R + 4 W + 2 W of PSD).  A workgroup is NP IIR waves -- env_bwd_kernel<SE, true>'s walk, line for
// line, except that a tile goes into LDS RAW and is rectified on the way into the cascades -- and NP FFT waves
// with the same pairwise hand-over as in chain_fwd_kernel: H1 "the tile holds the filtered samples", H2 "copied"
// (the IIR wave may overwrite the tile with the forward cascade's outputs).  nfft 2048 / hop 1024 only.
struct ChainBwdArgs {
    BwdArgs b;
    float *psd;
    long long psd_pitch;
    long long n_valid;
    const float *tables;
    float scale;
    long long units;
    int debug;
    int *fault;
};

template <int SE, int NP>
__global__ __launch_bounds__(128 * NP, NP / 2) void chain_bwd_kernel(const SosPlanDev *__restrict__ P0, ChainBwdArgs a)
{
    constexpr int DE = 2 * SE;
    constexpr int NFFT = 2048, M = NFFT / 2, F = M + 1, MP = M + M / 16;
    constexpr int R1 = 16, R2 = 16, R3 = 4;
    constexpr int TW2 = (R2 - 1) * R1, TW3 = R1 * R2, TWN = M / 2 + 1, NTAB = TW2 + TW3 + TWN + M;
    __shared__ float4 tiles[NP][64 * 8];
    __shared__ float2 fbs[NP][MP];
    __shared__ float2 tab[NTAB];
    __shared__ int ready[NP], taken[NP];
    __shared__ int abort_wg;
    if (threadIdx.x < NP) { ready[threadIdx.x] = 0; taken[threadIdx.x] = 0; }
    if (threadIdx.x == 0) abort_wg = 0;
    bool gave_up = false;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int pair = wave < NP ? wave : wave - NP;
    {
        const float2 *src = reinterpret_cast<const float2 *>(a.tables);
        for (int i = tid; i < NTAB; i += 128 * NP) tab[i] = src[i];
    }
    __syncthreads();

    const long long unit = (long long)blockIdx.x * NP + pair;
    const bool unit_ok = unit < a.units;
    const int seg = unit_ok ? (int)(unit % a.b.n_seg) : 0;
    const long long ch = unit_ok ? unit / a.b.n_seg : 0;
    const long long T = a.b.T;
    const int edge = a.b.edge;
    // reversed tile index rt = n_tiles-1 - (p / TILE): the unit owns rt in [rt_lo, rt_hi)
    const long long rt_lo = (long long)seg * a.b.seg_tiles;
    long long rt_hi = rt_lo + a.b.seg_tiles;
    if (rt_hi > a.b.n_tiles) rt_hi = a.b.n_tiles;
    long long rt_start = rt_lo - a.b.warm_tiles;
    if (rt_start < 0) rt_start = 0;
    if (!unit_ok) rt_hi = rt_start;

    if (wave < NP) {
        // ================= IIR role: env_bwd_kernel<SE, true> with the hand-overs added ==================
        float4 *lds = tiles[pair];
        float *ldsf = reinterpret_cast<float *>(lds);
        const float *in = a.b.in + ch * a.b.in_pitch;
        float *out = a.b.out + ch * a.b.out_pitch;
        const double *ckpt = a.b.ckpt + ch * a.b.ckpt_pitch;
        double cb_[DE];
#pragma unroll
        for (int r = 0; r < DE; r++) cb_[r] = 0.0;
        v4f nx[8], nck[SE];
        bool pre = false;
        auto fetch = [&](long long tidx) {
#pragma unroll
            for (int k = 0; k < 8; k++) nx[k] = asm_load16(in + tidx * TILE + 256 * k + 4 * lane);
#pragma unroll
            for (int i = 0; i < SE; i++) nck[i] = asm_load16(ckpt + tidx * DE + 2 * i);
        };
        const long long top_full = T / TILE - 1;                     // host guarantees >= 0
        auto prefetchable = [&](long long tidx) { return tidx >= 0 && tidx <= top_full; };
        {
            const long long t0 = a.b.n_tiles - 1 - rt_start;
            pre = rt_start < rt_hi && prefetchable(t0);
            fetch(pre ? t0 : top_full);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const double rgain = a.b.gain;
        int it = 0;
        for (long long rt = rt_start; rt < rt_hi; rt++) {
            it++;
            const long long tidx = a.b.n_tiles - 1 - rt;
            const long long tile = tidx * TILE;
            double cfw_[DE];
            // the FFT wave must have its copy of the previous tile before this one goes into LDS
            if (it > 1) CHAIN_WAIT_FOR(taken, it - 1, it);
            const bool fast = pre;
            if (fast) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    asm volatile("" : "+v"(nx[k]));
                    lds[lds_slot(8 * k + (lane >> 3), lane & 7)] = make_float4(nx[k].x, nx[k].y, nx[k].z, nx[k].w);
                }
#pragma unroll
                for (int i = 0; i < SE; i++) {
                    asm volatile("" : "+v"(nck[i]));
                    const v2d d = __builtin_bit_cast(v2d, nck[i]);
                    cfw_[2 * i] = d.x; cfw_[2 * i + 1] = d.y;
                }
            } else {
                // a tile that touches T: untracked loads from clamped addresses, zeros past T; RAW, like the fast path
#pragma unroll 1
                for (int k = 0; k < 8; k++) {
                    const long long p = tile + 256 * k + 4 * lane;
                    v4f t;
                    t.x = asm_load4(in + (p < T ? p : T - 1));
                    t.y = asm_load4(in + (p + 1 < T ? p + 1 : T - 1));
                    t.z = asm_load4(in + (p + 2 < T ? p + 2 : T - 1));
                    t.w = asm_load4(in + (p + 3 < T ? p + 3 : T - 1));
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("" : "+v"(t));
                    lds[lds_slot(8 * k + (lane >> 3), lane & 7)] =
                        make_float4(p < T ? t.x : 0.f, p + 1 < T ? t.y : 0.f, p + 2 < T ? t.z : 0.f, p + 3 < T ? t.w : 0.f);
                }
                v4f ck[SE];
#pragma unroll
                for (int i = 0; i < SE; i++) ck[i] = asm_load16(ckpt + tidx * DE + 2 * i);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < SE; i++) {
                    asm volatile("" : "+v"(ck[i]));
                    const v2d d = __builtin_bit_cast(v2d, ck[i]);
                    cfw_[2 * i] = d.x; cfw_[2 * i + 1] = d.y;
                }
            }
            WAVE_SYNC();
            CHAIN_POST(ready, it);                                  // H1: the tile holds the filtered samples
            {
                pre = rt + 1 < rt_hi && prefetchable(tidx - 1);
                fetch(pre ? tidx - 1 : top_full);
            }
            if (fast) {
                // ---- forward cascade again, from the state that entered this tile; rectification on the way in.
                // Phase 1 only reads the tile: H2 is needed before phase 3 overwrites it.
#define CASC_S SE
#define CASC_PLAN() PLAN_OF(P0)
#define CASC_CARRY cfw_
#define CASC_IN(v) (a.b.rectify ? fabsf(v) : (v))
#define CASC_GAIN rgain
#define CASC_ROLLED_GROUPS
#define CASC_STAMP(n) do { if ((n) == 1) CHAIN_WAIT_FOR(taken, it, it); } while (0)
#include "sos_cascade.inc"
#undef CASC_STAMP
#undef CASC_IN
#undef CASC_CARRY
            } else {
                CHAIN_WAIT_FOR(taken, it, it);                      // H2, then the tile is this wave's alone
                if (a.b.rectify) {
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        float4 v = lds[lds_slot(lane, q)];
                        v = make_float4(fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w));
                        lds[lds_slot(lane, q)] = v;
                    }
                }
                float ra = asm_load4(in + (T - 1));
                float rb = asm_load4(in + (lane < edge ? T - 2 - lane : 0));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(ra));
                asm volatile("" : "+v"(rb));
                WAVE_SYNC();
                // right odd extension ext[T + i] = 2 r(T-1) - r(T-2-i), i < edge
                if (lane < edge) {
                    const long long pj = T + lane;
                    if (pj >= tile && pj < tile + TILE) {
                        if (a.b.rectify) { ra = fabsf(ra); rb = fabsf(rb); }
                        ldsf[lds_float_index((int)(pj - tile))] = 2.f * ra - rb;
                    }
                }
                WAVE_SYNC();
#define CASC_CARRY cfw_
#define CASC_IN(v) (v)
#include "sos_cascade.inc"
#undef CASC_IN
#undef CASC_CARRY
            }
            WAVE_SYNC();
            if (rt == 0) {
                // scipy: backward pass starts from zi * y_fwd[-1]; pad the rest of the tile with it
                const int last = (int)(T + edge - 1 - tile);
                const float v0 = ldsf[lds_float_index(last)];
                WAVE_SYNC();
                for (int s2 = last + 1 + lane; s2 < TILE; s2 += 64) ldsf[lds_float_index(s2)] = v0;
                const SosPlanDev *P = PLAN_OF(P0);
#pragma unroll
                for (int r = 0; r < DE; r++) cb_[r] = P->zi[r] * (double)v0;
                WAVE_SYNC();
            }
            // ---- backward cascade over the forward outputs, last sample first (the gain belongs to the forward one)
#undef CASC_GAIN
#define CASC_CARRY cb_
#define CASC_IN(v) (v)
#define CASC_REVERSED
#include "sos_cascade.inc"
#undef CASC_REVERSED
#undef CASC_ROLLED_GROUPS
#undef CASC_S
#undef CASC_PLAN
#undef CASC_CARRY
#undef CASC_IN
            WAVE_SYNC();
            if (rt >= rt_lo) {
                if (tile + TILE <= T) {
                    // interior tile: exactly 8 vector stores, then the counted wait for the prefetch
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        float4 v = lds[lds_slot(8 * k + (lane >> 3), lane & 7)];
                        if (a.b.clamp) {
                            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        }
                        f4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
                        *reinterpret_cast<f4u *>(out + (tile + 256 * k + 4 * lane)) = t;
                    }
                    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                } else {
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        float4 v = lds[lds_slot(8 * k + (lane >> 3), lane & 7)];
                        if (a.b.clamp) {
                            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        }
                        store_four(out, tile + 256 * k + 4 * lane, v, 0, T, 0);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // warm-up tile: no stores to count
            }
            WAVE_SYNC();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // ================= FFT role: frame 2t+1 = second half of tile t + first half of tile t+1 ===========
        __builtin_amdgcn_s_setprio(3);
        float2 *fb = fbs[pair];
        const float *tlf = reinterpret_cast<const float *>(tiles[pair]);
        const float2 *tw2 = tab, *tw3 = tab + TW2, *twn = tab + TW2 + TW3, *win = tab + TW2 + TW3 + TWN;
        float *oc = a.psd + ch * a.psd_pitch;
        v2f cur_[16], nxt_[8];                     // this tile; the first half of the tile walked before it
#pragma unroll
        for (int j = 0; j < 16; j++) cur_[j] = (v2f){0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; j++) nxt_[j] = (v2f){0.f, 0.f};
        bool have_next = false;
        int it = 0;
        for (long long rt = rt_start; rt < rt_hi; rt++) {
            it++;
            const long long tidx = a.b.n_tiles - 1 - rt;
            CHAIN_WAIT_FOR(ready, it, it);
#pragma unroll
            for (int j = 0; j < 16; j++)
                cur_[j] = *reinterpret_cast<const v2f *>(tlf + lds_float_index(2 * lane + 128 * j));
            {
                const bool withhold = (a.debug & 8) && blockIdx.x == 0 && pair == 0 && rt == rt_start;
                if (!withhold) CHAIN_POST(taken, it);
            }
            const long long f = 2 * tidx + 1;
            if (rt >= rt_lo && have_next && f < a.n_valid) {
                v2f w[16];
#pragma unroll
                for (int i = 0; i < 8; i++) { w[i] = cur_[8 + i]; w[8 + i] = nxt_[i]; }
                psd_frame<NFFT, 64, R1, R2, R3, false>(w, fb, tw2, tw3, twn, win, lane, a.scale, true, oc + f * (long long)F,
                                                       nullptr);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) nxt_[j] = cur_[j];
            have_next = true;
        }
    }
}

#undef CHAIN_WAIT_FOR
#undef CHAIN_POST
#undef CHAIN_FAIR

__global__ void zero_rows_kernel(float *__restrict__ y, long long y_pitch, long long n, float value)
{
    long long ch = blockIdx.y;
    float *yo = y + ch * y_pitch;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        yo[i] = value;
}

// the single-wave sweeps: up to "sos_waves_per_cu" (16) waves per CU, four SIMDs per CU
// ("sos_waves_min" = w, experiments: force w waves per CU by making every level below it cost the same)
void plan_segments(const hipdsp_ctx *ctx, long long N, long long channels, long long warm,
                   long long *seg_len, int *n_seg)
{
    const int w_max = ctx->sos_waves_per_cu > 0 ? ctx->sos_waves_per_cu : 16;
    hd_plan_segments_occ(ctx->n_cus, w_max, ctx->sos_waves_min >= w_max ? 0 : 4, ctx->max_segments, N, channels, warm, seg_len, n_seg);
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

int launch_scan(hipdsp_ctx *ctx, const SosPlanDev *dev, int S, SeqArgs a, long long channels, long long warm)
{
    if (a.gain == 0.0) a.gain = 1.0;
    plan_segments(ctx, a.N, channels, warm, &a.seg_len, &a.n_seg);
    a.units = channels * a.n_seg;
    long long blocks = (a.units + WPB - 1) / WPB;
    if (blocks > 0x7fffffffLL) {
        hipdsp_set_error("grid too large (%lld blocks)", blocks);
        return HIPDSP_ERR_INVALID;
    }
    dim3 grid((unsigned)blocks), block(64 * WPB);
    switch (S) {
    case 1: hipLaunchKernelGGL((sos_scan_kernel<1>), grid, block, 0, ctx->stream, dev, a); break;
    case 2: hipLaunchKernelGGL((sos_scan_kernel<2>), grid, block, 0, ctx->stream, dev, a); break;
    case 3: hipLaunchKernelGGL((sos_scan_kernel<3>), grid, block, 0, ctx->stream, dev, a); break;
    case 4: hipLaunchKernelGGL((sos_scan_kernel<4>), grid, block, 0, ctx->stream, dev, a); break;
    default:
        hipdsp_set_error("n_sections %d not in 1..%d", S, MAXS);
        return HIPDSP_ERR_UNSUPPORTED;
    }
    return hd_launch_status("sos_scan_kernel");
}

// Envelope by checkpoints: forward sweep (optionally with the band-pass in front), then the
// recomputing backward sweep.  `fplan` NULL: `x` is the trace to rectify.
int launch_env_ckpt(hipdsp_ctx *ctx, const SosPlanDev *fdev, const SosPlanDev *edev, int SF, int SE, long long warmF,
                    long long warmE, int edge, const float *x, long long x_pitch, float *yf,
                    long long yf_pitch, float *env, long long env_pitch, long long channels,
                    long long frames, long long skip, int rectify, double gain, int clamp, int phase)
{
    const long long n_tiles = (frames + edge + TILE - 1) / TILE;
    const long long ckpt_pitch = n_tiles * 2 * SE;
    void *work = nullptr;
    int rc = hipdsp_scratch(ctx, sizeof(double) * (size_t)ckpt_pitch * (size_t)channels, &work);
    if (rc != HIPDSP_OK) return rc;
    if (phase != 2) {
        CkptArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.in = x; fa.yf = yf; fa.ckpt = (double *)work;
        fa.in_pitch = x_pitch; fa.yf_pitch = yf_pitch; fa.ckpt_pitch = ckpt_pitch;
        fa.T = frames; fa.edge = edge; fa.rectify = rectify; fa.gain = rectify ? gain : 1.0;
        // only the band-pass warms up; the envelope's states are handed over exactly (env_fix_kernel), which needs
        // a cascade that forgets (a plan that does not decay is never cut into segments)
        long long warm = warmF;
        if (warmF >= (1LL << 40) || warmE >= (1LL << 40)) warm = 1LL << 50;
        plan_segments(ctx, frames, channels, warm, &fa.seg_len, &fa.n_seg);
        fa.units = channels * fa.n_seg;
        long long blocks = (fa.units + WPB - 1) / WPB;
        HD_REQUIRE(blocks <= 0x7fffffffLL, "grid too large");
        dim3 grid((unsigned)blocks), block(64 * WPB);
        const bool pf = ctx->sos_prefetch && frames >= 4 * TILE;
#define HD_CKPT(A, B)                                                                                       \
    case (A) * 8 + (B):                                                                                     \
        if (pf && (A) <= 2 && (B) <= 2)                                                                     \
            hipLaunchKernelGGL((sos_ckpt_kernel<A, B, ((A) <= 2 && (B) <= 2)>), grid, block, 0, ctx->stream, fdev, edev, fa); \
        else                                                                                                \
            hipLaunchKernelGGL((sos_ckpt_kernel<A, B, false>), grid, block, 0, ctx->stream, fdev, edev, fa); \
        break
        switch (SF * 8 + SE) {
            HD_CKPT(0, 1); HD_CKPT(0, 2); HD_CKPT(0, 3); HD_CKPT(0, 4);
            HD_CKPT(1, 1); HD_CKPT(1, 2); HD_CKPT(1, 3); HD_CKPT(1, 4);
            HD_CKPT(2, 1); HD_CKPT(2, 2); HD_CKPT(2, 3); HD_CKPT(2, 4);
            HD_CKPT(3, 1); HD_CKPT(3, 2); HD_CKPT(3, 3); HD_CKPT(3, 4);
            HD_CKPT(4, 1); HD_CKPT(4, 2); HD_CKPT(4, 3); HD_CKPT(4, 4);
        default:
            hipdsp_set_error("n_sections %d / %d not in 0..%d / 1..%d", SF, SE, MAXS, MAXS);
            return HIPDSP_ERR_UNSUPPORTED;
        }
#undef HD_CKPT
        rc = hd_launch_status("sos_ckpt_kernel");
        if (rc != HIPDSP_OK) return rc;
        rc = launch_env_fix(ctx, edev, SE, (double *)work, ckpt_pitch, channels, fa.n_seg, fa.seg_len, n_tiles);
        if (rc != HIPDSP_OK) return rc;
    }
    if (phase == 1) return HIPDSP_OK;
    if (ctx->mid_event) HD_CHECK_HIP(hipEventRecord(ctx->mid_event, ctx->stream));
    if (frames - skip == 0) return HIPDSP_OK;
    BwdArgs b;
    memset(&b, 0, sizeof(b));
    b.in = SF > 0 ? yf : x; b.in_pitch = SF > 0 ? yf_pitch : x_pitch;
    b.out = env; b.out_pitch = env_pitch;
    b.ckpt = (const double *)work; b.ckpt_pitch = ckpt_pitch;
    b.T = frames; b.skip = skip; b.n_tiles = n_tiles; b.edge = edge;
    b.rectify = rectify; b.clamp = clamp; b.gain = rectify ? gain : 1.0;
    b.trace = ctx->sos_trace;
    b.trace_rows = ctx->sos_trace_rows;
    b.debug = ctx->sos_debug;
    b.fair = ctx->sos_fair;
    const long long used_tiles = n_tiles - skip / TILE;      // tiles below `skip` are never visited
    long long seg_len = 0;
    plan_segments(ctx, used_tiles * TILE, channels, warmE, &seg_len, &b.n_seg);
    b.seg_tiles = seg_len / TILE;
    b.warm_tiles = warmE / TILE;
    b.units = channels * b.n_seg;
    long long blocks = (b.units + WPB - 1) / WPB;            // four waves per workgroup: one per SIMD of a CU
    HD_REQUIRE(blocks <= 0x7fffffffLL, "grid too large");
    dim3 grid((unsigned)blocks), block(64 * WPB);
    if (ctx->sos_single_wave_wg) { grid = dim3((unsigned)b.units); block = dim3(64); }
    if (ctx->sos_single_wave_wg && SE == 1 && ctx->sos_prefetch && frames >= 4 * TILE && !ctx->sos_trace && !ctx->sos_no_pin) {
        hipLaunchKernelGGL((env_bwd_kernel<1, true, true, false, 1>), grid, block, 0, ctx->stream, edev, b);   // A/B
        return hd_launch_status("env_bwd_kernel");
    }
    if (ctx->sos_single_wave_wg) { grid = dim3((unsigned)blocks); block = dim3(64 * WPB); }
    switch (SE) {
    case 1:
        if (ctx->sos_prefetch && frames >= 4 * TILE) {
            if (ctx->sos_trace) hipLaunchKernelGGL((env_bwd_kernel<1, true, true, true>), grid, block, 0, ctx->stream, edev, b);
            else if (ctx->sos_no_pin) hipLaunchKernelGGL((env_bwd_kernel<1, true, false>), grid, block, 0, ctx->stream, edev, b);
            else hipLaunchKernelGGL((env_bwd_kernel<1, true>), grid, block, 0, ctx->stream, edev, b);
        } else hipLaunchKernelGGL((env_bwd_kernel<1, false>), grid, block, 0, ctx->stream, edev, b);
        break;
    case 2:
        if (ctx->sos_prefetch && frames >= 4 * TILE) hipLaunchKernelGGL((env_bwd_kernel<2, true>), grid, block, 0, ctx->stream, edev, b);
        else hipLaunchKernelGGL((env_bwd_kernel<2, false>), grid, block, 0, ctx->stream, edev, b);
        break;
    case 3: hipLaunchKernelGGL((env_bwd_kernel<3, false>), grid, block, 0, ctx->stream, edev, b); break;
    case 4: hipLaunchKernelGGL((env_bwd_kernel<4, false>), grid, block, 0, ctx->stream, edev, b); break;
    default:
        hipdsp_set_error("n_sections %d not in 1..%d", SE, MAXS);
        return HIPDSP_ERR_UNSUPPORTED;
    }
    return hd_launch_status("env_bwd_kernel");
}

}  // namespace

struct hipdsp_sosplan {
    SosPlanDev *host;      // pinned
    SosPlanDev *dev;
    hipEvent_t uploaded;
    bool valid;
};

extern "C" {

int hipdsp_sosplan_create(hipdsp_ctx *ctx, hipdsp_sosplan **out)
{
    HD_REQUIRE(ctx != nullptr && out != nullptr, "NULL argument");
    *out = nullptr;
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    hipdsp_sosplan *p = new hipdsp_sosplan();
    p->host = nullptr; p->dev = nullptr; p->uploaded = nullptr; p->valid = false;
    hipError_t e = hipHostMalloc((void **)&p->host, sizeof(SosPlanDev), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void **)&p->dev, sizeof(SosPlanDev));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->uploaded, hipEventDisableTiming);
    if (e != hipSuccess) {
        if (p->host) (void)hipHostFree(p->host);
        if (p->dev) (void)hipFree(p->dev);
        delete p;
        HD_CHECK_HIP(e);
    }
    memset(p->host, 0, sizeof(SosPlanDev));
    *out = p;
    return HIPDSP_OK;
}

int hipdsp_sosplan_destroy(hipdsp_ctx *ctx, hipdsp_sosplan *plan)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (!plan) return HIPDSP_OK;
    (void)hipStreamSynchronize(ctx->stream);
    if (plan->uploaded) (void)hipEventDestroy(plan->uploaded);
    if (plan->host) (void)hipHostFree(plan->host);
    if (plan->dev) (void)hipFree(plan->dev);
    delete plan;
    return HIPDSP_OK;
}

int hipdsp_sosplan_set_host(hipdsp_ctx *ctx, hipdsp_sosplan *plan, const double *host_sos,
                            int n_sections)
{
    HD_REQUIRE(ctx != nullptr && plan != nullptr && host_sos != nullptr, "NULL argument");
    if (n_sections < 1 || n_sections > MAXS) {
        hipdsp_set_error("n_sections %d not in 1..%d (cascade longer filters over several plans)",
                         n_sections, MAXS);
        return HIPDSP_ERR_UNSUPPORTED;
    }
    // the previous upload must have left the pinned staging block
    if (plan->valid) HD_CHECK_HIP(hipEventSynchronize(plan->uploaded));
    SosPlanDev tmp;
    int rc = hd_fill_plan(&tmp, host_sos, n_sections);
    if (rc != HIPDSP_OK) return rc;
    memcpy(plan->host, &tmp, sizeof(tmp));
    return HIPDSP_OK;
}

int hipdsp_sosplan_upload(hipdsp_ctx *ctx, hipdsp_sosplan *plan)
{
    HD_REQUIRE(ctx != nullptr && plan != nullptr, "NULL argument");
    HD_REQUIRE(plan->host->n_sections > 0, "plan has no coefficients yet");
    HD_CHECK_HIP(hipMemcpyAsync(plan->dev, plan->host, sizeof(SosPlanDev), hipMemcpyHostToDevice,
                                ctx->stream));
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (ctx->stream) (void)hipStreamIsCapturing(ctx->stream, &st);
    if (st == hipStreamCaptureStatusNone) {
        HD_CHECK_HIP(hipEventRecord(plan->uploaded, ctx->stream));
        plan->valid = true;
    }
    return HIPDSP_OK;
}

int hipdsp_sosplan_set(hipdsp_ctx *ctx, hipdsp_sosplan *plan, const double *host_sos, int n_sections)
{
    int rc = hipdsp_sosplan_set_host(ctx, plan, host_sos, n_sections);
    if (rc != HIPDSP_OK) return rc;
    return hipdsp_sosplan_upload(ctx, plan);
}

int hipdsp_sosplan_info(hipdsp_ctx *ctx, hipdsp_sosplan *plan, int64_t *warmup, int *edge)
{
    HD_REQUIRE(ctx != nullptr && plan != nullptr, "NULL argument");
    if (warmup) *warmup = plan->host->warm;
    if (edge) *edge = plan->host->edge;
    return HIPDSP_OK;
}

int hipdsp_sosfilt(hipdsp_ctx *ctx, const hipdsp_sosplan *plan, const float *x, int64_t x_pitch,
                   float *y, int64_t y_pitch, int64_t channels, int64_t frames, int64_t skip)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_REQUIRE(channels >= 0 && frames >= 0, "negative size");
    HD_REQUIRE(skip >= 0 && skip <= frames, "skip %lld not in [0, frames=%lld]", (long long)skip,
               (long long)frames);
    if (channels == 0 || frames - skip == 0) return HIPDSP_OK;
    HD_REQUIRE(x != nullptr && y != nullptr, "NULL data pointer");
    HD_REQUIRE(x_pitch >= frames && y_pitch >= frames - skip, "pitch smaller than row length");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    if (plan == nullptr) {
        long long n = frames - skip;
        unsigned gx = (unsigned)((n + 1023) / 1024 > 4096 ? 4096 : (n + 1023) / 1024);
        hipLaunchKernelGGL(copy_skip_kernel, dim3(gx, (unsigned)channels), dim3(256), 0, ctx->stream, x,
                           (long long)x_pitch, y, (long long)y_pitch, n, (long long)skip);
        return hd_launch_status("copy_skip_kernel");
    }
    HD_REQUIRE(plan->host->n_sections > 0, "plan has no coefficients");
    SeqArgs a;
    memset(&a, 0, sizeof(a));
    a.in = x; a.out = y; a.in_pitch = x_pitch; a.out_pitch = y_pitch;
    a.N = frames; a.skip = skip;
    return launch_scan(ctx, plan->dev, plan->host->n_sections, a, channels, plan->host->warm);
}

int hipdsp_sosfilt_envelope(hipdsp_ctx *ctx, const hipdsp_sosplan *fplan, const hipdsp_sosplan *eplan,
                            const float *x, int64_t x_pitch, float *yf, int64_t yf_pitch, float *env,
                            int64_t env_pitch, int64_t channels, int64_t frames, int rectify, double gain,
                            int clamp, int phase)
{
    HD_REQUIRE(ctx != nullptr && fplan != nullptr && eplan != nullptr, "NULL argument");
    HD_REQUIRE(phase >= 0 && phase <= 2, "phase must be 0 (both), 1 (forward) or 2 (backward)");
    HD_REQUIRE(channels >= 0 && frames >= 0, "negative size");
    HD_REQUIRE(fplan->host->n_sections > 0 && eplan->host->n_sections > 0, "plan has no coefficients");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    {   // phase 2 consumes what a forward sweep left behind: not if that sweep reported a fault
        const int frc = hd_device_fault(ctx);
        if (frc != HIPDSP_OK) return frc;
    }
    const int edge = eplan->host->edge;
    if (frames <= edge) {
        hipdsp_set_error("The length of the input vector x must be greater than padlen, which is %d.", edge);
        return HIPDSP_ERR_TOO_SHORT;
    }
    if (channels == 0) return HIPDSP_OK;
    HD_REQUIRE(x != nullptr && yf != nullptr && env != nullptr, "NULL data pointer");
    HD_REQUIRE(x_pitch >= frames && yf_pitch >= frames && env_pitch >= frames, "pitch smaller than row length");
    return launch_env_ckpt(ctx, fplan->dev, eplan->dev, fplan->host->n_sections, eplan->host->n_sections,
                           fplan->host->warm, eplan->host->warm, edge, x, x_pitch, yf, yf_pitch, env, env_pitch,
                           channels, frames, 0, rectify, gain, clamp, phase);
}

int hipdsp_chain_forward(hipdsp_ctx *ctx, const hipdsp_sosplan *fplan, const hipdsp_sosplan *eplan,
                         const float *x, int64_t x_pitch, float *yf, int64_t yf_pitch, int64_t channels,
                         int64_t frames, int rectify, double gain, int nfft, int hop, double fs, float *psd,
                         float *db_out, int64_t frames_out, int64_t psd_pitch, int64_t spec_frames)
{
    HD_REQUIRE(ctx != nullptr && fplan != nullptr, "NULL argument");
    HD_REQUIRE(channels >= 0 && frames >= 0 && frames_out >= 0, "negative size");
    HD_REQUIRE(spec_frames >= 0 && spec_frames <= frames, "spec_frames %lld not in [0, frames=%lld]",
               (long long)spec_frames, (long long)frames);
    HD_REQUIRE(fs > 0, "fs must be positive");
    const int SF = fplan->host->n_sections, SE = eplan ? eplan->host->n_sections : 0;
    HD_REQUIRE(SF > 0 && (SE > 0 || eplan == nullptr), "plan has no coefficients");
    // shapes the kernel is built for: frames that are register windows of a 2048-sample tile
    const bool shape_ok = (nfft == 2048 && (hop == 1024 || hop == 512)) || (nfft == 1024 && (hop == 512 || hop == 256)) ||
                          (nfft == 512 && hop == 256) || (nfft == 256 && hop == 128);
    if (!shape_ok || SF > 4 || SE > 2 || frames < 4 * TILE ||
        fplan->host->warm >= (1LL << 40) || (eplan && eplan->host->warm >= (1LL << 40))) {
        hipdsp_set_error("the fused forward sweep covers nfft/hop 2048/1024, 2048/512, 1024/512, 1024/256, 512/256 and 256/128, a band-pass of "
                         "at most four and an envelope of at most two sections that decay, and traces of at least %d "
                         "frames: use hipdsp_sosfilt_envelope + hipdsp_spectrogram", 4 * TILE);
        return HIPDSP_ERR_UNSUPPORTED;
    }
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    {   // an earlier launch on this context may have reported a fault that nobody has looked at yet
        const int frc = hd_device_fault(ctx);
        if (frc != HIPDSP_OK) return frc;
    }
    const int edge = eplan ? eplan->host->edge : 0;
    if (channels == 0) return HIPDSP_OK;
    HD_REQUIRE(channels <= 65535, "more than 65535 channels");     // grid.y of the zero-tail launch
    HD_REQUIRE(x != nullptr && yf != nullptr && (psd != nullptr || frames_out == 0), "NULL data pointer");
    HD_REQUIRE(x_pitch >= frames && yf_pitch >= frames, "pitch smaller than row length");
    const long long F = nfft / 2 + 1;
    if (psd_pitch == 0) psd_pitch = frames_out * F;
    HD_REQUIRE(psd_pitch >= frames_out * F, "psd_pitch smaller than one channel");
    // frames inside the trace, as in hipdsp_spectrogram (bufferedspectrogram.py:46-49); the spectrogram may be
    // handed fewer samples than the filter produces (spec_frames: BufferedData.load_buffer's one frame "after")
    const long long sframes = spec_frames > 0 ? spec_frames : frames;
    long long nsource = (frames_out - 1) * (long long)hop + nfft;
    if (nsource > sframes) nsource = sframes;
    long long n_valid = 0;
    if (frames_out > 0 && nsource >= nfft) n_valid = (nsource - (nfft - hop)) / hop;
    if (n_valid > frames_out) n_valid = frames_out;
    double wss = 0.0;
    for (int i = 0; i < nfft; i++) {
        const double w = 0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)nfft);
        wss += w * w;
    }
    ChainArgs a;
    memset(&a, 0, sizeof(a));
    int rc = hd_fft_tables(ctx, nfft, &a.tables);
    if (rc != HIPDSP_OK) return rc;
    const long long n_tiles = (frames + edge + TILE - 1) / TILE;
    const long long ckpt_pitch = n_tiles * 2 * SE;                 // the layout the backward sweep expects
    void *work = nullptr;
    if (SE > 0) {
        rc = hipdsp_scratch(ctx, sizeof(double) * (size_t)ckpt_pitch * (size_t)channels, &work);
        if (rc != HIPDSP_OK) return rc;
    }
    a.c.in = x; a.c.yf = yf; a.c.ckpt = (double *)work;
    a.c.in_pitch = x_pitch; a.c.yf_pitch = yf_pitch; a.c.ckpt_pitch = ckpt_pitch;
    a.c.T = frames; a.c.edge = edge; a.c.rectify = rectify; a.c.gain = rectify ? gain : 1.0;
    a.psd = psd; a.db = db_out; a.psd_pitch = psd_pitch; a.n_valid = n_valid;
    a.scale = (float)(1.0 / (fs * wss));
    a.warm_total = fplan->host->warm;              // the envelope's states are handed over exactly (env_fix_kernel)
    a.debug = ctx->chain_debug;
    a.fault = ctx->fault_dev;
    a.split = (ctx->chain_split_frames && nfft == 2048 && hop == 1024 && !db_out) ? 1 : 0;
    constexpr int P = 8;                                           // IIR waves (and FFT waves) per workgroup, one per CU
    // (hipdsp_chain_plan reports exactly this segmentation)
    plan_segments_chain(ctx, frames, channels, a.warm_total, &a.c.seg_len, &a.c.n_seg);
    a.units = channels * a.c.n_seg;
    a.n_iter = (int)((a.warm_total + a.c.seg_len + edge + TILE - 1) / TILE) + 1;
    long long blocks = (a.units + P - 1) / P;
    {
        const long long cus = ctx->n_cus - ctx->chain_reserve_cus > 0 ? ctx->n_cus - ctx->chain_reserve_cus : 1;
        if (a.units < cus * P) {                                   // not every pair of the chip gets a unit: spread them
            blocks = a.units < cus ? a.units : cus;
            a.unit_stride = blocks;
        }
    }
    HD_REQUIRE(blocks <= 0x7fffffffLL, "grid too large");
    if (frames_out > n_valid) {                                    // zero tail (bufferedspectrogram.py:59)
        const long long n = (frames_out - n_valid) * F;
        unsigned gx = (unsigned)((n + 1023) / 1024 > 4096 ? 4096 : (n + 1023) / 1024);
        hipLaunchKernelGGL(zero_rows_kernel, dim3(gx, (unsigned)channels), dim3(256), 0, ctx->stream,
                           psd + n_valid * F, (long long)psd_pitch, n, 0.f);
        if (db_out && !(ctx->chain_debug & 32))          // (bit 32: db_out is the stamp buffer of the diagnostic build)
            hipLaunchKernelGGL(zero_rows_kernel, dim3(gx, (unsigned)channels), dim3(256), 0, ctx->stream,
                               db_out + n_valid * F, (long long)psd_pitch, n, -INFINITY);
    }
    dim3 grid((unsigned)blocks), block(128 * P);
    const SosPlanDev *edev_ = eplan ? eplan->dev : nullptr;
    const bool flags = (ctx->chain_debug & 4) == 0;       // bit 4: workgroup barriers instead of the pairwise flags
    if ((ctx->chain_debug & 32) && db_out && SF == 2 && SE == 1 && flags && nfft == 2048 && hop == 1024) {
        // diagnostic build: db_out receives 16 clock sums per wave (needs >= blocks * 16 * 16 * 8 bytes)
        hipLaunchKernelGGL((chain_fwd_kernel<2, 1, P, true, false, 2048, 1024, true>), grid, block, 0, ctx->stream,
                           fplan->dev, edev_, a);
        return hd_launch_status("chain_fwd_kernel");
    }
    // nfft 2048 / hop 1024 with plans of up to two sections: every variant (barriers for the tests, fused dB
    // output for the display path); the other shapes and longer band-passes: pairwise flags, PSD only
#define HD_CHAIN_FULL(A, B)                                                                                         \
    case (A) * 8 + (B):                                                                                            \
        if (db_out) {                                                                                              \
            if (flags) hipLaunchKernelGGL((chain_fwd_kernel<A, B, P, true, true>), grid, block, 0, ctx->stream, fplan->dev, edev_, a);  \
            else hipLaunchKernelGGL((chain_fwd_kernel<A, B, P, false, true>), grid, block, 0, ctx->stream, fplan->dev, edev_, a);       \
        } else {                                                                                                   \
            if (flags) hipLaunchKernelGGL((chain_fwd_kernel<A, B, P, true, false>), grid, block, 0, ctx->stream, fplan->dev, edev_, a); \
            else hipLaunchKernelGGL((chain_fwd_kernel<A, B, P, false, false>), grid, block, 0, ctx->stream, fplan->dev, edev_, a);      \
        }                                                                                                          \
        break
#define HD_CHAIN_LONG(A, B)                                                                                         \
    case (A) * 8 + (B):                                                                                            \
        if (db_out) hipLaunchKernelGGL((chain_fwd_kernel<A, B, P, true, true>), grid, block, 0, ctx->stream, fplan->dev, edev_, a); \
        else hipLaunchKernelGGL((chain_fwd_kernel<A, B, P, true, false>), grid, block, 0, ctx->stream, fplan->dev, edev_, a);       \
        break
#define HD_CHAIN_SHAPE(A, B, N, H)                                                                                  \
    case (A) * 8 + (B):                                                                                            \
        if (db_out) hipLaunchKernelGGL((chain_fwd_kernel<A, B, P, true, true, N, H>), grid, block, 0, ctx->stream, fplan->dev, edev_, a); \
        else hipLaunchKernelGGL((chain_fwd_kernel<A, B, P, true, false, N, H>), grid, block, 0, ctx->stream, fplan->dev, edev_, a); \
        break
#define HD_CHAIN_ALL(N, H)                                                                                          \
    switch (SF * 8 + SE) {                                                                                         \
        HD_CHAIN_SHAPE(1, 0, N, H); HD_CHAIN_SHAPE(2, 0, N, H); HD_CHAIN_SHAPE(3, 0, N, H); HD_CHAIN_SHAPE(4, 0, N, H); \
        HD_CHAIN_SHAPE(1, 1, N, H); HD_CHAIN_SHAPE(1, 2, N, H); HD_CHAIN_SHAPE(2, 1, N, H); HD_CHAIN_SHAPE(2, 2, N, H); \
        HD_CHAIN_SHAPE(3, 1, N, H); HD_CHAIN_SHAPE(3, 2, N, H); HD_CHAIN_SHAPE(4, 1, N, H); HD_CHAIN_SHAPE(4, 2, N, H); \
    }
    if (nfft == 2048 && hop == 1024) {
        if ((!flags) && (SF > 2 || SE == 0)) {
            hipdsp_set_error("the barrier variant of the fused sweep is built for plans of one or two sections");
            return HIPDSP_ERR_UNSUPPORTED;
        }
        switch (SF * 8 + SE) {
            HD_CHAIN_FULL(1, 1); HD_CHAIN_FULL(1, 2); HD_CHAIN_FULL(2, 1); HD_CHAIN_FULL(2, 2);
            HD_CHAIN_LONG(3, 1); HD_CHAIN_LONG(3, 2); HD_CHAIN_LONG(4, 1); HD_CHAIN_LONG(4, 2);
            HD_CHAIN_LONG(1, 0); HD_CHAIN_LONG(2, 0); HD_CHAIN_LONG(3, 0); HD_CHAIN_LONG(4, 0);
        }
    } else if (nfft == 2048 && hop == 512) {
        HD_CHAIN_ALL(2048, 512)
    } else if (nfft == 1024 && hop == 512) {
        HD_CHAIN_ALL(1024, 512)
    } else if (nfft == 1024 && hop == 256) {
        HD_CHAIN_ALL(1024, 256)
    } else if (nfft == 512) {
        HD_CHAIN_ALL(512, 256)
    } else {
        HD_CHAIN_ALL(256, 128)
    }
#undef HD_CHAIN_FULL
#undef HD_CHAIN_LONG
#undef HD_CHAIN_SHAPE
#undef HD_CHAIN_ALL
    rc = hd_launch_status("chain_fwd_kernel");
    if (rc != HIPDSP_OK || SE == 0) return rc;
    return launch_env_fix(ctx, eplan->dev, SE, (double *)work, ckpt_pitch, channels, a.c.n_seg, a.c.seg_len, n_tiles);
}

int hipdsp_envelope_multi(hipdsp_ctx *ctx, const hipdsp_sosplan *const *plans, int n_plans, const float *x,
                          int64_t x_pitch, float *y, int64_t y_pitch, int64_t channels, int64_t frames, int64_t skip,
                          int rectify, double gain, int clamp)
{
    HD_REQUIRE(ctx != nullptr && plans != nullptr, "NULL argument");
    HD_REQUIRE(n_plans >= 1 && n_plans <= 16, "n_plans %d not in 1..16", n_plans);
    HD_REQUIRE(channels >= 0 && frames >= 0, "negative size");
    HD_REQUIRE(skip >= 0 && skip <= frames, "skip %lld not in [0, frames=%lld]", (long long)skip, (long long)frames);
    HD_REQUIRE(channels <= 65535, "more than 65535 channels");
    // pad length and the DC gain in front of every plan, from the whole cascade (scipy sosfiltfilt / sosfilt_zi)
    int total = 0, nb = 0, na = 0;
    double gain_before[16];
    double g = 1.0;
    for (int p = 0; p < n_plans; p++) {
        HD_REQUIRE(plans[p] != nullptr && plans[p]->host->n_sections > 0, "plan %d has no coefficients", p);
        gain_before[p] = g;
        const SosPlanDev *h = plans[p]->host;
        for (int sct = 0; sct < h->n_sections; sct++) {
            const double b0 = h->coef[sct][0], b1 = h->coef[sct][1], b2 = h->coef[sct][2];
            const double a1 = h->coef[sct][3], a2 = h->coef[sct][4];
            if (b2 == 0.0) nb++;
            if (a2 == 0.0) na++;
            g *= (b0 + b1 + b2) / (1.0 + a1 + a2);
        }
        total += h->n_sections;
    }
    const int edge = 3 * (2 * total + 1 - (nb < na ? nb : na));
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    if (frames <= edge) {
        hipdsp_set_error("The length of the input vector x must be greater than padlen, which is %d.", edge);
        return HIPDSP_ERR_TOO_SHORT;
    }
    if (channels == 0 || frames - skip == 0) return HIPDSP_OK;
    HD_REQUIRE(x != nullptr && y != nullptr, "NULL data pointer");
    HD_REQUIRE(x_pitch >= frames && y_pitch >= frames - skip, "pitch smaller than row length");
    const long long N = frames + 2LL * edge;
    float *buf[2] = {nullptr, nullptr}, *ref = nullptr;
    int rc = hipdsp_malloc(ctx, sizeof(float) * (size_t)N * (size_t)channels, (void **)&buf[0]);
    if (rc == HIPDSP_OK) rc = hipdsp_malloc(ctx, sizeof(float) * (size_t)N * (size_t)channels, (void **)&buf[1]);
    if (rc == HIPDSP_OK) rc = hipdsp_malloc(ctx, sizeof(float) * (size_t)channels, (void **)&ref);
    auto cleanup = [&]() {
        (void)hipdsp_free(ctx, buf[0]);
        (void)hipdsp_free(ctx, buf[1]);
        (void)hipdsp_free(ctx, ref);
    };
    if (rc != HIPDSP_OK) { cleanup(); return rc; }
    const unsigned gx = (unsigned)((N + 1023) / 1024 > 4096 ? 4096 : (N + 1023) / 1024);
    const dim3 grid(gx, (unsigned)channels);
    hipLaunchKernelGGL(odd_ext_kernel, grid, dim3(256), 0, ctx->stream, x, (long long)x_pitch, (long long)frames, edge,
                       rectify, buf[0], N);
    const double in_gain = rectify ? gain : 1.0;           // what the forward pass is fed is in_gain * buf[0]
    int cur = 0;
    for (int pass = 0; pass < 2 && rc == HIPDSP_OK; pass++) {
        // initial state of every section: its zi times the first sample of what the whole cascade is fed
        hipLaunchKernelGGL(first_sample_kernel, dim3((unsigned)((channels + 255) / 256)), dim3(256), 0, ctx->stream,
                           buf[cur], N, (long long)channels, ref);
        for (int p = 0; p < n_plans && rc == HIPDSP_OK; p++) {
            SeqArgs a;
            memset(&a, 0, sizeof(a));
            a.in = buf[cur]; a.out = buf[cur ^ 1]; a.in_pitch = N; a.out_pitch = N;
            a.N = N; a.skip = 0;
            // forward pass: the input gain on the first plan's numerator, and in every plan's initial state
            a.zi_ref = ref; a.zi_ref_pitch = 1; a.zi_scale = gain_before[p] * (pass == 0 ? in_gain : 1.0);
            a.gain = (pass == 0 && p == 0) ? in_gain : 1.0;
            rc = launch_scan(ctx, plans[p]->dev, plans[p]->host->n_sections, a, channels, plans[p]->host->warm);
            cur ^= 1;
        }
        if (rc != HIPDSP_OK) break;
        if (pass == 0) {
            hipLaunchKernelGGL(flip_kernel, grid, dim3(256), 0, ctx->stream, buf[cur], N, N, 0LL, N, 0, buf[cur ^ 1], N);
            cur ^= 1;
        } else {
            // undo the reversal, drop the extensions and the first `skip` frames, clamp
            const long long n = frames - skip;
            const unsigned gy = (unsigned)((n + 1023) / 1024 > 4096 ? 4096 : (n + 1023) / 1024);
            hipLaunchKernelGGL(flip_kernel, dim3(gy, (unsigned)channels), dim3(256), 0, ctx->stream, buf[cur], N, N,
                               (long long)edge + (long long)skip, n, clamp, y, (long long)y_pitch);
        }
    }
    if (rc == HIPDSP_OK) rc = hd_launch_status("envelope_multi kernels");
    cleanup();
    return rc;
}

int hipdsp_chain_backward(hipdsp_ctx *ctx, const hipdsp_sosplan *eplan, const float *yf, int64_t yf_pitch, float *env,
                          int64_t env_pitch, int64_t channels, int64_t frames, int rectify, double gain, int clamp,
                          int nfft, int hop, double fs, float *psd, int64_t frames_out, int64_t psd_pitch)
{
    HD_REQUIRE(ctx != nullptr && eplan != nullptr, "NULL argument");
    HD_REQUIRE(channels >= 0 && frames >= 0 && frames_out >= 0, "negative size");
    HD_REQUIRE(fs > 0, "fs must be positive");
    const int SE = eplan->host->n_sections;
    HD_REQUIRE(SE > 0, "plan has no coefficients");
    if (nfft != 2048 || hop != 1024 || SE > 2 || frames < 4 * TILE || eplan->host->warm >= (1LL << 40)) {
        hipdsp_set_error("the backward sweep with the odd frames fused in covers nfft 2048 / hop 1024, envelope plans of at "
                         "most two decaying sections and traces of at least %d frames", 4 * TILE);
        return HIPDSP_ERR_UNSUPPORTED;
    }
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    {
        const int frc = hd_device_fault(ctx);
        if (frc != HIPDSP_OK) return frc;
    }
    const int edge = eplan->host->edge;
    if (frames <= edge) {
        hipdsp_set_error("The length of the input vector x must be greater than padlen, which is %d.", edge);
        return HIPDSP_ERR_TOO_SHORT;
    }
    if (channels == 0) return HIPDSP_OK;
    HD_REQUIRE(channels <= 65535, "more than 65535 channels");
    HD_REQUIRE(yf != nullptr && env != nullptr && (psd != nullptr || frames_out == 0), "NULL data pointer");
    HD_REQUIRE(yf_pitch >= frames && env_pitch >= frames, "pitch smaller than row length");
    const long long F = nfft / 2 + 1;
    if (psd_pitch == 0) psd_pitch = frames_out * F;
    HD_REQUIRE(psd_pitch >= frames_out * F, "psd_pitch smaller than one channel");
    long long nsource = (frames_out - 1) * (long long)hop + nfft;
    if (nsource > frames) nsource = frames;
    long long n_valid = 0;
    if (frames_out > 0 && nsource >= nfft) n_valid = (nsource - (nfft - hop)) / hop;
    if (n_valid > frames_out) n_valid = frames_out;
    double wss = 0.0;
    for (int i = 0; i < nfft; i++) {
        const double w = 0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)nfft);
        wss += w * w;
    }
    ChainBwdArgs a;
    memset(&a, 0, sizeof(a));
    int rc = hd_fft_tables(ctx, nfft, &a.tables);
    if (rc != HIPDSP_OK) return rc;
    const long long n_tiles = (frames + edge + TILE - 1) / TILE;
    const long long ckpt_pitch = n_tiles * 2 * SE;                 // where the forward sweep parked the tile states
    void *work = nullptr;
    rc = hipdsp_scratch(ctx, sizeof(double) * (size_t)ckpt_pitch * (size_t)channels, &work);
    if (rc != HIPDSP_OK) return rc;
    a.b.in = yf; a.b.in_pitch = yf_pitch;
    a.b.out = env; a.b.out_pitch = env_pitch;
    a.b.ckpt = (const double *)work; a.b.ckpt_pitch = ckpt_pitch;
    a.b.T = frames; a.b.skip = 0; a.b.n_tiles = n_tiles; a.b.edge = edge;
    a.b.rectify = rectify; a.b.clamp = clamp; a.b.gain = rectify ? gain : 1.0;
    constexpr int P = 8;
    long long seg_len = 0;
    plan_segments_chain(ctx, n_tiles * TILE, channels, eplan->host->warm, &seg_len, &a.b.n_seg);
    a.b.seg_tiles = seg_len / TILE;
    a.b.warm_tiles = eplan->host->warm / TILE;
    a.psd = psd; a.psd_pitch = psd_pitch; a.n_valid = n_valid;
    a.scale = (float)(1.0 / (fs * wss));
    a.units = channels * a.b.n_seg;
    a.debug = ctx->chain_debug;
    a.fault = ctx->fault_dev;
    const long long blocks = (a.units + P - 1) / P;
    HD_REQUIRE(blocks <= 0x7fffffffLL, "grid too large");
    dim3 grid((unsigned)blocks), block(128 * P);
    if (SE == 1) hipLaunchKernelGGL((chain_bwd_kernel<1, P>), grid, block, 0, ctx->stream, eplan->dev, a);
    else hipLaunchKernelGGL((chain_bwd_kernel<2, P>), grid, block, 0, ctx->stream, eplan->dev, a);
    return hd_launch_status("chain_bwd_kernel");
}

int hipdsp_chain_backward_plan(hipdsp_ctx *ctx, const hipdsp_sosplan *eplan, int64_t channels, int64_t frames,
                               int64_t *first_border, int64_t *segment_frames, int *n_segments)
{
    HD_REQUIRE(ctx != nullptr && eplan != nullptr, "NULL argument");
    HD_REQUIRE(first_border != nullptr && segment_frames != nullptr && n_segments != nullptr, "NULL output");
    HD_REQUIRE(channels >= 1 && frames >= 1, "bad size");
    HD_REQUIRE(eplan->host->n_sections > 0, "plan has no coefficients");
    const long long n_tiles = (frames + eplan->host->edge + TILE - 1) / TILE;
    long long len = 0;
    int n = 0;
    plan_segments_chain(ctx, n_tiles * TILE, channels, eplan->host->warm, &len, &n);
    // segment s (walked from the END of the trace) covers frames [n_tiles*TILE - (s+1)*len, n_tiles*TILE - s*len)
    *first_border = n_tiles * TILE - len;
    *segment_frames = len;
    *n_segments = n;
    return HIPDSP_OK;
}

int hipdsp_chain_plan(hipdsp_ctx *ctx, const hipdsp_sosplan *fplan, const hipdsp_sosplan *eplan,
                      int64_t channels, int64_t frames, int64_t *segment_frames, int *n_segments)
{
    HD_REQUIRE(ctx != nullptr && fplan != nullptr, "NULL argument");      // (eplan may be NULL: no envelope)
    HD_REQUIRE(segment_frames != nullptr && n_segments != nullptr, "NULL output");
    HD_REQUIRE(channels >= 1 && frames >= 1, "bad size");
    HD_REQUIRE(fplan->host->n_sections > 0 && (eplan == nullptr || eplan->host->n_sections > 0), "plan has no coefficients");
    long long len = 0;
    int n = 0;
    plan_segments_chain(ctx, frames, channels, fplan->host->warm, &len, &n);
    *segment_frames = len;
    *n_segments = n;
    return HIPDSP_OK;
}

int hipdsp_envelope(hipdsp_ctx *ctx, const hipdsp_sosplan *plan, const float *x, int64_t x_pitch,
                    float *y, int64_t y_pitch, int64_t channels, int64_t frames, int64_t skip,
                    int rectify, double gain, int clamp)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_REQUIRE(channels >= 0 && frames >= 0, "negative size");
    HD_REQUIRE(skip >= 0 && skip <= frames, "skip %lld not in [0, frames=%lld]", (long long)skip,
               (long long)frames);
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    if (plan == nullptr) {
        long long n = frames - skip;
        if (channels == 0 || n == 0) return HIPDSP_OK;
        HD_REQUIRE(y != nullptr && y_pitch >= n, "bad output");
        unsigned gx = (unsigned)((n + 1023) / 1024 > 4096 ? 4096 : (n + 1023) / 1024);
        hipLaunchKernelGGL(zero_rows_kernel, dim3(gx, (unsigned)channels), dim3(256), 0, ctx->stream, y,
                           (long long)y_pitch, n, 0.f);
        return hd_launch_status("zero_rows_kernel");
    }
    HD_REQUIRE(plan->host->n_sections > 0, "plan has no coefficients");
    const int edge = plan->host->edge;
    if (frames <= edge) {
        hipdsp_set_error("The length of the input vector x must be greater than padlen, which is %d.",
                         edge);
        return HIPDSP_ERR_TOO_SHORT;
    }
    if (channels == 0) return HIPDSP_OK;
    HD_REQUIRE(x != nullptr && y != nullptr, "NULL data pointer");
    HD_REQUIRE(x_pitch >= frames && y_pitch >= frames - skip, "pitch smaller than row length");
    return launch_env_ckpt(ctx, nullptr, plan->dev, 0, plan->host->n_sections, 0, plan->host->warm, edge, x,
                           x_pitch, nullptr, 0, y, y_pitch, channels, frames, skip, rectify, gain, clamp, 0);
}

}  // extern "C"
