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
#include "sos_device.h"
#include <cmath>
#include <vector>

namespace {
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
    if (a.flags != nullptr && lane == 0) a.flags[unit] = state_not_finite(carry) ? 1 : 0;   // FloodArgs
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
    // The envelope starts at p' = env0 (sos_device.h: GridShift): the unit whose range holds that tile starts it from
    // the true state there, units in front of it have no envelope work, units behind it start from zero state at `lo`.
    const long long lead = a.lead;
    const long long env_tile0 = a.env0 - a.env0 % TILE;
    const bool env_true = lo <= env_tile0 && env_tile0 < hi;
    const long long env_start = env_true ? env_tile0 : (lo > env_tile0 ? lo : (1LL << 62));
    long long start = SF > 0 ? lo : (env_start < hi ? env_start : hi);      // (no band-pass: nothing to do in front of the envelope)
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
        pre = start < loop_end && start + TILE <= T && start >= lead;
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
            // a tile that reaches past T (or holds the `lead` samples in front of the trace): untracked loads from
            // clamped addresses, zeros outside [lead, T)
#pragma unroll 1
            for (int k = 0; k < 8; k++) {
                const long long p = tile + 256 * k + 4 * lane;
                auto at = [&](long long q) { return in + (q < lead ? lead : (q < T ? q : T - 1)); };
                auto ok = [&](long long q) { return q >= lead && q < T; };
                v4f t;
                t.x = asm_load4(at(p));
                t.y = asm_load4(at(p + 1));
                t.z = asm_load4(at(p + 2));
                t.w = asm_load4(at(p + 3));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(t));
                lds[lds_slot(8 * k + (lane >> 3), lane & 7)] =
                    make_float4(ok(p) ? t.x : 0.f, ok(p + 1) ? t.y : 0.f, ok(p + 2) ? t.z : 0.f, ok(p + 3) ? t.w : 0.f);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++)
                lds[lds_slot(8 * k + (lane >> 3), lane & 7)] = load_four_from(in, tile + 256 * k + 4 * lane, lead, T);
        }
        WAVE_SYNC();
        if (PREFETCH) {
            const long long next = tile + TILE;
            pre = next < loop_end && next + TILE <= T;          // (next > 0 >= ... : never the tile with the lead)
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
                if (tile >= lo && tile + TILE <= hi && tile >= lead) {
                    // interior tile: exactly 8 vector stores, then the counted wait
                    v4f rows[8];
                    tile_rows_from_lds(lds, lane, rows);
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const v4f v = rows[k];
                        f4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
                        *reinterpret_cast<f4u *>(yf + tile + 256 * k + 4 * lane) = t;
                    }
                    if (PREFETCH) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                } else {
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        store_four(yf, tile + 256 * k + 4 * lane, lds[lds_slot(8 * k + (lane >> 3), lane & 7)],
                                   lo > lead ? lo : lead, hi, 0);
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
        if (env_true && tile == env_start && a.env0 > 0) {
            // the envelope starts inside this tile: left odd extension and its steady state in front of it, in the
            // tile itself (env_left_fill; chain_fwd_kernel has the same lines)
            const SosPlanDev *P = PLAN_OF(PE0);
            const float e0 = env_left_fill(ldsf, lane, (int)(a.env0 - tile), edge);
            const double x0 = a.gain * (double)e0;
#pragma unroll
            for (int r = 0; r < DE; r++) ce_[r] = P->zi[r] * x0;
        } else if (env_true && tile == env_start) {
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
        // (the trace's very last tile advances the state too: slot n_tiles receives the state the channel ends with,
        // where the backward sweep looks for a non-finite one, see FloodArgs)
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
            // (the last segment: into the slot behind the last tile)
            if (lane == 0) {
#pragma unroll
                for (int r = 0; r < DE; r++) ckpt[(tile / TILE + 1) * DE + r] = ce_[r];
            }
            break;
        }
        WAVE_SYNC();
    }
    if constexpr (SF > 0) {
        if (a.flags != nullptr && lane == 0) a.flags[unit] = state_not_finite(cf_) ? 1 : 0;   // FloodArgs
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
// Non-finite input (FloodArgs): the block first overwrites what the band-pass segments behind a non-finite one wrote,
// and if any segment's envelope state ended non-finite it leaves NaN in slot n_tiles for the backward sweep.
template <int SE>
__global__ __launch_bounds__(256) void env_fix_kernel(const SosPlanDev *__restrict__ P0, double *ckpt_all, long long ckpt_pitch,
                                                      int n_seg, long long seg_tiles, long long n_tiles, FloodArgs flood,
                                                      int first_seg)
{
    // (first_seg: the segment the envelope starts in -- it holds the true states; the segments in front of it left nothing)
    constexpr int D = 2 * SE;
    double *ckpt = ckpt_all + (long long)blockIdx.x * ckpt_pitch;
    flood_channel(flood, blockIdx.x);
    {
        __shared__ int any_bad;
        if (threadIdx.x == 0) any_bad = 0;
        __syncthreads();
        // e_j, the end state of segment j: slot (j + 1) seg_tiles, the last one: slot n_tiles
        for (long long j = first_seg + threadIdx.x; j < n_seg; j += blockDim.x) {
            const double *e = ckpt + (j + 1 < n_seg ? (j + 1) * seg_tiles : n_tiles) * D;
            bool bad = false;
#pragma unroll
            for (int r = 0; r < D; r++) bad = bad || !(fabs(e[r]) <= 1.7976931348623157e308);
            if (bad) atomicOr(&any_bad, 1);
        }
        __syncthreads();
        if (any_bad && threadIdx.x == 0) ckpt[n_tiles * D] = __builtin_nan("");
        __syncthreads();
    }
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
    for (long long k_hi = n_seg - 1; k_hi >= first_seg + 1; k_hi -= blockDim.x) {
        const long long k = k_hi - threadIdx.x;
        double S[D];
#pragma unroll
        for (int r = 0; r < D; r++) S[r] = 0.0;
        if (k >= first_seg + 1) {
            // Horner from the oldest term: S = e_(k-J) ; S = P S + e_(k-J+1) ; ... ; + e_(k-1); e_j sits in slot (j+1) seg_tiles
            long long j = k - terms;
            if (j < first_seg) j = first_seg;
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
        if (k >= first_seg + 1) {
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
                   int n_seg, long long seg_len, long long n_tiles, const FloodArgs *flood, int first_seg = 0)
{
    if (n_seg <= 1) return HIPDSP_OK;       // (one segment: nothing to hand over, nothing to flood, slot n_tiles is true)
    dim3 grid((unsigned)channels), block(256);
    const long long seg_tiles = seg_len / TILE;
    FloodArgs fl;
    memset(&fl, 0, sizeof(fl));
    if (flood) fl = *flood;
    switch (SE) {
    case 1: hipLaunchKernelGGL((env_fix_kernel<1>), grid, block, 0, ctx->stream, edev, ckpt, ckpt_pitch, n_seg, seg_tiles, n_tiles, fl, first_seg); break;
    case 2: hipLaunchKernelGGL((env_fix_kernel<2>), grid, block, 0, ctx->stream, edev, ckpt, ckpt_pitch, n_seg, seg_tiles, n_tiles, fl, first_seg); break;
    case 3: hipLaunchKernelGGL((env_fix_kernel<3>), grid, block, 0, ctx->stream, edev, ckpt, ckpt_pitch, n_seg, seg_tiles, n_tiles, fl, first_seg); break;
    case 4: hipLaunchKernelGGL((env_fix_kernel<4>), grid, block, 0, ctx->stream, edev, ckpt, ckpt_pitch, n_seg, seg_tiles, n_tiles, fl, first_seg); break;
    default: return HIPDSP_OK;
    }
    return hd_launch_status("env_fix_kernel");
}

// WPB waves per workgroup, each an independent (channel, segment) unit with an LDS tile of its own and no workgroup
// barrier anywhere.  With WPB = 4 the hardware puts the four waves of a workgroup on the four SIMDs of ONE CU, so every
// SIMD of the chip carries the same number of these persistent waves whatever else the dispatcher has seen before;
// single-wave workgroups (WPB = 1) land wherever the dispatcher's round-robin stands -- a tiny copy kernel in front of
// the launch (the spectrogram tile of the multi-GPU step) left some SIMDs with three waves and others with one:
// 2.96 -> 3.8 ms at 32 channels (profiles/r03_forcedist_*), and probably the "two modes" of round 2.
// REGW (one- and two-section plans): the forward cascade's outputs stay in registers as float64 (w_[32] per lane) and
// the backward cascade reads them there (CASC_X, walking lanes downwards: CASC_DOWN) -- 96 conversions, 8 LDS stores
// and 16 LDS loads less per tile and lane, 185-203 VGPRs instead of 135-147: two waves a SIMD, which is what the planner
// asks for anyway (launch_env_bwd).  It only pays with the scalar tables under control: left alone hipcc fetched both
// G groups before using either and put the lane-0 branch of the state fold between fetch and use (264 SGPR spills,
// 6.0-6.1 ms against 5.9); with CASC_CARRY_AFTER_F and the opaque `last` below it has 60 spills and 15 % fewer
// instructions per tile: 8 % fewer cycles (SQ_BUSY_CYCLES, profiles/r03_bwd_regw_ab.log), 1-5 % less time free-running.
template <int SE, bool PREFETCH, bool PIN = true, bool TRACE = false, int WPB_ = WPB, bool REGW = (SE <= 2)>
__global__ __launch_bounds__(64 * WPB_, REGW ? 2 : 1) void env_bwd_kernel(const SosPlanDev *__restrict__ P0, BwdArgs a)
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

    // A channel whose forward sweep ended non-finite is NaN everywhere (sosfiltfilt's backward pass starts from that
    // end; sos_device.h: FloodArgs): the unit fills its share of the output instead of sweeping.
    {
        double end_state[DE];
#pragma unroll
        for (int r = 0; r < DE; r++) end_state[r] = ckpt[a.n_tiles * DE + r];
        if (state_not_finite(end_state)) {
            long long p_lo = (a.n_tiles - rt_hi) * TILE, p_hi = (a.n_tiles - rt_lo) * TILE;
            if (p_lo < a.skip) p_lo = a.skip;
            if (p_hi > T) p_hi = T;
            for (long long p = p_lo + lane; p < p_hi; p += 64) out[p - a.skip] = __builtin_nanf("");
            return;
        }
    }

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
    auto prefetchable = [&](long long tidx) {
        return tidx >= 0 && tidx <= top_full && tidx * TILE + TILE > a.skip && tidx * TILE >= a.lead;
    };
    // the tile the envelope starts in (GridShift; -1: it starts with the trace, nothing to rebuild)
    const long long env_tile0 = a.env0 > 0 ? a.env0 - a.env0 % TILE : -1;
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
            // a tile that touches T (the top one or two of a channel) or holds the `lead` samples in front of the
            // trace.  Every load of this loop is
            // an untracked asm load: hipcc's own vmcnt(0) for a tracked one would also drain the
            // prefetch issued further down.  Clamped addresses are always valid; samples outside [lead, T)
            // become zero.
            const long long lead = a.lead;
#pragma unroll 1
            for (int k = 0; k < 8; k++) {
                const long long p = tile + 256 * k + 4 * lane;
                auto at = [&](long long q) { return in + (q < lead ? lead : (q < T ? q : T - 1)); };
                auto ok = [&](long long q) { return q >= lead && q < T; };
                v4f t;
                t.x = asm_load4(at(p));
                t.y = asm_load4(at(p + 1));
                t.z = asm_load4(at(p + 2));
                t.w = asm_load4(at(p + 3));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(t));
                float4 v = make_float4(ok(p) ? t.x : 0.f, ok(p + 1) ? t.y : 0.f, ok(p + 2) ? t.z : 0.f,
                                       ok(p + 3) ? t.w : 0.f);
                if (a.rectify) v = make_float4(fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w));
                lds[lds_slot(8 * k + (lane >> 3), lane & 7)] = v;
            }
            v4f ck[SE];
#pragma unroll
            for (int i = 0; i < SE; i++) ck[i] = asm_load16(ckpt + tidx * DE + 2 * i);
            float ra = asm_load4(in + (T - 1));
            float rb = asm_load4(in + (lane < edge ? T - 2 - lane : T - 1));
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
                float4 v = load_four_from(in, tile + 256 * k + 4 * lane, a.lead, T);
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
        if (tile == env_tile0) {
            // the tile the envelope starts in: rebuild what the forward sweep filtered there -- the left odd extension
            // and its first value in front of it (the tile's state, zi * that value, is in the tile states)
            (void)env_left_fill(ldsf, lane, (int)(a.env0 - tile), edge);
        }
        TRACE_AT(0);                               // tile from the prefetch registers into LDS
        if (PREFETCH) {
            pre = rt + 1 < rt_hi && prefetchable(tidx - 1);
            fetch(pre ? tidx - 1 : top_full);
        }
        TRACE_AT(1);                               // prefetch of the next tile issued
        // ---- forward cascade again, from the state that entered this tile
#define CASC_S SE
#define CASC_PLAN() PLAN_OF(P0)
#define CASC_IN(v) (v)
#define CASC_PIN_GROUPS PIN
      if constexpr (REGW) {
#define CASC_CARRY_AFTER_F
        double w_[L];
#define CASC_CARRY cfw_
#define CASC_GAIN a.gain
#define CASC_NO_STORE
#define CASC_TAP(j, e, y) w_[(j)] = (y)
#include "sos_cascade.inc"
#undef CASC_TAP
#undef CASC_NO_STORE
#undef CASC_GAIN
#undef CASC_CARRY
        WAVE_SYNC();
        TRACE_AT(2);
        if (rt == 0) {
            int last = (int)(T + edge - 1 - tile);
            asm volatile("" : "+v"(last));              // or its 32 lane masks are hoisted out of the sweep into SGPRs
            const int lrow = last >> 5, lj = last & 31;
            double v0 = 0.0;
#pragma unroll
            for (int j = 0; j < L; j++) v0 = (j == lj) ? w_[j] : v0;
            v0 = __shfl(v0, lrow, 64);
#pragma unroll
            for (int j = 0; j < L; j++) w_[j] = (L * lane + j > last) ? v0 : w_[j];
            const SosPlanDev *P = PLAN_OF(P0);
#pragma unroll
            for (int r = 0; r < DE; r++) cb_[r] = P->zi[r] * v0;
        }
#define CASC_CARRY cb_
#define CASC_DOWN
#define CASC_X(j) w_[L - 1 - (j)]
#include "sos_cascade.inc"
#undef CASC_X
#undef CASC_DOWN
#undef CASC_CARRY
#undef CASC_CARRY_AFTER_F
      } else {
#define CASC_CARRY cfw_
#define CASC_GAIN a.gain
#include "sos_cascade.inc"
#undef CASC_GAIN
#undef CASC_CARRY
        WAVE_SYNC();
        TRACE_AT(2);
        if (rt == 0) {
            const int last = (int)(T + edge - 1 - tile);
            const float v0 = ldsf[lds_float_index(last)];
            WAVE_SYNC();
            for (int s2 = last + 1 + lane; s2 < TILE; s2 += 64) ldsf[lds_float_index(s2)] = v0;
            const SosPlanDev *P = PLAN_OF(P0);
#pragma unroll
            for (int r = 0; r < DE; r++) cb_[r] = P->zi[r] * (double)v0;
            WAVE_SYNC();
        }
#define CASC_CARRY cb_
#define CASC_REVERSED
#include "sos_cascade.inc"
#undef CASC_REVERSED
#undef CASC_CARRY
      }
#undef CASC_PIN_GROUPS
#undef CASC_S
#undef CASC_PLAN
#undef CASC_IN
        WAVE_SYNC();
        TRACE_AT(3);                               // backward cascade
        if (rt >= rt_lo) {
            if (tile >= a.skip && tile + TILE <= T) {
                // interior tile: exactly 8 vector stores (max(x, 0) as one instruction: fmaxf() costs a second one
                // that only quiets signalling NaNs)
                v4f rows[8];
                tile_rows_from_lds(lds, lane, rows);
                if (a.clamp) {
                    const long long tdst = (a.debug & 1) ? a.skip : tile;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const v4f v = rows[k];
                        f4u t; t.x = max_zero(v.x); t.y = max_zero(v.y); t.z = max_zero(v.z); t.w = max_zero(v.w);
                        *reinterpret_cast<f4u *>(out + (tdst + 256 * k + 4 * lane - a.skip)) = t;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const v4f v = rows[k];
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
// (both: four values and ONE 16-byte access per thread, no loop -- the form that runs at the device's copy rate, see
// copy_skip_kernel; f4u only needs 4-byte alignment, so `edge` may be anything)
__global__ __launch_bounds__(256) void odd_ext_kernel(const float *__restrict__ x, long long x_pitch, long long T, int edge, int rectify,
                                                      float *__restrict__ out, long long out_pitch)
{
    const long long ch = blockIdx.y;
    const float *xc = x + ch * x_pitch;
    float *oc = out + ch * out_pitch;
    const long long N = T + 2LL * edge;
    auto r = [&](long long k) { const float v = xc[k]; return rectify ? fabsf(v) : v; };
    const long long i = 4 * ((long long)blockIdx.x * 256 + threadIdx.x);
    if (i >= N) return;
    if (i >= edge && i + 4 <= edge + T) {                          // inside the trace: a straight (rectified) copy
        f4u v = *reinterpret_cast<const f4u *>(xc + (i - edge));
        if (rectify) { v.x = fabsf(v.x); v.y = fabsf(v.y); v.z = fabsf(v.z); v.w = fabsf(v.w); }
        *reinterpret_cast<f4u *>(oc + i) = v;
        return;
    }
    for (long long j = i; j < i + 4 && j < N; j++) {
        float v;
        if (j < edge) v = 2.f * r(0) - r(edge - j);
        else if (j < edge + T) v = r(j - edge);
        else v = 2.f * r(T - 1) - r(T - 2 - (j - edge - T));
        oc[j] = v;
    }
}

// y[c][i] = x[c][N - 1 - (first + i)], i < n, optionally clamped at zero: time reversal (and the final trim)
__global__ __launch_bounds__(256) void flip_kernel(const float *__restrict__ x, long long x_pitch, long long N, long long first, long long n,
                                                   int clamp, float *__restrict__ y, long long y_pitch)
{
    const long long ch = blockIdx.y;
    const float *xc = x + ch * x_pitch;
    float *yc = y + ch * y_pitch;
    // env[env < 0] = 0 (bufferedenvelope.py:41): NaN stays NaN, fmaxf would drop it
    auto cl = [&](float v) { return (clamp && v < 0.f) ? 0.f : v; };
    const long long i = 4 * ((long long)blockIdx.x * 256 + threadIdx.x);
    if (i >= n) return;
    if (i + 4 <= n) {
        const f4u v = *reinterpret_cast<const f4u *>(xc + (N - 1 - (first + i) - 3));   // sources i+3, i+2, i+1, i
        f4u o; o.x = cl(v.w); o.y = cl(v.z); o.z = cl(v.y); o.w = cl(v.x);
        *reinterpret_cast<f4u *>(yc + i) = o;
        return;
    }
    for (long long j = i; j < n; j++) yc[j] = cl(xc[N - 1 - (first + j)]);
}

// ref[c] = x[c][0]
__global__ void first_sample_kernel(const float *__restrict__ x, long long x_pitch, long long channels, float *__restrict__ ref)
{
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < channels) ref[c] = x[c * x_pitch];
}

// pass-through / zero fill for the sos-is-None branches
// (ONE 16-byte access per thread and no loop, like the copy probe: reads and writes then interleave at the finest grain
// and the copy runs at the device's copy rate -- one float per thread in a grid-stride loop reached 4.5 TB/s, four floats
// per thread in such a loop 4.8; f4u is a float4 that only needs 4-byte alignment, whatever `skip` and the pitches are)
__global__ __launch_bounds__(256) void copy_skip_kernel(const float *__restrict__ x, long long x_pitch, float *__restrict__ y,
                                                        long long y_pitch, long long n, long long skip)
{
    const long long ch = blockIdx.y;
    const float *xi = x + ch * x_pitch + skip;
    float *yo = y + ch * y_pitch;
    const long long n4 = n / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        *reinterpret_cast<f4u *>(yo + 4 * i) = *reinterpret_cast<const f4u *>(xi + 4 * i);
    } else if (i == n4) {
        for (long long k = 4 * n4; k < n; k++) yo[k] = xi[k];
    }
}

int launch_scan(hipdsp_ctx *ctx, const SosPlanDev *dev, int S, SeqArgs a, long long channels, long long warm)
{
    if (a.gain == 0.0) a.gain = 1.0;
    plan_segments(ctx, a.N, channels, warm, &a.seg_len, &a.n_seg, -1);
    a.units = channels * a.n_seg;
    long long blocks = (a.units + WPB - 1) / WPB;
    if (blocks > 0x7fffffffLL) {
        hipdsp_set_error("grid too large (%lld blocks)", blocks);
        return HIPDSP_ERR_INVALID;
    }
    dim3 grid((unsigned)blocks), block(64 * WPB);
    if (a.n_seg > 1) {                                    // non-finite samples must not be forgotten at a segment border
        const int frc = hd_seg_flags(ctx, (size_t)a.units, &a.flags);
        if (frc != HIPDSP_OK) return frc;
    }
    switch (S) {
    case 1: hipLaunchKernelGGL((sos_scan_kernel<1>), grid, block, 0, ctx->stream, dev, a); break;
    case 2: hipLaunchKernelGGL((sos_scan_kernel<2>), grid, block, 0, ctx->stream, dev, a); break;
    case 3: hipLaunchKernelGGL((sos_scan_kernel<3>), grid, block, 0, ctx->stream, dev, a); break;
    case 4: hipLaunchKernelGGL((sos_scan_kernel<4>), grid, block, 0, ctx->stream, dev, a); break;
    default:
        hipdsp_set_error("n_sections %d not in 1..%d", S, MAXS);
        return HIPDSP_ERR_UNSUPPORTED;
    }
    const int rc = hd_launch_status("sos_scan_kernel");
    if (rc != HIPDSP_OK || a.flags == nullptr) return rc;
    FloodArgs fl;
    memset(&fl, 0, sizeof(fl));
    fl.flags = a.flags; fl.n_seg = a.n_seg; fl.seg_len = a.seg_len;
    fl.y = a.out; fl.y_pitch = a.out_pitch; fl.T = a.N; fl.skip = a.skip;
    return launch_flood(ctx, fl, channels);
}

// Envelope by checkpoints: forward sweep (optionally with the band-pass in front), then the
// recomputing backward sweep.  `fplan` NULL: `x` is the trace to rectify.
int launch_env_ckpt(hipdsp_ctx *ctx, const SosPlanDev *fdev, const SosPlanDev *edev, int SF, int SE, long long warmF,
                    long long warmE, int edge, const float *x, long long x_pitch, float *yf,
                    long long yf_pitch, float *env, long long env_pitch, long long channels,
                    long long frames, long long skip, int rectify, double gain, int clamp, int phase,
                    GridShift gs = GridShift{0, 0, 0})
{
    // `gs`: the sweep's tile grid when the envelope does not start with the trace (sos_device.h: GridShift): the kernels
    // walk T = frames + lead samples through pointers shifted by -lead; `skip` counts from the envelope's first sample
    const long long real_frames = frames;
    x -= gs.lead;
    if (yf) yf -= gs.lead;
    frames += gs.lead;
    skip += gs.env0;
    const long long n_tiles = (frames + edge + TILE - 1) / TILE;
    const long long ckpt_pitch = (n_tiles + 1) * 2 * SE;     // (+ 1: the state the channel ends with, FloodArgs)
    void *work = nullptr;
    int rc = phase == 2 ? hd_scratch_parked(ctx, sizeof(double) * (size_t)ckpt_pitch * (size_t)channels, &work)
                        : hipdsp_scratch(ctx, sizeof(double) * (size_t)ckpt_pitch * (size_t)channels, &work);
    if (rc != HIPDSP_OK) return rc;
    if (phase != 2) {
        CkptArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.in = x; fa.yf = yf; fa.ckpt = (double *)work;
        fa.in_pitch = x_pitch; fa.yf_pitch = yf_pitch; fa.ckpt_pitch = ckpt_pitch;
        fa.T = frames; fa.edge = edge; fa.rectify = rectify; fa.gain = rectify ? gain : 1.0;
        fa.lead = gs.lead; fa.env0 = gs.env0;
        hd_note_sweep(ctx, gs.lead, gs.env0, real_frames, channels, SE);
        // only the band-pass warms up; the envelope's states are handed over exactly (env_fix_kernel), which needs
        // a cascade that forgets (a plan that does not decay is never cut into segments)
        long long warm = warmF;
        if (warmF >= (1LL << 40) || warmE >= (1LL << 40)) warm = 1LL << 50;
        plan_segments(ctx, frames, channels, warm, &fa.seg_len, &fa.n_seg, -2);
        fa.units = channels * fa.n_seg;
        long long blocks = (fa.units + WPB - 1) / WPB;
        HD_REQUIRE(blocks <= 0x7fffffffLL, "grid too large");
        dim3 grid((unsigned)blocks), block(64 * WPB);
        if (SF > 0 && fa.n_seg > 1) {
            rc = hd_seg_flags(ctx, (size_t)fa.units, &fa.flags);
            if (rc != HIPDSP_OK) return rc;
        }
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
        FloodArgs fl;
        memset(&fl, 0, sizeof(fl));
        fl.flags = fa.flags; fl.n_seg = fa.n_seg; fl.seg_len = fa.seg_len;
        fl.y = yf ? yf + gs.lead : nullptr; fl.y_pitch = yf_pitch; fl.T = frames; fl.skip = gs.lead;
        rc = launch_env_fix(ctx, edev, SE, (double *)work, ckpt_pitch, channels, fa.n_seg, fa.seg_len, n_tiles, &fl,
                            (int)((gs.env0 - gs.env0 % TILE) / fa.seg_len));
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
    b.lead = gs.lead; b.env0 = gs.env0;
    b.rectify = rectify; b.clamp = clamp; b.gain = rectify ? gain : 1.0;
    b.trace = ctx->sos_trace;
    b.trace_rows = ctx->sos_trace_rows;
    b.debug = ctx->sos_debug;
    b.fair = ctx->sos_fair;
    const long long used_tiles = n_tiles - skip / TILE;      // tiles below `skip` are never visited
    long long seg_len = 0;
    plan_segments(ctx, used_tiles * TILE, channels, warmE, &seg_len, &b.n_seg, 4, SE <= 2 ? 8 : 16);   // env_bwd_kernel: REGW
    b.seg_tiles = seg_len / TILE;
    b.warm_tiles = warmE / TILE;
    b.units = channels * b.n_seg;
    long long blocks = (b.units + WPB - 1) / WPB;            // four waves per workgroup: one per SIMD of a CU
    HD_REQUIRE(blocks <= 0x7fffffffLL, "grid too large");
    dim3 grid((unsigned)blocks), block(64 * WPB);
#ifdef HIPDSP_WITH_ENVSPLIT
    if (ctx->sos_split && SE <= 2 && frames >= 4 * TILE) return hd_launch_env_bwd_split(ctx, edev, SE, b);   // (A/B: envsplit.hip, make SPLIT=1)
#endif
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

int hd_launch_env_fix(hipdsp_ctx *ctx, const SosPlanDev *edev, int SE, double *ckpt, long long ckpt_pitch, long long channels,
                      int n_seg, long long seg_len, long long n_tiles, const FloodArgs *flood, int first_seg)
{
    return launch_env_fix(ctx, edev, SE, ckpt, ckpt_pitch, channels, n_seg, seg_len, n_tiles, flood, first_seg);
}

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
    HD_NO_OVERLAP(x, x_pitch, frames, y, y_pitch, frames - skip, channels, "x and y");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    if (plan == nullptr) {
        long long n = frames - skip;
        const long long blocks = (n / 4 + 1 + 255) / 256;        // one thread per float4, one more for the tail
        HD_REQUIRE(blocks <= 0x7fffffffLL && channels <= 65535, "grid too large");
        const unsigned gx = (unsigned)blocks;
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
                            int clamp, int phase, int64_t env_first)
{
    HD_REQUIRE(ctx != nullptr && fplan != nullptr && eplan != nullptr, "NULL argument");
    HD_REQUIRE(phase >= 0 && phase <= 2, "phase must be 0 (both), 1 (forward) or 2 (backward)");
    HD_REQUIRE(channels >= 0 && frames >= 0, "negative size");
    HD_REQUIRE(env_first >= 0 && env_first <= frames, "env_first %lld not in [0, frames=%lld]", (long long)env_first,
               (long long)frames);
    HD_REQUIRE(fplan->host->n_sections > 0 && eplan->host->n_sections > 0, "plan has no coefficients");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    {   // phase 2 consumes what a forward sweep left behind: not if that sweep reported a fault
        const int frc = hd_device_fault(ctx);
        if (frc != HIPDSP_OK) return frc;
    }
    const int edge = eplan->host->edge;
    if (frames - env_first <= edge) {
        hipdsp_set_error("The length of the input vector x must be greater than padlen, which is %d.", edge);
        return HIPDSP_ERR_TOO_SHORT;
    }
    if (channels == 0) return HIPDSP_OK;
    HD_REQUIRE(x != nullptr && yf != nullptr && env != nullptr, "NULL data pointer");
    HD_REQUIRE(x_pitch >= frames && yf_pitch >= frames && env_pitch >= frames - env_first, "pitch smaller than row length");
    if (phase != 2) HD_NO_OVERLAP(x, x_pitch, frames, yf, yf_pitch, frames, channels, "x and yf");
    if (phase != 1) HD_NO_OVERLAP(yf, yf_pitch, frames, env, env_pitch, frames - env_first, channels, "yf and env");
    GridShift gs;
    if (phase == 2) {
        // the tile states in the scratch belong to the grid of the forward sweep that left them
        // (hipdsp_chain_forward or phase 1): same trace, same envelope start
        const long long lead = ctx->sweep_lead;
        if (ctx->sweep_frames != frames || ctx->sweep_channels != channels || ctx->sweep_sections != eplan->host->n_sections ||
            ctx->sweep_env0 - lead != env_first) {
            hipdsp_set_error("hipdsp_sosfilt_envelope(phase = 2): the last forward sweep on this context left tile states for "
                             "%lld channels x %lld frames, %d sections, envelope from frame %lld -- not for %lld x %lld, %d, %lld",
                             ctx->sweep_channels, ctx->sweep_frames, ctx->sweep_sections,
                             ctx->sweep_env0 - ctx->sweep_lead, (long long)channels, (long long)frames,
                             eplan->host->n_sections, (long long)env_first);
            return HIPDSP_ERR_INVALID;
        }
        gs.lead = lead; gs.env0 = ctx->sweep_env0; gs.frame_off = 0;
    } else {
        gs = hd_grid_shift(0, env_first, 0, edge, true);
    }
    HD_REQUIRE(gs.lead < TILE, "grid shift %lld", gs.lead);
    return launch_env_ckpt(ctx, fplan->dev, eplan->dev, fplan->host->n_sections, eplan->host->n_sections,
                           fplan->host->warm, eplan->host->warm, edge, x, x_pitch, yf, yf_pitch, env, env_pitch,
                           channels, frames, 0, rectify, gain, clamp, phase, gs);
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
    const long long N = frames + 2LL * edge;                 // (x is fully consumed before y is written: they may overlap)
    // Two temporaries of the slab's size and one float per channel, from the context's scratch (it grows once and stays;
    // taken from hipdsp_malloc they were larger than any block its cache keeps by default, and every call paid a
    // hipMalloc + hipFree of tens of GB -- 1.1-1.2 s at BASELINE configs[2]'s size on some boxes, profiles/r03_entry_points.log).
    // Nothing below uses the scratch for anything else (sos_scan_kernel parks no tile states).
    float *buf[2] = {nullptr, nullptr}, *ref = nullptr;
    const size_t slab = ((sizeof(float) * (size_t)N * (size_t)channels + 255) / 256) * 256;
    void *work = nullptr;
    int rc = hipdsp_scratch(ctx, 2 * slab + sizeof(float) * (size_t)channels, &work);
    if (rc != HIPDSP_OK) return rc;
    buf[0] = (float *)work;
    buf[1] = (float *)((char *)work + slab);
    ref = (float *)((char *)work + 2 * slab);
    auto cleanup = [&]() {};
    HD_REQUIRE((N + 1023) / 1024 <= 0x7fffffffLL, "grid too large");
    const unsigned gx = (unsigned)((N + 1023) / 1024);            // 256 threads x 4 values per block, no loop
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
            const unsigned gy = (unsigned)((n + 1023) / 1024);
            hipLaunchKernelGGL(flip_kernel, dim3(gy, (unsigned)channels), dim3(256), 0, ctx->stream, buf[cur], N, N,
                               (long long)edge + (long long)skip, n, clamp, y, (long long)y_pitch);
        }
    }
    if (rc == HIPDSP_OK) rc = hd_launch_status("envelope_multi kernels");
    cleanup();
    return rc;
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
    HD_NO_OVERLAP(x, x_pitch, frames, y, y_pitch, frames - skip, channels, "x and y");
    return launch_env_ckpt(ctx, nullptr, plan->dev, 0, plan->host->n_sections, 0, plan->host->warm, edge, x,
                           x_pitch, nullptr, 0, y, y_pitch, channels, frames, skip, rectify, gain, clamp, 0);
}

}  // extern "C"
