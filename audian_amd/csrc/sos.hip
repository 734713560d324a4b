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
#include <cmath>
#include <vector>

namespace {

constexpr int L = 32;               // samples per lane per tile
constexpr int TILE = 64 * L;        // samples per wave per tile
constexpr int MAXS = HIPDSP_MAX_SECTIONS;
constexpr int MAXD = 2 * MAXS;      // state dimension

struct SosPlanDev {
    double coef[MAXS][5];           // b0 b1 b2 a1 a2
    double G[L][MAXD];              // A^(L-1-j) B
    double M[6][MAXD][MAXD];        // A^(L*2^k)
    double zi[MAXD];                // scipy sosfilt_zi, flattened (z0,z1) per section
    long long warm;                 // warm-up samples, multiple of TILE
    int n_sections;
    int edge;                       // sosfiltfilt pad length
};

enum { MODE_FILT = 0, MODE_ENV_FWD = 1, MODE_ENV_BWD = 2 };

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 4-byte aligned float4

struct SeqArgs {
    const float *in;        // channel 0 of the physical input
    float *out;             // channel 0 of the physical output
    long long in_pitch, out_pitch;
    long long T;            // frames of the user's trace
    long long N;            // logical sequence length (T, or T + 2*edge)
    long long seg_len;      // multiple of TILE
    int n_seg;
    long long skip;         // nbefore
    int edge;
    int rectify;
    float gain;
    int clamp;
};

__device__ __forceinline__ long long opaque_zero()
{
    int z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    return (long long)z;
}

__device__ __forceinline__ int lds_slot(int row, int q) { return row * 8 + (q ^ ((row >> 1) & 7)); }

// One logical sample (slow path: sequence borders, odd extension).
template <int MODE>
__device__ __forceinline__ float load_one(const SeqArgs &a, const float *in, long long i)
{
    if (i < 0 || i >= a.N) return 0.f;
    if (MODE == MODE_FILT) return in[i];
    if (MODE == MODE_ENV_BWD) return in[a.N - 1 - i];
    // MODE_ENV_FWD: odd extension of r(j) = gain*|x[j]| (scipy odd_ext)
    long long j = i - a.edge;
    auto r = [&](long long k) { float v = in[k]; return a.rectify ? a.gain * fabsf(v) : v; };
    if (j < 0) return 2.f * r(0) - r(-j);
    if (j >= a.T) return 2.f * r(a.T - 1) - r(2 * a.T - 2 - j);
    return r(j);
}

template <int MODE>
__device__ __forceinline__ float4 load_four(const SeqArgs &a, const float *in, long long p)
{
    float4 v;
    bool fast;
    if (MODE == MODE_FILT) fast = (p >= 0 && p + 4 <= a.N);
    else if (MODE == MODE_ENV_BWD) fast = (p >= 0 && p + 4 <= a.N);
    else fast = (p >= a.edge && p + 4 <= a.edge + a.T);
    if (fast) {
        if (MODE == MODE_FILT) {
            f4u t = *reinterpret_cast<const f4u *>(in + p);
            v = make_float4(t.x, t.y, t.z, t.w);
        } else if (MODE == MODE_ENV_BWD) {
            f4u t = *reinterpret_cast<const f4u *>(in + (a.N - 4 - p));
            v = make_float4(t.w, t.z, t.y, t.x);
        } else {
            f4u t = *reinterpret_cast<const f4u *>(in + (p - a.edge));
            if (a.rectify)
                v = make_float4(a.gain * fabsf(t.x), a.gain * fabsf(t.y), a.gain * fabsf(t.z),
                                a.gain * fabsf(t.w));
            else
                v = make_float4(t.x, t.y, t.z, t.w);
        }
    } else {
        v.x = load_one<MODE>(a, in, p);
        v.y = load_one<MODE>(a, in, p + 1);
        v.z = load_one<MODE>(a, in, p + 2);
        v.w = load_one<MODE>(a, in, p + 3);
    }
    return v;
}

// Store logical samples p..p+3 restricted to [lo, hi) (the wave's own segment).
template <int MODE>
__device__ __forceinline__ void store_four(const SeqArgs &a, float *out, long long p, float4 v,
                                           long long lo, long long hi)
{
    if (MODE == MODE_ENV_BWD && a.clamp) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    // logical range that maps onto the physical output
    long long olo, ohi, phys;      // phys index of logical p (for BWD: of logical p+3)
    if (MODE == MODE_FILT) { olo = a.skip; ohi = a.N; }
    else if (MODE == MODE_ENV_FWD) { olo = 0; ohi = a.N; }
    else { olo = a.edge; ohi = a.N - a.edge - a.skip; }   // t = N-1-i-edge in [skip, T)
    if (olo < lo) olo = lo;
    if (ohi > hi) ohi = hi;
    if (p >= olo && p + 4 <= ohi) {
        f4u t;
        if (MODE == MODE_ENV_BWD) {
            phys = a.N - 1 - (p + 3) - a.edge - a.skip;
            t.x = v.w; t.y = v.z; t.z = v.y; t.w = v.x;
        } else {
            phys = (MODE == MODE_FILT) ? p - a.skip : p;
            t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
        }
        *reinterpret_cast<f4u *>(out + phys) = t;
    } else {
        float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            long long i = p + k;
            if (i >= olo && i < ohi) {
                if (MODE == MODE_FILT) out[i - a.skip] = e[k];
                else if (MODE == MODE_ENV_FWD) out[i] = e[k];
                else out[a.N - 1 - i - a.edge - a.skip] = e[k];
            }
        }
    }
}

template <int S, int MODE>
__global__ __launch_bounds__(64) void sos_scan_kernel(const SosPlanDev *__restrict__ P0, SeqArgs a)
{
    // The plan tables (coefficients, G, M: up to ~3 KB) are wave-uniform and must come
    // through scalar loads (s_load -> SGPR operands of v_fma_f64).  Hoisting them out of
    // the tile loop would need ~500 SGPRs and spill through v_writelane; a "memory"
    // clobber would demote them to per-lane vector loads.  So every use goes through
    // PLAN(): the same pointer plus an opaque, always-zero scalar that the compiler must
    // assume changes each time, which pins the s_load next to its use.
#define PLAN() (reinterpret_cast<const SosPlanDev *>(reinterpret_cast<const char *>(P0) + opaque_zero()))
    constexpr int D = 2 * S;
    __shared__ float4 lds[64 * 8];
    const int lane = threadIdx.x;
    const int seg = blockIdx.x % a.n_seg;
    const long long ch = blockIdx.x / a.n_seg;
    const float *in = a.in + ch * a.in_pitch;
    float *out = a.out + ch * a.out_pitch;

    const long long lo = (long long)seg * a.seg_len;
    long long hi = lo + a.seg_len;
    if (hi > a.N) hi = a.N;
    long long start = lo - P0->warm;
    const bool true_init = start <= 0;
    if (start < 0) start = 0;

    double carry[D];
#pragma unroll
    for (int r = 0; r < D; r++) carry[r] = 0.0;
    if (MODE != MODE_FILT && true_init) {
        // scipy sosfiltfilt: zi * x_ext[0] (forward) / zi * y_fwd[-1] (backward)
        double v0 = (double)load_one<MODE>(a, in, 0);
#pragma unroll
        for (int r = 0; r < D; r++) carry[r] = P0->zi[r] * v0;
    }

    for (long long tile = start; tile < hi; tile += TILE) {
        // ---- HBM -> LDS (coalesced 16 B per lane), LDS -> registers (row per lane)
        // (A register prefetch of the next tile with a hand-counted vmcnt was measured and
        // bought nothing: the kernel already runs at the device's read+write copy rate.)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            long long p = tile + 256 * k + 4 * lane;
            lds[lds_slot(8 * k + (lane >> 3), lane & 7)] = load_four<MODE>(a, in, p);
        }
        __syncthreads();

#define CASC_S S
#define CASC_PLAN() PLAN()
#define CASC_CARRY carry
#define CASC_IN(v) (v)
#include "sos_cascade.inc"
#undef CASC_S
#undef CASC_PLAN
#undef CASC_CARRY
#undef CASC_IN
        __syncthreads();
        if (tile + TILE > lo) {      // warm-up tiles produce no output
#pragma unroll
            for (int k = 0; k < 8; k++) {
                long long p = tile + 256 * k + 4 * lane;
                float4 v = lds[lds_slot(8 * k + (lane >> 3), lane & 7)];
                store_four<MODE>(a, out, p, v, lo, hi);
            }
        }
        __syncthreads();
    }
}

// ---- fused band-pass + envelope forward pass ----------------------------------------
// Batch chains (whole slab: filter -> envelope of the SAME slab) re-read the filtered trace
// only to rectify it and run the envelope's forward pass.  This kernel does both cascades
// on the tile while it is in LDS: read x once, write the filtered trace and the forward
// scratch (12 B per sample instead of 8 + 8).  The backward pass stays the plain
// MODE_ENV_BWD launch.  Sequence handling of scipy's sosfiltfilt in sample coordinates:
//   * left odd extension: `edge` serial steps before tile 0 from zi * ext[0] (wave-uniform);
//   * right odd extension: the samples T .. T+edge-1 of the last tile(s) are replaced by the
//     extension values, so the cascade itself produces the padded outputs;
//   * scratch index = sample index + edge, exactly what the backward kernel expects.
struct FusedArgs {
    const float *in;
    float *yf, *w;
    long long in_pitch, yf_pitch, w_pitch;
    long long T, seg_len;
    int n_seg, edge, rectify;
    float gain;
};

__device__ __forceinline__ int lds_float_index(int s) { return lds_slot(s >> 5, (s & 31) >> 2) * 4 + (s & 3); }

template <int SF, int SE>
__global__ __launch_bounds__(64) void sos_fused_kernel(const SosPlanDev *__restrict__ PF0,
                                                       const SosPlanDev *__restrict__ PE0, FusedArgs a)
{
#define PLANF() (reinterpret_cast<const SosPlanDev *>(reinterpret_cast<const char *>(PF0) + opaque_zero()))
#define PLANE() (reinterpret_cast<const SosPlanDev *>(reinterpret_cast<const char *>(PE0) + opaque_zero()))
    constexpr int DF = 2 * SF, DE = 2 * SE;
    __shared__ float4 lds[64 * 8];
    __shared__ float rprev[64];            // rectified samples of the previous tile's last two rows
    float *ldsf = reinterpret_cast<float *>(lds);
    const int lane = threadIdx.x;
    const int seg = blockIdx.x % a.n_seg;
    const long long ch = blockIdx.x / a.n_seg;
    const float *in = a.in + ch * a.in_pitch;
    float *yf = a.yf + ch * a.yf_pitch;
    float *w = a.w + ch * a.w_pitch;
    const long long T = a.T;
    const int edge = a.edge;

    const long long lo = (long long)seg * a.seg_len;
    long long hi = lo + a.seg_len;
    const bool last_seg = hi >= T;
    if (hi > T) hi = T;
    long long env_start = lo - PE0->warm;
    const bool env_true = env_start <= 0;
    if (env_start < 0) env_start = 0;
    long long start = env_start - PF0->warm;
    if (start < 0) start = 0;               // zero state at sample 0 is the filter's true state
    const long long loop_end = last_seg ? T + edge : hi;   // the right extension may need a tile more

    double cf_[DF], ce_[DE];
#pragma unroll
    for (int r = 0; r < DF; r++) cf_[r] = 0.0;
#pragma unroll
    for (int r = 0; r < DE; r++) ce_[r] = 0.0;
    rprev[lane] = 0.f;

    for (long long tile = start; tile < loop_end; tile += TILE) {
        // ---- x tile -> LDS
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const long long p = tile + 256 * k + 4 * lane;
            float4 v;
            if (p + 4 <= T) {
                f4u t = *reinterpret_cast<const f4u *>(in + p);
                v = make_float4(t.x, t.y, t.z, t.w);
            } else {
                v.x = p < T ? in[p] : 0.f;
                v.y = p + 1 < T ? in[p + 1] : 0.f;
                v.z = p + 2 < T ? in[p + 2] : 0.f;
                v.w = p + 3 < T ? in[p + 3] : 0.f;
            }
            lds[lds_slot(8 * k + (lane >> 3), lane & 7)] = v;
        }
        __syncthreads();
        // ---- band-pass cascade: rows now hold the filtered samples
#define CASC_S SF
#define CASC_PLAN() PLANF()
#define CASC_CARRY cf_
#define CASC_IN(v) (v)
#include "sos_cascade.inc"
#undef CASC_S
#undef CASC_PLAN
#undef CASC_CARRY
#undef CASC_IN
        __syncthreads();
        if (tile + TILE > lo && tile < hi) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const long long p = tile + 256 * k + 4 * lane;
                const float4 v = lds[lds_slot(8 * k + (lane >> 3), lane & 7)];
                if (p >= lo && p + 4 <= hi) {
                    f4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
                    *reinterpret_cast<f4u *>(yf + p) = t;
                } else {
                    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (p + q >= lo && p + q < hi) yf[p + q] = e[q];
                }
            }
        }
        if (tile < env_start) { __syncthreads(); continue; }    // band-pass warm-up only

        // ---- envelope input in place: r = gain*|y|, then the odd extension past T
        if (a.rectify) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                float4 v = lds[lds_slot(lane, q)];
                v = make_float4(a.gain * fabsf(v.x), a.gain * fabsf(v.y), a.gain * fabsf(v.z), a.gain * fabsf(v.w));
                lds[lds_slot(lane, q)] = v;
            }
        }
        __syncthreads();
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
            __syncthreads();
            if (lane < edge && pj >= tile && pj < tile + TILE) ldsf[lds_float_index((int)(pj - tile))] = pv;
            __syncthreads();
        }
        if (env_true && tile == 0) {
            // left odd extension: ext[i] = 2 r(0) - r(edge - i), i < edge, from zi * ext[0];
            // wave-uniform serial steps, lane 0 stores the padded outputs
            const SosPlanDev *P = PLANE();
            const float r0 = ldsf[lds_float_index(0)];
            const double x0 = (double)(2.f * r0 - ldsf[lds_float_index(edge)]);
#pragma unroll
            for (int r = 0; r < DE; r++) ce_[r] = P->zi[r] * x0;
            for (int i = 0; i < edge; i++) {
                double cur = (double)(2.f * r0 - ldsf[lds_float_index(edge - i)]);
#pragma unroll
                for (int s2 = 0; s2 < SE; s2++) {
                    const double y = fma(P->coef[s2][0], cur, ce_[2 * s2]);
                    ce_[2 * s2] = fma(-P->coef[s2][3], y, fma(P->coef[s2][1], cur, ce_[2 * s2 + 1]));
                    ce_[2 * s2 + 1] = fma(-P->coef[s2][4], y, P->coef[s2][2] * cur);
                    cur = y;
                }
                if (lane == 0) w[i] = (float)cur;
            }
        }
        // keep the last two rows for an extension that reaches back over the tile border
        {
            const float4 keep0 = lds[lds_slot(62 + ((lane >> 3) & 1), lane & 7)];
            __syncthreads();
            if (lane < 16) {
                rprev[4 * lane] = keep0.x; rprev[4 * lane + 1] = keep0.y;
                rprev[4 * lane + 2] = keep0.z; rprev[4 * lane + 3] = keep0.w;
            }
        }
        // ---- envelope forward cascade
#define CASC_S SE
#define CASC_PLAN() PLANE()
#define CASC_CARRY ce_
#define CASC_IN(v) (v)
#include "sos_cascade.inc"
#undef CASC_S
#undef CASC_PLAN
#undef CASC_CARRY
#undef CASC_IN
        __syncthreads();
        {
            // scratch index = sample + edge; this wave owns samples [lo, hi) (+ the extension)
            const long long wlo = lo, whi = last_seg ? T + edge : hi;
            if (tile + TILE > wlo) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const long long p = tile + 256 * k + 4 * lane;
                    const float4 v = lds[lds_slot(8 * k + (lane >> 3), lane & 7)];
                    if (p >= wlo && p + 4 <= whi) {
                        f4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
                        *reinterpret_cast<f4u *>(w + p + edge) = t;
                    } else {
                        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (p + q >= wlo && p + q < whi) w[p + q + edge] = e[q];
                    }
                }
            }
        }
        __syncthreads();
    }
#undef PLANF
#undef PLANE
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

__global__ void zero_rows_kernel(float *__restrict__ y, long long y_pitch, long long n)
{
    long long ch = blockIdx.y;
    float *yo = y + ch * y_pitch;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        yo[i] = 0.f;
}

// ---- host-side plan mathematics (float64) -----------------------------------

struct Mat {
    int d;
    double v[MAXD][MAXD];
};

Mat mat_identity(int d)
{
    Mat m; m.d = d;
    for (int i = 0; i < MAXD; i++) for (int j = 0; j < MAXD; j++) m.v[i][j] = (i == j && i < d) ? 1.0 : 0.0;
    return m;
}

Mat mat_mul(const Mat &a, const Mat &b)
{
    Mat m = mat_identity(a.d);
    for (int i = 0; i < a.d; i++)
        for (int j = 0; j < a.d; j++) {
            double s = 0.0;
            for (int k = 0; k < a.d; k++) s += a.v[i][k] * b.v[k][j];
            m.v[i][j] = s;
        }
    return m;
}

double mat_norm_inf(const Mat &a)
{
    double n = 0.0;
    for (int i = 0; i < a.d; i++) {
        double s = 0.0;
        for (int j = 0; j < a.d; j++) s += fabs(a.v[i][j]);
        if (!(s <= n)) n = s;      // NaN propagates as "large"
    }
    return n;
}

// One time step of the cascade (same arithmetic order as the kernel / scipy).
void cascade_step(const double coef[MAXS][5], int S, double *z, double x)
{
    double cur = x;
    for (int s = 0; s < S; s++) {
        double y = coef[s][0] * cur + z[2 * s];
        z[2 * s] = coef[s][1] * cur - coef[s][3] * y + z[2 * s + 1];
        z[2 * s + 1] = coef[s][2] * cur - coef[s][4] * y;
        cur = y;
    }
}

int fill_plan(SosPlanDev *p, const double *sos, int S)
{
    memset(p, 0, sizeof(*p));
    p->n_sections = S;
    const int D = 2 * S;
    for (int s = 0; s < S; s++) {
        const double *c = sos + 6 * s;
        if (c[3] != 1.0) {
            hipdsp_set_error("sos[%d][3] (a0) must be 1, got %g", s, c[3]);
            return HIPDSP_ERR_INVALID;
        }
        for (int k = 0; k < 6; k++)
            if (!std::isfinite(c[k])) {
                hipdsp_set_error("sos[%d][%d] is not finite", s, k);
                return HIPDSP_ERR_INVALID;
            }
        p->coef[s][0] = c[0]; p->coef[s][1] = c[1]; p->coef[s][2] = c[2];
        p->coef[s][3] = c[4]; p->coef[s][4] = c[5];
    }
    // state-space (A, B): columns of A from unit states with zero input, B from unit input
    Mat A = mat_identity(D);
    double B[MAXD] = {0};
    for (int c = 0; c < D; c++) {
        double z[MAXD] = {0};
        z[c] = 1.0;
        cascade_step(p->coef, S, z, 0.0);
        for (int r = 0; r < D; r++) A.v[r][c] = z[r];
    }
    {
        double z[MAXD] = {0};
        cascade_step(p->coef, S, z, 1.0);
        for (int r = 0; r < D; r++) B[r] = z[r];
    }
    // G[j] = A^(L-1-j) B
    {
        double g[MAXD];
        for (int r = 0; r < D; r++) g[r] = B[r];
        for (int j = L - 1; j >= 0; j--) {
            for (int r = 0; r < D; r++) p->G[j][r] = g[r];
            double t[MAXD] = {0};
            for (int r = 0; r < D; r++)
                for (int c = 0; c < D; c++) t[r] += A.v[r][c] * g[c];
            for (int r = 0; r < D; r++) g[r] = t[r];
        }
    }
    // M[k] = A^(L*2^k); keep squaring up to A^TILE for the warm-up search
    Mat pw = A;                                   // A^1
    for (int k = 0; k < 5; k++) pw = mat_mul(pw, pw);   // A^32 = A^L
    static_assert(L == 32, "plan assumes L == 32");
    for (int k = 0; k < 6; k++) {
        for (int r = 0; r < D; r++)
            for (int c = 0; c < D; c++) p->M[k][r][c] = pw.v[r][c];
        pw = mat_mul(pw, pw);
    }
    // pw == A^(L*64) == A^TILE.  warm = TILE * (smallest m with ||A^(TILE*m)|| < 2^-60)
    const double tol = ldexp(1.0, -60);
    const int MAXBITS = 40;
    std::vector<Mat> pows;
    pows.push_back(pw);
    long long m = 1;
    int top = 0;
    while (!(mat_norm_inf(pows[top]) < tol) && top < MAXBITS) {
        pows.push_back(mat_mul(pows[top], pows[top]));
        top++;
        m <<= 1;
    }
    if (!(mat_norm_inf(pows[top]) < tol)) {
        m = 1LL << 50;                            // does not decay: never segment
    } else if (top > 0) {
        // binary refinement: largest q with ||A^(TILE*q)|| >= tol, answer q + 1
        Mat acc = mat_identity(D);
        long long q = 0;
        for (int k = top - 1; k >= 0; k--) {
            Mat cand = mat_mul(acc, pows[k]);
            if (!(mat_norm_inf(cand) < tol)) { acc = cand; q += 1LL << k; }
        }
        m = q + 1;
    }
    p->warm = m * TILE;
    // scipy sosfilt_zi
    double scale = 1.0;
    for (int s = 0; s < S; s++) {
        const double *c = sos + 6 * s;
        double b0 = c[0], b1 = c[1], b2 = c[2], a1 = c[4], a2 = c[5];
        double B0 = b1 - a1 * b0, B1 = b2 - a2 * b0;
        double m00 = 1.0 + a1, m01 = -1.0, m10 = a2, m11 = 1.0;
        double det = m00 * m11 - m01 * m10;
        p->zi[2 * s] = scale * (B0 * m11 - m01 * B1) / det;
        p->zi[2 * s + 1] = scale * (m00 * B1 - m10 * B0) / det;
        scale *= (c[0] + c[1] + c[2]) / (c[3] + c[4] + c[5]);
    }
    // scipy sosfiltfilt: edge = 3*ntaps, ntaps = 2S+1 - min(#b2==0, #a2==0)
    int nb = 0, na = 0;
    for (int s = 0; s < S; s++) {
        if (sos[6 * s + 2] == 0.0) nb++;
        if (sos[6 * s + 5] == 0.0) na++;
    }
    p->edge = 3 * (2 * S + 1 - (nb < na ? nb : na));
    return HIPDSP_OK;
}

// Choose the number of time segments per channel.  One wave per (channel, segment);
// a segment costs its own length plus the warm-up it re-reads, and waves run in rounds
// of `slots` resident waves: minimise rounds * (segment + warm-up).
void plan_segments(const hipdsp_ctx *ctx, long long N, long long channels, long long warm,
                   long long *seg_len, int *n_seg)
{
    const long long slots = (long long)ctx->n_cus * (ctx->sos_waves_per_cu > 0 ? ctx->sos_waves_per_cu : 16);
    long long max_seg = (N + TILE - 1) / TILE;            // at least one tile per segment
    if (warm >= (1LL << 40)) max_seg = 1;                 // non-decaying filter: never segment
    if (ctx->max_segments > 0 && max_seg > ctx->max_segments) max_seg = ctx->max_segments;
    if (max_seg > 65536) max_seg = 65536;
    if (max_seg < 1) max_seg = 1;
    long long best_n = 1, best_len = (N + TILE - 1) / TILE * TILE;
    double best_cost = -1.0;
    // candidates: segment counts that fill whole rounds, and powers of two below one round
    for (long long rounds = 0; rounds <= 64; rounds++) {
        for (int half = 0; half < (rounds == 0 ? 16 : 1); half++) {
            long long n = rounds == 0 ? ((slots / channels) >> half) : rounds * slots / channels;
            if (n < 1) n = 1;
            if (n > max_seg) n = max_seg;
            long long len = ((N + n - 1) / n + TILE - 1) / TILE * TILE;
            if (len < TILE) len = TILE;
            long long cnt = (N + len - 1) / len;
            long long r = (channels * cnt + slots - 1) / slots;
            double cost = (double)r * (double)(len + (cnt > 1 ? warm : 0));
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_n = cnt; best_len = len; }
        }
    }
    *seg_len = best_len;
    *n_seg = (int)best_n;
}

template <int MODE>
int launch_scan(hipdsp_ctx *ctx, const hipdsp_sosplan *plan, const SosPlanDev *dev, int S, SeqArgs a,
                long long channels, long long warm)
{
    plan_segments(ctx, a.N, channels, warm, &a.seg_len, &a.n_seg);
    long long blocks = channels * a.n_seg;
    if (blocks > 0x7fffffffLL) {
        hipdsp_set_error("grid too large (%lld blocks)", blocks);
        return HIPDSP_ERR_INVALID;
    }
    dim3 grid((unsigned)blocks), block(64);
    switch (S) {
    case 1: hipLaunchKernelGGL((sos_scan_kernel<1, MODE>), grid, block, 0, ctx->stream, dev, a); break;
    case 2: hipLaunchKernelGGL((sos_scan_kernel<2, MODE>), grid, block, 0, ctx->stream, dev, a); break;
    case 3: hipLaunchKernelGGL((sos_scan_kernel<3, MODE>), grid, block, 0, ctx->stream, dev, a); break;
    case 4: hipLaunchKernelGGL((sos_scan_kernel<4, MODE>), grid, block, 0, ctx->stream, dev, a); break;
    default:
        hipdsp_set_error("n_sections %d not in 1..%d", S, MAXS);
        return HIPDSP_ERR_UNSUPPORTED;
    }
    (void)plan;
    return hd_launch_status("sos_scan_kernel");
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
    int rc = fill_plan(&tmp, host_sos, n_sections);
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

int hipdsp_sos_plan_host(const double *host_sos, int n_sections, int64_t *warmup, int *edge, double *zi)
{
    HD_REQUIRE(host_sos != nullptr, "host_sos is NULL");
    if (n_sections < 1 || n_sections > MAXS) {
        hipdsp_set_error("n_sections %d not in 1..%d", n_sections, MAXS);
        return HIPDSP_ERR_UNSUPPORTED;
    }
    SosPlanDev tmp;
    int rc = fill_plan(&tmp, host_sos, n_sections);
    if (rc != HIPDSP_OK) return rc;
    if (warmup) *warmup = tmp.warm;
    if (edge) *edge = tmp.edge;
    if (zi)
        for (int k = 0; k < 2 * n_sections; k++) zi[k] = tmp.zi[k];
    return HIPDSP_OK;
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
    a.T = frames; a.N = frames; a.skip = skip; a.edge = 0; a.rectify = 0; a.gain = 1.f; a.clamp = 0;
    return launch_scan<MODE_FILT>(ctx, plan, plan->dev, plan->host->n_sections, a, channels,
                                  plan->host->warm);
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
    const int edge = eplan->host->edge;
    if (frames <= edge) {
        hipdsp_set_error("The length of the input vector x must be greater than padlen, which is %d.", edge);
        return HIPDSP_ERR_TOO_SHORT;
    }
    if (channels == 0) return HIPDSP_OK;
    HD_REQUIRE(x != nullptr && yf != nullptr && env != nullptr, "NULL data pointer");
    HD_REQUIRE(x_pitch >= frames && yf_pitch >= frames && env_pitch >= frames, "pitch smaller than row length");
    const long long N = frames + 2LL * edge;
    const long long wpitch = (N + 3) / 4 * 4;
    void *work = nullptr;
    int rc = hipdsp_scratch(ctx, sizeof(float) * (size_t)wpitch * (size_t)channels, &work);
    if (rc != HIPDSP_OK) return rc;
    FusedArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.in = x; fa.yf = yf; fa.w = (float *)work;
    fa.in_pitch = x_pitch; fa.yf_pitch = yf_pitch; fa.w_pitch = wpitch;
    fa.T = frames; fa.edge = edge; fa.rectify = rectify; fa.gain = (float)gain;
    long long warm = fplan->host->warm + eplan->host->warm;
    if (fplan->host->warm >= (1LL << 40) || eplan->host->warm >= (1LL << 40)) warm = 1LL << 50;
    plan_segments(ctx, frames, channels, warm, &fa.seg_len, &fa.n_seg);
    long long blocks = channels * fa.n_seg;
    HD_REQUIRE(blocks <= 0x7fffffffLL, "grid too large");
    dim3 grid((unsigned)blocks), block(64);
    const int SF = fplan->host->n_sections, SE = eplan->host->n_sections;
    if (phase != 2) {
#define HD_FUSED(A, B)                                                                                  \
    case (A) * 8 + (B):                                                                                 \
        hipLaunchKernelGGL((sos_fused_kernel<A, B>), grid, block, 0, ctx->stream, fplan->dev, eplan->dev, fa); \
        break
    switch (SF * 8 + SE) {
        HD_FUSED(1, 1); HD_FUSED(1, 2); HD_FUSED(1, 3); HD_FUSED(1, 4);
        HD_FUSED(2, 1); HD_FUSED(2, 2); HD_FUSED(2, 3); HD_FUSED(2, 4);
        HD_FUSED(3, 1); HD_FUSED(3, 2); HD_FUSED(3, 3); HD_FUSED(3, 4);
        HD_FUSED(4, 1); HD_FUSED(4, 2); HD_FUSED(4, 3); HD_FUSED(4, 4);
    default:
        hipdsp_set_error("n_sections %d / %d not in 1..%d", SF, SE, MAXS);
        return HIPDSP_ERR_UNSUPPORTED;
    }
#undef HD_FUSED
    rc = hd_launch_status("sos_fused_kernel");
    if (rc != HIPDSP_OK) return rc;
    }
    if (phase == 1) return HIPDSP_OK;
    if (ctx->mid_event) HD_CHECK_HIP(hipEventRecord(ctx->mid_event, ctx->stream));
    // backward pass over the reversed scratch -> env (trim the padding, clamp)
    SeqArgs a;
    memset(&a, 0, sizeof(a));
    a.T = frames; a.N = N; a.skip = 0; a.edge = edge;
    a.rectify = rectify; a.gain = (float)gain; a.clamp = clamp;
    a.in = (const float *)work; a.in_pitch = wpitch; a.out = env; a.out_pitch = env_pitch;
    return launch_scan<MODE_ENV_BWD>(ctx, eplan, eplan->dev, SE, a, channels, eplan->host->warm);
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
                           (long long)y_pitch, n);
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
    const long long N = frames + 2LL * edge;
    const long long wpitch = (N + 3) / 4 * 4;
    void *work = nullptr;
    int rc = hipdsp_scratch(ctx, sizeof(float) * (size_t)wpitch * (size_t)channels, &work);
    if (rc != HIPDSP_OK) return rc;
    SeqArgs a;
    memset(&a, 0, sizeof(a));
    a.T = frames; a.N = N; a.skip = skip; a.edge = edge;
    a.rectify = rectify; a.gain = (float)gain; a.clamp = clamp;
    // forward pass over the odd-extended, rectified input -> scratch
    a.in = x; a.in_pitch = x_pitch; a.out = (float *)work; a.out_pitch = wpitch;
    rc = launch_scan<MODE_ENV_FWD>(ctx, plan, plan->dev, plan->host->n_sections, a, channels,
                                   plan->host->warm);
    if (rc != HIPDSP_OK) return rc;
    if (ctx->mid_event) HD_CHECK_HIP(hipEventRecord(ctx->mid_event, ctx->stream));
    if (frames - skip == 0) return HIPDSP_OK;
    // backward pass over the reversed scratch -> y (trim edge, skip, clamp)
    a.in = (const float *)work; a.in_pitch = wpitch; a.out = y; a.out_pitch = y_pitch;
    return launch_scan<MODE_ENV_BWD>(ctx, plan, plan->dev, plan->host->n_sections, a, channels,
                                     plan->host->warm);
}

}  // extern "C"
