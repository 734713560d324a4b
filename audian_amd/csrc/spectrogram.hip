// Framed Hann STFT -> one-sided PSD for gfx950: BufferedSpectrogram.process of
// bendalab/audian (src/audian/bufferedspectrogram.py:45-59), i.e.
// scipy.signal.spectrogram(window='hann', detrend='constant', scaling='density',
// mode='psd') per channel, with the optional fused decibel epilogue
// (thunderlab decibel, src/audian/specitem.py:36).
#include "common.h"
#include "fft_device.h"
#include <cmath>
#include <type_traits>

namespace {


// ---- generic path: any power-of-two nfft in [8, 8192] ---------------------------
// One 256-thread workgroup per (frame, channel); radix-2 Stockham autosort in LDS.  Every size
// has a faster kernel below; this one stays as their in-tree cross-check ("force_generic_fft").
__global__ __launch_bounds__(256) void spec_generic_kernel(
    const float *__restrict__ x, long long x_pitch, long long n_valid, long long frames_out,
    long long out_pitch, int nfft, int hop, float scale, float *__restrict__ out,
    float *__restrict__ db_out)
{
    extern __shared__ float2 fftbuf[];        // 2 * nfft
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const long long frame = blockIdx.x;
    const long long ch = blockIdx.y;
    const int F = nfft / 2 + 1;
    const long long obase = ch * out_pitch + frame * (long long)F;
    if (frame >= n_valid) {                   // zero tail (bufferedspectrogram.py:59)
        for (int f = tid; f < F; f += 256) {
            out[obase + f] = 0.f;
            if (db_out) db_out[obase + f] = -INFINITY;
        }
        return;
    }
    const float *seg = x + ch * x_pitch + frame * (long long)hop;
    // detrend='constant': subtract the frame mean -- sum and subtraction in float64 (the streamed kernels get the same
    // effect from a pivot near the mean, spec_pack.h; this kernel is a cross-check path and can afford the plain way)
    double s = 0.0;
    for (int i = tid; i < nfft; i += 256) s += (double)seg[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const double mean = (red[0] + red[1] + red[2] + red[3]) / (double)nfft;
    float2 *in = fftbuf, *ou = fftbuf + nfft;
    for (int i = tid; i < nfft; i += 256) {
        float w = 0.5f - 0.5f * cospif(2.0f * (float)i / (float)nfft);   // periodic Hann
        in[i] = make_float2((float)((double)seg[i] - mean) * w, 0.f);
    }
    __syncthreads();
    const int half = nfft >> 1;
    for (int Ns = 1; Ns < nfft; Ns <<= 1) {
        for (int j = tid; j < half; j += 256) {
            int k = j & (Ns - 1);
            float sn, cs;
            sincospif(-(float)k / (float)Ns, &sn, &cs);
            float2 v0 = in[j], v1 = in[j + half];
            float2 t = make_float2(v1.x * cs - v1.y * sn, v1.x * sn + v1.y * cs);
            int j0 = ((j - k) << 1) + k;
            ou[j0] = make_float2(v0.x + t.x, v0.y + t.y);
            ou[j0 + Ns] = make_float2(v0.x - t.x, v0.y - t.y);
        }
        __syncthreads();
        float2 *tmp = in; in = ou; ou = tmp;
    }
    for (int f = tid; f < F; f += 256) {
        float2 v = in[f];
        float p = (v.x * v.x + v.y * v.y) * scale;
        if (f != 0 && f != F - 1) p *= 2.f;
        out[obase + f] = p;
        if (db_out) db_out[obase + f] = to_db(p);
    }
}


// ---- fast path: nfft in {256, 512, 1024, 2048, 4096} ------------------------------
// The real FFT of a frame is taken as ONE complex FFT of half the length
// (z[n] = x[2n] + i x[2n+1], M = nfft/2) plus a split step.  A frame is owned by LPF
// lanes of one wave, PPL = M/LPF points per lane, and runs as three Stockham stages of
// radix R1 x R2 x R3 = M: every butterfly is an in-register DFT, the exchanges between
// stages go through a per-frame LDS buffer and need no barrier because the whole frame
// lives in one wave (LDS operations of a wave execute in order).  Stage 1 reads the
// samples straight from HBM (8 B per lane, contiguous over lanes), subtracts the frame
// mean and applies the LDS-staged Hann window.  Twiddles come from LDS tables computed on
// the host in float64.

// Three workgroups per CU (<= 168 VGPRs); the dB epilogue does not fit in that (it spilled 44
// registers to scratch) and runs two per CU instead, which measured 15 % faster than spilling.
template <int NFFT, int LPF, int R1, int R2, int R3, int WAVES, bool DB, int REUSE>
__global__ __launch_bounds__(64 * WAVES, (NFFT <= 2048 ? (DB ? 2 : 3) : 1) * WAVES / 4) void spec_fast_kernel(
    const float *__restrict__ x, long long x_pitch, long long n_valid, long long frames_out,
    long long out_pitch, int hop, float scale, const float *__restrict__ tables, float *__restrict__ out,
    float *__restrict__ db_out, int frames_per_wave)
{
    constexpr int M = NFFT / 2;
    constexpr int PPL = M / LPF;
    constexpr int G = 64 / LPF;              // frames processed side by side in one wave
    constexpr int F = M + 1;
    constexpr int MP = M + M / 16;           // padded frame buffer
    static_assert(R1 * R2 * R3 == M, "radices must multiply to M");
    static_assert(PPL % R1 == 0 && PPL % R2 == 0 && PPL % R3 == 0, "radix must divide points per lane");
    constexpr int NW = WAVES;                // waves per workgroup
    constexpr int TW2 = (R2 - 1) * R1;       // stage-2 twiddles [t-1][k], k < R1
    constexpr int TW3 = R1 * R2;             // stage-3 twiddles W_M^k, k < R1*R2 (powers on the fly)
    constexpr int TWN = M / 2 + 1;           // split step: exp(-2 pi i k / NFFT)
    constexpr int NTAB = TW2 + TW3 + TWN + M;   // + window as M float2
    __shared__ float2 smem[NTAB + NW * G * MP];
    const float2 *tw2 = smem;
    const float2 *tw3 = smem + TW2;
    const float2 *twn = smem + TW2 + TW3;
    const float2 *win = smem + TW2 + TW3 + TWN;      // (w[2n], w[2n+1])
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int g = lane / LPF, l = lane % LPF;
    float2 *fb = smem + NTAB + (wave * G + g) * MP;

    {   // device tables are stored in exactly this order
        const float2 *src = reinterpret_cast<const float2 *>(tables);
        for (int i = tid; i < NTAB; i += 64 * NW) smem[i] = src[i];
    }
    __syncthreads();

    const long long ch = blockIdx.y;
    const float *xc = x + ch * x_pitch;
    float *oc = out + ch * out_pitch;
    float *dc = DB ? db_out + ch * out_pitch : nullptr;
    // first frame of this wave; wave-uniform (scalar) when a frame takes the whole wave
    const long long first = ((long long)blockIdx.x * NW + wave) * (long long)frames_per_wave * G;

    // Raw samples of one frame: the lane keeps z[n], n = l + LPF*u + t*M/R1, as FOUR quarters of the frame
    // (quarter k: R1/4 * k <= t < R1/4 * (k + 1), i.e. samples [k NFFT/4, (k + 1) NFFT/4)).  The loads are inline
    // asm so that the wait for them can be counted by hand: the next frame is fetched as soon as the window has
    // consumed this frame's raw registers, i.e. BEFORE this frame's stores, and `s_waitcnt vmcnt(NST)` at the
    // top of the next frame retires the loads while the NST younger stores stay in flight (hipcc would wait
    // vmcnt(0), i.e. for every store acknowledgement).
    // REUSE = 2 (hop == NFFT/2, one frame per wave): the upper half of frame f IS the lower half of frame f + 1
    // in the same lane (n' = n - M/2 <=> t' = t - R1/2), so only the new half is fetched and the register
    // sets rotate by two quarters per frame: each sample is requested once instead of twice (PMC: 22.1 GB ->
    // 14.8 GB read per launch).  REUSE = 4 (hop == NFFT/4, the 75 % overlap of BASELINE configs[1]): the sets
    // rotate by one quarter and one quarter is fetched -- once instead of four times.
    // (the dB epilogue needs the registers of the long windows: it would spill; 512 and the 1024 without register reuse have them)
    constexpr bool EARLY_PF = !DB || REUSE == 4;
    static_assert(R1 % 4 == 0, "quarters of the first radix");
    constexpr int R1Q = R1 / 4;
    constexpr int QP = PPL / 4;                            // points per quarter
    v2f rq0[QP], rq1[QP], rq2[QP], rq3[QP];
#pragma unroll
    for (int i = 0; i < QP; i++) {
        rq0[i] = (v2f){0.f, 0.f}; rq1[i] = (v2f){0.f, 0.f}; rq2[i] = (v2f){0.f, 0.f}; rq3[i] = (v2f){0.f, 0.f};
    }
    float pvs = 0.f;                                       // pivot of the frame mean, per lane group (body())
    auto fetch_quarter = [&](long long frame, auto which, v2f *dst) {
        constexpr int K = decltype(which)::value;
        const float *seg = xc + frame * (long long)hop + 2 * l;
#pragma unroll
        for (int u = 0; u < PPL / R1; u++)
#pragma unroll
            for (int t = 0; t < R1Q; t++) {
                constexpr int CH = 512;                    // float2 per 4096-byte window
                const int n0 = LPF * u + (K * R1Q + t) * (M / R1);     // compile-time after unrolling
                asm_load8(dst[u * R1Q + t], seg + 2 * (n0 / CH) * CH, (n0 % CH) * 8);
            }
    };
    using Q0 = std::integral_constant<int, 0>;
    using Q1 = std::integral_constant<int, 1>;
    using Q2 = std::integral_constant<int, 2>;
    using Q3 = std::integral_constant<int, 3>;
    // stores behind a prefetch in the steady-state loop; one less than issued, so the wait
    // stays sufficient even if the compiler ever merged two of them
    constexpr int NST0 = (DB ? 2 : 1) * (PPL + 1) - 1;
    constexpr int NST = NST0 > 63 ? 63 : NST0;        // vmcnt is a 6-bit field

    const long long last_valid = n_valid > 0 ? n_valid - 1 : 0;
    const int partner = g * LPF + ((LPF - l) & (LPF - 1));

    // One frame per lane group from the raw halves (lo, hi).  `keep` masks the stores of
    // lane groups whose frame is not valid (only in the one mixed iteration of a wave).
    // With PF the next frame is prefetched (REUSE 2 / 4: only its new half / quarter); the steady-state
    // body has no divergent branch around its stores so that their count is exact.
    auto body = [&](long long frame, bool keep, auto full, auto pf, auto waitn, v2f *qa, v2f *qb, v2f *qc, v2f *qd) {
        constexpr bool FULL = decltype(full)::value;
        constexpr bool PF = decltype(pf)::value;
        constexpr int WAITN = decltype(waitn)::value;
        if (WAITN >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN < 0 ? 0 : WAITN) : "memory");
        float *o = oc + frame * (long long)F;
        float *od = DB ? dc + frame * (long long)F : nullptr;
        float2 v[PPL];
        v2f acc = {0.f, 0.f};                             // even and odd samples side by side (v_pk_add_f32)
        // the frame mean relative to a PIVOT, the mean of the frame this lane group transformed before (`pvs`; the run's
        // first frame: the two steps in front of the first call) -- chain.hip's psd_frame has the two cases that ask for it
        auto gsum = [&](float sum) {
            if (LPF == 64) return wave_sum(sum);
#pragma unroll
            for (int d = LPF / 2; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
            return sum;
        };
        // TWOSTEP (the variants without register reuse, i.e. every hop but nfft / 2 and nfft / 4 -- hop = nfft among them: the
        // frame before need not overlap this one at all): the frame's own rough mean as its pivot, in every frame.  Behind a
        // step in the level the differences to the frame before are the size of the step, and at 3e-8 of the step their
        // per-sample rounding alone is 3e-4 of a flat frame's peak for 5000 sigma; the differences to the frame's own mean
        // are the size of what the frame holds.
        constexpr bool TWOSTEP = REUSE == 1;
        if (TWOSTEP) {
            const v2f p02 = {pvs, pvs};
            v2f a0 = {0.f, 0.f};
#pragma unroll
            for (int u = 0; u < PPL / R1; u++)
#pragma unroll
                for (int t = 0; t < R1; t++) {
                    v2f *const qs[4] = {qa, qb, qc, qd};
                    v2f &r = qs[t / R1Q][u * R1Q + t % R1Q];
                    asm volatile("" : "+v"(r));           // not before the counted wait
                    a0 += r - p02;
                }
            const float c = pvs + gsum(a0.x + a0.y) * (1.0f / (float)NFFT);
            pvs = (fabsf(c) <= 3.0e38f) ? c : pvs;
        }
        const v2f pivot2 = {pvs, pvs};
#pragma unroll
        for (int u = 0; u < PPL / R1; u++)
#pragma unroll
            for (int t = 0; t < R1; t++) {
                v2f *const qs[4] = {qa, qb, qc, qd};
                v2f &r = qs[t / R1Q][u * R1Q + t % R1Q];
                asm volatile("" : "+v"(r));               // not before the counted wait
                const v2f d = r - pivot2;
                v[u * R1 + t] = make_float2(d.x, d.y);
                acc += d;
            }
        const float s = gsum(acc.x + acc.y);
        const float mean = s * (1.0f / (float)NFFT);
        const v2f mean2 = {mean, mean};
        {
            const float c = pvs + mean;                   // the next frame's pivot (a NaN or Inf in this frame: unchanged)
            pvs = (fabsf(c) <= 3.0e38f) ? c : pvs;
        }
        // CORR (the variants WITH register reuse; the others have taken two steps): what the subtraction leaves.  After a
        // step in the trace's level the differences to the mean of the frame before are all large, `mean` is good to 6e-8
        // of THEM, and the Hann window puts that error times nfft / 2 into bins 0 and 1 (9e-5 of a flat frame's peak for a
        // step of 1000 sigma at 2048 / 1024).  The detrended samples are summed once more; their mean m1 under the window
        // is m1 nfft / 2 in bin 0, -m1 nfft / 4 in bin 1 and nothing elsewhere: the split step removes it.
        constexpr bool CORR = !TWOSTEP;
        v2f rest = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < PPL / R1; u++)
#pragma unroll
            for (int t = 0; t < R1; t++) {
                const v2f w = as_v2f(win[l + LPF * u + t * (M / R1)]);
                float2 &e = v[u * R1 + t];
                const v2f q = as_v2f(e) - mean2;
                if (CORR) rest += q;
                e = as_f2(q * w);
            }
        const float corr = CORR ? 0.5f * gsum(rest.x + rest.y) : 0.f;      // m1 nfft / 2
        // the raw registers are dead from here on: request the next frame now, so that the whole
        // FFT of this one hides the latency
        if (PF && EARLY_PF) {
            const long long nf = frame + G;
            const long long cf = nf < last_valid ? nf : last_valid;
            if (REUSE == 4) {
                fetch_quarter(cf, Q3(), qa);              // qb, qc, qd stay: they are the next frame's first three quarters
            } else if (REUSE == 2) {
                fetch_quarter(cf, Q2(), qa);              // qc, qd stay: they are the next lower half
                fetch_quarter(cf, Q3(), qb);
            } else {
                fetch_quarter(cf, Q0(), qa); fetch_quarter(cf, Q1(), qb);
                fetch_quarter(cf, Q2(), qc); fetch_quarter(cf, Q3(), qd);
            }
        }
        stockham_stage<R1, 1, M, LPF, false, true>(v, fb, tw2, l);
        stockham_stage<R2, R1, M, LPF, true, true>(v, fb, tw2, l);
        stockham_stage<R3, R1 * R2, M, LPF, true, false, true>(v, fb, tw3, l);
        if (PF && !EARLY_PF) {
            const long long nf = frame + G;
            const long long cf = nf < last_valid ? nf : last_valid;
            if (REUSE == 4) {
                fetch_quarter(cf, Q3(), qa);              // qb, qc, qd stay: they are the next frame's first three quarters
            } else if (REUSE == 2) {
                fetch_quarter(cf, Q2(), qa);              // qc, qd stay: they are the next lower half
                fetch_quarter(cf, Q3(), qb);
            } else {
                fetch_quarter(cf, Q0(), qa); fetch_quarter(cf, Q1(), qb);
                fetch_quarter(cf, Q2(), qc); fetch_quarter(cf, Q3(), qd);
            }
        }
        // Now v[u*R3 + t] = Z[k], k = l + LPF*m, m = u + NB3*t.  Split step for m < PPL/2
        // (k < M/2): X[k] = E + W^k O, X[M-k] = conj(E - W^k O); the partner bin Z[M-k] sits
        // in lane LPF-l at m' = PPL-1-m (lane 0: in itself at m' = PPL-m) and comes over
        // with ds_bpermute instead of a third trip through LDS memory.
        constexpr int NB3 = PPL / R3;
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
            // E = (zk + conj zm)/2, O = -i (zk - conj zm)/2; the halves go into the scale.
            // X[k] = E + W^k O, X[M-k] = conj(E - W^k O): real parts in `re`, imaginary in `im`
            const v2f e = pk_add_conj(as_v2f(zk), as_v2f(zm));
            const v2f t = pk_cmul_negi(pk_sub_conj(as_v2f(zk), as_v2f(zm)), as_v2f(twn[k]));
            v2f re = pk_sumdiff_x(e, t);
            const v2f im = pk_sumdiff_y(e, t);
            if (CORR && m == 0) re.x += (l == 1) ? corr : 0.f;       // bin 1 (re: twice its real part)
            const v2f pw = (re * re + im * im) * hscale2;
            float pk = pw.x, pm = pw.y;
            if (m == 0) {
                // bin 0 pairs with itself: DC = re + im, Nyquist = re - im, not doubled
                const float dc0 = zk.x + zk.y - corr, ny = zk.x - zk.y;
                pk = (l == 0) ? dc0 * dc0 * scale : pk;
                pm = (l == 0) ? ny * ny * scale : pm;
            }
            if (FULL || keep) {
                o[k] = pk;
                o[M - k] = pm;
                if (DB) { od[k] = to_db(pk); od[M - k] = to_db(pm); }
            }
            pk_last = pk;
        }
        {   // bin M/2 pairs with itself (lane 0, m = PPL/2); the other lanes repeat their
            // last store so that the instruction is unconditional
            constexpr int mh = PPL / 2;
            const float2 z = v[(mh % NB3) * R3 + mh / NB3];
            const float ph = 2.f * scale * (z.x * z.x + z.y * z.y);
            const int kk = (l == 0) ? M / 2 : l + LPF * (PPL / 2 - 1);
            const float pv = (l == 0) ? ph : pk_last;
            if (FULL || keep) {
                o[kk] = pv;
                if (DB) od[kk] = to_db(pv);
            }
        }
    };

    using T_ = std::true_type;
    using F_ = std::false_type;
    using W0 = std::integral_constant<int, 0>;
    using WN = std::integral_constant<int, NST>;
    using WX = std::integral_constant<int, -1>;
    // iterations in which every lane group of the wave has a valid frame
    long long n_main = (n_valid - first) / G;
    if (n_main < 0) n_main = 0;
    if (n_main > frames_per_wave) n_main = frames_per_wave;
    {
        const long long f0 = first + g;
        const long long c0 = f0 < last_valid ? f0 : last_valid;
        fetch_quarter(c0, Q0(), rq0); fetch_quarter(c0, Q1(), rq1);
        fetch_quarter(c0, Q2(), rq2); fetch_quarter(c0, Q3(), rq3);
    }
    {
        // the pivot of the run's first frame in two steps: a sample as the pivot of a rough mean, that mean as the pivot
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < QP; i++) {
            asm volatile("" : "+v"(rq0[i])); asm volatile("" : "+v"(rq1[i]));
            asm volatile("" : "+v"(rq2[i])); asm volatile("" : "+v"(rq3[i]));
        }
        float p0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(rq0[0].x)));
        p0 = (fabsf(p0) <= 3.0e38f) ? p0 : 0.f;
        const v2f p02 = {p0, p0};
        v2f a0 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < QP; i++) a0 += ((rq0[i] - p02) + (rq1[i] - p02)) + ((rq2[i] - p02) + (rq3[i] - p02));
        float s0 = a0.x + a0.y;
        if (LPF == 64) {
            s0 = wave_sum(s0);
        } else {
#pragma unroll
            for (int d = LPF / 2; d >= 1; d >>= 1) s0 += __shfl_xor(s0, d, 64);
        }
        const float c = p0 + s0 * (1.0f / (float)NFFT);
        pvs = (fabsf(c) <= 3.0e38f) ? c : p0;
    }
    int it = 0;
    // (the steady-state loop sits inside this branch so that no path of the generated code can
    // reach a counted wait without the stores it counts: tools/check_prefetch_isa.py walks all
    // paths and knows nothing about n_main)
    if (n_main > 0) {
        body(first + g, true, T_(), T_(), W0(), rq0, rq1, rq2, rq3);
        it = 1;
        const int nm = (int)n_main;
        auto fr = [&](int i) { return first + (long long)i * G + g; };
        if (REUSE == 4) {
            // the register sets rotate by one quarter per frame: (a, b, c, d) -> (b, c, d, a) -> ...
            for (; it + 3 < nm; it += 4) {
                body(fr(it), true, T_(), T_(), WN(), rq1, rq2, rq3, rq0);
                body(fr(it + 1), true, T_(), T_(), WN(), rq2, rq3, rq0, rq1);
                body(fr(it + 2), true, T_(), T_(), WN(), rq3, rq0, rq1, rq2);
                body(fr(it + 3), true, T_(), T_(), WN(), rq0, rq1, rq2, rq3);
            }
            if (it < nm) { body(fr(it), true, T_(), T_(), WN(), rq1, rq2, rq3, rq0); it++; }
            if (it < nm) { body(fr(it), true, T_(), T_(), WN(), rq2, rq3, rq0, rq1); it++; }
            if (it < nm) { body(fr(it), true, T_(), T_(), WN(), rq3, rq0, rq1, rq2); it++; }
        } else if (REUSE == 2) {
            // ... by two quarters: (a, b, c, d) -> (c, d, a, b) -> (a, b, c, d)
            for (; it + 1 < nm; it += 2) {
                body(fr(it), true, T_(), T_(), WN(), rq2, rq3, rq0, rq1);
                body(fr(it + 1), true, T_(), T_(), WN(), rq0, rq1, rq2, rq3);
            }
            if (it < nm) { body(fr(it), true, T_(), T_(), WN(), rq2, rq3, rq0, rq1); it++; }
        } else {
            for (; it < nm; it++) body(fr(it), true, T_(), T_(), WN(), rq0, rq1, rq2, rq3);
        }
    }
    // retire the last prefetch before anything else may reuse its registers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < QP; i++) {
        asm volatile("" : "+v"(rq0[i])); asm volatile("" : "+v"(rq1[i]));
        asm volatile("" : "+v"(rq2[i])); asm volatile("" : "+v"(rq3[i]));
    }
    // at most one mixed iteration (G > 1 only, never with REUSE > 1), then the zero tail
    // (bufferedspectrogram.py:59)
    for (; it < frames_per_wave; it++) {
        const long long frame = first + (long long)it * G + g;
        if (first + (long long)it * G >= frames_out) break;
        if (G > 1 && it == (int)n_main && first + (long long)it * G < n_valid)
            body(frame < last_valid ? frame : last_valid, frame < n_valid, F_(), F_(), WX(), rq0, rq1, rq2, rq3);
        if (frame >= n_valid && frame < frames_out) {
            float *o = oc + frame * (long long)F;
            for (int f = l; f < F; f += LPF) {
                o[f] = 0.f;
                if (DB) dc[frame * (long long)F + f] = -INFINITY;
            }
        }
    }
}

// ---- two-stage fast path (M = R1 x R2): one LDS exchange per frame -------------------
// LPF lanes own a frame, PPL = M/LPF = max(R1, R2) points per lane.  Stage 1 (radix R1)
// takes its inputs straight from HBM, stage 2 (radix R2) leaves lane l with the bins
// k = l + LPF*m in natural order, and the split step fetches the partner bin M-k from
// lane LPF-l with ds_bpermute (no third trip through LDS memory).
template <int NFFT, int LPF, int R1, int R2, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void spec2_kernel(
    const float *__restrict__ x, long long x_pitch, long long n_valid, long long frames_out,
    long long out_pitch, int hop, float scale, const float *__restrict__ tables, float *__restrict__ out,
    float *__restrict__ db_out, int frames_per_wave)
{
    constexpr int M = NFFT / 2;
    constexpr int PPL = M / LPF;
    constexpr int G = 64 / LPF;
    constexpr int F = M + 1;
    constexpr int MP = M + M / 16;
    constexpr int NB1 = PPL / R1, NB2 = PPL / R2;
    static_assert(R1 * R2 == M, "radices must multiply to M");
    static_assert(PPL % R1 == 0 && PPL % R2 == 0, "radix must divide points per lane");
    constexpr int TW2 = (R2 - 1) * R1;       // stage-2 twiddles [t-1][k], k < R1
    constexpr int TWN = M / 2 + 1;
    constexpr int NTAB = TW2 + TWN + M;      // + window as M float2
    __shared__ float2 smem[NTAB + WAVES * G * MP];
    const float2 *tw2 = smem;
    const float2 *twn = smem + TW2;
    const float2 *win = smem + TW2 + TWN;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int g = lane / LPF, l = lane % LPF;
    float2 *fb = smem + NTAB + (wave * G + g) * MP;
    {
        const float2 *src = reinterpret_cast<const float2 *>(tables);
        for (int i = tid; i < NTAB; i += 64 * WAVES) smem[i] = src[i];
    }
    __syncthreads();

    const long long ch = blockIdx.y;
    const float *xc = x + ch * x_pitch;
    // each lane group walks its own run of consecutive frames
    const long long first = (((long long)blockIdx.x * WAVES + wave) * G + g) * (long long)frames_per_wave;
    const int partner = g * LPF + ((LPF - l) & (LPF - 1));

    for (int it = 0; it < frames_per_wave; it++) {
        const long long frame = first + it;
        const bool active = frame < frames_out;
        const bool valid = frame < n_valid;
        const long long obase = ch * out_pitch + frame * (long long)F;
        if (active && !valid) {                              // zero tail
            for (int f = l; f < F; f += LPF) {
                out[obase + f] = 0.f;
                if (db_out) db_out[obase + f] = -INFINITY;
            }
        }
        // lanes without a frame still take part in the cross-lane steps below
        const float *seg = xc + (valid ? frame : 0) * (long long)hop;
        float2 v[PPL];
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < NB1; u++)
#pragma unroll
            for (int t = 0; t < R1; t++) {
                const int n = l + LPF * u + t * (M / R1);
                f2u r;
                if (!valid) { r.x = 0.f; r.y = 0.f; }
                else r = *reinterpret_cast<const f2u *>(seg + 2 * n);
                v[u * R1 + t] = make_float2(r.x, r.y);
                s += r.x + r.y;
            }
#pragma unroll
        for (int d = LPF / 2; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
        // the frame mean in two steps (this cross-check kernel can afford the plain way: the rough mean is subtracted, what
        // is left is summed again and its mean subtracted too; spec_pack.h has the cases a single float32 sum does not survive)
        const float mean0 = s * (1.0f / (float)NFFT);
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < PPL; i++) {
            v[i].x -= mean0; v[i].y -= mean0;
            s1 += v[i].x + v[i].y;
        }
#pragma unroll
        for (int d = LPF / 2; d >= 1; d >>= 1) s1 += __shfl_xor(s1, d, 64);
        const float mean = s1 * (1.0f / (float)NFFT);
#pragma unroll
        for (int u = 0; u < NB1; u++)
#pragma unroll
            for (int t = 0; t < R1; t++) {
                const float2 w = win[l + LPF * u + t * (M / R1)];
                float2 &e = v[u * R1 + t];
                e = make_float2((e.x - mean) * w.x, (e.y - mean) * w.y);
            }
        stockham_stage<R1, 1, M, LPF, false, true>(v, fb, tw2, l);
        stockham_stage<R2, R1, M, LPF, true, false>(v, fb, tw2, l);
        // now v[u*R2 + t] = Z[k], k = (l + LPF*u) + t*R1 = l + LPF*m with m = u + NB2*t
        // split step for m < PPL/2 (k < M/2): partner bin M-k sits in lane LPF-l at
        // m' = PPL-1-m (lane 0: in itself at m' = PPL-m)
#pragma unroll
        for (int m = 0; m < PPL / 2; m++) {
            const int k = l + LPF * m;
            const float2 zk = v[(m % NB2) * R2 + m / NB2];
            const int mp = PPL - 1 - m;
            const float2 zsrc = v[(mp % NB2) * R2 + mp / NB2];
            float2 zm;
            zm.x = __shfl(zsrc.x, partner, 64);
            zm.y = __shfl(zsrc.y, partner, 64);
            if (m > 0) {
                const int m0 = PPL - m;                      // lane 0 pairs inside itself
                const float2 z0 = v[(m0 % NB2) * R2 + m0 / NB2];
                if (l == 0) zm = z0;
            }
            float pk, pm;
            if (m == 0 && l == 0) {
                const float a = zk.x + zk.y, b = zk.x - zk.y;    // DC and Nyquist, not doubled
                pk = a * a * scale;
                pm = b * b * scale;
            } else {
                const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
                const float2 o = make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
                const float2 t = cmul(o, twn[k]);
                const float2 a = cadd(e, t), b = csub(e, t);
                pk = 2.f * scale * (a.x * a.x + a.y * a.y);
                pm = 2.f * scale * (b.x * b.x + b.y * b.y);
            }
            if (valid) {
                out[obase + k] = pk;
                out[obase + M - k] = pm;
                if (db_out) { db_out[obase + k] = to_db(pk); db_out[obase + M - k] = to_db(pm); }
            }
        }
        if (l == 0 && valid) {                               // k = M/2 pairs with itself
            constexpr int mh = PPL / 2;
            const float2 z = v[(mh % NB2) * R2 + mh / NB2];
            const float p = 2.f * scale * (z.x * z.x + z.y * z.y);
            out[obase + M / 2] = p;
            if (db_out) db_out[obase + M / 2] = to_db(p);
        }
    }
}

// ---- workgroup path: nfft 8192, 16384 and 32768 ------------------------------------------
// One frame per workgroup of LPF = 256 or 512 threads at a time: the same three Stockham stages
// as spec_fast_kernel with LPF "lanes" (16 or 32 points per thread), so the exchanges cross
// waves and every stage boundary is a workgroup barrier; a stage that loads AND stores needs one
// between the two as well (the exchange is in place).  The partner bin of the split step lives in
// another wave, so the transform takes one more trip through LDS in natural order.  Window and
// split twiddles stay in global memory (L2): with them in LDS only one workgroup would fit.
template <int NFFT, int LPF, int R1, int R2, int R3, int OCC, bool DB>
__global__ __launch_bounds__(LPF, OCC) void spec_wg_kernel(
    const float *__restrict__ x, long long x_pitch, long long n_valid, long long frames_out,
    long long out_pitch, int hop, float scale, const float *__restrict__ tables, float *__restrict__ out,
    float *__restrict__ db_out, int frames_per_block)
{
    constexpr int M = NFFT / 2;
    constexpr int PPL = M / LPF;
    constexpr int F = M + 1;
    constexpr int MP = M + M / 16;
    static_assert(R1 * R2 * R3 == M, "radices must multiply to M");
    static_assert(PPL % R1 == 0 && PPL % R2 == 0 && PPL % R3 == 0, "radix must divide points per thread");
    constexpr int NB3 = PPL / R3;
    constexpr int TW2 = (R2 - 1) * R1;
    constexpr int TW3 = R1 * R2;
    constexpr int TWN = M / 2 + 1;
    __shared__ float2 smem[TW2 + TW3 + MP];
    __shared__ float red[LPF / 64];
    const float2 *tw2 = smem;
    const float2 *tw3 = smem + TW2;
    float2 *fb = smem + TW2 + TW3;
    const float2 *gtab = reinterpret_cast<const float2 *>(tables);
    const float2 *twn = gtab + TW2 + TW3;
    const float2 *win = twn + TWN;
    const int l = threadIdx.x;
    const int wave = l >> 6;
    for (int i = l; i < TW2 + TW3; i += LPF) smem[i] = gtab[i];
    __syncthreads();

    const long long ch = blockIdx.y;
    const float *xc = x + ch * x_pitch;
    for (int it = 0; it < frames_per_block; it++) {
        const long long frame = (long long)blockIdx.x * frames_per_block + it;   // workgroup-uniform
        if (frame >= frames_out) break;
        const long long obase = ch * out_pitch + frame * (long long)F;
        if (frame >= n_valid) {                              // zero tail (bufferedspectrogram.py:59)
            for (int f = l; f < F; f += LPF) {
                out[obase + f] = 0.f;
                if (DB) db_out[obase + f] = -INFINITY;
            }
            continue;
        }
        const float *seg = xc + frame * (long long)hop;
        float2 v[PPL];
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < PPL / R1; u++)
#pragma unroll
            for (int t = 0; t < R1; t++) {
                const int n = l + LPF * u + t * (M / R1);
                const f2u r = *reinterpret_cast<const f2u *>(seg + 2 * n);
                v[u * R1 + t] = make_float2(r.x, r.y);
                s += r.x + r.y;
            }
        s = wave_sum(s);
        if ((l & 63) == 0) red[wave] = s;
        __syncthreads();          // also: the previous frame's partner reads of fb are done
        float total = 0.f;
#pragma unroll
        for (int w = 0; w < LPF / 64; w++) total += red[w];
        // (the frame mean in two steps, as in spec2_kernel)
        const float mean0 = total * (1.0f / (float)NFFT);
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < PPL; i++) {
            v[i].x -= mean0; v[i].y -= mean0;
            s1 += v[i].x + v[i].y;
        }
        s1 = wave_sum(s1);
        __syncthreads();          // (red[] has been read)
        if ((l & 63) == 0) red[wave] = s1;
        __syncthreads();
        float total1 = 0.f;
#pragma unroll
        for (int w = 0; w < LPF / 64; w++) total1 += red[w];
        const float mean = total1 * (1.0f / (float)NFFT);
#pragma unroll
        for (int u = 0; u < PPL / R1; u++)
#pragma unroll
            for (int t = 0; t < R1; t++) {
                const float2 w = win[l + LPF * u + t * (M / R1)];
                float2 &e = v[u * R1 + t];
                e = make_float2((e.x - mean) * w.x, (e.y - mean) * w.y);
            }
        stockham_stage<R1, 1, M, LPF, false, true>(v, fb, tw2, l);
        __syncthreads();
        stockham_stage<R2, R1, M, LPF, true, false>(v, fb, tw2, l);
        __syncthreads();
        stockham_stage<R2, R1, M, LPF, false, true, false, false>(v, fb, tw2, l);
        __syncthreads();
        stockham_stage<R3, R1 * R2, M, LPF, true, false, true>(v, fb, tw3, l);
        __syncthreads();
        // v[(m % NB3) * R3 + m / NB3] = Z[k], k = l + LPF m; natural order into LDS for the
        // partner bins Z[M-k]
#pragma unroll
        for (int m = 0; m < PPL; m++) fb[pad16(l) + LPF * m + LPF * m / 16] = v[(m % NB3) * R3 + m / NB3];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < PPL / 2; m++) {
            const int k = l + LPF * m;
            const float2 zk = v[(m % NB3) * R3 + m / NB3];
            const float2 zm = fb[pad16((M - k) & (M - 1))];
            float pk, pm;
            if (m == 0 && l == 0) {
                const float a = zk.x + zk.y, b = zk.x - zk.y;    // DC and Nyquist, not doubled
                pk = a * a * scale;
                pm = b * b * scale;
            } else {
                const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
                const float2 o = make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
                const float2 t = cmul(o, twn[k]);
                const float2 a = cadd(e, t), b = csub(e, t);
                pk = 2.f * scale * (a.x * a.x + a.y * a.y);
                pm = 2.f * scale * (b.x * b.x + b.y * b.y);
            }
            out[obase + k] = pk;
            out[obase + M - k] = pm;
            if (DB) { db_out[obase + k] = to_db(pk); db_out[obase + M - k] = to_db(pm); }
        }
        if (l == 0) {                                        // k = M/2 pairs with itself
            const float2 z = v[((PPL / 2) % NB3) * R3 + (PPL / 2) / NB3];
            const float p = 2.f * scale * (z.x * z.x + z.y * z.y);
            out[obase + M / 2] = p;
            if (DB) db_out[obase + M / 2] = to_db(p);
        }
    }
}

// ---- direct path: any nfft that is not a power of two ---------------------------------
// The reference clamps nfft to len(source)//2 (bufferedspectrogram.py:88-90), which on a short
// recording yields a non power of two.  Rare and small, so a plain O(nfft^2) DFT does: one
// thread per bin, the detrended, windowed frame staged through LDS in chunks, twiddles from
// sincospif on the exact fraction (k*n mod nfft)/nfft.
__global__ __launch_bounds__(256) void spec_direct_kernel(
    const float *__restrict__ x, long long x_pitch, long long n_valid, long long frames_out,
    long long out_pitch, int nfft, int hop, float scale, float *__restrict__ out,
    float *__restrict__ db_out)
{
    constexpr int CHUNK = 2048;
    __shared__ float xs[CHUNK];
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const long long frame = blockIdx.y;
    const long long ch = blockIdx.z;
    const int F = nfft / 2 + 1;
    const int k = blockIdx.x * 256 + tid;
    float *o = out + ch * out_pitch + frame * (long long)F;
    float *od = db_out ? db_out + ch * out_pitch + frame * (long long)F : nullptr;
    if (frame >= n_valid) {
        if (k < F) { o[k] = 0.f; if (od) od[k] = -INFINITY; }
        return;
    }
    const float *seg = x + ch * x_pitch + frame * (long long)hop;
    double s = 0.0;
    for (int i = tid; i < nfft; i += 256) s += (double)seg[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const double mean = (red[0] + red[1] + red[2] + red[3]) / (double)nfft;   // (subtracted in float64: an offset of any size)
    float re = 0.f, im = 0.f;
    long long idx = 0;                       // (k * n) mod nfft, advanced incrementally
    for (int base = 0; base < nfft; base += CHUNK) {
        const int len = nfft - base < CHUNK ? nfft - base : CHUNK;
        __syncthreads();
        for (int i = tid; i < len; i += 256) {
            const int n = base + i;
            const float w = 0.5f - 0.5f * cospif(2.0f * (float)n / (float)nfft);
            xs[i] = (float)((double)seg[n] - mean) * w;
        }
        __syncthreads();
        if (k < F) {
            for (int i = 0; i < len; i++) {
                float sn, cs;
                sincospif(-2.0f * (float)idx / (float)nfft, &sn, &cs);
                re = fmaf(xs[i], cs, re);
                im = fmaf(xs[i], sn, im);
                idx += k;
                if (idx >= nfft) idx -= nfft;
            }
        }
    }
    if (k < F) {
        float p = (re * re + im * im) * scale;
        const bool edge_bin = (k == 0) || ((nfft % 2 == 0) && k == F - 1);
        if (!edge_bin) p *= 2.f;
        o[k] = p;
        if (od) od[k] = to_db(p);
    }
}

// ---- large path: nfft = 2^14 ... 2^19 (the rest of the reference's nfft combo box,
// src/audian/databrowser.py:516) -----------------------------------------------------
// A frame no longer fits into LDS, so the half-length complex FFT (M = nfft/2 = N1*N2) runs
// as Bailey's four-step algorithm over a global scratch, 16 sub-transforms per workgroup so
// that every global access moves 128 contiguous bytes:
//   P0  mean of every frame                                   (detrend='constant')
//   P1  for each n2: FFT over n1 of w*(x - mean) at n = N2*n1 + n2, times W_M^(n2*k1) -> B[k1][n2]
//   P2  for each k1: FFT over n2 of B[k1][.]                  -> Z[k1 + N1*k2]
//   P3  split step + PSD scaling (+ dB)                       -> out
// Twiddles come from sincospif on exact dyadic arguments.  This path is about coverage of the
// reference's parameter range, not about the roofline: it moves ~6x the algorithmic bytes.

constexpr int BIG_W = 16;                 // sub-transforms per workgroup

__global__ __launch_bounds__(256) void big_mean_kernel(const float *__restrict__ x, long long x_pitch,
                                                       long long frames_valid, long long item0, int nfft,
                                                       int hop, double *__restrict__ mean)
{
    __shared__ double red[4];
    const long long item = item0 + blockIdx.x;
    const long long ch = item / frames_valid, frame = item % frames_valid;
    const float *seg = x + ch * x_pitch + frame * (long long)hop;
    // four independent partial sums: the loop is a chain of load latencies otherwise (nfft is a
    // multiple of 1024 on this path)
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int i = threadIdx.x; i < nfft; i += 1024) {
        s0 += (double)seg[i];
        s1 += (double)seg[i + 256];
        s2 += (double)seg[i + 512];
        s3 += (double)seg[i + 768];
    }
    double s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) mean[blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) / (double)nfft;
}

// BIG_W Stockham transforms of length n (power of two) side by side in LDS, radix-4 stages and
// one radix-2 stage when log2(n) is odd: element e of transform j at buf[j*n + e]; returns the
// buffer that holds the result.  `tw` (n entries of LDS) receives exp(-2 pi i m / n) once; a
// stage of radix R at Ns takes W^(k q) from entry k q n / (R Ns) instead of a sincospif per point.
__device__ float2 *big_fft_lds(float2 *a, float2 *b, float2 *tw, int n)
{
    for (int m = threadIdx.x; m < n; m += 256) {
        float sn, cs;
        sincospif(-2.0f * (float)m / (float)n, &sn, &cs);
        tw[m] = make_float2(cs, sn);
    }
    __syncthreads();
    const int quarter = n >> 2;
    int Ns = 1;
    for (; Ns * 4 <= n; Ns <<= 2) {
        const int stride = quarter / Ns;
        for (int i = threadIdx.x; i < BIG_W * quarter; i += 256) {
            const int j = i / quarter, e = i - j * quarter;
            const int k = e & (Ns - 1);
            const float2 *in = a + j * n + e;
            const float2 v0 = in[0];
            const float2 v1 = cmul(in[quarter], tw[k * stride]);
            const float2 v2 = cmul(in[2 * quarter], tw[2 * k * stride]);
            const float2 v3 = cmul(in[3 * quarter], tw[3 * k * stride]);
            const float2 t0 = cadd(v0, v2), t1 = csub(v0, v2), t2 = cadd(v1, v3), t3 = mul_negi(csub(v1, v3));
            float2 *o = b + j * n + ((e - k) << 2) + k;
            o[0] = cadd(t0, t2);
            o[Ns] = cadd(t1, t3);
            o[2 * Ns] = csub(t0, t2);
            o[3 * Ns] = csub(t1, t3);
        }
        __syncthreads();
        float2 *tmp = a; a = b; b = tmp;
    }
    if (Ns < n) {                              // one radix-2 stage is left: Ns == n / 2
        const int half = n >> 1;
        for (int i = threadIdx.x; i < BIG_W * half; i += 256) {
            const int j = i / half, e = i - j * half;
            const float2 v0 = a[j * n + e];
            const float2 t = cmul(a[j * n + e + half], tw[e]);
            b[j * n + e] = cadd(v0, t);
            b[j * n + e + half] = csub(v0, t);
        }
        __syncthreads();
        float2 *tmp = a; a = b; b = tmp;
    }
    return a;
}

__global__ __launch_bounds__(256) void big_pass1_kernel(const float *__restrict__ x, long long x_pitch,
                                                        long long frames_valid, long long item0, int nfft,
                                                        int hop, int N1, int N2, const double *__restrict__ mean,
                                                        float2 *__restrict__ B)
{
    extern __shared__ float2 big_lds[];                    // 2 * BIG_W * N1 + N1
    const long long item = item0 + blockIdx.y;
    const long long ch = item / frames_valid, frame = item % frames_valid;
    const float *seg = x + ch * x_pitch + frame * (long long)hop;
    const int n2_0 = blockIdx.x * BIG_W;
    const int M = nfft >> 1;
    const double mu = mean[blockIdx.y];                    // (float64 mean, subtracted in float64: an offset of any size)
    float2 *a = big_lds, *b = big_lds + BIG_W * N1;
    for (int i = threadIdx.x; i < BIG_W * N1; i += 256) {
        const int n1 = i / BIG_W, j = i - n1 * BIG_W;
        const int n = N2 * n1 + n2_0 + j;                  // complex sample index
        const float x0 = seg[2 * n], x1 = seg[2 * n + 1];
        const float w0 = 0.5f - 0.5f * cospif((float)(2 * n) / (float)M);        // 2 pi (2n) / nfft
        const float w1 = 0.5f - 0.5f * cospif((float)(2 * n + 1) / (float)M);
        a[j * N1 + n1] = make_float2((float)((double)x0 - mu) * w0, (float)((double)x1 - mu) * w1);
    }
    __syncthreads();
    float2 *r = big_fft_lds(a, b, big_lds + 2 * BIG_W * N1, N1);
    float2 *Bi = B + (long long)blockIdx.y * M;
    for (int i = threadIdx.x; i < BIG_W * N1; i += 256) {
        const int k1 = i / BIG_W, j = i - k1 * BIG_W;
        const int n2 = n2_0 + j;
        const int t = (int)(((long long)n2 * k1) & (M - 1));
        float sn, cs;
        sincospif(-2.0f * (float)t / (float)M, &sn, &cs);
        const float2 v = r[j * N1 + k1];
        Bi[(long long)k1 * N2 + n2] = make_float2(v.x * cs - v.y * sn, v.x * sn + v.y * cs);
    }
}

__global__ __launch_bounds__(256) void big_pass2_kernel(const float2 *__restrict__ B, int M, int N1, int N2,
                                                        float2 *__restrict__ Z)
{
    extern __shared__ float2 big_lds[];                    // 2 * BIG_W * N2 + N2
    const int k1_0 = blockIdx.x * BIG_W;
    const float2 *Bi = B + (long long)blockIdx.y * M;
    float2 *a = big_lds, *b = big_lds + BIG_W * N2;
    for (int i = threadIdx.x; i < BIG_W * N2; i += 256) {
        const int j = i / N2, n2 = i - j * N2;
        a[j * N2 + n2] = Bi[(long long)(k1_0 + j) * N2 + n2];
    }
    __syncthreads();
    float2 *r = big_fft_lds(a, b, big_lds + 2 * BIG_W * N2, N2);
    float2 *Zi = Z + (long long)blockIdx.y * M;
    for (int i = threadIdx.x; i < BIG_W * N2; i += 256) {
        const int k2 = i / BIG_W, j = i - k2 * BIG_W;
        Zi[(long long)k2 * N1 + k1_0 + j] = r[j * N2 + k2];
    }
}

__global__ __launch_bounds__(256) void big_split_kernel(const float2 *__restrict__ Z, int M,
                                                        long long frames_valid, long long frames_out,
                                                        long long out_pitch, long long item0, float scale,
                                                        float *__restrict__ out, float *__restrict__ db_out)
{
    const long long item = item0 + blockIdx.y;
    const long long ch = item / frames_valid, frame = item % frames_valid;
    const long long F = (long long)M + 1;
    float *o = out + ch * out_pitch + frame * F;
    float *od = db_out ? db_out + ch * out_pitch + frame * F : nullptr;
    const float2 *Zi = Z + (long long)blockIdx.y * M;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k > M / 2) return;
    const float2 zk = Zi[k];
    const float2 zm = Zi[(M - k) & (M - 1)];
    float pk, pm;
    if (k == 0) {
        const float a = zk.x + zk.y, b = zk.x - zk.y;
        pk = a * a * scale;
        pm = b * b * scale;
    } else {
        float sn, cs;
        sincospif(-(float)k / (float)M, &sn, &cs);         // exp(-2 pi i k / nfft)
        const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        const float2 od2 = make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
        const float2 t = make_float2(od2.x * cs - od2.y * sn, od2.x * sn + od2.y * cs);
        const float2 a = make_float2(e.x + t.x, e.y + t.y), b = make_float2(e.x - t.x, e.y - t.y);
        pk = 2.f * scale * (a.x * a.x + a.y * a.y);
        pm = 2.f * scale * (b.x * b.x + b.y * b.y);
    }
    o[k] = pk;
    if (od) od[k] = to_db(pk);
    if (k != M - k) {
        o[M - k] = pm;
        if (od) od[M - k] = to_db(pm);
    }
}

__global__ void big_zero_tail_kernel(float *__restrict__ out, float *__restrict__ db_out, long long out_pitch,
                                     long long F, long long frames_valid, long long frames_out)
{
    const long long ch = blockIdx.y;
    const long long n = (frames_out - frames_valid) * F;
    float *o = out + ch * out_pitch + frames_valid * F;
    float *od = db_out ? db_out + ch * out_pitch + frames_valid * F : nullptr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        o[i] = 0.f;
        if (od) od[i] = -INFINITY;
    }
}

int run_big(hipdsp_ctx *ctx, const float *x, long long x_pitch, long long channels, long long n_valid,
            long long frames_out, long long out_pitch, int nfft, int hop, float scale, float *out, float *db_out)
{
    const int M = nfft / 2;
    int m = 0;
    while ((1 << m) < M) m++;
    const int N1 = 1 << ((m + 1) / 2), N2 = 1 << (m / 2);
    const long long F = (long long)M + 1;
    if (frames_out > n_valid) {
        hipLaunchKernelGGL(big_zero_tail_kernel, dim3(256, (unsigned)channels), dim3(256), 0, ctx->stream, out, db_out,
                           out_pitch, F, n_valid, frames_out);
        int rc = hd_launch_status("big_zero_tail_kernel");
        if (rc != HIPDSP_OK) return rc;
    }
    const long long items = channels * n_valid;
    if (items == 0) return HIPDSP_OK;
    // scratch: mean | B | Z for one batch of (channel, frame) items, at most ~512 MiB
    long long batch = (512LL << 20) / (16LL * M + 8);
    if (batch < 1) batch = 1;
    if (batch > items) batch = items;
    if (batch > 65535) batch = 65535;
    const size_t off_b = (size_t)((batch * 8 + 255) / 256 * 256);
    const size_t bytes = off_b + 2 * (size_t)batch * (size_t)M * sizeof(float2);
    void *work = nullptr;
    int rc = hipdsp_scratch(ctx, bytes, &work);
    if (rc != HIPDSP_OK) return rc;
    double *mean = (double *)work;
    float2 *B = (float2 *)((char *)work + off_b);
    float2 *Z = B + (size_t)batch * M;
    const size_t lds1 = (2 * (size_t)BIG_W * N1 + N1) * sizeof(float2), lds2 = (2 * (size_t)BIG_W * N2 + N2) * sizeof(float2);
    HD_CHECK_HIP(hipFuncSetAttribute((const void *)big_pass1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    HD_CHECK_HIP(hipFuncSetAttribute((const void *)big_pass2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    for (long long item0 = 0; item0 < items; item0 += batch) {
        const unsigned nb = (unsigned)(items - item0 < batch ? items - item0 : batch);
        hipLaunchKernelGGL(big_mean_kernel, dim3(nb), dim3(256), 0, ctx->stream, x, x_pitch, n_valid, item0, nfft, hop, mean);
        hipLaunchKernelGGL(big_pass1_kernel, dim3((unsigned)(N2 / BIG_W), nb), dim3(256), lds1, ctx->stream, x, x_pitch,
                           n_valid, item0, nfft, hop, N1, N2, mean, B);
        hipLaunchKernelGGL(big_pass2_kernel, dim3((unsigned)(N1 / BIG_W), nb), dim3(256), lds2, ctx->stream, B, M, N1, N2, Z);
        hipLaunchKernelGGL(big_split_kernel, dim3((unsigned)((M / 2 + 1 + 255) / 256), nb), dim3(256), 0, ctx->stream, Z, M,
                           n_valid, frames_out, out_pitch, item0, scale, out, db_out);
        rc = hd_launch_status("big spectrogram kernels");
        if (rc != HIPDSP_OK) return rc;
    }
    return HIPDSP_OK;
}

// Consecutive frames one wave (lane group) walks: long runs amortise the table load
// and keep the 50 % overlap in cache, short runs keep small inputs spread over the chip.
int frames_per_wave(const hipdsp_ctx *ctx, long long channels, long long frames_out, int G)
{
    if (ctx->spec_fpw > 0) return ctx->spec_fpw;
    long long groups = (long long)ctx->n_cus * 32 * G;        // lane groups we want busy
    long long f = channels * frames_out / groups;
    if (f < 4) f = 4;
    if (f > 64) f = 64;
    return (int)f;
}

template <int NFFT, int LPF, int R1, int R2, int R3, int WAVES>
int launch_fast(hipdsp_ctx *ctx, const float *x, long long x_pitch, long long channels, long long n_valid,
                long long frames_out, long long out_pitch, int hop, float scale, const float *tables,
                float *out, float *db_out)
{
    constexpr int G = 64 / LPF;
    const int fpw = frames_per_wave(ctx, channels, frames_out, G);
    long long per_block = (long long)WAVES * fpw * G;
    long long bx = (frames_out + per_block - 1) / per_block;
    // 50 % / 75 % overlap with one frame per wave: reuse the overlapped half / three quarters from registers
    // (measured, tools/spec_reuse_ab.py, 64 ch x 120 s: 1024/256 3.17 -> 3.92 TB/s, 2048/512 3.29 -> 3.66, 1024/512
    // 4.28 -> 4.54, 2048/1024 equal; at nfft 4096 -- one workgroup per CU -- the rotating register sets cost
    // more than the saved requests: 3.46 -> 2.75 TB/s at hop 2048, so that size fetches every frame whole)
    const int reuse = (LPF != 64 || NFFT > 2048 || ctx->spec_no_half) ? 1
                                                                      : (hop * 2 == NFFT ? 2 : (hop * 4 == NFFT ? 4 : 1));
    dim3 grid((unsigned)bx, (unsigned)channels), block(64 * WAVES);
#define HD_SPEC_LAUNCH(DBV, REUSEV)                                                                 \
    hipLaunchKernelGGL((spec_fast_kernel<NFFT, LPF, R1, R2, R3, WAVES, DBV, REUSEV>), grid, block, 0, \
                       ctx->stream, x, x_pitch, n_valid, frames_out, out_pitch, hop, scale, tables,   \
                       out, db_out, fpw)
    if (LPF == 64 && reuse == 2) {
        if (db_out) HD_SPEC_LAUNCH(true, (LPF == 64 ? 2 : 1));
        else HD_SPEC_LAUNCH(false, (LPF == 64 ? 2 : 1));
    } else if (LPF == 64 && reuse == 4) {
        if (db_out) HD_SPEC_LAUNCH(true, (LPF == 64 ? 4 : 1));
        else HD_SPEC_LAUNCH(false, (LPF == 64 ? 4 : 1));
    } else {
        if (db_out) HD_SPEC_LAUNCH(true, 1);
        else HD_SPEC_LAUNCH(false, 1);
    }
#undef HD_SPEC_LAUNCH
    return hd_launch_status("spec_fast_kernel");
}

// tw2 | tw3 | twn | window for one nfft, computed in float64 on the host in the order the
// kernel keeps them in LDS
int fft_tables(hipdsp_ctx *ctx, int nfft, int R1, int R2, int R3, const float **dev)
{
    int lg = 0;
    while ((1 << lg) < nfft) lg++;
    if (!ctx->fft_tables[lg]) {
        const int M = nfft / 2;
        const int TW2 = (R2 - 1) * R1, TW3 = R1 * R2, TWN = M / 2 + 1;
        size_t n = 2 * (size_t)(TW2 + TW3 + TWN + M);
        float *h = new float[n];
        float *p = h;
        for (int t = 1; t < R2; t++)
            for (int k = 0; k < R1; k++) {
                double a = -2.0 * M_PI * (double)(k * t) / (double)(R1 * R2);
                *p++ = (float)cos(a); *p++ = (float)sin(a);
            }
        for (int k = 0; k < R1 * R2; k++) {
            double a = -2.0 * M_PI * (double)k / (double)M;
            *p++ = (float)cos(a); *p++ = (float)sin(a);
        }
        (void)R3;
        for (int k = 0; k <= M / 2; k++) {
            double a = -2.0 * M_PI * (double)k / (double)nfft;
            *p++ = (float)cos(a); *p++ = (float)sin(a);
        }
        for (int i = 0; i < nfft; i++) *p++ = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)nfft));
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (ctx->stream) (void)hipStreamIsCapturing(ctx->stream, &st);
        if (st != hipStreamCaptureStatusNone) {
            delete[] h;
            hipdsp_set_error("first spectrogram call for nfft %d during stream capture; run it once before", nfft);
            return HIPDSP_ERR_INVALID;
        }
        void *d = nullptr;
        hipError_t e = hipMalloc(&d, n * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice);
        delete[] h;
        if (e != hipSuccess) {
            if (d) (void)hipFree(d);
            HD_CHECK_HIP(e);
        }
        ctx->fft_tables[lg] = d;
    }
    *dev = (const float *)ctx->fft_tables[lg];
    return HIPDSP_OK;
}

// tw2 | twn | window for the two-stage kernel
int fft_tables2(hipdsp_ctx *ctx, int nfft, int R1, int R2, const float **dev)
{
    int lg = 0;
    while ((1 << lg) < nfft) lg++;
    if (!ctx->fft_tables2[lg]) {
        const int M = nfft / 2;
        const int TW2 = (R2 - 1) * R1, TWN = M / 2 + 1;
        size_t n = 2 * (size_t)(TW2 + TWN + M);
        float *h = new float[n];
        float *p = h;
        for (int t = 1; t < R2; t++)
            for (int k = 0; k < R1; k++) {
                double a = -2.0 * M_PI * (double)(k * t) / (double)M;
                *p++ = (float)cos(a); *p++ = (float)sin(a);
            }
        for (int k = 0; k <= M / 2; k++) {
            double a = -2.0 * M_PI * (double)k / (double)nfft;
            *p++ = (float)cos(a); *p++ = (float)sin(a);
        }
        for (int i = 0; i < nfft; i++) *p++ = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)nfft));
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (ctx->stream) (void)hipStreamIsCapturing(ctx->stream, &st);
        if (st != hipStreamCaptureStatusNone) {
            delete[] h;
            hipdsp_set_error("first spectrogram call for nfft %d during stream capture; run it once before", nfft);
            return HIPDSP_ERR_INVALID;
        }
        void *d = nullptr;
        hipError_t e = hipMalloc(&d, n * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice);
        delete[] h;
        if (e != hipSuccess) {
            if (d) (void)hipFree(d);
            HD_CHECK_HIP(e);
        }
        ctx->fft_tables2[lg] = d;
    }
    *dev = (const float *)ctx->fft_tables2[lg];
    return HIPDSP_OK;
}

template <int NFFT, int LPF, int R1, int R2, int WAVES>
int run_fast2(hipdsp_ctx *ctx, const float *x, long long x_pitch, long long channels, long long n_valid,
              long long frames_out, long long out_pitch, int hop, float scale, float *out, float *db_out)
{
    const float *tables = nullptr;
    int rc = fft_tables2(ctx, NFFT, R1, R2, &tables);
    if (rc != HIPDSP_OK) return rc;
    constexpr int G = 64 / LPF;
    const int fpw = frames_per_wave(ctx, channels, frames_out, G);
    long long per_block = (long long)WAVES * G * fpw;
    long long bx = (frames_out + per_block - 1) / per_block;
    hipLaunchKernelGGL((spec2_kernel<NFFT, LPF, R1, R2, WAVES>), dim3((unsigned)bx, (unsigned)channels),
                       dim3(64 * WAVES), 0, ctx->stream, x, x_pitch, n_valid, frames_out, out_pitch, hop, scale,
                       tables, out, db_out, fpw);
    return hd_launch_status("spec2_kernel");
}

template <int NFFT, int LPF, int R1, int R2, int R3, int WAVES>
int run_fast(hipdsp_ctx *ctx, const float *x, long long x_pitch, long long channels, long long n_valid,
             long long frames_out, long long out_pitch, int hop, float scale, float *out, float *db_out)
{
    const float *tables = nullptr;
    int rc = fft_tables(ctx, NFFT, R1, R2, R3, &tables);
    if (rc != HIPDSP_OK) return rc;
    return launch_fast<NFFT, LPF, R1, R2, R3, WAVES>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch,
                                                     hop, scale, tables, out, db_out);
}

#include "spec_pack.h"
#include "spec_wgs.h"
#include "spec_chip.h"
#include "spec_chipx.h"

template <int NFFT, int LPF, int R1, int R2, int R3, int OCC>
int run_wg(hipdsp_ctx *ctx, const float *x, long long x_pitch, long long channels, long long n_valid,
           long long frames_out, long long out_pitch, int hop, float scale, float *out, float *db_out)
{
    const float *tables = nullptr;
    int rc = fft_tables(ctx, NFFT, R1, R2, R3, &tables);
    if (rc != HIPDSP_OK) return rc;
    // a few frames per workgroup once there are many more frames than the chip holds workgroups
    long long fpb = frames_out * channels / ((long long)ctx->n_cus * 8);
    if (fpb < 1) fpb = 1;
    if (fpb > 8) fpb = 8;
    const dim3 grid((unsigned)((frames_out + fpb - 1) / fpb), (unsigned)channels);
    if (db_out)
        hipLaunchKernelGGL((spec_wg_kernel<NFFT, LPF, R1, R2, R3, OCC, true>), grid, dim3(LPF), 0, ctx->stream, x, x_pitch,
                           n_valid, frames_out, out_pitch, hop, scale, tables, out, db_out, (int)fpb);
    else
        hipLaunchKernelGGL((spec_wg_kernel<NFFT, LPF, R1, R2, R3, OCC, false>), grid, dim3(LPF), 0, ctx->stream, x, x_pitch,
                           n_valid, frames_out, out_pitch, hop, scale, tables, out, db_out, (int)fpb);
    return hd_launch_status("spec_wg_kernel");
}

}  // namespace

int hd_fft_tables(hipdsp_ctx *ctx, int nfft, const float **dev)
{
    if (nfft == 2048) return fft_tables(ctx, 2048, 16, 16, 4, dev);
    if (nfft == 1024) return fft_tables(ctx, 1024, 8, 8, 8, dev);
    if (nfft == 512) return fft_tables(ctx, 512, 8, 8, 4, dev);
    if (nfft == 256) return fft_tables(ctx, 256, 8, 4, 4, dev);
    hipdsp_set_error("no three-stage table set for nfft %d", nfft);
    return HIPDSP_ERR_UNSUPPORTED;
}

extern "C" int hipdsp_spectrogram(hipdsp_ctx *ctx, const float *x, int64_t x_pitch, int64_t channels,
                                  int64_t frames, int nfft, int hop, double fs, float *out,
                                  float *db_out, int64_t frames_out, int64_t out_pitch)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_REQUIRE(channels >= 0 && frames >= 0 && frames_out >= 0, "negative size");
    HD_REQUIRE(nfft >= 8, "nfft %d < 8", nfft);
    HD_REQUIRE(hop >= 1 && hop <= nfft, "hop %d not in [1, nfft=%d]", hop, nfft);
    HD_REQUIRE(fs > 0, "fs must be positive");
    const bool pow2 = (nfft & (nfft - 1)) == 0;
    if ((pow2 && nfft > (1 << 19)) || (!pow2 && nfft > (1 << 17))) {
        hipdsp_set_error("nfft %d: powers of two up to 524288 and other sizes up to 131072 are implemented",
                         nfft);
        return HIPDSP_ERR_UNSUPPORTED;
    }
    if (channels == 0 || frames_out == 0) return HIPDSP_OK;
    HD_REQUIRE(out != nullptr, "out is NULL");
    if (out_pitch == 0) out_pitch = frames_out * (long long)(nfft / 2 + 1);
    HD_REQUIRE(out_pitch >= frames_out * (long long)(nfft / 2 + 1), "out_pitch smaller than one channel");
    HD_REQUIRE(channels <= 65535, "more than 65535 channels");
    HD_REQUIRE(frames_out <= 0x7fffffffLL, "too many frames");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    // bufferedspectrogram.py:46-49 and scipy _spectral_helper segment count
    long long nsource = (frames_out - 1) * (long long)hop + nfft;
    if (nsource > frames) nsource = frames;
    long long n_valid = 0;
    if (nsource >= nfft) n_valid = (nsource - (nfft - hop)) / hop;
    if (n_valid > frames_out) n_valid = frames_out;
    if (n_valid > 0) HD_REQUIRE(x != nullptr && x_pitch >= frames, "bad input");
    if (n_valid == 0) {
        // a slab shorter than one window (bufferedspectrogram.py:47-49: dest[:] = 0): nothing may read x -- the
        // framed kernels request frame 0 before they look at the frame count (found by tools/fuzz_stress.py as a
        // memory fault when the slab ended at the end of a mapping)
        hipLaunchKernelGGL(big_zero_tail_kernel, dim3(256, (unsigned)channels), dim3(256), 0, ctx->stream, out, db_out,
                           (long long)out_pitch, (long long)(nfft / 2 + 1), 0LL, (long long)frames_out);
        return hd_launch_status("big_zero_tail_kernel");
    }
    double wss = 0.0;
    for (int i = 0; i < nfft; i++) {
        double w = 0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)nfft);
        wss += w * w;
    }
    float scale = (float)(1.0 / (fs * wss));
    if (!pow2) {
        HD_REQUIRE(frames_out <= 65535, "too many frames for the direct DFT path");
        hipLaunchKernelGGL(spec_direct_kernel, dim3((unsigned)((nfft / 2 + 1 + 255) / 256), (unsigned)frames_out,
                                                    (unsigned)channels), dim3(256), 0, ctx->stream, x,
                           (long long)x_pitch, n_valid, (long long)frames_out, (long long)out_pitch, nfft, hop,
                           scale, out, db_out);
        return hd_launch_status("spec_direct_kernel");
    }
    if (!ctx->force_generic_fft && nfft <= 4096) {
        // spec_kernel: 0 = default choice per size, 2 = two-stage kernel, 3 = three-stage kernel
        const int want = ctx->spec_kernel;
        switch (nfft) {
        // short windows: 64 (one frame per lane), 32, 16, 8 or 4 frames side by side in a wave that streams a run of
        // consecutive frames through LDS (spec_pack.h); "spec_kernel" 2 / 3: the kernels they replaced, as cross-checks
        case 8:
            if (want == 2) return run_fast2<8, 1, 2, 2, 4>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
            return run_pack<8, 1, 2, 2, 1>(ctx, x, x_pitch, channels, frames, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
        case 16:
            if (want == 2) return run_fast2<16, 1, 4, 2, 4>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
            return run_pack<16, 1, 4, 2, 1>(ctx, x, x_pitch, channels, frames, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
        case 32:
            if (want == 2) return run_fast2<32, 2, 8, 2, 4>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
            return run_pack<32, 2, 8, 2, 1>(ctx, x, x_pitch, channels, frames, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
        case 64:
            if (want == 2) return run_fast2<64, 4, 8, 4, 4>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
            return run_pack<64, 4, 8, 4, 1>(ctx, x, x_pitch, channels, frames, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
        case 128:
            if (want == 2) return run_fast2<128, 8, 8, 8, 4>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
            return run_pack<128, 8, 8, 8, 1>(ctx, x, x_pitch, channels, frames, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
        case 256:
            if (want == 2) return run_fast2<256, 8, 16, 8, 4>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
            if (want == 3) return run_fast<256, 16, 8, 4, 4, 4>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
            return run_pack<256, 16, 8, 4, 4>(ctx, x, x_pitch, channels, frames, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
        case 512:
            if (want == 2) return run_fast2<512, 16, 16, 16, 4>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
            // the stream through LDS as well (round 5: with the dB image 13 % ahead at 50 % overlap, 27 % at 75 %, 36 % at hop 100;
            // profiles/r05at_spec_kernel_ab.log) -- except for the PSD alone at 50 % overlap, where the kernel it replaces ("spec_kernel"
            // 3) is 5 % ahead
            if (want == 3 || (want == 0 && db_out == nullptr && 2 * hop == nfft))
                return run_fast<512, 32, 8, 8, 4, 4>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
            return run_pack<512, 32, 8, 8, 4>(ctx, x, x_pitch, channels, frames, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
        case 1024:
            if (want == 2) return run_fast2<1024, 16, 32, 16, 4>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
            // with the dB image: the stream through LDS, whose staged 16-byte stores carry both images (separate processes, 64 ch
            // x 120 s: hop 100 12.2 -> 8.7 ms, hop 700 1.74 -> 1.53, hop = nfft 1.32 -> 1.24; in one process level to 3 % ahead at
            // every hop, nfft / 2 and nfft / 4 -- where the other kernel reuses its registers -- included: 1.96 -> 1.90, 3.55 ->
            // 3.50; without the dB image it loses 3-20 %: profiles/r05at_spec_kernel_ab.log); "spec_kernel" 3: for the PSD alone too
            if (want == 3 || (want == 0 && db_out != nullptr))
                return run_pack<1024, 64, 8, 8, 8>(ctx, x, x_pitch, channels, frames, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
            return run_fast<1024, 64, 8, 8, 8, 4>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
        case 2048:
            if (want == 2) return run_fast2<2048, 32, 32, 32, 8>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
            return run_fast<2048, 64, 16, 16, 4, 4>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
        case 4096:
            if (want == 2) return run_wg<4096, 128, 16, 16, 8, 3>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
            // (the streamed workgroup: 3.0 TB/s, level with the one-wave-per-frame kernel, "spec_kernel" 3; 4.1 against 2.9
            // with the dB image next to the PSD)
            if (want != 3) return run_wgs<4096, 128, 16, 16, 8, 3>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
            return run_fast<4096, 64, 16, 16, 8, 4>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
        }
    }
    // long windows: a workgroup streams a run of frames through LDS (spec_wgs.h); "spec_kernel" 2: the kernel it replaced
    const bool old_wg = ctx->spec_kernel == 2;
    if (!ctx->force_generic_fft && nfft == 8192)
        return old_wg ? run_wg<8192, 256, 16, 16, 16, 3>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out)
                      : run_wgs<8192, 256, 16, 16, 16, 3>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
    if (!ctx->force_generic_fft && nfft == 16384)
        return old_wg ? run_wg<16384, 256, 16, 16, 32, 2>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out)
                      : run_wgs<16384, 256, 16, 16, 32, 2>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
    if (!ctx->force_generic_fft && nfft == 32768)
        return old_wg
                   ? run_wg<32768, 512, 16, 32, 32, 2>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out)
                      : run_wgs<32768, 512, 16, 32, 32, 1>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
    // 65536: the frame in the registers of one workgroup (spec_chip.h); "spec_kernel" 2: the four-step path through HBM
    if (!ctx->force_generic_fft && nfft == 65536 && !old_wg)
        return run_chip<false>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
    // 131072: two such workgroups per frame, one for the even and one for the odd bins
    if (!ctx->force_generic_fft && nfft == 131072 && !old_wg)
        return run_chip<true>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
    // 262144, 524288: a radix-4 / radix-8 step in front of the same transform, the residues that pair with each other in one task
    if (!ctx->force_generic_fft && nfft == 262144 && !old_wg)
        return run_chipx<4>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
    if (!ctx->force_generic_fft && nfft == 524288 && !old_wg)
        return run_chipx<8>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, out, db_out);
    if (nfft > 8192)
        return run_big(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, nfft, hop, scale, out, db_out);
    size_t lds = sizeof(float2) * 2 * (size_t)nfft;
    if (lds > 48 * 1024)
        HD_CHECK_HIP(hipFuncSetAttribute((const void *)spec_generic_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(spec_generic_kernel, dim3((unsigned)frames_out, (unsigned)channels), dim3(256), lds,
                       ctx->stream, x, (long long)x_pitch, n_valid, (long long)frames_out, (long long)out_pitch,
                       nfft, hop, scale, out, db_out);
    return hd_launch_status("spec_generic_kernel");
}
