// spec_pack.h -- the short windows of BufferedSpectrogram.process (nfft 8 ... 256: the reference's default is 256 / 128,
// src/audian/bufferedspectrogram.py:14-16; its selector offers 2^3 ... 2^19, src/audian/databrowser.py:516) for any hop;
// since round 5 also nfft 512 (except the PSD alone at 50 % overlap) and nfft 1024 with the dB image (spectrogram.hip's
// dispatch has the measurements): G x nfft <= 1024 samples is all it asks for.
// Included by spectrogram.hip inside its anonymous namespace.
//
// A frame of a short window is a few hundred bytes in and out: kernels that let every lane group fetch ITS frame from
// HBM (8 bytes per lane, every sample requested once per frame it is part of) and store ITS bins (4 bytes per lane in
// runs of LPF lanes) move the right bytes in the wrong shape -- 256 / 128 ran at 4.2 TB/s, 64 / 32 at 2.1, 8 / 4 at 0.7
// (profiles/r04a_spec_sizes_before.log).  Here a wave owns a RUN of consecutive frames of one channel and treats both
// sides as streams:
//   in    the samples of the run go through a per-wave ring in LDS, 4 KB (1024 samples) per fetch as one 16-byte load
//         per lane and instruction, each sample exactly once whatever the overlap; the lane groups of a batch of
//         G = 64 / LPF consecutive frames pick their frames out of the ring (their span is at most 1024 samples);
//   fft   as before: LPF lanes per frame, in-register DFTs, one or two Stockham exchanges through a per-frame LDS buffer;
//   out   the G frames' bins are consecutive in HBM ((frames, F) rows per channel): they are staged in LDS (over the
//         frame buffers, which are dead by then) and leave as 16-byte stores per lane, dB next to them when asked for.
// Frames behind the last valid one are zero (bufferedspectrogram.py:59), staged and stored the same way.
#pragma once

typedef float f4p __attribute__((ext_vector_type(4), aligned(4)));     // float4 that only needs 4-byte alignment

// sum over the LPF (a power of two <= 16) lanes of a lane group, in every lane of the group: DPP adds inside the VALU
// (the first steps of wave_sum) instead of trips through the LDS crossbar
template <int LPF>
__device__ __forceinline__ float group_sum(float v)
{
    static_assert(LPF == 1 || LPF == 2 || LPF == 4 || LPF == 8 || LPF == 16 || LPF == 32 || LPF == 64, "lanes per frame");
    if (LPF == 64) return wave_sum(v);
    if (LPF >= 2) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    if (LPF >= 4) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    if (LPF >= 8) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));   // row_half_mirror
    if (LPF >= 16) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));  // row_mirror
    if (LPF == 32) v += __shfl_xor(v, 16, 64);            // (the other row of the half wave)
    return v;
}

// the value of lane (LPF - l) mod LPF of the same lane group.  Sixteen lanes are one DPP row: mirror (l -> 15 - l), then
// rotate right by one (l -> l - 1): two VALU moves per dword instead of a ds_bpermute
template <int LPF>
__device__ __forceinline__ float2 group_partner(float2 z, int partner)
{
    if (LPF == 1) return z;
    if (LPF == 16) {
        auto one = [](float x) {
            const int m = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xf, 0xf, false);      // row_mirror
            return __int_as_float(__builtin_amdgcn_update_dpp(0, m, 0x121, 0xf, 0xf, false));             // row_ror:1
        };
        return make_float2(one(z.x), one(z.y));
    }
    return make_float2(__shfl(z.x, partner, 64), __shfl(z.y, partner, 64));
}

template <int NFFT, int LPF, int R1, int R2, int R3, bool DB>
__global__ __launch_bounds__(256, NFFT >= 1024 ? 2 : 3) void spec_pack_kernel(
    const float *__restrict__ x, long long x_pitch, long long frames, long long n_valid, long long frames_out,
    long long out_pitch, int hop, float scale, const float *__restrict__ tables, float *__restrict__ out,
    float *__restrict__ db_out, int batches_per_wave, int debug)
{
    // debug (option "spec_debug", measurements only, results wrong): 1 = no global stores, 2 = no fetches, 4 = no transform

    constexpr int M = NFFT / 2, PPL = M / LPF, G = 64 / LPF, F = M + 1, MP = M + M / 16;
    constexpr bool THREE = R3 > 1;
    constexpr bool INLANE = R3 == 4 && PPL == 8 && (R1 * R2) % 32 == 0;    // two last-stage butterflies per lane: see the split step
    constexpr int NS3 = R1 * R2;                           // (= M / 4 there)
    static_assert(R1 * R2 * R3 == M && PPL % R1 == 0 && PPL % R2 == 0 && PPL % R3 == 0, "radices");
    constexpr int TW2 = (R2 - 1) * R1, TW3 = THREE ? R1 * R2 : 0, TWN = M / 2 + 1, NTAB = TW2 + TW3 + TWN + M;
    constexpr int RB = 2048;                               // ring: a batch's span (<= 1024 samples) + one fetch
    constexpr int CHUNK = 1024;
    static_assert((G - 1) * NFFT + NFFT <= CHUNK, "a batch must fit half the ring");
    constexpr int FBW0 = 2 * G * MP > G * F ? 2 * G * MP : G * F;
    constexpr int FBW = (FBW0 + 3) & ~3;
    __shared__ float2 tab[NTAB];
    __shared__ __attribute__((aligned(16))) float ring_all[4][RB + 4];   // [RB] mirrors [0]: a pair never wraps
    __shared__ __attribute__((aligned(16))) float fb_all[4][FBW];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int g = lane / LPF, l = lane % LPF;
    {
        const float2 *src = reinterpret_cast<const float2 *>(tables);
        for (int i = tid; i < NTAB; i += 256) tab[i] = src[i];
    }
    __syncthreads();
    const float2 *tw2 = tab, *tw3 = tab + TW2, *twn = tab + TW2 + TW3, *win = tab + TW2 + TW3 + TWN;
    float *ring = ring_all[wave];
    float *stage = fb_all[wave];
    float2 *fb = reinterpret_cast<float2 *>(fb_all[wave]) + g * MP;

    const long long ch = blockIdx.y;
    const float *xc = x + ch * x_pitch;
    float *oc = out + ch * out_pitch;
    float *dc = DB ? db_out + ch * out_pitch : nullptr;
    const long long f0 = ((long long)blockIdx.x * 4 + wave) * (long long)batches_per_wave * G;
    if (f0 >= frames_out) return;                          // (no workgroup barrier below)
    // samples [ring_end - RB, ring_end) of the channel may sit in the ring; a fetch is the 1024 samples from a multiple of
    // 1024 (one half of the ring, 16-byte aligned in the row; a run's first fetch starts up to 4 KB in front of its
    // first sample -- samples of the same row that the run before this one has just pulled through L2)
    long long ring_end = (f0 * (long long)hop) & ~(long long)(CHUNK - 1);
    const int partner = g * LPF + ((LPF - l) & (LPF - 1));
    // what batch `fbx` needs to see in the ring, and what the whole run will need (nothing is fetched beyond that)
    auto need_of = [&](long long fbx) -> long long {
        long long nvx = n_valid - fbx;
        if (nvx > G) nvx = G;
        return nvx > 0 ? (fbx + nvx - 1) * (long long)hop + NFFT : 0;
    };
    long long run_end = 0;
    {
        long long lastf = f0 + (long long)batches_per_wave * G;
        if (lastf > n_valid) lastf = n_valid;
        if (lastf > f0) run_end = (lastf - 1) * (long long)hop + NFFT;
    }
    // The next fetch is always in flight: it is requested as soon as the one before it has gone into the ring and
    // consumed when the ring runs short -- between a batch's arithmetic and its stores, a batch or two later.
    float4 pend[CHUNK / 256];
    bool pend_ok = false;
    auto request = [&](long long pos) {
        if (pos + CHUNK <= frames) {                           // (wave-uniform) the whole fetch lies inside the trace
#pragma unroll
            for (int k = 0; k < CHUNK / 256; k++) {
                const f4p t = *reinterpret_cast<const f4p *>(xc + pos + 256 * k + 4 * lane);
                pend[k] = make_float4(t.x, t.y, t.z, t.w);
            }
        } else {
#pragma unroll
            for (int k = 0; k < CHUNK / 256; k++) {
                const long long p = pos + 256 * k + 4 * lane;
                pend[k].x = p < frames ? xc[p] : 0.f;
                pend[k].y = p + 1 < frames ? xc[p + 1] : 0.f;
                pend[k].z = p + 2 < frames ? xc[p + 2] : 0.f;
                pend[k].w = p + 3 < frames ? xc[p + 3] : 0.f;
            }
        }
    };
    auto refill = [&](long long need) {
        while (ring_end < need) {
            if (debug & 2) { ring_end += CHUNK; continue; }
            if (!pend_ok) request(ring_end);
            const int rbase = (int)(ring_end & (RB - 1));      // 0 or CHUNK: no wrap inside a fetch
#pragma unroll
            for (int k = 0; k < CHUNK / 256; k++)
                *reinterpret_cast<float4 *>(ring + rbase + 256 * k + 4 * lane) = pend[k];
            if (rbase == 0 && lane == 0) ring[RB] = pend[0].x;
            ring_end += CHUNK;
            pend_ok = ring_end < run_end;
            if (pend_ok) request(ring_end);
        }
    };
    refill(need_of(f0));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    float pvs = 0.f;                                        // pivot of the frame mean, per lane group
    for (int b = 0; b < batches_per_wave; b++) {
        const long long fb0 = f0 + (long long)b * G;       // first frame of the batch (wave-uniform)
        if (fb0 >= frames_out) break;
        const long long left = frames_out - fb0;
        const int nfr = left < G ? (int)left : G;          // frames to write
        long long nv = n_valid - fb0;
        const int nval = nv <= 0 ? 0 : (nv < G ? (int)nv : G);   // of which hold a spectrum
        const bool valid = g < nval;
        // ---- the batch's frames, one per lane group, out of the ring
        float2 v[PPL];
        constexpr bool CORRP = NFFT >= 128;
        float corr = 0.f;
        {
            const int r0 = (int)(((fb0 + (valid ? g : 0)) * (long long)hop) & (RB - 1)) + 2 * l;
            // The frame mean (detrend='constant') relative to a PIVOT: on a trace that is an offset plus something small --
            // raw data of the reference's default session, a filter's decaying transient -- a float32 sum of the samples
            // carries an error of 1e-7 of the OFFSET into bins 0 and 1 of every frame (tools/fuzz_stress.py, seed 10268:
            // 1.0e-4 of the frame's peak); the sum of the differences to a value near the mean carries 1e-7 of the small
            // part.  The pivot is the mean of the frame this lane group transformed in the batch before (`pvs`), not a
            // sample: a frame that starts on a pulse has that sample under a window weight of zero (chain.hip's psd_frame
            // has the case).  The run's first batch takes two steps: the batch's first sample (one LDS word, the same for
            // every lane; zero if it is not finite) as the pivot of a rough mean, that mean as the pivot.
            // (Windows of up to 64 samples take the two steps in every batch: the mean of so few samples follows a single pulse
            // among them -- 1/8 of it at nfft 8 -- and would be a poor pivot for the frame that comes G frames later.)
            if (b == 0 || NFFT <= 64) {
                float p0 = ring[(int)((fb0 * (long long)hop) & (RB - 1))];
                p0 = (fabsf(p0) <= 3.0e38f) ? p0 : 0.f;
                const v2f p02 = {p0, p0};
                v2f a0 = {0.f, 0.f};
#pragma unroll
                for (int u = 0; u < PPL / R1; u++)
#pragma unroll
                    for (int t = 0; t < R1; t++) {
                        const int ri = (r0 + 2 * (LPF * u + t * (M / R1))) & (RB - 1);
                        a0 += (v2f){ring[ri], ring[ri + 1]} - p02;
                    }
                const float c = p0 + group_sum<LPF>(a0.x + a0.y) * (1.0f / (float)NFFT);
                pvs = (fabsf(c) <= 3.0e38f) ? c : p0;
            }
            const v2f pivot2 = {pvs, pvs};
            v2f acc = {0.f, 0.f};
#pragma unroll
            for (int u = 0; u < PPL / R1; u++)
#pragma unroll
                for (int t = 0; t < R1; t++) {
                    const int ri = (r0 + 2 * (LPF * u + t * (M / R1))) & (RB - 1);
                    const v2f e = (v2f){ring[ri], ring[ri + 1]} - pivot2;
                    v[u * R1 + t] = as_f2(e);
                    acc += e;
                }
            const float mean = group_sum<LPF>(acc.x + acc.y) * (1.0f / (float)NFFT);
            const v2f mean2 = {mean, mean};
            {
                const float c = pvs + mean;                          // the next batch's pivot (a NaN or Inf in this frame: unchanged)
                pvs = (fabsf(c) <= 3.0e38f) ? c : pvs;
            }
            // CORRP (nfft 128, 256): what the subtraction leaves.  The pivot here is G frames old; after a step in the level
            // the differences of up to G flat frames are all the size of the step, their mean is good to 6e-8 of THAT, and
            // the Hann window puts the error times nfft / 2 into bins 0 and 1 (2e-4 of the frame's peak behind 275 sigma at
            // nfft 256).  The detrended samples are summed once more and their mean taken out of those two bins at the
            // split step (m1 nfft / 2 and -m1 nfft / 4 under the periodic Hann window), as in spec_wgs.h.
            v2f rest = {0.f, 0.f};
#pragma unroll
            for (int u = 0; u < PPL / R1; u++)
#pragma unroll
                for (int t = 0; t < R1; t++) {
                    float2 &e = v[u * R1 + t];
                    const v2f q = as_v2f(e) - mean2;
                    if (CORRP) rest += q;
                    e = as_f2(q * as_v2f(win[l + LPF * u + t * (M / R1)]));
                }
            if (CORRP) corr = 0.5f * group_sum<LPF>(rest.x + rest.y);
        }
        constexpr int RL = THREE ? R3 : R2, NBL = PPL / RL;          // the last stage's radix
        float *sg = stage + g * F;
        const bool full = nval == G;                                 // (wave-uniform: the masks only in a run's last batch)
        const v2f hscale2 = {0.5f * scale, 0.5f * scale};
        // one pair of bins: X[k] = E + W^k O, X[M-k] = conj(E - W^k O) from Z[k] and Z[M-k]; the halves go into the scale
        auto pair_psd = [&](float2 zk, float2 zm, int k, float &pk, float &pm, float re_add = 0.f) {
            const v2f e = pk_add_conj(as_v2f(zk), as_v2f(zm));
            const v2f t = pk_cmul_negi(pk_sub_conj(as_v2f(zk), as_v2f(zm)), as_v2f(twn[k]));
            v2f re = pk_sumdiff_x(e, t);
            const v2f im = pk_sumdiff_y(e, t);
            re.x += re_add;                                          // (bin 1 of lane 1: `re` is twice its real part)
            const v2f pw = (re * re + im * im) * hscale2;
            pk = pw.x; pm = pw.y;
        };
        if (debug & 4) {
#pragma unroll
            for (int m = 0; m < PPL; m++) { sg[l + LPF * m] = v[m].x; }
            if (l == 0) sg[M] = v[0].y;
        } else if constexpr (INLANE) {
            // nfft 256 (M = 128 = 8 x 4 x 4, sixteen lanes per frame; 512 = 8 x 8 x 4 with 32 lanes alike): the last stage's
            // butterfly j (j < NS3 = M / 4) produces the bins j + NS3 t, and the partner of bin k is bin M - k: butterfly j
            // pairs with butterfly NS3 - j.  Which butterflies a lane takes is only a matter of the LDS addresses it
            // loads from, so lane l takes j = l AND j = NS3 - l (lane 0: the two self-paired ones, 0 and NS3 / 2) and
            // finds every partner in its own registers -- no exchange between lanes in the split step at all.
            stockham_stage<R1, 1, M, LPF, false, true>(v, fb, tw2, l);
            stockham_stage<R2, R1, M, LPF, true, true>(v, fb, tw2, l);
            const int jb = (l == 0) ? NS3 / 2 : NS3 - l;
            float2 A[4], B[4];
            {
                const int pa = pad16(l), pb = pad16(jb);
#pragma unroll
                for (int t = 0; t < 4; t++) {                        // pad16(j + NS3 t) = pad16(j) + NS3 t + NS3 t / 16
                    A[t] = fb[pa + NS3 * t + NS3 * t / 16];
                    B[t] = fb[pb + NS3 * t + NS3 * t / 16];
                }
                const v2f wa = as_v2f(tw3[l]), wb = as_v2f(tw3[jb]);
                v2f qa = wa, qb = wb;
#pragma unroll
                for (int t = 1; t < 4; t++) {
                    A[t] = as_f2(pk_cmul(as_v2f(A[t]), qa));
                    B[t] = as_f2(pk_cmul(as_v2f(B[t]), qb));
                    if (t < 3) { qa = pk_cmul(qa, wa); qb = pk_cmul(qb, wb); }
                }
                dft<4>(A);
                dft<4>(B);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // pairs (low bin k < M / 2, its partner M - k): lanes >= 1: (A0, B3) k = l; (A1, B2) k = NS3 + l; (B1, A2)
            // k = 2 NS3 - l; (B0, A3) k = NS3 - l.  Lane 0: A0 = DC and Nyquist; (A1, A3) k = NS3; (B1, B2) k = 3 NS3 / 2;
            // (B0, B3) k = NS3 / 2; A2 = bin M / 2, which pairs with itself.
            const bool l0 = l == 0;
            float pk[4], pm[4];
            int kk[4];
            kk[0] = l; kk[1] = NS3 + l; kk[2] = l0 ? 3 * NS3 / 2 : 2 * NS3 - l; kk[3] = l0 ? NS3 / 2 : NS3 - l;
            pair_psd(A[0], B[3], kk[0], pk[0], pm[0], (CORRP && l == 1) ? corr : 0.f);
            pair_psd(A[1], l0 ? A[3] : B[2], kk[1], pk[1], pm[1]);
            pair_psd(B[1], l0 ? B[2] : A[2], kk[2], pk[2], pm[2]);
            pair_psd(B[0], l0 ? B[3] : A[3], kk[3], pk[3], pm[3]);
            {
                const float dc0 = A[0].x + A[0].y - corr, ny = A[0].x - A[0].y;     // DC and Nyquist, not doubled
                pk[0] = l0 ? dc0 * dc0 * scale : pk[0];
                pm[0] = l0 ? ny * ny * scale : pm[0];
            }
            float ph = 2.f * scale * (A[2].x * A[2].x + A[2].y * A[2].y);
            if (!full) {
#pragma unroll
                for (int t = 0; t < 4; t++) { pk[t] = valid ? pk[t] : 0.f; pm[t] = valid ? pm[t] : 0.f; }
                ph = valid ? ph : 0.f;
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                sg[kk[t]] = pk[t];
                sg[M - kk[t]] = pm[t];
            }
            if (l0) sg[M / 2] = ph;
        } else {
        if constexpr (THREE) {
            stockham_stage<R1, 1, M, LPF, false, true>(v, fb, tw2, l);
            stockham_stage<R2, R1, M, LPF, true, true>(v, fb, tw2, l);
            stockham_stage<R3, R1 * R2, M, LPF, true, false, true>(v, fb, tw3, l);
        } else {
            stockham_stage<R1, 1, M, LPF, false, true>(v, fb, tw2, l);
            stockham_stage<R2, R1, M, LPF, true, false>(v, fb, tw2, l);
        }
        // v[(m % NBL) * RL + m / NBL] = Z[k], k = l + LPF m.  Split step for m < PPL / 2 (k < M / 2); the partner bin
        // M - k sits in lane LPF - l at m' = PPL - 1 - m (lane 0: in itself at m' = PPL - m).  The bins go into the
        // staging area (the frame buffers are dead: every lane's last LDS load of the transform has been issued, and
        // LDS operations of a wave execute in order).  Same packed arithmetic as spec_fast_kernel.
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float pk_last = 0.f;
#pragma unroll
        for (int m = 0; m < PPL / 2; m++) {
            const int k = l + LPF * m;
            const float2 zk = v[(m % NBL) * RL + m / NBL];
            const int mp = PPL - 1 - m;
            const float2 zsrc = v[(mp % NBL) * RL + mp / NBL];
            float2 zm = group_partner<LPF>(zsrc, partner);
            if (m > 0) {
                const int m0 = PPL - m;                              // lane 0 pairs inside itself
                const float2 z0 = v[(m0 % NBL) * RL + m0 / NBL];
                zm = (l == 0) ? z0 : zm;
            }
            float pk, pm;
            pair_psd(zk, zm, k, pk, pm, (CORRP && m == 0 && l == 1) ? corr : 0.f);
            if (m == 0) {
                const float dc0 = zk.x + zk.y - corr, ny = zk.x - zk.y;     // DC and Nyquist, not doubled
                pk = (l == 0) ? dc0 * dc0 * scale : pk;
                pm = (l == 0) ? ny * ny * scale : pm;
            }
            if (!full) { pk = valid ? pk : 0.f; pm = valid ? pm : 0.f; }
            sg[k] = pk;
            sg[M - k] = pm;
            pk_last = pk;
        }
        {   // bin M / 2 pairs with itself (lane 0); the other lanes repeat their last store
            constexpr int mh = PPL / 2;
            const float2 z = v[(mh % NBL) * RL + mh / NBL];
            float ph = 2.f * scale * (z.x * z.x + z.y * z.y);
            if (!full) ph = valid ? ph : 0.f;
            sg[(l == 0) ? M / 2 : l + LPF * (PPL / 2 - 1)] = (l == 0) ? ph : pk_last;
        }
        }
        // the ring for the NEXT batch (this batch has read its frames): the fetch consumed here was requested a batch or
        // two ago, and this batch's stores are not yet in the queue behind it
        if (b + 1 < batches_per_wave) refill(need_of(fb0 + G));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- nfr * F consecutive floats of the channel's spectrogram
        {
            float *o = oc + ((debug & 1) ? 0 : fb0 * (long long)F);    // (1: every batch lands on the channel's first frames)
            float *od = DB ? dc + ((debug & 1) ? 0 : fb0 * (long long)F) : nullptr;
            auto put4 = [&](int i) {
                const float4 p = *reinterpret_cast<const float4 *>(stage + i);
                f4p t; t.x = p.x; t.y = p.y; t.z = p.z; t.w = p.w;
                *reinterpret_cast<f4p *>(o + i) = t;
                if (DB) {
                    f4p d; d.x = to_db(p.x); d.y = to_db(p.y); d.z = to_db(p.z); d.w = to_db(p.w);
                    *reinterpret_cast<f4p *>(od + i) = d;
                }
            };
            if (nfr == G) {
                constexpr int TOTAL = G * F, NV = TOTAL / 4;         // whole float4s, then up to three floats
#pragma unroll
                for (int k = 0; k < (NV + 63) / 64; k++) {
                    const int i = 4 * (lane + 64 * k);
                    if (k < NV / 64 || lane + 64 * k < NV) put4(i);
                }
                if (lane < TOTAL - 4 * NV) {
                    const float e = stage[4 * NV + lane];
                    o[4 * NV + lane] = e;
                    if (DB) od[4 * NV + lane] = to_db(e);
                }
            } else {
                const int total = nfr * F;
                for (int i = 4 * lane; i < total; i += 256) {
                    if (i + 4 <= total) {
                        put4(i);
                    } else {
                        for (int q = 0; i + q < total; q++) {
                            o[i + q] = stage[i + q];
                            if (DB) od[i + q] = to_db(stage[i + q]);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// batches of G frames one wave walks: long runs amortise the table load and the first fetch, short ones keep small
// inputs spread over the chip (12 waves per CU resident)
template <int NFFT, int LPF, int R1, int R2, int R3>
int run_pack(hipdsp_ctx *ctx, const float *x, long long x_pitch, long long channels, long long frames, long long n_valid,
             long long frames_out, long long out_pitch, int hop, float scale, float *out, float *db_out)
{
    const float *tables = nullptr;
    int rc = R3 > 1 ? fft_tables(ctx, NFFT, R1, R2, R3, &tables) : fft_tables2(ctx, NFFT, R1, R2, &tables);
    if (rc != HIPDSP_OK) return rc;
    constexpr int G = 64 / LPF;
    const long long batches = (frames_out + G - 1) / G;
    long long bpw = ctx->spec_fpw > 0 ? ctx->spec_fpw : channels * batches / ((long long)ctx->n_cus * 48);
    if (bpw < 1) bpw = 1;
    if (bpw > 64) bpw = 64;
    const long long per_block = 4 * bpw;
    const long long bx = (batches + per_block - 1) / per_block;
    HD_REQUIRE(bx <= 0x7fffffffLL, "grid too large");
    const dim3 grid((unsigned)bx, (unsigned)channels), block(256);
    if (db_out)
        hipLaunchKernelGGL((spec_pack_kernel<NFFT, LPF, R1, R2, R3, true>), grid, block, 0, ctx->stream, x, x_pitch, frames,
                           n_valid, frames_out, out_pitch, hop, scale, tables, out, db_out, (int)bpw, ctx->spec_debug);
    else
        hipLaunchKernelGGL((spec_pack_kernel<NFFT, LPF, R1, R2, R3, false>), grid, block, 0, ctx->stream, x, x_pitch, frames,
                           n_valid, frames_out, out_pitch, hop, scale, tables, out, db_out, (int)bpw, ctx->spec_debug);
    return hd_launch_status("spec_pack_kernel");
}
