// spec_chip.h -- nfft 65536 and 131072 of BufferedSpectrogram.process (the reference's selector offers 2^3 ... 2^19,
// src/audian/databrowser.py:516) with the whole frame ON CHIP.  Included by spectrogram.hip inside its anonymous namespace.
//
// A frame of 65536 samples is 32768 complex points of the half-length transform: 256 KB, more than the 160 KB of LDS, which
// is why round 1 ran it as Bailey's four-step algorithm through a scratch in HBM (52 bytes of traffic per complex point
// against 8 algorithmic: 0.5 TB/s, profiles/r04k_spec_pmc_traffic.txt).  But the REGISTERS of one 512-thread workgroup hold
// it: 64 points per thread, M = 32 x 32 x 32, two radix-32 butterflies per thread and stage, and each of the two exchanges
// between stages goes through LDS in two halves of 16384 points (128 KB + padding) -- conveniently a thread's first
// butterfly always writes into the lower half of the index space and its second into the upper one (j < 512 or not, at both
// exchanges), so an exchange is: write half A, barrier, read A, barrier, write half B, barrier, read B, barrier.  No third
// exchange for the split step: the last stage's butterfly j produces the bins j + 1024 t, the partner of bin k is bin
// 32768 - k, so butterfly j pairs with butterfly 1024 - j, and which butterflies a thread takes at the last stage is only a
// matter of the LDS addresses it reads: thread l takes j = l and 1024 - l (thread 0: the two self-paired ones, 0 and 512).
// Window, stage twiddles and split twiddles are computed on the fly (v_cos_f32 / v_sin_f32 on exact fractions of a turn
// for one base angle each, the rest by angle addition with compile-time constants or by powers), the frame mean relative to
// its first sample and then to that mean (two steps: the sample may be a pulse).  One frame per workgroup, the workgroups
// numbered so that an XCD works on consecutive frames: HBM sees each sample once (the half a frame shares with its
// neighbour comes out of the XCD's L2) and each bin once.
#pragma once

// X2: nfft 131072 as TWO such workgroups per frame.  One radix-2 step of decimation in frequency in front of the transform:
// with z1, z2 the windowed halves of the frame (as complex points), the even bins of the half-length spectrum are the
// 32768-point transform of z1 + z2 and the odd ones that of (z1 - z2) W_65536^n -- workgroup `res` = 0 / 1 forms its input
// from both halves on the way in (the two of a frame are neighbours on an XCD and share the frame in L2) and owns the bins
// k = 2 k' + res.  The split step's partner of bin k is 65536 - k: even with even (sub-index k' with 32768 - k', as at
// 65536), odd with odd (k' with 32767 - k': butterfly j with 1023 - j), so both workgroups finish on their own.  The window
// of the second half is one minus the window of the first (Hann, half a period on), its even sample's angle is the angle
// of W^n.  The frame mean never touches the samples here: its pivot is the mean of 2048 samples spread over the frame, and
// what the sum of the differences says is left comes out of bins 0 and 1 at the split step (m nfft / 2 and -m nfft / 4
// under the Hann window, nothing elsewhere).
template <bool DB, bool X2>
__global__ __launch_bounds__(512, 2) void spec_chip_kernel(
    const float *__restrict__ x, long long x_pitch, long long n_valid, long long frames_out, long long out_pitch, int hop,
    float scale, float *__restrict__ out, float *__restrict__ db_out, int frames_per_block, long long runs_per_channel,
    long long total_runs)
{
    constexpr int NFFT = X2 ? 131072 : 65536, M = 32768, MF = NFFT / 2, F = MF + 1, SB = X2 ? 2 : 1;   // M: the transform; MF + 1 bins
    constexpr int LPF = 512, R = 32, Q = M / R, H = M / 2;
    __shared__ float2 xb[H + H / 32];
    __shared__ float red[2][LPF / 64];
    const int l = threadIdx.x, lane = l & 63, wave = l >> 6;
    // exchange buffer index of element e: e + e / 32 (the writes of one instruction are 32 elements apart).  The constant
    // parts are multiples of 32 everywhere, so each access is a per-lane base plus a compile-time offset.
    auto pidx = [](int e) { return e + (e >> 5); };
    const unsigned loff = 8u * (unsigned)l;
    const int wr1 = 33 * l;                                  // pidx(32 l + t) = 33 l + t
    const int rd1 = pidx(l);                                 // pidx(l + 512 b + 1024 t) = pidx(l) + 528 b + 1056 t
    typedef float f2q __attribute__((ext_vector_type(2), aligned(4)));
    // Workgroups are dealt to the eight XCDs round-robin by their linear id; each XCD has its own L2.  The runs of frames
    // are numbered so that an XCD works on CONSECUTIVE runs at any time: the half a frame shares with its neighbour is
    // then fetched by two CUs of one XCD within microseconds of each other -- one trip to HBM, not two.
    const long long per_xcd = (total_runs + 7) / 8;
    const long long wx = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((long long)(blockIdx.x >> 3) >= per_xcd || wx >= total_runs) return;
    const int res = X2 ? (int)(wx & 1) : 0;                  // which bins of the frame: k = SB k' + res
    const long long w = X2 ? wx >> 1 : wx;
    const long long ch = w / runs_per_channel;
    const float *xc = x + ch * x_pitch;
    float *oc = out + ch * out_pitch;
    float *dc = DB ? db_out + ch * out_pitch : nullptr;
    const long long fbeg = (w % runs_per_channel) * frames_per_block;
    long long nh = frames_out - fbeg;
    if (nh > frames_per_block) nh = frames_per_block;
    long long nvl = n_valid - fbeg;
    const int nv = nvl <= 0 ? 0 : (nvl < nh ? (int)nvl : (int)nh);

    // frame-invariant twiddle bases of this thread (exact fractions of a turn)
    const int k2 = l & 31;                                   // stage 2: W_1024^(k2 t), the same k2 for both butterflies
    const int j3a = l, j3b = res ? 1023 - l : ((l == 0) ? 512 : 1024 - l);      // stage 3 butterflies: a pair of partners
    auto cis = [](float turn) { return make_float2(__builtin_amdgcn_cosf(turn), -__builtin_amdgcn_sinf(turn)); };   // exp(-2 pi i turn)
    float2 w2 = cis((float)k2 * (1.0f / 1024.0f));
    float2 w3a = cis((float)j3a * (1.0f / 32768.0f)), w3b = cis((float)j3b * (1.0f / 32768.0f));
    float2 sa = cis((float)(SB * j3a + res) * (1.0f / (float)NFFT)), sb = cis((float)(SB * j3b + res) * (1.0f / (float)NFFT));   // split twiddles of the first bins

    // (one frame; RES, which bins of it, as a compile-time constant: a run-time `res` inside the unrolled loops left a branch
    // per iteration and per-lane 64-bit addresses behind)
    auto frame_body = [&](int it, auto res_c) {
        constexpr int RES = decltype(res_c)::value;
        const long long frame = fbeg + it;
        const float *seg = xc + frame * (long long)hop;
        float pivot = seg[0];                                // (the first of the two steps of the frame mean, below)
        pivot = (fabsf(pivot) <= 3.0e38f) ? pivot : 0.f;
        int zero = 0;                                        // (keeps the frame-invariant window and twiddle powers out of the
        asm volatile("" : "+v"(zero));                       // loop's preheader: hoisted, they would fill the register file)
        asm volatile("" : "+v"(w2.x), "+v"(w2.y), "+v"(w3a.x), "+v"(w3a.y), "+v"(w3b.x), "+v"(w3b.y));
        asm volatile("" : "+v"(sa.x), "+v"(sa.y), "+v"(sb.x), "+v"(sb.y));
        // ---- stage 1 inputs: z[n] = (x[2n], x[2n+1]), n = j + 1024 t, butterflies j = l and l + 512
        float2 v0[R], v1[R];
        if constexpr (X2) {
            {
                // the pivot: the mean of 2048 samples spread over the frame, none of them on its borders (a pulse among them
                // moves it by 1/2048 of itself; what the pivot misses of the mean is corrected below, to the accuracy of a
                // float32 sum of 131072 differences -- the closer, the better)
                float p = (seg[64 * l + 32] + seg[64 * (l + 512) + 32]) + (seg[64 * (l + 1024) + 32] + seg[64 * (l + 1536) + 32]);
                p = wave_sum(p);
                if (lane == 0) red[0][wave] = p;
                __syncthreads();
                float tot = 0.f;
#pragma unroll
                for (int w8 = 0; w8 < LPF / 64; w8++) tot += red[0][w8];
                tot *= 1.0f / (float)(4 * LPF);
                pivot = (fabsf(tot) <= 3.0e38f) ? tot : pivot;
            }
            float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};    // (eight partial sums: smaller partial sums, smaller roundings)
            auto stage_in = [&](auto bc, float2 *v) {
                constexpr int b = decltype(bc)::value, TB = 16;
                // n = l + 512 b + 1024 t: the even sample's window angle 2 n / nfft IS the angle of W_65536^n; the odd
                // sample's is 1 / nfft of a turn further on; + t / 64 of a turn per t
                const float a0 = (float)(l + 512 * b + zero) * (1.0f / 65536.0f);
                const float a1 = (float)(2 * (l + 512 * b) + 1 + zero) * (1.0f / 131072.0f);
                const float c0 = __builtin_amdgcn_cosf(a0), s0 = __builtin_amdgcn_sinf(a0);
                const float c1 = __builtin_amdgcn_cosf(a1), s1 = __builtin_amdgcn_sinf(a1);
                // (loads as SGPR base + 32-bit lane offset in inline asm: from C++ hipcc adds the lane offset to the frame's
                // address FIRST -- the subexpression all 128 loads share -- and then pays a 64-bit VALU addition and a register
                // pair per load.  The wait for them is by hand, the empty asm statements keep their uses behind it.)
#pragma unroll
              for (int th = 0; th < R; th += TB) {             // (sixteen inputs at a time: 64 registers of loads in flight)
                v2f pr[TB], qr[TB];
#pragma unroll
                for (int t = th; t < th + TB; t++) {
                    const float *b1 = seg + 2 * Q * t + 1024 * b, *b2 = seg + NFFT / 2 + 2 * Q * t + 1024 * b;
                    // (s_nop 4: a base that came out of an SGPR spill slot by v_readlane_b32 just before needs five wait states
                    // in front of a VMEM instruction that reads it, and hipcc cannot see into the asm -- spec_chipx.h ran into it)
                    asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" : "=v"(pr[t - th]) : "v"(loff), "s"(b1) : "memory");
                    asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" : "=v"(qr[t - th]) : "v"(loff), "s"(b2) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int t = th; t < th + TB; t++) {
                    asm volatile("" : "+v"(pr[t - th]), "+v"(qr[t - th]));
                    const v2f p = pr[t - th], q = qr[t - th];
                    const float d1x = p.x - pivot, d1y = p.y - pivot, d2x = q.x - pivot, d2y = q.y - pivot;
                    acc8[t % 8] += (d1x + d1y) + (d2x + d2y);
                    const float ct = wgs_cos64(t), st = wgs_sin64(t);
                    const float cx = c0 * ct - s0 * st, sx = s0 * ct + c0 * st, cy = c1 * ct - s1 * st;
                    const float wx1 = 0.5f - 0.5f * cx, wy1 = 0.5f - 0.5f * cy;       // the first half's window; the second half's is 1 - it
                    const float e1x = wx1 * d1x, e1y = wy1 * d1y, e2x = (1.0f - wx1) * d2x, e2y = (1.0f - wy1) * d2y;
                    if constexpr (RES == 0) {
                        v[t] = make_float2(e1x + e2x, e1y + e2y);
                    } else {
                        const float dx = e1x - e2x, dy = e1y - e2y;                  // times exp(-2 pi i n / 65536) = cx - i sx
                        v[t] = make_float2(dx * cx + dy * sx, dy * cx - dx * sx);
                    }
                    asm volatile("" : "+v"(v[t].x), "+v"(v[t].y));       // (here and now: left alone hipcc parks the differences in scratch
                                                                         // memory and forms the inputs when the first butterfly asks for them)
                }
                __builtin_amdgcn_sched_barrier(0);
              }
            };
            stage_in(std::integral_constant<int, 0>(), v0);
            __builtin_amdgcn_sched_barrier(0);
            stage_in(std::integral_constant<int, 1>(), v1);
            float acc = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
            acc = wave_sum(acc);
            if (lane == 0) red[1][wave] = acc;               // (read at the split step, eight barriers on)
        } else {
            float s = 0.f;
    #pragma unroll
            for (int t = 0; t < R; t++) {
                // (a uniform base per load plus one 32-bit lane offset: 64 lane addresses of 64 bits would be half the registers)
                const f2q a = *reinterpret_cast<const f2q *>(reinterpret_cast<const char *>(seg + 2 * Q * t) + loff);
                const f2q b = *reinterpret_cast<const f2q *>(reinterpret_cast<const char *>(seg + 2 * Q * t + 1024) + loff);
                v0[t] = make_float2(a.x - pivot, a.y - pivot);
                v1[t] = make_float2(b.x - pivot, b.y - pivot);
                s += (v0[t].x + v0[t].y) + (v1[t].x + v1[t].y);
            }
            s = wave_sum(s);
            if (lane == 0) red[0][wave] = s;
            __syncthreads();
            float total = 0.f;
    #pragma unroll
            for (int w = 0; w < LPF / 64; w++) total += red[0][w];
            // The frame mean in TWO steps: the mean of the differences to the frame's first sample is good to 6e-8 of ITS size,
            // and that sample may be a pulse a thousand times the rest of the frame under a window weight of zero (chain.hip's
            // psd_frame has the case; every workgroup starts a new frame here, there is no frame before it to take a pivot
            // from): subtract it, take the mean of what is left -- small whatever the sample was -- and subtract that too.
            const float mean0 = total * (1.0f / (float)NFFT);
            float s1 = 0.f;
    #pragma unroll
            for (int t = 0; t < R; t++) {
                v0[t].x -= mean0; v0[t].y -= mean0; v1[t].x -= mean0; v1[t].y -= mean0;
                s1 += (v0[t].x + v0[t].y) + (v1[t].x + v1[t].y);
            }
            s1 = wave_sum(s1);
            if (lane == 0) red[1][wave] = s1;
            __syncthreads();
            float total1 = 0.f;
    #pragma unroll
            for (int w = 0; w < LPF / 64; w++) total1 += red[1][w];
            const float mean = total1 * (1.0f / (float)NFFT);
            {
                // periodic Hann 0.5 - 0.5 cos(2 pi i / nfft) at i = 2n, 2n + 1: n = l + 1024 t is t / 32 of a turn further on,
                // the second butterfly 1 / 64 of a turn
                const float a0 = (float)(2 * l + zero) * (1.0f / (float)NFFT), a1 = (float)(2 * l + 1 + zero) * (1.0f / (float)NFFT);
                const float c0 = __builtin_amdgcn_cosf(a0), s0 = __builtin_amdgcn_sinf(a0);
                const float c1 = __builtin_amdgcn_cosf(a1), s1 = __builtin_amdgcn_sinf(a1);
    #pragma unroll
                for (int t = 0; t < R; t++) {
                    const float ca = wgs_cos64(2 * t), sn = wgs_sin64(2 * t);
                    const float cb = wgs_cos64(2 * t + 1), sb2 = wgs_sin64(2 * t + 1);
                    v0[t].x = (v0[t].x - mean) * (0.5f - 0.5f * (c0 * ca - s0 * sn));
                    v0[t].y = (v0[t].y - mean) * (0.5f - 0.5f * (c1 * ca - s1 * sn));
                    v1[t].x = (v1[t].x - mean) * (0.5f - 0.5f * (c0 * cb - s0 * sb2));
                    v1[t].y = (v1[t].y - mean) * (0.5f - 0.5f * (c1 * cb - s1 * sb2));
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);       // (one butterfly at a time: two interleaved ones do not fit the registers)
        dft<32>(v0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- exchange 1 -> 2: butterfly j's outputs are elements 32 j + t'; stage 2's butterfly j2 reads j2 + 1024 t.
        // (The first butterfly's half is on its way into LDS while the VALU transforms the second one.)
#pragma unroll
        for (int t = 0; t < R; t++) xb[wr1 + t] = v0[t];
        __builtin_amdgcn_sched_barrier(0);
        dft<32>(v1);
        __builtin_amdgcn_sched_barrier(0);
        float2 u0[R], u1[R];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < R / 2; t++) {
            u0[t] = xb[rd1 + 1056 * t];
            u1[t] = xb[rd1 + 528 + 1056 * t];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < R; t++) xb[wr1 + t] = v1[t];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < R / 2; t++) {
            u0[R / 2 + t] = xb[rd1 + 1056 * t];
            u1[R / 2 + t] = xb[rd1 + 528 + 1056 * t];
        }
        __syncthreads();
        // ---- stage 2: W_1024^(k2 t) by powers (the same for both butterflies)
        {
            v2f w = as_v2f(w2);
#pragma unroll
            for (int t = 1; t < R; t++) {
                u0[t] = as_f2(pk_cmul(as_v2f(u0[t]), w));
                u1[t] = as_f2(pk_cmul(as_v2f(u1[t]), w));
                if (t + 1 < R) w = pk_cmul(w, as_v2f(w2));
            }
        }
        __builtin_amdgcn_sched_barrier(0);       // (one butterfly at a time: two interleaved ones do not fit the registers)
        dft<32>(u0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- exchange 2 -> 3: butterfly j2's output t2 is element (j2 / 32) 1024 + j2 % 32 + 32 t2
        const int wr2 = (l >> 5) * (Q + 32) + (l & 31);      // pidx((l / 32) 1024 + l % 32 + 32 t) = this + 33 t
#pragma unroll
        for (int t = 0; t < R; t++) xb[wr2 + 33 * t] = u0[t];
        __builtin_amdgcn_sched_barrier(0);
        dft<32>(u1);
        __builtin_amdgcn_sched_barrier(0);
        float2 ya[R], yb[R];
        const int ra = pidx(j3a), rb = pidx(j3b);            // pidx(j3 + 1024 t) = pidx(j3) + 1056 t
        __syncthreads();
#pragma unroll
        for (int t = 0; t < R / 2; t++) {
            ya[t] = xb[ra + 1056 * t];
            yb[t] = xb[rb + 1056 * t];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < R; t++) xb[wr2 + 33 * t] = u1[t];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < R / 2; t++) {
            ya[R / 2 + t] = xb[ra + 1056 * t];
            yb[R / 2 + t] = xb[rb + 1056 * t];
        }
        __syncthreads();
        // ---- stage 3: W_32768^(j3 t) by powers
        {
            v2f wa = as_v2f(w3a), wb = as_v2f(w3b);
#pragma unroll
            for (int t = 1; t < R; t++) {
                ya[t] = as_f2(pk_cmul(as_v2f(ya[t]), wa));
                yb[t] = as_f2(pk_cmul(as_v2f(yb[t]), wb));
                if (t + 1 < R) { wa = pk_cmul(wa, as_v2f(w3a)); wb = pk_cmul(wb, as_v2f(w3b)); }
            }
        }
        __builtin_amdgcn_sched_barrier(0);       // (one butterfly at a time: two interleaved ones do not fit the registers)
        dft<32>(ya);
        __builtin_amdgcn_sched_barrier(0);
        dft<32>(yb);
        __builtin_amdgcn_sched_barrier(0);
        // ---- split step and PSD: ya[t] = Z[j3a + 1024 t], yb[t] = Z[j3b + 1024 t].  Pairs (low bin k <= M / 2, partner
        // M - k), t' < 16: (ya[t'], yb[31 - t']) k = l + 1024 t' and (yb[t'], ya[31 - t']) k = 1024 - l + 1024 t'; thread 0:
        // ya[0] = DC and Nyquist, (ya[t'], ya[32 - t']) k = 1024 t', (yb[t'], yb[31 - t']) k = 512 + 1024 t', ya[16] = bin
        // M / 2, which pairs with itself.  exp(-2 pi i k / nfft): the first bin's angle plus t' / 64 of a turn.
        float *o = oc + frame * (long long)F;
        float *od = DB ? dc + frame * (long long)F : nullptr;
        const bool l0 = l == 0;
        const v2f hscale2 = {0.5f * scale, 0.5f * scale};
        auto pair_psd = [&](float2 zk, float2 zm, float2 tw, float &pk, float &pm) {
            const v2f e = pk_add_conj(as_v2f(zk), as_v2f(zm));
            const v2f t = pk_cmul_negi(pk_sub_conj(as_v2f(zk), as_v2f(zm)), as_v2f(tw));
            const v2f re = pk_sumdiff_x(e, t), im = pk_sumdiff_y(e, t);
            const v2f pw = (re * re + im * im) * hscale2;
            pk = pw.x; pm = pw.y;
        };
        // bin SB (j + 1024 t) + res and its partner MF minus that: a uniform base per store plus a 32-bit lane offset (as for
        // the loads); SB = 2 at nfft 131072, where this workgroup owns every second bin
        auto put = [&](int t, unsigned lo, unsigned lm, float pk, float pm) {
            *reinterpret_cast<float *>(reinterpret_cast<char *>(o + SB * Q * t + RES) + lo) = pk;
            *reinterpret_cast<float *>(reinterpret_cast<char *>(o + (MF - SB * Q) - SB * Q * t + RES) + lm) = pm;
            if (DB) {
                *reinterpret_cast<float *>(reinterpret_cast<char *>(od + SB * Q * t + RES) + lo) = to_db(pk);
                *reinterpret_cast<float *>(reinterpret_cast<char *>(od + (MF - SB * Q) - SB * Q * t + RES) + lm) = to_db(pm);
            }
        };
        // X2: half the sum of the differences to the pivot = m nfft / 2, what the frame mean left in bin 0 (and, halved and
        // negated, in bin 1: `re` below is twice that bin's real part)
        float corr = 0.f;
        if constexpr (X2) {
#pragma unroll
            for (int w8 = 0; w8 < LPF / 64; w8++) corr += red[1][w8];
            corr *= 0.5f;
        }
        if constexpr (X2 && RES == 1) {
            // odd bins: k = 2 (l + 1024 t) + 1 from (ya[t], yb[31 - t]), its partner 65536 - k from the same pair; no bin
            // pairs with itself
            const unsigned lo1 = 8u * (unsigned)l, lm1 = 8u * (unsigned)(1023 - l);
#pragma unroll
            for (int t = 0; t < R; t++) {
                const float ct = wgs_cos64(t), st = wgs_sin64(t);
                const float2 twa = make_float2(sa.x * ct + sa.y * st, sa.y * ct - sa.x * st);
                const v2f e = pk_add_conj(as_v2f(ya[t]), as_v2f(yb[31 - t]));
                const v2f tt = pk_cmul_negi(pk_sub_conj(as_v2f(ya[t]), as_v2f(yb[31 - t])), as_v2f(twa));
                v2f re = pk_sumdiff_x(e, tt);
                const v2f im = pk_sumdiff_y(e, tt);
                if (t == 0) re.x += l0 ? corr : 0.f;                 // bin 1
                const v2f pw = (re * re + im * im) * hscale2;
                // (the partner of k = 2 (l + 1024 t) + 1 is (65536 - 2048 - 2048 t) + 2 (1023 - l) + 1)
                *reinterpret_cast<float *>(reinterpret_cast<char *>(o + SB * Q * t + 1) + lo1) = pw.x;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(o + (MF - SB * Q) - SB * Q * t + 1) + lm1) = pw.y;
                if (DB) {
                    *reinterpret_cast<float *>(reinterpret_cast<char *>(od + SB * Q * t + 1) + lo1) = to_db(pw.x);
                    *reinterpret_cast<float *>(reinterpret_cast<char *>(od + (MF - SB * Q) - SB * Q * t + 1) + lm1) = to_db(pw.y);
                }
            }
        } else {
            const unsigned loa = 4u * SB * (unsigned)j3a, lma = 4u * SB * (unsigned)(Q - j3a), lob = 4u * SB * (unsigned)j3b,
                           lmb = 4u * SB * (unsigned)(Q - j3b);
#pragma unroll
            for (int t = 0; t < R / 2; t++) {
                // exp(-2 pi i (k0 + SB 1024 t) / nfft) = base * exp(-2 pi i t / 64)
                const float ct = wgs_cos64(t), st = wgs_sin64(t);
                const float2 twa = make_float2(sa.x * ct + sa.y * st, sa.y * ct - sa.x * st);
                const float2 twb = make_float2(sb.x * ct + sb.y * st, sb.y * ct - sb.x * st);
                float pk, pm;
                // (selects of VALUES: `l0 ? ya[i] : yb[j]` is a select of addresses and keeps the arrays in scratch memory)
                const float2 pa0 = ya[t == 0 ? 0 : 32 - t], pa1 = yb[31 - t], pb0 = yb[31 - t], pb1 = ya[31 - t];
                pair_psd(ya[t], make_float2(l0 ? pa0.x : pa1.x, l0 ? pa0.y : pa1.y), twa, pk, pm);
                if (t == 0) {
                    const float dc0 = ya[0].x + ya[0].y - corr, ny = ya[0].x - ya[0].y;     // DC and Nyquist, not doubled
                    pk = l0 ? dc0 * dc0 * scale : pk;
                    pm = l0 ? ny * ny * scale : pm;
                }
                put(t, loa, lma, pk, pm);
                pair_psd(yb[t], make_float2(l0 ? pb0.x : pb1.x, l0 ? pb0.y : pb1.y), twb, pk, pm);
                put(t, lob, lmb, pk, pm);
            }
            if (l0) {
                const float ph = 2.f * scale * (ya[16].x * ya[16].x + ya[16].y * ya[16].y);
                o[MF / 2] = ph;
                if (DB) od[MF / 2] = to_db(ph);
            }
        }
        __syncthreads();                                     // (red[] and the exchange buffer are reused by the next frame)
    };
    if constexpr (X2) {                                      // (one frame per workgroup: no loop for the compiler to hoist out of)
        if (nv > 0) {
            if (res) frame_body(0, std::integral_constant<int, 1>());
            else frame_body(0, std::integral_constant<int, 0>());
        }
    } else {
        for (int it = 0; it < nv; it++) frame_body(it, std::integral_constant<int, 0>());
    }
    // frames behind the last valid one (bufferedspectrogram.py:59)
    for (long long frame = fbeg + nv; res == 0 && frame < fbeg + nh; frame++) {
        float *o = oc + frame * (long long)F;
        for (int f = l; f < F; f += LPF) {
            o[f] = 0.f;
            if (DB) dc[frame * (long long)F + f] = -INFINITY;
        }
    }
}

template <bool X2>
inline int run_chip(hipdsp_ctx *ctx, const float *x, long long x_pitch, long long channels, long long n_valid,
                    long long frames_out, long long out_pitch, int hop, float scale, float *out, float *db_out)
{
    // (a run of ONE frame: neighbours in time are then neighbours on the XCD, and the half they share is in L2; a longer run
    // only amortises a thread's five base twiddles and re-reads its own shared halves 30 us later, from HBM)
    long long fpb = (!X2 && ctx->spec_fpw > 0) ? ctx->spec_fpw : 1;
    if (fpb > 16) fpb = 16;
    const long long runs = (frames_out + fpb - 1) / fpb, total = runs * channels * (X2 ? 2 : 1);
    HD_REQUIRE(total <= 0x7ffffff0LL, "too many frames");
    const dim3 grid((unsigned)(((total + 7) / 8) * 8));
    if (db_out)
        hipLaunchKernelGGL((spec_chip_kernel<true, X2>), grid, dim3(512), 0, ctx->stream, x, x_pitch, n_valid, frames_out,
                           out_pitch, hop, scale, out, db_out, (int)fpb, runs, total);
    else
        hipLaunchKernelGGL((spec_chip_kernel<false, X2>), grid, dim3(512), 0, ctx->stream, x, x_pitch, n_valid, frames_out,
                           out_pitch, hop, scale, out, db_out, (int)fpb, runs, total);
    return hd_launch_status("spec_chip_kernel");
}
