// spec_chipx.h -- nfft 262144 and 524288 of BufferedSpectrogram.process (the last two entries of the reference's selector,
// src/audian/databrowser.py:516) on the frame-on-chip transform of spec_chip.h.  Included by spectrogram.hip inside its
// anonymous namespace.
//
// RX = nfft / 65536 = 4 or 8.  One step of decimation in frequency of radix RX in front of the 32768-point transform: with
// z_q, q < RX, the windowed RX-ths of the frame (as complex points), the bins k = RX k' + r of the half-length spectrum
// are the 32768-point transform of (sum_q z_q w^(q r)) W^(n r), w = exp(-2 pi i / RX), W = exp(-2 pi i / (32768 RX)) -- one
// workgroup pass per residue r, its input formed from all RX parts of the frame on the way in.  The split step's partner
// of bin k is bin 32768 RX - k: residue 0 pairs with itself (sub-index k' with 32768 - k', as at 65536), residue RX / 2 with
// itself (k' with 32767 - k'), and residue r with residue RX - r (k' with 32767 - k').  So a frame is RX / 2 + 1 TASKS:
// residue 0, residue RX / 2, and RX / 2 - 1 tasks of TWO passes each (r, then RX - r).  Between its two passes such a task
// parks the first pass's 64 points per thread -- there is no room for them on chip next to the second transform -- in the
// 128 floats per thread of the OUTPUT that are this thread's to write anyway (its bins of both residues): same thread,
// same addresses, written, read back and then overwritten with the result; no scratch, no synchronisation with anybody.
// The windows of the RX parts are one window turned by q / RX of a period, all twiddles come from v_cos_f32 / v_sin_f32 on
// exact fractions of a turn (three pairs per point), the frame mean is handled as in spec_chip.h's X2 path (pivot = mean
// of 2048 samples spread over the frame, the rest taken out of bins 0 and 1 at the split step).  HBM sees the frame once
// per task through L2 (the tasks of a frame are neighbours on an XCD) and each bin once, plus the parking.
#pragma once

template <bool DB, int RX>
__global__ __launch_bounds__(512, 2) void spec_chipx_kernel(
    const float *__restrict__ x, long long x_pitch, long long n_valid, long long frames_out, long long out_pitch, int hop,
    float scale, float *__restrict__ out, float *__restrict__ db_out, long long total_tasks)
{
    static_assert(RX == 4 || RX == 8, "radix of the step in front of the transform");
    constexpr int NFFT = 65536 * RX, M = 32768, MF = NFFT / 2, F = MF + 1, SB = RX, NT = RX / 2 + 1;
    constexpr int LPF = 512, R = 32, Q = M / R, H = M / 2;
    __shared__ float2 xb[H + H / 32];
    __shared__ float red[2][LPF / 64];
    const int l = threadIdx.x, lane = l & 63, wave = l >> 6;
    auto pidx = [](int e) { return e + (e >> 5); };
    const unsigned loff = 8u * (unsigned)l;
    const int wr1 = 33 * l, rd1 = pidx(l);
    // XCD-aware numbering (spec_chip.h): an XCD works on consecutive tasks, i.e. on the tasks of one frame and its neighbours
    const long long per_xcd = (total_tasks + 7) / 8;
    const long long wx = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((long long)(blockIdx.x >> 3) >= per_xcd || wx >= total_tasks) return;
    const int tsk = (int)(wx % NT);
    const long long w = wx / NT;                             // (channel, frame)
    const long long ch = w / frames_out, frame = w % frames_out;
    const float *xc = x + ch * x_pitch;
    float *o = out + ch * out_pitch + frame * (long long)F;
    float *od = DB ? db_out + ch * out_pitch + frame * (long long)F : nullptr;
    if (frame >= n_valid) {                                  // frames behind the last valid one (bufferedspectrogram.py:59)
        if (tsk == 0)
            for (int f = l; f < F; f += LPF) {
                o[f] = 0.f;
                if (DB) od[f] = -INFINITY;
            }
        return;
    }
    // task 0: residue 0; task 1: residue RX / 2; task 2 + i: residues 1 + i and RX - 1 - i
    const int kind = tsk == 0 ? 0 : (tsk == 1 ? 1 : 2);
    const int r0 = tsk == 0 ? 0 : (tsk == 1 ? RX / 2 : tsk - 1);
    const int npass = kind == 2 ? 2 : 1;
    const float *seg = xc + frame * (long long)hop;
    auto cis = [](float turn) { return make_float2(__builtin_amdgcn_cosf(turn), -__builtin_amdgcn_sinf(turn)); };   // exp(-2 pi i turn)

    // the pivot of the frame mean: 2048 samples spread over the frame, none of them on its borders
    float pivot;
    {
        constexpr int ST = NFFT / 2048;
        float p = (seg[ST * l + ST / 2] + seg[ST * (l + 512) + ST / 2]) + (seg[ST * (l + 1024) + ST / 2] + seg[ST * (l + 1536) + ST / 2]);
        p = wave_sum(p);
        if (lane == 0) red[0][wave] = p;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < LPF / 64; w8++) tot += red[0][w8];
        tot *= 1.0f / 2048.0f;
        pivot = (fabsf(tot) <= 3.0e38f) ? tot : 0.f;
    }

    const int k2 = l & 31;                                   // stage 2: W_1024^(k2 t)
    const int j3a = l, j3b = kind == 0 ? ((l == 0) ? 512 : 1024 - l) : 1023 - l;     // stage 3 butterflies: a pair of partners
    const float2 w2 = cis((float)k2 * (1.0f / 1024.0f));
    const float2 w3a = cis((float)j3a * (1.0f / 32768.0f)), w3b = cis((float)j3b * (1.0f / 32768.0f));
    const int wr2 = (l >> 5) * (Q + 32) + (l & 31);
    const int ra = pidx(j3a), rb = pidx(j3b);
    const unsigned lo_l = 4u * RX * (unsigned)l, lo_m = 4u * RX * (unsigned)(1023 - l);      // lane offsets of bins RX (l + ..) and their partners
    // The places of bin k = RX (l + 1024 t) + res and of its partner MF - k = (MF - RX 1024 (t + 1)) + RX (1023 - l) + (RX - res)
    // as a uniform base (SGPR pair) and a 32-bit lane offset; stores and loads in inline asm (from C++ hipcc forms the 128
    // lane addresses of a thread in front of everything and spills them).
    auto base_k = [&](float *base, int t, int res) { return base + SB * Q * t + res; };
    auto base_m = [&](float *base, int t, int res) { return base + (MF - SB * Q) - SB * Q * t + (RX - res); };
    // (s_nop 4: the base may have come out of an SGPR spill slot by v_readlane_b32 the instruction before, and a VMEM
    // instruction that reads an SGPR a VALU instruction has just written needs five wait states -- hipcc inserts them for
    // its own instructions and cannot see into these.  Without them the first run of this kernel took stale bases.)
    auto st_s = [](float *sbase, unsigned voff, float v) {
        asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
    };
    auto ld_s = [](float &v, const float *sbase, unsigned voff) {       // (untracked: s_waitcnt vmcnt(0) by hand before the use)
        asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
    };

    float2 ya[R], yb[R];
    float corr = 0.f;
    for (int pass = 0; pass < npass; pass++) {
        const int r = pass == 0 ? r0 : RX - r0;              // the residue of this pass
        // (what does not depend on the pass -- window angles, load addresses -- is not to be computed in front of the loop
        // and kept: 256 values and 512 addresses.  An always-zero offset and a copy of the frame's address the compiler
        // cannot see through keep them inside.)
        int zero = 0;
        asm volatile("" : "+v"(zero));
        const float *segp = seg;
        asm volatile("" : "+s"(segp));
        // w^(q r) = exp(-2 pi i q r / RX)
        float2 om[RX];
#pragma unroll
        for (int q = 0; q < RX; q++) om[q] = cis((float)((q * r) & (RX - 1)) * (1.0f / (float)RX));
        float2 v0[R], v1[R];
        float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        auto stage_in = [&](auto bc, float2 *v) {
            constexpr int b = decltype(bc)::value, TB = 16 / RX;          // 16 loads at a time (32: no faster)
#pragma unroll
            for (int th = 0; th < R; th += TB) {
                v2f ld[TB * RX];
#pragma unroll
                for (int t = th; t < th + TB; t++)
#pragma unroll
                    for (int q = 0; q < RX; q++) {
                        const float *bq = segp + 2 * (Q * t + 512 * b) + q * (NFFT / RX);
                        asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" : "=v"(ld[(t - th) * RX + q]) : "v"(loff), "s"(bq) : "memory");
                    }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int t = th; t < th + TB; t++) {
                    // complex point n = l + 512 b + 1024 t of the frame's first RX-th: window angles of its two samples
                    // (2 n and 2 n + 1 of nfft), the angle n r / (32768 RX) of W^(n r); part q: the window turned by q / RX
                    const int n = l + 512 * b + 1024 * t + zero;
                    const float ax = (float)(2 * n) * (1.0f / (float)NFFT), ay = (float)(2 * n + 1) * (1.0f / (float)NFFT);
                    const float cx = __builtin_amdgcn_cosf(ax), sx = __builtin_amdgcn_sinf(ax);
                    const float cy = __builtin_amdgcn_cosf(ay), sy = __builtin_amdgcn_sinf(ay);
                    const float2 tw = cis((float)((n * r) & (MF - 1)) * (1.0f / (float)MF));
                    float ux = 0.f, uy = 0.f;
#pragma unroll
                    for (int q = 0; q < RX; q++) {
                        asm volatile("" : "+v"(ld[(t - th) * RX + q]));
                        const v2f p = ld[(t - th) * RX + q];
                        const float dx = p.x - pivot, dy = p.y - pivot;
                        acc8[(t * RX + q) % 8] += dx + dy;
                        // cos(a + 2 pi q / RX) = cos a cq - sin a sq with compile-time cq, sq (multiples of 1/8 turn = 8/64)
                        const float cq = wgs_cos64(q * (64 / RX)), sq = wgs_sin64(q * (64 / RX));
                        const float ex = (0.5f - 0.5f * (cx * cq - sx * sq)) * dx, ey = (0.5f - 0.5f * (cy * cq - sy * sq)) * dy;
                        ux += ex * om[q].x - ey * om[q].y;
                        uy += ex * om[q].y + ey * om[q].x;
                    }
                    v[t] = make_float2(ux * tw.x - uy * tw.y, ux * tw.y + uy * tw.x);
                    asm volatile("" : "+v"(v[t].x), "+v"(v[t].y));       // (here and now, spec_chip.h)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        stage_in(std::integral_constant<int, 0>(), v0);
        __builtin_amdgcn_sched_barrier(0);
        stage_in(std::integral_constant<int, 1>(), v1);
        {
            float acc = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
            acc = wave_sum(acc);
            if (lane == 0) red[1][wave] = acc;               // (read after the transform, eight barriers on; the same in both passes)
        }
        // ---- the 32768-point transform of spec_chip.h: 32 x 32 x 32, two exchanges through LDS in halves
        __builtin_amdgcn_sched_barrier(0);
        dft<32>(v0);
        __builtin_amdgcn_sched_barrier(0);
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
        {
            v2f wq = as_v2f(w2);
#pragma unroll
            for (int t = 1; t < R; t++) {
                u0[t] = as_f2(pk_cmul(as_v2f(u0[t]), wq));
                u1[t] = as_f2(pk_cmul(as_v2f(u1[t]), wq));
                if (t + 1 < R) wq = pk_cmul(wq, as_v2f(w2));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        dft<32>(u0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < R; t++) xb[wr2 + 33 * t] = u0[t];
        __builtin_amdgcn_sched_barrier(0);
        dft<32>(u1);
        __builtin_amdgcn_sched_barrier(0);
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
        {
            v2f wa = as_v2f(w3a), wb = as_v2f(w3b);
#pragma unroll
            for (int t = 1; t < R; t++) {
                ya[t] = as_f2(pk_cmul(as_v2f(ya[t]), wa));
                yb[t] = as_f2(pk_cmul(as_v2f(yb[t]), wb));
                if (t + 1 < R) { wa = pk_cmul(wa, as_v2f(w3a)); wb = pk_cmul(wb, as_v2f(w3b)); }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        dft<32>(ya);
        __builtin_amdgcn_sched_barrier(0);
        dft<32>(yb);
        __builtin_amdgcn_sched_barrier(0);
        // ya[t] = Z_r[j3a + 1024 t], yb[t] = Z_r[j3b + 1024 t]
#pragma unroll
        for (int w8 = 0; w8 < LPF / 64; w8++) corr += (pass == 0) ? red[1][w8] : 0.f;
        if (kind == 2 && pass == 0) {
            // park this pass's points in this thread's own places of the output: ya[t] in the places of bin
            // RX (l + 1024 t) + r and its partner, yb[t] in those of bin RX (l + 1024 t) + RX - r and its partner
#pragma unroll
            for (int t = 0; t < R; t++) {
                st_s(base_k(o, t, r), lo_l, ya[t].x); st_s(base_m(o, t, r), lo_m, ya[t].y);
                st_s(base_k(o, t, RX - r), lo_l, yb[t].x); st_s(base_m(o, t, RX - r), lo_m, yb[t].y);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        }
        __syncthreads();                                     // (red[] and the exchange buffer are written again by the next pass)
    }
    corr *= 0.5f;                                            // m nfft / 2: what the frame mean left in bin 0 (and, halved and negated, in bin 1)

    // ---- split step and PSD
    const bool l0 = l == 0;
    const v2f hscale2 = {0.5f * scale, 0.5f * scale};
    // X[k] and X[MF - k] (times two: the halves go into the scale) from Z[k] and Z[MF - k], tw = exp(-2 pi i k / nfft)
    auto pair_psd = [&](float2 zk, float2 zm, float2 tw, float re_add, float &pk, float &pm) {
        const v2f e = pk_add_conj(as_v2f(zk), as_v2f(zm));
        const v2f tt = pk_cmul_negi(pk_sub_conj(as_v2f(zk), as_v2f(zm)), as_v2f(tw));
        v2f re = pk_sumdiff_x(e, tt);
        const v2f im = pk_sumdiff_y(e, tt);
        re.x += re_add;
        const v2f pw = (re * re + im * im) * hscale2;
        pk = pw.x; pm = pw.y;
    };
    auto put2 = [&](int t, int res, float pk, float pm) {    // bin RX (l + 1024 t) + res and its partner
        st_s(base_k(o, t, res), lo_l, pk);
        st_s(base_m(o, t, res), lo_m, pm);
        if (DB) { st_s(base_k(od, t, res), lo_l, to_db(pk)); st_s(base_m(od, t, res), lo_m, to_db(pm)); }
    };
    // exp(-2 pi i (RX j + res + RX 1024 t) / nfft) = base * exp(-2 pi i t / 64)
    auto tw_of = [&](float2 base, int t) {
        const float ct = wgs_cos64(t), st = wgs_sin64(t);
        return make_float2(base.x * ct + base.y * st, base.y * ct - base.x * st);
    };
    if (kind == 0) {
        // residue 0: the pairing of spec_chip.h at 65536 -- t' < 16: (ya[t'], yb[31 - t']) k' = l + 1024 t' and (yb[t'], ya[31 - t'])
        // k' = 1024 - l + 1024 t'; thread 0: ya[0] = DC and Nyquist, (ya[t'], ya[32 - t']), (yb[t'], yb[31 - t']), ya[16] = the
        // bin in the middle, which pairs with itself
        const float2 sa = cis((float)(RX * j3a) * (1.0f / (float)NFFT)), sb = cis((float)(RX * j3b) * (1.0f / (float)NFFT));
        const unsigned loa = 4u * RX * (unsigned)j3a, lma = 4u * RX * (unsigned)(Q - j3a), lob = 4u * RX * (unsigned)j3b,
                       lmb = 4u * RX * (unsigned)(Q - j3b);
        auto put0 = [&](int t, unsigned lo, unsigned lm, float pk, float pm) {
            st_s(o + SB * Q * t, lo, pk);
            st_s(o + (MF - SB * Q) - SB * Q * t, lm, pm);
            if (DB) { st_s(od + SB * Q * t, lo, to_db(pk)); st_s(od + (MF - SB * Q) - SB * Q * t, lm, to_db(pm)); }
        };
#pragma unroll
        for (int t = 0; t < R / 2; t++) {
            float pk, pm;
            const float2 pa0 = ya[t == 0 ? 0 : 32 - t], pa1 = yb[31 - t], pb0 = yb[31 - t], pb1 = ya[31 - t];
            pair_psd(ya[t], make_float2(l0 ? pa0.x : pa1.x, l0 ? pa0.y : pa1.y), tw_of(sa, t), 0.f, pk, pm);
            if (t == 0) {
                const float dc0 = ya[0].x + ya[0].y - corr, ny = ya[0].x - ya[0].y;     // DC and Nyquist, not doubled
                pk = l0 ? dc0 * dc0 * scale : pk;
                pm = l0 ? ny * ny * scale : pm;
            }
            put0(t, loa, lma, pk, pm);
            pair_psd(yb[t], make_float2(l0 ? pb0.x : pb1.x, l0 ? pb0.y : pb1.y), tw_of(sb, t), 0.f, pk, pm);
            put0(t, lob, lmb, pk, pm);
        }
        if (l0) {
            const float ph = 2.f * scale * (ya[16].x * ya[16].x + ya[16].y * ya[16].y);
            o[MF / 2] = ph;
            if (DB) od[MF / 2] = to_db(ph);
        }
    } else if (kind == 1) {
        // residue RX / 2 pairs with itself: k = RX (l + 1024 t) + RX / 2 from (ya[t], yb[31 - t])
        const float2 sa = cis((float)(RX * l + RX / 2) * (1.0f / (float)NFFT));
#pragma unroll
        for (int t = 0; t < R; t++) {
            float pk, pm;
            pair_psd(ya[t], yb[31 - t], tw_of(sa, t), 0.f, pk, pm);
            put2(t, RX / 2, pk, pm);
        }
    } else {
        // residues r0 (parked: Pa, Pb) and RX - r0 (ya, yb): k1 = RX (l + 1024 t) + r0 from (Pa[t], yb[31 - t]) and
        // k2 = RX (l + 1024 t) + RX - r0 from (ya[t], Pb[31 - t])
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        float2 pq[R];
        {
            // family 1: Pa out of its places, then each pair's two results into the places its Pa[t] came from
#pragma unroll
            for (int t = 0; t < R; t++) { ld_s(pq[t].x, base_k(o, t, r0), lo_l); ld_s(pq[t].y, base_m(o, t, r0), lo_m); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < R; t++) asm volatile("" : "+v"(pq[t].x), "+v"(pq[t].y));
            const float2 s1 = cis((float)(RX * l + r0) * (1.0f / (float)NFFT));
#pragma unroll
            for (int t = 0; t < R; t++) {
                float pk, pm;
                pair_psd(pq[t], yb[31 - t], tw_of(s1, t), (t == 0 && l0 && r0 == 1) ? corr : 0.f, pk, pm);
                put2(t, r0, pk, pm);
            }
        }
        {
            // family 2: pair t needs Pb[31 - t] and writes into Pb[t]'s places -- all of Pb first
#pragma unroll
            for (int t = 0; t < R; t++) { ld_s(pq[t].x, base_k(o, t, RX - r0), lo_l); ld_s(pq[t].y, base_m(o, t, RX - r0), lo_m); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < R; t++) asm volatile("" : "+v"(pq[t].x), "+v"(pq[t].y));
            const float2 s2 = cis((float)(RX * l + RX - r0) * (1.0f / (float)NFFT));
#pragma unroll
            for (int t = 0; t < R; t++) {
                float pk, pm;
                pair_psd(ya[t], pq[31 - t], tw_of(s2, t), 0.f, pk, pm);
                put2(t, RX - r0, pk, pm);
            }
        }
    }
}

template <int RX>
inline int run_chipx(hipdsp_ctx *ctx, const float *x, long long x_pitch, long long channels, long long n_valid,
                     long long frames_out, long long out_pitch, int hop, float scale, float *out, float *db_out)
{
    const long long total = frames_out * channels * (RX / 2 + 1);
    HD_REQUIRE(total <= 0x7ffffff0LL, "too many frames");
    const dim3 grid((unsigned)(((total + 7) / 8) * 8));
    if (db_out)
        hipLaunchKernelGGL((spec_chipx_kernel<true, RX>), grid, dim3(512), 0, ctx->stream, x, x_pitch, n_valid, frames_out,
                           out_pitch, hop, scale, out, db_out, total);
    else
        hipLaunchKernelGGL((spec_chipx_kernel<false, RX>), grid, dim3(512), 0, ctx->stream, x, x_pitch, n_valid, frames_out,
                           out_pitch, hop, scale, out, db_out, total);
    return hd_launch_status("spec_chipx_kernel");
}
