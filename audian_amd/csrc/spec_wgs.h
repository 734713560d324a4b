// spec_wgs.h -- the long windows of BufferedSpectrogram.process that one workgroup holds in LDS (nfft 4096 ... 32768;
// the reference's selector offers them, src/audian/databrowser.py:516, and 4096-16384 are ordinary choices for
// bioacoustic recordings), as a stream.  Included by spectrogram.hip inside its anonymous namespace.
//
// spec_wg_kernel (round 1) let every thread fetch its first-stage inputs from HBM -- 8 bytes per lane, every sample once
// per frame it is part of, the window and the split twiddles from global tables (more bytes through L2 than the frame
// itself) -- and only then started the transform: 2.5 / 1.9 / 1.5 TB/s at 8192 / 16384 / 32768 (profiles/r04a_*).  Here a
// workgroup walks a RUN of consecutive frames:
//   * the raw frame goes into LDS in natural order with 16-byte loads of consecutive lanes; at 50 % overlap (HALF) the
//     upper half of a frame stays in the registers of the thread that fetched it and becomes the lower half of the next
//     frame: each sample leaves HBM once;
//   * the next frame's samples are requested while this frame is transformed, and consumed -- written into LDS --
//     between this frame's last LDS read and its global stores, so that the wait for them is not also a wait for stores;
//   * the Hann window and the split-step twiddles are computed where they are used: v_cos_f32 / v_sin_f32 on exact
//     fractions of a turn for a thread's first input and first bin (|error| < 1e-6 against the 1e-4 of the parity bar),
//     the others by the angle-addition formulas with compile-time constants (a thread's inputs are 1 / R1 of a turn
//     apart in the window's argument, its bins LPF / nfft of a turn in the twiddle's): nothing but samples and bins
//     crosses L2, and a frame costs a thread six to twelve transcendental instructions instead of 48-96.
#pragma once

// cos and sin of j / 64 of a turn as compile-time constants (the steps between a thread's first-stage inputs, 1 / R1 of a
// turn in the window's argument, and between its bins, LPF / nfft of a turn in the split twiddle's, are multiples of it)
constexpr float WGS_C64[17] = {1.f, 0.99518472667219688624f, 0.98078528040323044913f, 0.95694033573220886494f,
                               0.92387953251128675613f, 0.88192126434835502971f, 0.83146961230254523708f,
                               0.77301045336273696081f, 0.70710678118654752440f, 0.63439328416364549822f,
                               0.55557023301960222474f, 0.47139673682599764856f, 0.38268343236508977173f,
                               0.29028467725446236764f, 0.19509032201612826785f, 0.09801714032956060199f, 0.f};
constexpr float wgs_cos64(int j)
{
    j = ((j % 64) + 64) % 64;
    return j <= 16 ? WGS_C64[j] : (j <= 32 ? -WGS_C64[32 - j] : (j <= 48 ? -WGS_C64[j - 32] : WGS_C64[64 - j]));
}
constexpr float wgs_sin64(int j) { return wgs_cos64(j - 16); }

template <int NFFT, int LPF, int R1, int R2, int R3, int OCC, bool DB, bool HALF>
__global__ __launch_bounds__(LPF, OCC) void spec_wgs_kernel(
    const float *__restrict__ x, long long x_pitch, long long n_valid, long long frames_out, long long out_pitch, int hop,
    float scale, const float *__restrict__ tables, float *__restrict__ out, float *__restrict__ db_out, int frames_per_block)
{
    constexpr int M = NFFT / 2, PPL = M / LPF, F = M + 1, MP = M + M / 16;
    static_assert(R1 * R2 * R3 == M && PPL % R1 == 0 && PPL % R2 == 0 && PPL % R3 == 0, "radices");
    constexpr int NB3 = PPL / R3;
    constexpr int TW2 = (R2 - 1) * R1, TW3 = R1 * R2;
    constexpr int NQ = NFFT / 4 / LPF;                 // 16-byte pieces of a frame per thread
    constexpr int NQN = HALF ? NQ / 2 : NQ;            // ... of which are fetched per frame
    constexpr int NW = LPF / 64;
    // PIPE: the next frame's samples are in flight while this frame is transformed (32 more registers per thread at 16
    // points per thread); at 32 points per thread the registers do not go that far (106-150 spilled), so those sizes
    // fetch the next frame once this one's bins are on their way -- the other workgroup of the CU covers the wait
    constexpr bool PIPE = PPL <= 16;
    constexpr bool KEEPW = PPL <= 16 && LPF >= 256;     // (nfft 8192; at 4096 the 32 registers are not there: spills)
    __shared__ __attribute__((aligned(16))) float2 smem[TW2 + TW3 + MP];
    __shared__ float red[NW], red2[NW];
    const float2 *tw2 = smem, *tw3 = smem + TW2;
    float2 *fb = smem + TW2 + TW3;                     // (16-byte aligned: TW2 + TW3 is even)
    static_assert((TW2 + TW3) % 2 == 0, "frame buffer alignment");
    float4 *raw4 = reinterpret_cast<float4 *>(fb);     // the raw frame, natural order, unpadded
    const float2 *raw2 = fb;
    const int l = threadIdx.x, lane = l & 63, wave = l >> 6;
    {
        const float2 *gtab = reinterpret_cast<const float2 *>(tables);
        for (int i = l; i < TW2 + TW3; i += LPF) smem[i] = gtab[i];
    }
    const long long ch = blockIdx.y;
    const float *xc = x + ch * x_pitch;
    float *oc = out + ch * out_pitch;
    float *dc = DB ? db_out + ch * out_pitch : nullptr;
    const long long fbeg = (long long)blockIdx.x * frames_per_block;
    long long nh = frames_out - fbeg;
    if (nh > frames_per_block) nh = frames_per_block;
    long long nvl = n_valid - fbeg;
    const int nv = nvl <= 0 ? 0 : (nvl < nh ? (int)nvl : (int)nh);      // frames of this run that hold a spectrum

    typedef float f4q __attribute__((ext_vector_type(4), aligned(4)));
    float4 keep[HALF ? NQ / 2 : 1], nx[NQN];
    auto fetch = [&](long long frame) {               // the pieces of `frame` this thread does not hold yet
        const float *seg = xc + frame * (long long)hop + (HALF ? NFFT / 2 : 0);
#pragma unroll
        for (int j = 0; j < NQN; j++) {
            const f4q t = *reinterpret_cast<const f4q *>(seg + 4 * (l + LPF * j));
            nx[j] = make_float4(t.x, t.y, t.z, t.w);
        }
    };
    // The frame mean is taken relative to a PIVOT: on an offset plus something small a float32 sum of the samples carries
    // 1e-7 of the OFFSET into bins 0 and 1, the sum of the differences to a value near the mean 1e-7 of the small part
    // (spec_pack.h has the case that showed it).  The pivot of a frame is the MEAN OF THE FRAME BEFORE IT; the run's first
    // frame finds its own in two steps (the run's first sample -- zero if that is not finite -- as the pivot of a rough
    // mean): a sample alone can be a pulse a thousand times the baseline under a window weight of zero (chain.hip's
    // psd_frame has that case).  A half that is kept for the next frame stays a difference to the pivot it was fetched
    // under; the bookkeeping is two scalars (`pivot`, `delta`) and a per-half mean at the window.
    float p0 = (nv > 0) ? xc[fbeg * (long long)hop] : 0.f;
    p0 = (fabsf(p0) <= 3.0e38f) ? p0 : 0.f;
    float pivot = p0, delta = 0.f;
    // the frame, as differences to the pivot, into LDS from (keep | nx); leaves the upper half in `keep` (differences as
    // well: a sample meets the pivot once); returns the thread's share of the sum of the differences
    auto put_raw = [&]() -> float {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        auto add = [&](float4 v) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; };
        auto rel = [&](float4 v) { return make_float4(v.x - pivot, v.y - pivot, v.z - pivot, v.w - pivot); };
        if (HALF) {
#pragma unroll
            for (int j = 0; j < NQ / 2; j++) {
                const float4 d = rel(nx[j]);
                raw4[l + LPF * j] = keep[j];
                raw4[l + LPF * (j + NQ / 2)] = d;
                add(keep[j]); add(d);
                keep[j] = d;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NQ; j++) { const float4 d = rel(nx[j]); raw4[l + LPF * j] = d; add(d); }
        }
        return (acc.x + acc.y) + (acc.z + acc.w);
    };
    auto post_sum = [&](float s) {
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
    };

    // periodic Hann 0.5 - 0.5 cos(2 pi i / nfft) at i = 2n, 2n + 1, n = n0 + t M / R1, n0 = l + LPF u: the angle of input t is
    // that of input 0 plus t / R1 of a turn (the argument of v_cos_f32 / v_sin_f32 is in turns), so two transcendental
    // pairs and the angle-addition formulas with compile-time constants give a thread's R1 window pairs
    static_assert(64 % R1 == 0, "first-stage inputs a multiple of 1/64 turn apart");
    auto window_of = [&](int u, int zero, float2 *w) {
        const int n0 = l + LPF * u + zero;
        const float a0 = (float)(2 * n0) * (1.0f / (float)NFFT), a1 = (float)(2 * n0 + 1) * (1.0f / (float)NFFT);
        const float c0 = __builtin_amdgcn_cosf(a0), s0 = __builtin_amdgcn_sinf(a0);
        const float c1 = __builtin_amdgcn_cosf(a1), s1 = __builtin_amdgcn_sinf(a1);
#pragma unroll
        for (int t = 0; t < R1; t++) {
            constexpr int STEP = 64 / R1;
            const float ct = wgs_cos64(STEP * t), st = wgs_sin64(STEP * t);
            w[t] = make_float2(0.5f - 0.5f * (c0 * ct - s0 * st), 0.5f - 0.5f * (c1 * ct - s1 * st));
        }
    };
    // KEEPW: where the registers allow (16 points per thread), the thread's window values stay in registers for the whole run
    // of frames; elsewhere they are recomputed per frame (an opaque zero in their argument keeps hipcc from hoisting them)
    float2 wkeep[KEEPW ? PPL : 1];
    if (KEEPW) {
#pragma unroll
        for (int u = 0; u < PPL / R1; u++) window_of(u, 0, wkeep + (KEEPW ? u * R1 : 0));
    }

    if (nv > 0) {
        if (HALF) {
            // the first frame whole: its lower half through `keep`
            const float *seg = xc + fbeg * (long long)hop;
#pragma unroll
            for (int j = 0; j < NQ / 2; j++) {
                const f4q t = *reinterpret_cast<const f4q *>(seg + 4 * (l + LPF * j));
                keep[j] = make_float4(t.x, t.y, t.z, t.w);
            }
        }
        fetch(fbeg);
        {
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f);
            auto add0 = [&](float4 v) { a0.x += v.x - p0; a0.y += v.y - p0; a0.z += v.z - p0; a0.w += v.w - p0; };
            if (HALF) {
#pragma unroll
                for (int j = 0; j < NQ / 2; j++) add0(keep[j]);
            }
#pragma unroll
            for (int j = 0; j < NQN; j++) add0(nx[j]);
            post_sum((a0.x + a0.y) + (a0.z + a0.w));
            __syncthreads();                           // (and the tables)
            float total0 = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) total0 += red[w];
            const float c = p0 + total0 * (1.0f / (float)NFFT);
            pivot = (fabsf(c) <= 3.0e38f) ? c : p0;
            __syncthreads();                           // (red[] is written again below)
            if (HALF) {
#pragma unroll
                for (int j = 0; j < NQ / 2; j++)
                    keep[j] = make_float4(keep[j].x - pivot, keep[j].y - pivot, keep[j].z - pivot, keep[j].w - pivot);
            }
        }
        post_sum(put_raw());
        if (PIPE && nv > 1) fetch(fbeg + 1);
        __syncthreads();                               // B0
        for (int it = 0; it < nv; it++) {
            const long long frame = fbeg + it;
            float total = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) total += red[w];
            // the halves of the frame in LDS are differences to the pivots in force when they were fetched: the upper half's
            // is `pivot`, the lower half's `pivot - delta`
            const float mean = total * (1.0f / (float)NFFT) - 0.5f * delta;      // of (sample - pivot)
            // (an always-zero offset the compiler cannot see through: the window and the split twiddles do not depend on
            // the frame, and left alone hipcc computes them once, in front of the loop -- 96 registers that then live
            // through every frame)
            int zero = 0;
            asm volatile("" : "+v"(zero));
            float2 v[PPL];
            // What the subtraction leaves: `mean` is a float32 and the mean of the differences as good as their sum -- after
            // a step in the trace's level the differences to the frame before are all large, and 6e-8 of THEM times nfft / 2
            // would sit in bins 0 and 1 of a frame that is flat.  The detrended samples are summed once more (small
            // whatever came before); their mean m1 times the Hann window is m1 nfft / 2 in bin 0 and -m1 nfft / 4 in bin
            // 1 and nothing anywhere else, and the split step takes it out of those two bins.
            v2f rest = {0.f, 0.f};
#pragma unroll
            for (int u = 0; u < PPL / R1; u++) {
                float2 wl[R1];
                if (!KEEPW) window_of(u, zero, wl);
#pragma unroll
                for (int t = 0; t < R1; t++) {
                    const float2 r = raw2[l + LPF * u + t * (M / R1)];
                    const float2 w = KEEPW ? wkeep[u * R1 + t] : wl[t];
                    // (input t < R1 / 2: the frame's lower half, whose pivot is `delta` behind -- taken off in a subtraction of
                    // its own: exact where it matters, i.e. where delta is large against what is left; one rounded
                    // `mean + delta` for the whole half would be a step of half an ulp of delta in the middle of the frame)
                    v2f q = {r.x, r.y};
                    if (HALF && t < R1 / 2) q = q - (v2f){delta, delta};
                    q = q - (v2f){mean, mean};
                    rest += q;
                    v[u * R1 + t] = make_float2(q.x * w.x, q.y * w.y);
                }
            }
            {
                const float s1 = wave_sum(rest.x + rest.y);
                if (lane == 0) red2[wave] = s1;            // (read at the split step, four barriers on; written again a frame later)
            }
            stockham_stage<R1, 1, M, LPF, false, false>(v, fb, tw2, l);                  // first butterflies
            __syncthreads();                           // B1: every raw sample has been read
            stockham_stage<R1, 1, M, LPF, false, true, false, false>(v, fb, tw2, l);     // ... their stores
            __syncthreads();
            stockham_stage<R2, R1, M, LPF, true, false>(v, fb, tw2, l);
            __syncthreads();
            stockham_stage<R2, R1, M, LPF, false, true, false, false>(v, fb, tw2, l);
            __syncthreads();
            stockham_stage<R3, R1 * R2, M, LPF, true, false, true>(v, fb, tw3, l);
            __syncthreads();
            // v[(m % NB3) * R3 + m / NB3] = Z[k], k = l + LPF m; natural order into LDS for the partner bins Z[M-k] -- only
            // the upper half of the bins (and bin 0) is anybody's partner
#pragma unroll
            for (int m = 0; m < PPL; m++)
                if (m == 0 || m >= PPL / 2) fb[pad16(l) + LPF * m + LPF * m / 16] = v[(m % NB3) * R3 + m / NB3];
            __syncthreads();
            float *o = oc + frame * (long long)F;
            float *od = DB ? dc + frame * (long long)F : nullptr;
            float pk[PPL / 2], pm[PPL / 2], ph = 0.f;
            // bins k = l + LPF m and M - k: a uniform base per store plus a 32-bit lane offset (a 64-bit address per lane and
            // store costs two VALU instructions and two registers each)
            const unsigned lo4 = 4u * (unsigned)l, lm4 = 4u * (unsigned)(LPF - l);
            auto put = [&](int m, float a, float b) {
                *reinterpret_cast<float *>(reinterpret_cast<char *>(o + LPF * m) + lo4) = a;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(o + (M - LPF) - LPF * m) + lm4) = b;
                if (DB) {
                    *reinterpret_cast<float *>(reinterpret_cast<char *>(od + LPF * m) + lo4) = to_db(a);
                    *reinterpret_cast<float *>(reinterpret_cast<char *>(od + (M - LPF) - LPF * m) + lm4) = to_db(b);
                }
            };
            // exp(-2 pi i k / nfft) at k = l + LPF m: bin 0's angle plus m LPF / nfft of a turn
            static_assert((64 * LPF) % NFFT == 0, "bins a multiple of 1/64 turn apart");
            const float tb = (float)(l + zero) * (1.0f / (float)NFFT);
            const float cb = __builtin_amdgcn_cosf(tb), sb = __builtin_amdgcn_sinf(tb);
            const v2f hscale2 = {0.5f * scale, 0.5f * scale};
            float corr = 0.f;                                        // m1 nfft / 2 = half the sum of what the subtraction left
#pragma unroll
            for (int w = 0; w < NW; w++) corr += red2[w];
            corr *= 0.5f;
#pragma unroll
            for (int m = 0; m < PPL / 2; m++) {
                const int k = l + LPF * m;
                const float2 zk = v[(m % NB3) * R3 + m / NB3];
                const float2 zm = fb[pad16((M - k) & (M - 1))];
                constexpr int BSTEP = 64 * LPF / NFFT;
                const float cm = wgs_cos64(BSTEP * m), sm = wgs_sin64(BSTEP * m);
                const v2f tw = {cb * cm - sb * sm, -(sb * cm + cb * sm)};                         // exp(-2 pi i k / nfft)
                // X[k] = E + W O and X[M-k] = conj(E - W O) with 2 E = Z[k] + conj(Z[M-k]), 2 O = -i (Z[k] - conj(Z[M-k])), in
                // packed arithmetic (spec_pack.h's pair_psd)
                const v2f e = pk_add_conj(as_v2f(zk), as_v2f(zm));
                const v2f t = pk_cmul_negi(pk_sub_conj(as_v2f(zk), as_v2f(zm)), tw);
                v2f re = pk_sumdiff_x(e, t);
                const v2f im = pk_sumdiff_y(e, t);
                if (m == 0) re.x += (l == 1) ? corr : 0.f;           // bin 1 (re is twice its real part; the window's bin 1 is -nfft / 4)
                const v2f pw = (re * re + im * im) * hscale2;
                pk[m] = pw.x;
                pm[m] = pw.y;
                if (m == 0) {
                    const float a = zk.x + zk.y - corr, b = zk.x - zk.y;    // DC (the window's bin 0 is nfft / 2) and Nyquist, not doubled
                    pk[m] = (l == 0) ? a * a * scale : pk[m];
                    pm[m] = (l == 0) ? b * b * scale : pm[m];
                }
                if (!PIPE) put(m, pk[m], pm[m]);                     // (nothing is held back: the bins leave at once)
            }
            if (l == 0) {                                            // k = M / 2 pairs with itself
                const float2 z = v[((PPL / 2) % NB3) * R3 + (PPL / 2) / NB3];
                ph = 2.f * scale * (z.x * z.x + z.y * z.y);
                if (!PIPE) {
                    o[M / 2] = ph;
                    if (DB) od[M / 2] = to_db(ph);
                }
            }
            if (!PIPE && it + 1 < nv) fetch(frame + 1);
            __syncthreads();                           // B7: the partner bins have been read, the buffer is free
            // the next frame goes into LDS now (PIPE: its fetch is older than this frame's stores, which follow, and the
            // one after it is requested)
            if (it + 1 < nv) {
                // the next frame's pivot: this frame's mean (a NaN or Inf in this frame: unchanged); what is kept of this
                // frame stays as it is, `delta` remembers by how much the pivot moved
                const float c = pivot + mean;
                const float pn = (fabsf(c) <= 3.0e38f) ? c : pivot;
                float dl = pn - pivot;
                if (HALF && !((pivot + dl == pn) && (pn - dl == pivot))) {
                    // The kept half is x - pivot, the window wants x - pn = (x - pivot) - dl, and dl is not EXACTLY pn - pivot
                    // (a pivot that moves far -- a step in the level -- drops the low bits of the smaller of the two): the
                    // halves would disagree by half an ulp of dl, a step in the middle of the next frame that shows in bins 1,
                    // 3, 5 ... (tools/fuzz_stress.py with runs of 16 frames, nfft 32768: 1.6e-4 of a flat frame's peak behind a
                    // step of 275 sigma).  Rare: fetch the half again (it was read a frame ago) as differences to pn.
                    const float *segk = xc + (frame + 1) * (long long)hop;
#pragma unroll
                    for (int j = 0; j < (HALF ? NQ / 2 : 0); j++) {
                        const f4q t = *reinterpret_cast<const f4q *>(segk + 4 * (l + LPF * j));
                        keep[j] = make_float4(t.x - pn, t.y - pn, t.z - pn, t.w - pn);
                    }
                    dl = 0.f;
                }
                delta = HALF ? dl : 0.f;
                pivot = pn;
                post_sum(put_raw());
                if (PIPE && it + 2 < nv) fetch(frame + 2);
            }
            if (PIPE) {
#pragma unroll
                for (int m = 0; m < PPL / 2; m++) put(m, pk[m], pm[m]);
                if (l == 0) {
                    o[M / 2] = ph;
                    if (DB) od[M / 2] = to_db(ph);
                }
            }
            __syncthreads();                           // B0 of the next frame
        }
    }
    // frames behind the last valid one (bufferedspectrogram.py:59)
    for (long long frame = fbeg + nv; frame < fbeg + nh; frame++) {
        float *o = oc + frame * (long long)F;
        for (int f = l; f < F; f += LPF) {
            o[f] = 0.f;
            if (DB) dc[frame * (long long)F + f] = -INFINITY;
        }
    }
}

template <int NFFT, int LPF, int R1, int R2, int R3, int OCC>
int run_wgs(hipdsp_ctx *ctx, const float *x, long long x_pitch, long long channels, long long n_valid,
            long long frames_out, long long out_pitch, int hop, float scale, float *out, float *db_out)
{
    const float *tables = nullptr;
    int rc = fft_tables(ctx, NFFT, R1, R2, R3, &tables);
    if (rc != HIPDSP_OK) return rc;
    // long runs of frames per workgroup once there are many more frames than the chip holds workgroups
    long long fpb = ctx->spec_fpw > 0 ? ctx->spec_fpw : frames_out * channels / ((long long)ctx->n_cus * 4 * OCC);
    if (fpb < 1) fpb = 1;
    if (fpb > 32) fpb = 32;
    const dim3 grid((unsigned)((frames_out + fpb - 1) / fpb), (unsigned)channels);
    const bool half = hop * 2 == NFFT;
#define HD_WGS(DBV, HALFV)                                                                                                 \
    hipLaunchKernelGGL((spec_wgs_kernel<NFFT, LPF, R1, R2, R3, OCC, DBV, HALFV>), grid, dim3(LPF), 0, ctx->stream, x, x_pitch, \
                       n_valid, frames_out, out_pitch, hop, scale, tables, out, db_out, (int)fpb)
    if (db_out) { if (half) HD_WGS(true, true); else HD_WGS(true, false); }
    else { if (half) HD_WGS(false, true); else HD_WGS(false, false); }
#undef HD_WGS
    return hd_launch_status("spec_wgs_kernel");
}
