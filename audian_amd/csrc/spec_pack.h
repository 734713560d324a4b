// spec_pack.h -- the short windows of BufferedSpectrogram.process (nfft 8 ... 256: the reference's default is 256 / 128,
// src/audian/bufferedspectrogram.py:14-16; its selector offers 2^3 ... 2^19, src/audian/databrowser.py:516) for any hop.
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

template <int NFFT, int LPF, int R1, int R2, int R3, bool DB>
__global__ __launch_bounds__(256, 3) void spec_pack_kernel(
    const float *__restrict__ x, long long x_pitch, long long frames, long long n_valid, long long frames_out,
    long long out_pitch, int hop, float scale, const float *__restrict__ tables, float *__restrict__ out,
    float *__restrict__ db_out, int batches_per_wave)
{
    constexpr int M = NFFT / 2, PPL = M / LPF, G = 64 / LPF, F = M + 1, MP = M + M / 16;
    constexpr bool THREE = R3 > 1;
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
    // samples [ring_end - RB, ring_end) of the channel may sit in the ring; fetches are 1024 samples from a multiple of 4
    long long ring_end = (f0 * (long long)hop) & ~3LL;
    const int partner = g * LPF + ((LPF - l) & (LPF - 1));

    for (int b = 0; b < batches_per_wave; b++) {
        const long long fb0 = f0 + (long long)b * G;       // first frame of the batch (wave-uniform)
        if (fb0 >= frames_out) break;
        const long long left = frames_out - fb0;
        const int nfr = left < G ? (int)left : G;          // frames to write
        long long nv = n_valid - fb0;
        const int nval = nv <= 0 ? 0 : (nv < G ? (int)nv : G);   // of which hold a spectrum
        const bool valid = g < nval;
        if (nval > 0) {
            const long long need_end = (fb0 + nval - 1) * (long long)hop + NFFT;      // <= frames
            while (ring_end < need_end) {
#pragma unroll
                for (int k = 0; k < CHUNK / 256; k++) {
                    const long long p = ring_end + 256 * k + 4 * lane;
                    float4 v;
                    if (p + 4 <= frames) {
                        const f4p t = *reinterpret_cast<const f4p *>(xc + p);
                        v = make_float4(t.x, t.y, t.z, t.w);
                    } else {
                        v.x = p < frames ? xc[p] : 0.f;
                        v.y = p + 1 < frames ? xc[p + 1] : 0.f;
                        v.z = p + 2 < frames ? xc[p + 2] : 0.f;
                        v.w = p + 3 < frames ? xc[p + 3] : 0.f;
                    }
                    const int ri = (int)(p & (RB - 1));
                    *reinterpret_cast<float4 *>(ring + ri) = v;
                    if (ri == 0) ring[RB] = v.x;
                }
                ring_end += CHUNK;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // ---- the batch's frames, one per lane group, out of the ring
        float2 v[PPL];
        {
            const long long s0 = (fb0 + (valid ? g : 0)) * (long long)hop;
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < PPL / R1; u++)
#pragma unroll
                for (int t = 0; t < R1; t++) {
                    const int n = l + LPF * u + t * (M / R1);
                    const int ri = (int)((s0 + 2 * n) & (RB - 1));
                    const float a = ring[ri], c = ring[ri + 1];
                    v[u * R1 + t] = make_float2(a, c);
                    s += a + c;
                }
#pragma unroll
            for (int d = LPF / 2; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
            const float mean = s * (1.0f / (float)NFFT);
#pragma unroll
            for (int u = 0; u < PPL / R1; u++)
#pragma unroll
                for (int t = 0; t < R1; t++) {
                    const float2 w = win[l + LPF * u + t * (M / R1)];
                    float2 &e = v[u * R1 + t];
                    e = make_float2((e.x - mean) * w.x, (e.y - mean) * w.y);
                }
        }
        constexpr int RL = THREE ? R3 : R2, NBL = PPL / RL;          // the last stage's radix
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
        // LDS operations of a wave execute in order).
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float *sg = stage + g * F;
#pragma unroll
        for (int m = 0; m < PPL / 2; m++) {
            const int k = l + LPF * m;
            const float2 zk = v[(m % NBL) * RL + m / NBL];
            const int mp = PPL - 1 - m;
            const float2 zsrc = v[(mp % NBL) * RL + mp / NBL];
            float2 zm;
            if (LPF > 1) {
                zm.x = __shfl(zsrc.x, partner, 64);
                zm.y = __shfl(zsrc.y, partner, 64);
            } else {
                zm = zsrc;
            }
            if (m > 0) {
                const int m0 = PPL - m;                              // lane 0 pairs inside itself
                const float2 z0 = v[(m0 % NBL) * RL + m0 / NBL];
                if (l == 0) zm = z0;
            }
            float pk, pm;
            if (m == 0 && l == 0) {
                const float a = zk.x + zk.y, c = zk.x - zk.y;        // DC and Nyquist, not doubled
                pk = a * a * scale;
                pm = c * c * scale;
            } else {
                const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
                const float2 o = make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
                const float2 t = cmul(o, twn[k]);
                const float2 a = cadd(e, t), c = csub(e, t);
                pk = 2.f * scale * (a.x * a.x + a.y * a.y);
                pm = 2.f * scale * (c.x * c.x + c.y * c.y);
            }
            sg[k] = valid ? pk : 0.f;
            sg[M - k] = valid ? pm : 0.f;
        }
        if (l == 0) {                                                // k = M / 2 pairs with itself
            constexpr int mh = PPL / 2;
            const float2 z = v[(mh % NBL) * RL + mh / NBL];
            sg[M / 2] = valid ? 2.f * scale * (z.x * z.x + z.y * z.y) : 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- nfr * F consecutive floats of the channel's spectrogram
        {
            const int total = nfr * F;
            float *o = oc + fb0 * (long long)F;
            float *od = DB ? dc + fb0 * (long long)F : nullptr;
            for (int i = 4 * lane; i < total; i += 256) {
                const float4 p = *reinterpret_cast<const float4 *>(stage + i);
                if (i + 4 <= total) {
                    f4p t; t.x = p.x; t.y = p.y; t.z = p.z; t.w = p.w;
                    *reinterpret_cast<f4p *>(o + i) = t;
                    if (DB) {
                        f4p d; d.x = to_db(p.x); d.y = to_db(p.y); d.z = to_db(p.z); d.w = to_db(p.w);
                        *reinterpret_cast<f4p *>(od + i) = d;
                    }
                } else {
                    const float e[4] = {p.x, p.y, p.z, p.w};
                    for (int q = 0; q < 4 && i + q < total; q++) {
                        o[i + q] = e[q];
                        if (DB) od[i + q] = to_db(e[q]);
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
                           n_valid, frames_out, out_pitch, hop, scale, tables, out, db_out, (int)bpw);
    else
        hipLaunchKernelGGL((spec_pack_kernel<NFFT, LPF, R1, R2, R3, false>), grid, block, 0, ctx->stream, x, x_pitch, frames,
                           n_valid, frames_out, out_pitch, hop, scale, tables, out, db_out, (int)bpw);
    return hd_launch_status("spec_pack_kernel");
}
