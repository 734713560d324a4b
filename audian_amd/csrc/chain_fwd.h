// chain_fwd.h -- the fused forward sweep of the batch chain for gfx950 (chain_fwd_kernel: band-pass + envelope state
// sweep + spectrogram in one launch, IIR waves hand their tiles to FFT waves through LDS) and what its frames are made
// of.  Included by chain.hip (host side of hipdsp_chain_forward, the backward variant) and by one translation unit per
// window shape (chain_w<nfft>_<hop>.hip -> chain_shape.inc), which hold the kernel's instantiations: 150 of them in one
// unit took hipcc two and a half minutes, six units compile side by side.  Reference: BufferedFilter /
// BufferedSpectrogram / BufferedEnvelope.process of bendalab/audian (src/audian/bufferedfilter.py:31-36,
// bufferedspectrogram.py:45-59, bufferedenvelope.py:34-41) recomputed depth-first by BufferedData.recompute_all
// (buffereddata.py:149-153).
#pragma once
#include "sos_device.h"
// (STOCKHAM_LOADS_FIRST: a window's own translation unit may define it -- fft_device.h, stockham_stage)
#define STOCKHAM_SPLIT_MORE
#include "fft_device.h"
#include <cmath>

// the launchers of the per-shape translation units (chain_shape.inc); `args` is a ChainArgs of `args_size` bytes
#define HD_CHAIN_FWD_DECL(n, h)                                                                                     \
    int hd_chain_fwd_launch_##n##_##h(int SF, int SE, int flags, int want_db, int stamp, unsigned blocks, hipStream_t stream, \
                                      const SosPlanDev *fdev, const SosPlanDev *edev, const void *args, size_t args_size)
HD_CHAIN_FWD_DECL(2048, 1024);
HD_CHAIN_FWD_DECL(2048, 512);
HD_CHAIN_FWD_DECL(1024, 512);
HD_CHAIN_FWD_DECL(1024, 256);
HD_CHAIN_FWD_DECL(512, 256);
HD_CHAIN_FWD_DECL(256, 128);
#undef HD_CHAIN_FWD_DECL

#ifdef CHAIN_AB_PLAIN_SUM
#define CHAIN_MSUM_ADD(x) do { } while (0)
#define CHAIN_MSUM_ON false
#else
#define CHAIN_MSUM_ADD(x) msum_ += (x)
#define CHAIN_MSUM_ON true
#endif

namespace {

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
// Inclusive prefix sum of a float64 over the 64 lanes of a wave inside the VALU: row_shr 1, 2, 4, 8 within the rows of 16
// lanes (zero fill), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_f64(double x)
{
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_prefix_sum(double x)
{
    x += dpp_move_f64<0x111, 0xf>(x);          // row_shr:1
    x += dpp_move_f64<0x112, 0xf>(x);          // row_shr:2
    x += dpp_move_f64<0x114, 0xf>(x);          // row_shr:4
    x += dpp_move_f64<0x118, 0xf>(x);          // row_shr:8
    x += dpp_move_f64<0x142, 0xa>(x);          // row_bcast:15 -> rows 1, 3
    x += dpp_move_f64<0x143, 0xc>(x);          // row_bcast:31 -> rows 2, 3
    return x;
}
__device__ __forceinline__ double wave_read_lane63(double x)
{
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

struct ChainArgs {
    CkptArgs c;
    float *psd;               // (channels, frames_out, TILE/2 + 1)
    float *db;                // optional: decibel(psd), same layout
    long long psd_pitch;
    long long n_valid;        // frames that lie inside the trace
    long long tail_end;       // > n_valid: the FFT wave of a channel's last unit also writes the zero tail, frames
                              // [n_valid, tail_end) (bufferedspectrogram.py:59) -- a few frames, not worth a launch of their own
    const float *tables;      // tw2 | tw3 | twn | window of the 2048-point PSD kernel (fft_tables)
    float scale;              // 1 / (fs * sum w^2)
    float scale_half, scale_twice;   // (PsdScale)
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
    int frame_off;            // frame k of psd / db is frame k + frame_off of the sweep's grid (sos_device.h: GridShift)
};

// One frame of NFFT samples per group of LPF lanes (2048: LPF 64, radix 16 x 16 x 4; 1024: 64, 8 x 8 x 8; 512: two
// frames side by side in a wave, LPF 32, 8 x 8 x 4) from its PPL = NFFT / (2 LPF) values per lane (value t: samples
// 2l + 2 LPF t, + 1 of the frame) -> detrend, Hann, half-length complex FFT, split step, PSD.  `keep` masks the
// stores of a lane group whose frame does not exist.  Same arithmetic as spec_fast_kernel<NFFT, LPF, R1, R2, R3, ...>.
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };

// 1 / (fs sum w^2) and its half and double, from the host: gfx950's scalar unit has no float arithmetic, so 2 * scale
// formed in the kernel is a VGPR that lives across the whole tile loop (the 256-sample kernel kept it in scratch)
struct PsdScale { float one, half, twice; };

// Where a call's bins go: buffer descriptors (SGPRs) whose base is the frame of lane group 0 -- the frames of the other
// groups of a wave (512- and 256-sample windows: two and four side by side) lie `gstride` bytes further on each.  A bin is
// then  descriptor + one 32-bit VGPR offset + a compile-time constant, the addressing mode of buffer_store_dword; with
// 64-bit global pointers every lane carried two pointer pairs per frame (and, for the windows whose frame differs from
// lane group to lane group, a 64-bit multiply per call), which is what pushed the 256-sample kernel's LDS addresses
// into scratch.  num_records = 4 GiB - 1: a call reaches at most four frames beyond its base.
struct BinSink {
    __amdgpu_buffer_rsrc_t psd, db;
    int gstride;
    int soff;                 // bytes from the descriptors' base to lane group 0's frame of THIS call (wave-uniform: the
                              // store instruction's scalar offset -- the descriptors themselves are built once per tile)
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bin_rsrc(float *base)
{
    // (wave-uniform by construction, but hipcc sometimes does the 64-bit arithmetic behind it on the VALU, and a
    // descriptor in VGPRs costs a waterfall loop around EVERY store: pin it to scalar registers)
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float *>(((unsigned long long)hi << 32) | lo), 0, -1, 0x00027000);
}
template <bool DB>
__device__ __forceinline__ BinSink bin_sink(float *psd0, float *db0, int gstride)
{
    BinSink k;
    k.psd = bin_rsrc(psd0);
    k.db = bin_rsrc(DB ? db0 : psd0);
    k.gstride = gstride;
    k.soff = 0;
    return k;
}
__device__ __forceinline__ BinSink bin_at(BinSink k, int soff) { k.soff = soff; return k; }

// The transform behind the window: v = the frame's detrended, windowed values (value t of a lane), corr = what the split
// step still has to take out of bins 0 and 1 (half the sum of what the detrending left: a residual mean m under the periodic
// Hann window is m nfft / 2 in bin 0, -m nfft / 4 in bin 1 and nothing elsewhere).
template <int NFFT, int LPF, int R1, int R2, int R3, bool DB, class Hook = NoHook>
__device__ __forceinline__ void psd_frame_core(float2 *v, const float corr, float2 *fb, const float2 *tw2, const float2 *tw3,
                                               const float2 *twn, int lane, const PsdScale scale, bool keep,
                                               const BinSink &out, Hook hook = Hook());

// Detrending by an EXACT mean (round 5).  The IIR wave of the pair has every band-pass output in a float64 register in
// phase 3 (sos_cascade.inc: CASC_TAP) and adds them up there -- the reference's own arithmetic for detrend='constant'
// (scipy/signal/_spectral_py.py: a float64 mean of the float64 filtered trace, bufferedspectrogram.py:51-56) -- folds the
// rows of a frame once per tile and hands {hi, corr} over with the tile: hi = the mean rounded to float32, corr =
// nfft / 2 times the rest of it.  w - hi is exact where it matters (an offset plus something small: Sterbenz), the rest of
// the mean leaves the spectrum in the split step, and the FFT wave needs neither a pivot nor a sum nor a cross-lane
// reduction per frame (rounds 2-4: two sums and two reductions per frame on the critical path of the pair; at 256 / 128
// that was 32 reductions per tile and 37 % of the launch).
template <int NFFT, int LPF, int R1, int R2, int R3, bool DB, class Hook = NoHook>
__device__ __forceinline__ void psd_frame_mean(const v2f *w, float2 *fb, const float2 *tw2, const float2 *tw3,
                                               const float2 *twn, const float2 *win, int lane, const PsdScale scale, bool keep,
                                               const BinSink &out, const float2 mean_hc, Hook hook = Hook())
{
#pragma clang fp contract(fast)
    constexpr int M = NFFT / 2, PPL = M / LPF;
    const int l = lane % LPF;
    float2 v[PPL];
#ifdef CHAIN_AB_PLAIN_SUM      // (A/B only, never shipped: round 3's plain float32 frame sum in the FFT wave)
    v2f acc = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < R1; t++) acc += w[t];
    float sum = acc.x + acc.y;
    if (LPF == 64) sum = wave_sum(sum);
    else {
#pragma unroll
        for (int d = LPF / 2; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    }
    const float mean = sum * (1.0f / (float)NFFT);
    const v2f mean2 = {mean, mean};
#pragma unroll
    for (int t = 0; t < R1; t++) v[t] = as_f2((w[t] - mean2) * as_v2f(win[l + LPF * t]));
    hook(0);
    psd_frame_core<NFFT, LPF, R1, R2, R3, DB, Hook>(v, 0.f, fb, tw2, tw3, twn, lane, scale, keep, out, hook);
#else
    const v2f hi2 = {mean_hc.x, mean_hc.x};
#pragma unroll
    for (int t = 0; t < R1; t++) v[t] = as_f2((w[t] - hi2) * as_v2f(win[l + LPF * t]));
    hook(0);                                   // mean and window
    psd_frame_core<NFFT, LPF, R1, R2, R3, DB, Hook>(v, mean_hc.y, fb, tw2, tw3, twn, lane, scale, keep, out, hook);
#endif
}

template <int NFFT, int LPF, int R1, int R2, int R3, bool DB, class Hook = NoHook>
__device__ __forceinline__ void psd_frame_pivot(const v2f *w, float2 *fb, const float2 *tw2, const float2 *tw3,
                                          const float2 *twn, const float2 *win, int lane, const PsdScale scale, bool keep,
                                          const BinSink &out, float &piv, bool &have_piv, Hook hook = Hook())
{
#pragma clang fp contract(fast)
    constexpr int M = NFFT / 2, PPL = M / LPF;
    static_assert(PPL == R1 && R1 * R2 * R3 == M, "one first-stage butterfly per lane");
    const int l = lane % LPF;
    float2 v[PPL];
    v2f acc = {0.f, 0.f};
    // The frame mean (detrend='constant') relative to a PIVOT: on an offset plus something small (a low-pass only in front
    // of raw data with a DC offset, a decaying transient) a float32 sum of the samples carries 1e-7 of the OFFSET into
    // bins 0 and 1, the sum of the differences to a value near the mean 1e-7 of the small part (tools/fuzz_stress.py seed
    // 10268 was a frame of this sweep).  The pivot is the mean of the frame BEFORE this one in the wave's sequence (`piv`,
    // carried by the caller; a lane group's own), whatever single samples do: a frame that starts on a pulse a thousand
    // times its baseline -- pulse-type fish, clicks -- has that sample under a window weight of zero, and a pivot taken
    // from it left 6e-8 of the PULSE times nfft / 2 in bins 0 and 1 (test_spectrogram_of_pulses_at_the_frame_borders).
    // The first frame of a sequence takes two steps: a sample as the pivot of a rough mean, that mean as the pivot.
    // Non-finite candidates (a NaN or Inf in the frame) leave the pivot as it was.
    auto gsum = [&](float sum) {
        if (LPF == 64) return wave_sum(sum);
#pragma unroll
        for (int d = LPF / 2; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
        return sum;
    };
    if (!have_piv) {                            // (wave-uniform)
        float p0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(w[0].x)));
        p0 = (fabsf(p0) <= 3.0e38f) ? p0 : 0.f;
        const v2f p02 = {p0, p0};
        v2f a0 = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < R1; t++) a0 += w[t] - p02;
        const float c = p0 + gsum(a0.x + a0.y) * (1.0f / (float)NFFT);
        piv = (fabsf(c) <= 3.0e38f) ? c : p0;
        have_piv = true;
    }
    const v2f pivot2 = {piv, piv};
#pragma unroll
    for (int t = 0; t < R1; t++) {
        const v2f d = w[t] - pivot2;
        v[t] = make_float2(d.x, d.y);
        acc += d;
    }
    const float mean = gsum(acc.x + acc.y) * (1.0f / (float)NFFT);
    {
        const float c = piv + mean;
        piv = (fabsf(c) <= 3.0e38f) ? c : piv;
    }
    const v2f mean2 = {mean, mean};
    // What the subtraction leaves: after a step in the level the differences to the mean of the frame before are all large,
    // `mean` is good to 6e-8 of THEM, and the Hann window puts that error times nfft / 2 into bins 0 and 1.  The detrended
    // samples are summed once more; their mean m1 under the periodic Hann window is m1 nfft / 2 in bin 0, -m1 nfft / 4 in bin
    // 1 and nothing elsewhere, and the split step takes it out (spec_wgs.h, spec_pack.h and spec_fast_kernel do the same).
    v2f rest = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < R1; t++) {
        const v2f q = as_v2f(v[t]) - mean2;
        rest += q;
        v[t] = as_f2(q * as_v2f(win[l + LPF * t]));
    }
    const float corr = 0.5f * gsum(rest.x + rest.y);
    hook(0);                                   // mean and window
    psd_frame_core<NFFT, LPF, R1, R2, R3, DB, Hook>(v, corr, fb, tw2, tw3, twn, lane, scale, keep, out, hook);
}

template <int NFFT, int LPF, int R1, int R2, int R3, bool DB, class Hook>
__device__ __forceinline__ void psd_frame_core(float2 *v, const float corr, float2 *fb, const float2 *tw2, const float2 *tw3,
                                               const float2 *twn, int lane, const PsdScale scale, bool keep,
                                               const BinSink &out, Hook hook)
{
#pragma clang fp contract(fast)
    constexpr int M = NFFT / 2, PPL = M / LPF;
    static_assert(PPL == R1 && R1 * R2 * R3 == M, "one first-stage butterfly per lane");
    const int l = lane % LPF, g0 = (lane / LPF) * LPF;
    stockham_stage<R1, 1, M, LPF, false, true>(v, fb, tw2, l);
    hook(1);                                   // first butterflies, values on their way through LDS
    stockham_stage<R2, R1, M, LPF, true, true>(v, fb, tw2, l);
    hook(2);
    stockham_stage<R3, R1 * R2, M, LPF, true, false, true>(v, fb, tw3, l);
    hook(3);
    // v[u*R3 + t] = Z[k], k = l + LPF*m, m = u + NB3*t; partner bin Z[M-k] from lane LPF-l of the same group
    constexpr int NB3 = PPL / R3;
    const int partner = g0 + ((LPF - l) & (LPF - 1));
    // The two store streams of a lane, bins k = l + LPF m and M - k, as ONE 32-bit byte offset each (the lane's part, with
    // its group's frame) plus a compile-time constant behind the call's descriptor (BinSink).  Written as o[M - k] on a
    // 64-bit pointer, hipcc kept one index register per m alive across the whole tile loop and shifted each of them again
    // in every frame (negative immediates are not folded behind a VGPR offset): seven VALU instructions per frame.
    // (Opaque per call, then masked: a sum offset + constant that hipcc can see through is hoisted out of the tile loop
    // as a value of its own -- nine offset registers per frame --, and one whose sign it does not know is not folded into
    // the instruction's offset field.  The mask tells it the range: at most four frames of 1025 bins.)
    const int gframe = (LPF == 64) ? 0 : (lane / LPF) * out.gstride;
    int bo_k = 4 * l + gframe, bo_m = 4 * (M - l - LPF * (PPL / 2 - 1)) + gframe;
    asm volatile("" : "+v"(bo_k), "+v"(bo_m));
    bo_k &= 0xfffc; bo_m &= 0xfffc;
    const int soff = out.soff;
    auto st_bin = [soff](__amdgpu_buffer_rsrc_t r, int boff, int imm, float val) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(val), r, boff + imm, soff, 0);
    };
    float pk_last = 0.f;
    const v2f hscale2 = {scale.half, scale.half};
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
        v2f re = pk_sumdiff_x(e, t);
        const v2f im = pk_sumdiff_y(e, t);
        if (m == 0) re.x += (l == 1) ? corr : 0.f;                    // bin 1 (`re` is twice its real part)
        const v2f pw = __builtin_elementwise_fma(re, re, im * im) * hscale2;   // (spelled out: the DB and the PSD-only build must round alike)
        float pk = pw.x, pm = pw.y;
        if (m == 0) {
            const float dc0 = zk.x + zk.y - corr, ny = zk.x - zk.y;
            pk = (l == 0) ? dc0 * dc0 * scale.one : pk;
            pm = (l == 0) ? ny * ny * scale.one : pm;
        }
        if (LPF == 64 || keep) {
            st_bin(out.psd, bo_k, 4 * LPF * m, pk);                              // bin k
            st_bin(out.psd, bo_m, 4 * LPF * (PPL / 2 - 1 - m), pm);              // bin M - k
            if (DB) { st_bin(out.db, bo_k, 4 * LPF * m, to_db(pk)); st_bin(out.db, bo_m, 4 * LPF * (PPL / 2 - 1 - m), to_db(pm)); }
        }
        pk_last = pk;
    }
    {
        constexpr int mh = PPL / 2;
        const float2 z = v[(mh % NB3) * R3 + mh / NB3];
        const float ph = scale.twice * fmaf(z.x, z.x, z.y * z.y);
        // (bin M / 2 is lane 0's; the other lanes write their last bin once more, under the same instruction)
        const int bo_h = (l == 0) ? bo_k + 4 * (M / 2) : bo_k + 4 * LPF * (PPL / 2 - 1);        // (bo_k is 4 l + the group's frame)
        const float pv = (l == 0) ? ph : pk_last;
        if (LPF == 64 || keep) {
            st_bin(out.psd, bo_h, 0, pv);
            if (DB) st_bin(out.db, bo_h, 0, to_db(pv));
        }
    }
}

// STAMP (diagnostic build, "chain_debug" bit 32; results stay valid): every wave adds up the shader clocks
// it spends in each part of its loop body and leaves the 16 sums in a.db (which then is NOT a dB output).
#define FPT_OK(hop, g) ((2048 / (hop)) % (g) == 0)
// NFFT / HOP: the window lengths whose frames are register windows of a tile -- 2048 or 1024 samples, hops
// that divide the tile and are multiples of 128 samples (one register of the FFT wave's tile copy).
#ifndef CHAIN_FFT_STORES
#define CHAIN_FFT_STORES 1
#endif
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
    // Who writes an interior tile of the filtered trace to HBM: the FFT wave, out of the sixteen registers it has just
    // copied the tile into (round 5) -- the IIR wave, the pair's critical path (tools/chain_stamps.py), read the tile
    // back from LDS through the one register quad it had left, eight reads and eight stores one after the other.
    // Border tiles (segment ends, the `lead` samples) stay with the IIR wave and its bounds.
    constexpr bool FFT_STORES = CHAIN_FFT_STORES && FLAGS && G != 4;
    __shared__ float4 tiles[NP][64 * 8];
    __shared__ float rprevs[NP][64];
    __shared__ float2 fbs[NP][G * MP];
    __shared__ float2 tab[NTAB];
    // FLAGS: pairwise hand-over instead of the two workgroup barriers -- ready[p] counts the tiles IIR
    // wave p has finished, taken[p] the tiles FFT wave p has copied (monotonic, one writer each)
    __shared__ int ready[NP], taken[NP];
    // the frame means of a tile, {hi, corr} per frame that ENDS in it (psd_frame_mean), from the IIR wave to its FFT
    // wave; two sets, by the parity of the iteration: the FFT wave reads a frame's pair when it gets to the frame, and
    // the IIR wave cannot finish tile it + 2 (it cannot even load it) before the FFT wave has copied tile it + 1, i.e.
    // before it is through with the frames of tile it
    __shared__ float2 fmeans[2][NP][16];
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
    // no envelope warm-up: zero-state tile states + env_fix_kernel, exactly as in sos_ckpt_kernel.  The envelope
    // starts at p' = env0 (sos_device.h: GridShift): the unit whose range holds that tile starts it from the true
    // state there, units in front of it have no envelope work, units behind it start from zero state at `lo`.
    const long long lead = a.c.lead;
    const long long env_tile0 = a.c.env0 - a.c.env0 % TILE;
    const bool env_true = lo <= env_tile0 && env_tile0 < hi;
    const long long env_start = env_true ? env_tile0 : (lo > env_tile0 ? lo : (1LL << 62));
    long long start = lo - PF0->warm;
    if (start < 0) start = 0;
    long long loop_end = last_seg ? T + edge : hi;
    if (!unit_ok) loop_end = start;                     // a pair without a unit only takes the barriers
    // (iteration 0 is tile lo - warm_total: negative = idle iterations)
    // The loop bookkeeping in TILE INDICES of 32 bits (round 5): lo, seg_len and the warm-up are whole tiles, and gfx950's
    // scalar unit compares 32-bit values only -- every `tile < hi` on 64-bit sample positions was a VALU compare, and each
    // position two scalar registers of a kernel that has none to spare (hipcc parks what does not fit in VGPR lanes, and
    // every such VGPR is taken from the FFT role's 128: with one more of them round 4's builds read their LDS operands
    // one at a time).  tt = tile / TILE; only the paths of border tiles go back to sample positions.
    const int TT = (int)(T / TILE);                                  // whole tiles inside the trace: tt < TT <=> tile + TILE <= T
    const int lo_t = (int)(lo / TILE), start_t = (int)(start / TILE), base_t = lo_t - (int)(a.warm_total / TILE);
    const int end_t = (int)((loop_end + TILE - 1) / TILE);           // active <=> start_t <= tt < end_t
    const int hi_full_t = (int)(hi / TILE), hi_ceil_t = (int)((hi + TILE - 1) / TILE);
    const int lead_t = lead > 0 ? 1 : 0;                             // tt >= lead_t <=> tile >= lead (lead < TILE)
    const int env_start_t = env_start >= (1LL << 62) ? 0x7fffffff : (int)(env_start / TILE);

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
        // frame means: E[i] = (sum of the band-pass outputs of rows 0 .. i of the tile before) - (that tile's total), the
        // prefix sums of [previous tile | this tile] continued to the left (eprev_), float64
        double eprev_ = 0.0;
        v4f nx[8];
        bool pre = false;
        int pending = 0;
        const int top_full_t = TT - 1;                               // host guarantees >= 0
        auto fetch = [&](int t0) {
#pragma unroll
            for (int k = 0; k < 8; k++) nx[k] = asm_load16(in + (long long)t0 * TILE + 256 * k + 4 * lane);
        };
        // what iteration k + 1 will read: its tile if that is a full tile of this unit, else a dummy
        auto prefetchable = [&](int t0) { return t0 >= start_t && t0 < end_t && t0 < TT && t0 >= lead_t; };
        pre = prefetchable(base_t);
        fetch(pre ? base_t : top_full_t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        for (int it = 0; it < a.n_iter; it++) {
            const int tt = base_t + it;
            const bool active = tt >= start_t && tt < end_t;
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
                    // a tile that reaches past T (or holds the `lead` samples in front of the trace): untracked loads
                    // from clamped addresses, zeros outside [lead, T)
                    const long long tile = (long long)tt * TILE;
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
                }
            }
            WAVE_SYNC();
            // unconditional, like in sos_ckpt_kernel (a conditional fetch would turn `nx` into a phi
            // whose copies read registers with loads in flight)
            {
                const int next = tt + 1;
                pre = prefetchable(next);
                fetch(pre ? next : top_full_t);
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
            // the lane's row of band-pass outputs, added up in float64 -- the float32 values the FFT wave will see, not
            // the cascade's unrounded ones: what the detrending leaves must have mean zero, or the rounding of one large
            // sample (a pulse under a window weight of zero) is in bins 0 and 1 with nfft / 2 times its share of the mean
            double msum_ = 0.0;
            if (active && !(a.debug & 2)) {
                const double tgain = a.c.gain;
                const SosPlanDev *PEt = PLAN_OF(SE > 0 ? PE0 : PF0);
#define CASC_S SF
#define CASC_PLAN() PLAN_OF(PF0)
#define CASC_CARRY cf_
#define CASC_IN(v) (v)
#define CASC_ROLLED_GROUPS
#define CASC_UNIT_OK (SF <= 2)
#define CASC_STAMP(n) STAMP_AT(1 + (n))
#define CASC_TAP(j, e, y)                                                               \
    do {                                                                                \
        const double ed_ = (double)(e);                                                 \
        CHAIN_MSUM_ADD(ed_);                                                            \
        if constexpr (SE > 0) {                                                         \
            if (((j) & 3) == 0) PEt = PLAN_OF(PE0);                                     \
            const double rd_ = fabs(ed_);                                               \
            _Pragma("unroll") for (int r_ = 0; r_ < DE; r_++) etap[r_] = fma(PEt->G[(j) * DE + r_], rd_, etap[r_]); \
        }                                                                               \
    } while (0)
#include "sos_cascade.inc"
#undef CASC_TAP
#undef CASC_STAMP
#undef CASC_UNIT_OK
#undef CASC_ROLLED_GROUPS
#undef CASC_S
#undef CASC_PLAN
#undef CASC_CARRY
#undef CASC_IN
#pragma unroll
                for (int r = 0; r < DE; r++) etap[r] *= tgain;
            }
            if (active && CHAIN_MSUM_ON) {
                // ---- the means of the frames that end in this tile (psd_frame_mean): inclusive prefix sums of the rows
                // over the lanes, a frame = LR rows that end at row e: P[e] - E[e - LR], E continued into the tile before
                constexpr int LR = NFFT / 32, HR = HOP / 32;
                const double pfx = wave_prefix_sum(msum_);
                const double tot = wave_read_lane63(pfx);
                const double yrot = (LR == 64 || lane >= 64 - LR) ? eprev_ : pfx;
                const double xrot = LR == 64 ? yrot : __shfl(yrot, (lane - LR) & 63, 64);
                const double mean = (pfx - xrot) * (1.0 / (double)NFFT);
                eprev_ = pfx - tot;
                const float mhi = (float)mean;
                const float mcorr = (float)((mean - (double)mhi) * (0.5 * (double)NFFT));
                if (((lane + 1) & (HR - 1)) == 0) fmeans[it & 1][pair][(lane + 1) / HR - 1] = make_float2(mhi, mcorr);
            }
            WAVE_SYNC();
            if (FLAGS) { if (active) CHAIN_POST(ready, it + 1); }
            else __syncthreads();                              // B1: the tile holds the filtered samples
            if (FFT_STORES && active && tt >= lo_t && tt < hi_full_t && tt >= lead_t) {
                // interior tile, stored by the FFT wave: only the prefetch is in flight
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                STAMP_AT(4);
            } else if (active && tt >= lo_t && tt < hi_full_t && tt >= lead_t) {
                // interior tile: exactly 8 vector stores, then the counted wait
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const float4 v = lds[lds_slot(8 * k + (lane >> 3), lane & 7)];
                    f4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
                    *reinterpret_cast<f4u *>(yf + (long long)tt * TILE + 256 * k + 4 * lane) = t;
                }
                asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                STAMP_AT(4);                                   // H1, 8 stores of the filtered tile, wait for the prefetch
            } else {
                // border tile of the segment, warm-up or idle: whatever is stored, no stores to count
                if (active && tt >= lo_t && tt < hi_ceil_t) {
                    const long long tile = (long long)tt * TILE;
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        store_four(yf, tile + 256 * k + 4 * lane, lds[lds_slot(8 * k + (lane >> 3), lane & 7)],
                                   lo > lead ? lo : lead, hi, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // An interior tile of the envelope sweep (neither it nor the next one touches T, no left
            // extension) is not modified any more: the rectification rides on the cascade's input
            // and, with the flags, H2 is only needed before the NEXT tile goes into LDS.
            // (the last tile of a segment that is not the trace's last advances the state too: its end state is handed
            // to the next segment by env_fix_kernel)
            const bool last_tile = tt + 1 >= end_t;
            const bool quiet = SE > 0 && active && tt >= env_start_t && a.c.rectify && tt + 1 < TT &&
                               !(env_true && tt == env_start_t) && (!last_tile || !last_seg) && !(a.debug & 2);
            if (FLAGS) {
                if (active) {
                    if (quiet || SE == 0 || tt < env_start_t) pending = it + 1;   // (nothing touches the tile before the next one)
                    else CHAIN_WAIT_FOR(taken, it + 1, it);
                }
            } else {
                __syncthreads();                               // B2: the FFT wave has its copy
            }
            if constexpr (SE > 0) {
            if (quiet) {
                if ((tt > lo_t || env_true) && lane == 0) {     // (quiet: never the envelope's first tile)
#pragma unroll
                    for (int r = 0; r < DE; r++) ckpt[(long long)tt * DE + r] = ce_[r];
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
                    for (int r = 0; r < DE; r++) ckpt[(long long)(tt + 1) * DE + r] = ce_[r];
                }
                WAVE_SYNC();
            } else if (active && tt >= env_start_t && !(a.debug & 2)) {
                // ---- envelope input in place: r = |y| (the gain rides on the cascade), then the odd extension past T
                const long long tile = (long long)tt * TILE;
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
                if (tt >= TT) {
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
                if (env_true && tt == env_start_t && a.c.env0 > 0) {
                    // the envelope starts inside this tile, at q = env0 - tile (host: edge <= q < TILE - edge): scipy's
                    // left odd extension ext[i] = 2 r(q) - r(q + edge - i), i < edge, goes into the edge samples in
                    // front of q, and everything in front of THAT becomes ext[0] -- a constant input for which
                    // zi * ext[0], the state sosfiltfilt starts from, is the cascade's steady state: the tile is then
                    // scanned like any other and the state arrives at q as if the cascade had started at q - edge.
                    // The backward sweep rebuilds exactly this tile from the filtered trace (env_left_fill).
                    const SosPlanDev *Pz = PLAN_OF(PE0);
                    const float e0 = env_left_fill(ldsf, lane, (int)(a.c.env0 - tile), edge);
                    const double x0 = a.c.gain * (double)e0;
#pragma unroll
                    for (int r = 0; r < DE; r++) ce_[r] = Pz->zi[r] * x0;
                } else if (env_true && tt == env_start_t) {
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
                if ((tt > lo_t || env_true) && lane == 0) {
#pragma unroll
                    for (int r = 0; r < DE; r++) ckpt[(long long)tt * DE + r] = ce_[r];
                }
                {   // (the trace's very last tile advances the state too: slot n_tiles, see sos_device.h: FloodArgs)
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
                        for (int r = 0; r < DE; r++) ckpt[(long long)(tt + 1) * DE + r] = ce_[r];
                    }
                }
                WAVE_SYNC();
            }
            }   // SE > 0
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last (dummy) prefetch
        if (unit_ok && a.c.flags != nullptr && lane == 0) a.c.flags[unit] = state_not_finite(cf_) ? 1 : 0;   // FloodArgs
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
        // What the loop below needs of the kernel's arguments, as opaque scalar values: left as loads from the argument
        // block, hipcc re-fetches them INSIDE the frames when scalar registers run short (round 4's grid shift added
        // three) -- and with a scalar load in flight every wait for an LDS read becomes lgkmcnt(0), because scalar loads
        // return out of order: nineteen full waits per frame instead of counted ones, 4 % of the launch at 1024 / 256
        // and 256 / 128 (profiles/r05d_pmc_fwd_libs.txt: same instruction counts, longer launch).  An asm output
        // cannot be rematerialised from memory; at worst it is parked in a VGPR lane.
        PsdScale fscale = {a.scale, a.scale_half, a.scale_twice};
        int foff = a.frame_off, fsplit = a.split, fdebug = a.debug, n_iter = a.n_iter;
        int nvalid = (int)a.n_valid;                           // (the host keeps frames_out below 2^31 - 2^16)
        asm volatile("" : "+s"(fscale.one), "+s"(fscale.half), "+s"(fscale.twice), "+s"(foff), "+s"(fsplit), "+s"(fdebug),
                     "+s"(n_iter), "+s"(nvalid));
        if (!(fdebug & 64)) __builtin_amdgcn_s_setprio(3);
        float2 *fb = fbs[pair];
        const float *tlf = reinterpret_cast<const float *>(tiles[pair]);
        const float2 *tw2 = tab, *tw3 = tab + TW2, *twn = tab + TW2 + TW3, *win = tab + TW2 + TW3 + TWN;
        float *oc = a.psd + ch * a.psd_pitch;
        float *dc = DB ? a.db + ch * a.psd_pitch : nullptr;
        float *yfc = a.c.yf + ch * a.c.yf_pitch;
        const int st_lo_t = lo_t > lead_t ? lo_t : lead_t;
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
        for (int it = 0; it < n_iter; it++) {
            const int tt = base_t + it;
            const bool active = tt >= start_t && tt < end_t;
            // (which role stands above the other no longer matters once equals keep pace: IIR above FFT 10.95 ms,
            // both on the same two levels 10.83 ms, FFT above IIR 10.85 ms in one process)
            if (!(fdebug & 64)) CHAIN_FAIR(it, 3, 2);
            if (FLAGS) { if (active) CHAIN_WAIT_FOR(ready, it + 1, it); }
            else __syncthreads();                              // B1
            STAMP_AT(8);                                       // waited for the IIR wave's tile
            if (active) {
                if constexpr (G == 4) {
                    const int gq = lane >> 4, l16 = lane & 15;
#pragma unroll
                    for (int j = 0; j < 5; j++)
#pragma unroll
                        for (int tq = 0; tq < 4; tq++) {
                            const int b = 4 * gq - 1 + j;              // (-1: the block the last tile left in pv_)
                            const v2f h = *reinterpret_cast<const v2f *>(tlf + lds_float_index(128 * (b < 0 ? 0 : b) + 2 * (l16 + 16 * tq)));
                            bb_[4 * j + tq] = (j == 0 && b < 0) ? pv_[tq] : h;
                        }
                    // (the old block 15 has just gone into group 0's registers: the new one takes its place)
#pragma unroll
                    for (int tq = 0; tq < 4; tq++)
                        pv_[tq] = *reinterpret_cast<const v2f *>(tlf + lds_float_index(128 * 15 + 2 * (l16 + 16 * tq)));
                } else {
#pragma unroll
                    for (int j = 0; j < 16; j++)
                        cur_[j] = *reinterpret_cast<const v2f *>(tlf + lds_float_index(2 * lane + 128 * j));
                }
            }
            if (FLAGS) {                                        // (the release fence waits for the loads)
                // "chain_debug" bit 8 (fault-path test): FFT wave 0 of workgroup 0 withholds the hand-over of
                // its unit's first tile, so that IIR wave 0 runs into the timeout
                const bool withhold = (fdebug & 8) && blockIdx.x == 0 && pair == 0 && tt == start_t;
                if (active && !withhold) CHAIN_POST(taken, it + 1);
            }
            else __syncthreads();                              // B2
            if constexpr (FFT_STORES) {
                if (active && tt >= st_lo_t && tt < hi_full_t) {
                    float *dst = yfc + (long long)tt * TILE + 2 * lane;
#pragma unroll
                    for (int j = 0; j < 16; j++) {             // (x and yf are shifted by -lead: 4-byte alignment only)
                        f2u t; t.x = cur_[j].x; t.y = cur_[j].y;
                        *reinterpret_cast<f2u *>(dst + 128 * j) = t;
                    }
                }
            }
            STAMP_AT(9);                                       // tile copied, hand-over posted
            if (active) {
                const int t = tt;
                if (tt >= lo_t && tt < hi_ceil_t) {            // the unit that owns the tile writes its frames
                  // where the tile's frames go: descriptors based at the frame that ends HOP samples into the tile (number
                  // t FPT + 1 - NFFT / HOP - foff of the output; below zero in the first tile: never stored through), once
                  // per tile -- a frame is then a compile-time scalar offset of the store instruction, a lane group's
                  // frame (512- and 256-sample windows) a part of the lane's own offset
                  const long long f0 = (long long)(t * FPT + 1 - NFFT / HOP - foff) * F;
                  const BinSink tsink = bin_sink<DB>(oc + f0, dc + f0, (G == 4 ? 16 : 4) * F);
                  if (!(fdebug & 1)) {
                   if constexpr (G == 4) {
                    const int gq = lane >> 4;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        // lane group gq: frame m = 4 gq + q of the tile = its blocks q and q + 1
                        const int f = t * FPT + 4 * gq + q + 1 - NFFT / HOP - foff;
                        const bool keep = (4 * gq + q > 0 || have_prev) && f >= 0 && f < nvalid;
                        if (__builtin_amdgcn_ballot_w64(keep) != 0) {               // (wave-uniform: some group has a frame)
                            v2f w[8];
#pragma unroll
                            for (int i = 0; i < 8; i++) w[i] = bb_[4 * q + i];
                            // (the descriptors' base is the tile's frame 0, group gq's frame lies 4 gq + q frames further on)
                            psd_frame_mean<NFFT, LPF, R1, R2, R3, DB>(w, fb + gq * MP, tw2, tw3, twn, win, lane, fscale, keep,
                                                                      bin_at(tsink, 4 * F * q), fmeans[it & 1][pair][4 * gq + q]);
                        }
                    }
                   } else {
#pragma unroll
                    for (int m = 0; m < FPT; m += G) {
                        const int j0 = ((m + 1) * HOP - NFFT) / 128;          // compile-time after unrolling
                        auto reg = [&](int j) -> v2f { return j < 0 ? prv_[(PREV + j) < 0 ? 0 : (PREV + j)] : cur_[j < 0 ? 0 : j]; };
                        if constexpr (G == 1) {
                            const int f = t * FPT + m + 1 - NFFT / HOP - foff;
                            if ((j0 >= 0 || have_prev) && f >= 0 && f < nvalid && !(fsplit && (f & 1))) {
                                v2f w[PPL];
#pragma unroll
                                for (int i = 0; i < PPL; i++) w[i] = reg(j0 + i);
                                if (STAMP) {
                                    STAMP_AT(10);                  // (between the frames)
                                    auto hook = [&](int n) { STAMP_AT(11 + n); };
                                    psd_frame_mean<NFFT, LPF, R1, R2, R3, DB>(w, fb, tw2, tw3, twn, win, lane, fscale, true,
                                                                              bin_at(tsink, 4 * F * m), fmeans[it & 1][pair][m], hook);
                                    STAMP_AT(15);                  // split step, PSD, stores
                                } else {
                                    psd_frame_mean<NFFT, LPF, R1, R2, R3, DB>(w, fb, tw2, tw3, twn, win, lane, fscale, true,
                                                                              bin_at(tsink, 4 * F * m), fmeans[it & 1][pair][m]);
                                }
                            }
                        } else {
                            // two frames side by side: lanes 0-31 take frame m, lanes 32-63 frame m + 1 (HOP / 128 registers
                            // further on).  Value t of a lane is samples 2l + 64 t of ITS frame, i.e. register t / 2 of that
                            // frame, lower (t even) or upper (t odd) half of the wave: v_permlane32_swap of the two frames'
                            // registers gives {X.lo | Y.lo} and {X.hi | Y.hi} in one instruction per dword.
                            const int gq = lane / LPF;
                            const int f = t * FPT + m + gq + 1 - NFFT / HOP - foff;
                            const bool ok0 = (j0 >= 0 || have_prev), ok1 = (j0 + HOP / 128 >= 0 || have_prev);
                            const bool keep = (gq == 0 ? ok0 : ok1) && f >= 0 && f < nvalid;
                            const int fa = t * FPT + m + 1 - NFFT / HOP - foff;
                            if ((ok0 && fa >= 0 && fa < nvalid) || (ok1 && fa + 1 >= 0 && fa + 1 < nvalid)) {
                                v2f w[2 * PPL];
#pragma unroll
                                for (int u = 0; u < PPL; u++) {
                                    const v2f X = reg(j0 + u), Y = reg(j0 + HOP / 128 + u);
                                    const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_int(X.x), __float_as_int(Y.x), false, false);
                                    const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_int(X.y), __float_as_int(Y.y), false, false);
                                    w[2 * u] = (v2f){__int_as_float(rx[0]), __int_as_float(ry[0])};
                                    w[2 * u + 1] = (v2f){__int_as_float(rx[1]), __int_as_float(ry[1])};
                                }
                                psd_frame_mean<NFFT, LPF, R1, R2, R3, DB>(w, fb + gq * MP, tw2, tw3, twn, win, lane, fscale, keep,
                                                                          bin_at(tsink, 4 * F * m), fmeans[it & 1][pair][m + gq]);
                            }
                        }
                    }
                   }
                  } else if (t == -12345) oc[lane] = cur_[0].x + cur_[9].y + prv_[0].x + bb_[0].x;
                }
                if constexpr (G != 4) {
#pragma unroll
                    for (int j = 0; j < PREV; j++) prv_[j] = cur_[16 - PREV + j];
                }
                have_prev = true;
            }
            STAMP_AT(10);                                      // the tile's (at most) two frames
        }
        if (unit_ok && last_seg && a.tail_end > a.n_valid) {
            for (long long i = a.n_valid * F + lane; i < a.tail_end * F; i += 64) {
                oc[i] = 0.f;
                if (DB) dc[i] = -INFINITY;
            }
        }
    }
    if (STAMP && lane == 0) {
        long long *dst = reinterpret_cast<long long *>(a.db) + ((long long)blockIdx.x * 2 * NP + wave) * 16;
#pragma unroll
        for (int i = 0; i < 16; i++) dst[i] = st_acc[i];
    }
#undef STAMP_AT
}

}  // namespace
