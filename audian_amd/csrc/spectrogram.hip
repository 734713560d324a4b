// Framed Hann STFT -> one-sided PSD for gfx950: BufferedSpectrogram.process of
// bendalab/audian (src/audian/bufferedspectrogram.py:45-59), i.e.
// scipy.signal.spectrogram(window='hann', detrend='constant', scaling='density',
// mode='psd') per channel, with the optional fused decibel epilogue
// (thunderlab decibel, src/audian/specitem.py:36).
#include "common.h"
#include <cmath>

namespace {

constexpr float DB_MIN_POWER = 1e-20f;      // thunderlab decibel default min_power

__device__ __forceinline__ float to_db(float p)
{
    return (p <= DB_MIN_POWER) ? -INFINITY : 10.0f * log10f(p);
}

// ---- generic path: any power-of-two nfft in [8, 8192] ---------------------------
// One 256-thread workgroup per (frame, channel); radix-2 Stockham autosort in LDS.
__global__ __launch_bounds__(256) void spec_generic_kernel(
    const float *__restrict__ x, long long x_pitch, long long n_valid, long long frames_out,
    long long out_pitch, int nfft, int hop, float scale, float *__restrict__ out,
    float *__restrict__ db_out)
{
    extern __shared__ float2 fftbuf[];        // 2 * nfft
    __shared__ float red[4];
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
    // detrend='constant': subtract the frame mean
    float s = 0.f;
    for (int i = tid; i < nfft; i += 256) s += seg[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)nfft;
    float2 *in = fftbuf, *ou = fftbuf + nfft;
    for (int i = tid; i < nfft; i += 256) {
        float w = 0.5f - 0.5f * cospif(2.0f * (float)i / (float)nfft);   // periodic Hann
        in[i] = make_float2((seg[i] - mean) * w, 0.f);
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
// mean and applies the register-resident Hann window.  Twiddles come from LDS tables
// computed on the host in float64.

typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 mul_negi(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)

template <int R> __device__ __forceinline__ void dft(float2 *v);

template <> __device__ __forceinline__ void dft<2>(float2 *v)
{
    float2 a = v[0], b = v[1];
    v[0] = cadd(a, b); v[1] = csub(a, b);
}

template <> __device__ __forceinline__ void dft<4>(float2 *v)
{
    float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
    float2 t2 = cadd(v[1], v[3]), t3 = mul_negi(csub(v[1], v[3]));
    v[0] = cadd(t0, t2); v[2] = csub(t0, t2);
    v[1] = cadd(t1, t3); v[3] = csub(t1, t3);
}

template <> __device__ __forceinline__ void dft<8>(float2 *v)
{
    const float h = 0.70710678118654752440f;
    float2 e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
    dft<4>(e); dft<4>(o);
    o[1] = make_float2((o[1].x + o[1].y) * h, (o[1].y - o[1].x) * h);      // * W8^1
    o[2] = mul_negi(o[2]);                                                // * W8^2
    o[3] = make_float2((o[3].y - o[3].x) * h, -(o[3].x + o[3].y) * h);     // * W8^3
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = cadd(e[k], o[k]); v[k + 4] = csub(e[k], o[k]); }
}

template <> __device__ __forceinline__ void dft<16>(float2 *v)
{
    const float h = 0.70710678118654752440f;
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;   // cos, sin(pi/8)
    float2 e[8], o[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { e[k] = v[2 * k]; o[k] = v[2 * k + 1]; }
    dft<8>(e); dft<8>(o);
    o[1] = cmul(o[1], make_float2(c1, -s1));
    o[2] = make_float2((o[2].x + o[2].y) * h, (o[2].y - o[2].x) * h);
    o[3] = cmul(o[3], make_float2(s1, -c1));
    o[4] = mul_negi(o[4]);
    o[5] = cmul(o[5], make_float2(-s1, -c1));
    o[6] = make_float2((o[6].y - o[6].x) * h, -(o[6].x + o[6].y) * h);
    o[7] = cmul(o[7], make_float2(-c1, -s1));
#pragma unroll
    for (int k = 0; k < 8; k++) { v[k] = cadd(e[k], o[k]); v[k + 8] = csub(e[k], o[k]); }
}

__device__ __forceinline__ int pad16(int i) { return i + (i >> 4); }

// One Stockham stage on the PPL register values of this lane.
//   butterfly j = l + LPF*u reads in[j + t*M/R], twiddles by W^(k t), k = j % NS,
//   and writes out[(j/NS)*NS*R + k + t*NS].
template <int R, int NS, int M, int LPF, bool LOAD, bool STORE>
__device__ __forceinline__ void stockham_stage(float2 *v, float2 *fb, const float2 *twm, int l)
{
    constexpr int PPL = M / LPF;
    constexpr int NB = PPL / R;          // butterflies per lane
    // all loads of the stage come before any store: the exchange is in place and one
    // butterfly's outputs land on another butterfly's inputs
    if (LOAD) {
#pragma unroll
        for (int u = 0; u < NB; u++)
#pragma unroll
            for (int t = 0; t < R; t++) v[u * R + t] = fb[pad16(l + LPF * u + t * (M / R))];
    }
#pragma unroll
    for (int u = 0; u < NB; u++) {
        const int j = l + LPF * u;
        float2 *b = v + u * R;
        if (NS > 1) {
            const int k = j % NS;
#pragma unroll
            for (int t = 1; t < R; t++) b[t] = cmul(b[t], twm[k * t * (M / (NS * R))]);
        }
        dft<R>(b);
    }
    if (STORE) {
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const int j = l + LPF * u;
            const int k = j % NS;
            const int base = (j / NS) * NS * R + k;
#pragma unroll
            for (int t = 0; t < R; t++) fb[pad16(base + t * NS)] = v[u * R + t];
        }
    }
}

template <int NFFT, int LPF, int R1, int R2, int R3>
__global__ __launch_bounds__(256) void spec_fast_kernel(
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
    __shared__ float2 smem[M + (M / 2 + 1) + 4 * G * MP];
    float2 *twm = smem;                      // exp(-2 pi i m / M)
    float2 *twn = smem + M;                  // exp(-2 pi i k / NFFT), k <= M/2
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int g = lane / LPF, l = lane % LPF;
    float2 *fb = smem + M + (M / 2 + 1) + (wave * G + g) * MP;

    {   // tables: [window NFFT floats][twm M float2][twn M/2+1 float2]
        const float2 *src = reinterpret_cast<const float2 *>(tables + NFFT);
        for (int i = tid; i < M + M / 2 + 1; i += 256) smem[i] = src[i];
    }
    // Hann window for this lane's samples: n = j + t*M/R1, j = l + LPF*u
    float2 win[PPL];
    {
        const float2 *w2 = reinterpret_cast<const float2 *>(tables);
#pragma unroll
        for (int u = 0; u < PPL / R1; u++)
#pragma unroll
            for (int t = 0; t < R1; t++) win[u * R1 + t] = w2[l + LPF * u + t * (M / R1)];
    }
    __syncthreads();

    const long long ch = blockIdx.y;
    const float *xc = x + ch * x_pitch;
    const long long first = ((long long)blockIdx.x * 4 + wave) * (long long)frames_per_wave * G;

    for (int it = 0; it < frames_per_wave; it++) {
        const long long frame = first + (long long)it * G + g;
        if (frame >= frames_out) continue;               // uniform per lane group
        const long long obase = ch * out_pitch + frame * (long long)F;
        if (frame >= n_valid) {                          // zero tail
            for (int f = l; f < F; f += LPF) {
                out[obase + f] = 0.f;
                if (db_out) db_out[obase + f] = -INFINITY;
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
                f2u r = *reinterpret_cast<const f2u *>(seg + 2 * n);
                v[u * R1 + t] = make_float2(r.x, r.y);
                s += r.x + r.y;
            }
#pragma unroll
        for (int d = LPF / 2; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
        const float mean = s * (1.0f / (float)NFFT);
#pragma unroll
        for (int i = 0; i < PPL; i++)
            v[i] = make_float2((v[i].x - mean) * win[i].x, (v[i].y - mean) * win[i].y);

        stockham_stage<R1, 1, M, LPF, false, true>(v, fb, twm, l);
        stockham_stage<R2, R1, M, LPF, true, true>(v, fb, twm, l);
        stockham_stage<R3, R1 * R2, M, LPF, true, true>(v, fb, twm, l);

        // split step: X[k] = E + W^k O, X[M-k] = conj(E - W^k O)
#pragma unroll
        for (int q = 0; q < PPL / 2; q++) {
            const int k = l + LPF * q;
            const float2 zk = fb[pad16(k)];
            const float2 zm = fb[pad16((M - k) & (M - 1))];
            float pk, pm;
            if (k == 0) {
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
            if (db_out) { db_out[obase + k] = to_db(pk); db_out[obase + M - k] = to_db(pm); }
        }
        if (l == 0) {                                      // k = M/2 pairs with itself
            const float2 z = fb[pad16(M / 2)];
            const float p = 2.f * scale * (z.x * z.x + z.y * z.y);
            out[obase + M / 2] = p;
            if (db_out) db_out[obase + M / 2] = to_db(p);
        }
    }
}

template <int NFFT, int LPF, int R1, int R2, int R3>
int launch_fast(hipdsp_ctx *ctx, const float *x, long long x_pitch, long long channels, long long n_valid,
                long long frames_out, long long out_pitch, int hop, float scale, const float *tables,
                float *out, float *db_out)
{
    constexpr int G = 64 / LPF;
    const int fpw = 16;                                  // frames per wave (x G side by side)
    long long per_block = 4LL * fpw * G;
    long long bx = (frames_out + per_block - 1) / per_block;
    hipLaunchKernelGGL((spec_fast_kernel<NFFT, LPF, R1, R2, R3>), dim3((unsigned)bx, (unsigned)channels),
                       dim3(256), 0, ctx->stream, x, x_pitch, n_valid, frames_out, out_pitch, hop, scale,
                       tables, out, db_out, fpw);
    return hd_launch_status("spec_fast_kernel");
}

// window | twm | twn for one nfft, computed in float64 on the host
int fft_tables(hipdsp_ctx *ctx, int nfft, const float **dev)
{
    int lg = 0;
    while ((1 << lg) < nfft) lg++;
    if (!ctx->fft_tables[lg]) {
        const int M = nfft / 2;
        size_t n = (size_t)nfft + 2 * (size_t)M + 2 * (size_t)(M / 2 + 1);
        float *h = new float[n];
        for (int i = 0; i < nfft; i++) h[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)nfft));
        float *p = h + nfft;
        for (int m = 0; m < M; m++) {
            double a = -2.0 * M_PI * (double)m / (double)M;
            *p++ = (float)cos(a); *p++ = (float)sin(a);
        }
        for (int k = 0; k <= M / 2; k++) {
            double a = -2.0 * M_PI * (double)k / (double)nfft;
            *p++ = (float)cos(a); *p++ = (float)sin(a);
        }
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

}  // namespace

extern "C" int hipdsp_spectrogram(hipdsp_ctx *ctx, const float *x, int64_t x_pitch, int64_t channels,
                                  int64_t frames, int nfft, int hop, double fs, float *out,
                                  float *db_out, int64_t frames_out, int64_t out_pitch)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_REQUIRE(channels >= 0 && frames >= 0 && frames_out >= 0, "negative size");
    HD_REQUIRE(nfft >= 8, "nfft %d < 8", nfft);
    HD_REQUIRE(hop >= 1 && hop <= nfft, "hop %d not in [1, nfft=%d]", hop, nfft);
    HD_REQUIRE(fs > 0, "fs must be positive");
    if ((nfft & (nfft - 1)) != 0 || nfft > 8192) {
        hipdsp_set_error("nfft %d: only powers of two in [8, 8192] are implemented", nfft);
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
    double wss = 0.0;
    for (int i = 0; i < nfft; i++) {
        double w = 0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)nfft);
        wss += w * w;
    }
    float scale = (float)(1.0 / (fs * wss));
    if (!ctx->force_generic_fft && nfft >= 256 && nfft <= 4096) {
        const float *tables = nullptr;
        int rc = fft_tables(ctx, nfft, &tables);
        if (rc != HIPDSP_OK) return rc;
        switch (nfft) {
        case 256:  return launch_fast<256, 16, 8, 4, 4>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, tables, out, db_out);
        case 512:  return launch_fast<512, 32, 8, 8, 4>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, tables, out, db_out);
        case 1024: return launch_fast<1024, 64, 8, 8, 8>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, tables, out, db_out);
        case 2048: return launch_fast<2048, 64, 16, 16, 4>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, tables, out, db_out);
        case 4096: return launch_fast<4096, 64, 16, 16, 8>(ctx, x, x_pitch, channels, n_valid, frames_out, out_pitch, hop, scale, tables, out, db_out);
        }
    }
    size_t lds = sizeof(float2) * 2 * (size_t)nfft;
    if (lds > 48 * 1024)
        HD_CHECK_HIP(hipFuncSetAttribute((const void *)spec_generic_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(spec_generic_kernel, dim3((unsigned)frames_out, (unsigned)channels), dim3(256), lds,
                       ctx->stream, x, (long long)x_pitch, n_valid, (long long)frames_out, (long long)out_pitch,
                       nfft, hop, scale, out, db_out);
    return hd_launch_status("spec_generic_kernel");
}
