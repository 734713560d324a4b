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
    const float *__restrict__ x, long long x_pitch, long long n_valid, long long frames_out, int nfft,
    int hop, float scale, float *__restrict__ out, float *__restrict__ db_out)
{
    extern __shared__ float2 fftbuf[];        // 2 * nfft
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const long long frame = blockIdx.x;
    const long long ch = blockIdx.y;
    const int F = nfft / 2 + 1;
    const long long obase = (ch * frames_out + frame) * (long long)F;
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

}  // namespace

extern "C" int hipdsp_spectrogram(hipdsp_ctx *ctx, const float *x, int64_t x_pitch, int64_t channels,
                                  int64_t frames, int nfft, int hop, double fs, float *out,
                                  float *db_out, int64_t frames_out)
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
    size_t lds = sizeof(float2) * 2 * (size_t)nfft;
    if (lds > 48 * 1024)
        HD_CHECK_HIP(hipFuncSetAttribute((const void *)spec_generic_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(spec_generic_kernel, dim3((unsigned)frames_out, (unsigned)channels), dim3(256), lds,
                       ctx->stream, x, (long long)x_pitch, n_valid, (long long)frames_out, nfft, hop, scale,
                       out, db_out);
    return hd_launch_status("spec_generic_kernel");
}
