// Element-wise and layout kernels of libhip_dsp (gfx950): decibel, the
// (T,C) <-> planar conversions at the drop-in edge, and the synthetic generator.
#include "common.h"
#include <cmath>

namespace {

// one 16-byte access per thread and no loop (the form that reaches the device's copy rate, see copy_probe_kernel;
// as a grid-stride loop of single floats it ran at 5.4 TB/s); 4-byte aligned vectors: any pointer will do
typedef float db_f4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ float decibel_of(float v, float inv_ref, float min_power)
{
    return (v <= min_power) ? -INFINITY : 10.0f * log10f(v * inv_ref);
}
__global__ __launch_bounds__(256) void decibel_kernel(const float *__restrict__ p, float *__restrict__ out, long long n,
                                                      float inv_ref, float min_power)
{
    const long long n4 = n / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        const db_f4 v = *reinterpret_cast<const db_f4 *>(p + 4 * i);
        db_f4 r;
        r.x = decibel_of(v.x, inv_ref, min_power); r.y = decibel_of(v.y, inv_ref, min_power);
        r.z = decibel_of(v.z, inv_ref, min_power); r.w = decibel_of(v.w, inv_ref, min_power);
        *reinterpret_cast<db_f4 *>(out + 4 * i) = r;
    } else if (i == n4) {
        for (long long k = 4 * n4; k < n; k++) out[k] = decibel_of(p[k], inv_ref, min_power);
    }
}

// (rows, cols) -> (cols, rows) with optional dB, 32x32 LDS tiles (+1 pad).
template <bool DB>
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                        long long rows, long long cols, float inv_ref,
                                                        float min_power)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const long long r0 = (long long)blockIdx.y * 32, c0 = (long long)blockIdx.x * 32;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        long long r = r0 + ty + 8 * k, c = c0 + tx;
        float v = 0.f;
        if (r < rows && c < cols) {
            v = src[r * cols + c];
            if (DB) v = (v <= min_power) ? -INFINITY : 10.0f * log10f(v * inv_ref);
        }
        tile[ty + 8 * k][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        long long c = c0 + ty + 8 * k, r = r0 + tx;
        if (r < rows && c < cols) dst[c * rows + r] = tile[tx][ty + 8 * k];
    }
}

// Screen-resolution dB image of one channel: column c = max over the frames
// [start + c*step, min(start + (c+1)*step, stop)) of the (frames, F) slab (np.maximum.reduceat, NaN
// propagates), then dB, written transposed as (F, ncols).  32 bins x CPB columns per workgroup; a
// row of 32 lanes reads 128 contiguous bytes per frame, four frames in flight per thread.  CPB = 32 for short
// columns (each of the eight thread rows walks four columns); long columns (step >= 8) take CPB = 8 -- one column per
// thread row, four times the workgroups: an image is a few hundred workgroups at most, and a thread that walks
// 4 x 28 frames one load at a time leaves the chip idle (268 columns of 28 frames ran at 0.93 TB/s).
__device__ __forceinline__ float db_np_max(float v, float w) { return (w > v || w != w) ? w : v; }   // NaN wins, like np.maximum

template <int CPB>
__global__ __launch_bounds__(256) void db_image_decimate_kernel(const float *__restrict__ src,
                                                                float *__restrict__ dst, long long start,
                                                                long long stop, long long step, long long ncols,
                                                                long long F, float inv_ref, float min_power)
{
    __shared__ float tile[CPB][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 bins x 8 columns at a time
    const long long f0 = (long long)blockIdx.x * 32, c0 = (long long)blockIdx.y * CPB;
#pragma unroll
    for (int k = 0; k < CPB / 8; k++) {
        const long long c = c0 + ty + 8 * k, f = f0 + tx;
        float v = 0.f;
        if (c < ncols && f < F) {
            const long long a = start + c * step;
            const long long b = a + step < stop ? a + step : stop;
            const float *p = src + a * F + f;
            v = p[0];
            long long r = a + 1;
            for (; r + 4 <= b; r += 4) {                         // four independent loads in flight
                const float w0 = src[r * F + f], w1 = src[(r + 1) * F + f], w2 = src[(r + 2) * F + f],
                            w3 = src[(r + 3) * F + f];
                v = db_np_max(db_np_max(db_np_max(db_np_max(v, w0), w1), w2), w3);
            }
            for (; r < b; r++) v = db_np_max(v, src[r * F + f]);
            v = (v <= min_power) ? -INFINITY : 10.0f * log10f(v * inv_ref);
        }
        tile[ty + 8 * k][tx] = v;
    }
    __syncthreads();
    // 32 x CPB values out: consecutive lanes write consecutive columns of one bin
#pragma unroll
    for (int k = 0; k < CPB / 8; k++) {
        const int idx = threadIdx.x + 256 * k;
        const int cc = idx % CPB, ff = idx / CPB;
        const long long fo = f0 + ff, c = c0 + cc;
        if (fo < F && c < ncols) dst[fo * ncols + c] = tile[cc][ff];
    }
}

// (T, C) interleaved -> planar (C, pitch) float32
template <typename SRC>
__global__ __launch_bounds__(256) void pack_kernel(const SRC *__restrict__ src, float *__restrict__ dst,
                                                   long long pitch, long long T, long long C)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long long t0 = (long long)blockIdx.x * 32, c0 = (long long)blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        long long t = t0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (t < T && c < C) ? (float)src[t * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        long long c = c0 + ty + 8 * k, t = t0 + tx;
        if (t < T && c < C) dst[c * pitch + t] = tile[tx][ty + 8 * k];
    }
}

// planar (C, pitch) float32 -> (T, C) interleaved float64
__global__ __launch_bounds__(256) void unpack_kernel(const float *__restrict__ src, long long pitch,
                                                     double *__restrict__ dst, long long T, long long C)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long long t0 = (long long)blockIdx.x * 32, c0 = (long long)blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        long long c = c0 + ty + 8 * k, t = t0 + tx;
        tile[ty + 8 * k][tx] = (t < T && c < C) ? src[c * pitch + t] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        long long t = t0 + ty + 8 * k, c = c0 + tx;
        if (t < T && c < C) dst[t * C + c] = (double)tile[tx][ty + 8 * k];
    }
}

// (C, T', F) float32 -> (T', C, F) float64; one block per (frame, channel) row
__global__ void unpack_spectrum_kernel(const float *__restrict__ src, long long src_pitch,
                                       double *__restrict__ dst, long long frames, long long channels,
                                       long long F)
{
    const long long t = blockIdx.x, c = blockIdx.y;
    const float *s = src + c * src_pitch + t * F;
    double *d = dst + (t * channels + c) * F;
    for (long long f = threadIdx.x; f < F; f += blockDim.x) d[f] = (double)s[f];
}

// Ingest: interleaved little-endian signed PCM (frames, channels) with 2, 3 or 4
// bytes per sample -> planar float32 (channels, pitch) scaled by `scale` (1/2^(bits-1) gives
// the [-1, 1) floats that audioio/thunderlab's DataLoader hands to audian, data.py:172).
template <int BYTES>
__global__ __launch_bounds__(256) void pcm_unpack_kernel(const unsigned char *__restrict__ pcm,
                                                         float *__restrict__ dst, long long pitch,
                                                         long long T, long long C, float scale)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long long t0 = (long long)blockIdx.x * 32, c0 = (long long)blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long long t = t0 + ty + 8 * k, c = c0 + tx;
        float v = 0.f;
        if (t < T && c < C) {
            const unsigned char *p = pcm + (t * C + c) * BYTES;
            int iv;
            if (BYTES == 2) iv = (int)(short)((unsigned)p[0] | ((unsigned)p[1] << 8));
            else if (BYTES == 3) iv = ((int)(((unsigned)p[0] << 8) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 24))) >> 8;
            else iv = (int)((unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24));
            v = (float)((double)iv * (double)scale);
        }
        tile[ty + 8 * k][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long long c = c0 + ty + 8 * k, t = t0 + tx;
        if (t < T && c < C) dst[c * pitch + t] = tile[tx][ty + 8 * k];
    }
}

// The same for the shapes a recording usually has -- 16- or 32-bit samples, a multiple of 8 (4) channels -- with 16-byte
// accesses on both sides (the kernel above reads single BYTES: 3.2 TB/s of 6 B per int16 sample): a workgroup takes
// 128 frames x TC channels; a thread loads 16 bytes = 8 (4) channels of one frame, the values cross over through an
// LDS tile (rows of 129 floats: the transposed stores spread over the banks), and leave as 16-byte stores, 512 bytes
// per channel row.
template <int BYTES, int TC>
__global__ __launch_bounds__(256) void pcm_unpack_tile_kernel(const unsigned char *__restrict__ pcm, float *__restrict__ dst,
                                                              long long pitch, long long T, long long C, float scale)
{
    constexpr int TF = 128, P = TF + 1;
    constexpr int SPV = 16 / BYTES;                    // samples per 16-byte vector
    constexpr int VPF = TC / SPV;                      // vectors per frame of the tile
    constexpr int FPP = 256 / VPF;                     // frames per pass of the 256 threads
    __shared__ float tile[TC * P];
    const long long t0 = (long long)blockIdx.x * TF, c0 = (long long)blockIdx.y * TC;
    const int vi = threadIdx.x % VPF, fi = threadIdx.x / VPF;
    const double sc = (double)scale;
#pragma unroll
    for (int k = 0; k < (TF + FPP - 1) / FPP; k++) {
        const int f = fi + FPP * k;
        const long long t = t0 + f;
        if (f < TF && t < T) {
            const uint4 raw = *reinterpret_cast<const uint4 *>(pcm + ((t * C + c0) * BYTES + 16 * vi));
            const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int q = 0; q < SPV; q++) {
                int iv;
                if (BYTES == 2) iv = (int)(short)((w[q >> 1] >> (16 * (q & 1))) & 0xffffu);
                else iv = (int)w[q];
                tile[(SPV * vi + q) * P + f] = (float)((double)iv * sc);
            }
        }
    }
    __syncthreads();
    // TC rows of TF floats out: a thread stores four consecutive frames of one channel
#pragma unroll
    for (int k = 0; k < (TC * (TF / 4) + 255) / 256; k++) {
        const int idx = threadIdx.x + 256 * k;
        if (idx >= TC * (TF / 4)) break;
        const int c = idx / (TF / 4), f = 4 * (idx % (TF / 4));
        const long long t = t0 + f;
        float *o = dst + (c0 + c) * pitch + t;
        const float *r = tile + c * P + f;
        if (t + 4 <= T) {
            db_f4 v; v.x = r[0]; v.y = r[1]; v.z = r[2]; v.w = r[3];
            *reinterpret_cast<db_f4 *>(o) = v;
        } else {
            for (int q = 0; q < 4 && t + q < T; q++) o[q] = r[q];
        }
    }
}

// Playback chain (DataBrowser.play_region, databrowser.py:1711-1729): mean over a group of
// channels of frames [start, start + n), optionally times the heterodyne carrier
// sin(2 pi f k / rate) with k counted from the start of the region.
struct ChannelList {
    int count;
    int idx[64];
};

__global__ void channel_mean_kernel(const float *__restrict__ x, long long pitch, ChannelList ch, long long start,
                                    long long n, double cycles_per_sample, float *__restrict__ out)
{
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n;
         k += (long long)gridDim.x * blockDim.x) {
        double acc = 0.0;
        for (int c = 0; c < ch.count; c++) acc += (double)x[(long long)ch.idx[c] * pitch + start + k];
        float v = (float)(acc / (double)ch.count);
        if (cycles_per_sample != 0.0) {
            double ph = cycles_per_sample * (double)k;
            ph -= floor(ph);
            v *= sinpif(2.0f * (float)ph);
        }
        out[k] = v;
    }
}

__global__ void stride_copy_kernel(const float *__restrict__ x, long long step, long long m, float *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m;
         i += (long long)gridDim.x * blockDim.x)
        out[i] = x[i * step];
}

// Maximum of a non-negative float array (PSD values): workgroup reduction, then one integer
// atomicMax per workgroup on the bit pattern (order-preserving for floats >= 0).
__global__ __launch_bounds__(256) void max_nonneg_kernel(const float *__restrict__ x, long long n,
                                                         unsigned int *__restrict__ out)
{
    __shared__ float red[4];
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        m = (v > m || v != v) ? v : m;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float o = __shfl_xor(m, d, 64);
        m = (o > m || o != o) ? o : m;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) m = (red[w] > m || red[w] != red[w]) ? red[w] : m;
        atomicMax(out, __float_as_uint(m));
    }
}

// np.minimum / np.maximum semantics: a NaN in either operand wins
__device__ __forceinline__ float np_min(float a, float b) { return (a < b || a != a) ? a : b; }
__device__ __forceinline__ float np_max(float a, float b) { return (a > b || a != a) ? a : b; }

// Screen-resolution decimation of traces (TraceItem.update_plot, src/audian/traceitem.py:55-61;
// compresseddata.py:48-52): out[c][2i] = min, out[c][2i+1] = max of x[c][start + i*step :
// min(start + (i+1)*step, stop)]  (np.minimum/maximum.reduceat over arange(0, stop-start, step)).
// A pure read stream: whatever the step, the samples leave HBM as 16-byte accesses of consecutive lanes
// (round 1's kernel let GL lanes stride over a segment with 4-byte loads: 3.6 TB/s at step 28800, 1.4 at step 32).
// NaN: np.minimum / np.maximum return NaN if either operand is one; v_min / v_max return the other operand, so a
// flag rides along and the segment's pair becomes NaN at the end.
typedef float mm_f4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float mm_f2 __attribute__((ext_vector_type(2), aligned(4)));
struct MinMax {
    float mn, mx;
    bool bad;
    __device__ __forceinline__ void take(float v) { mn = fminf(mn, v); mx = fmaxf(mx, v); bad = bad || v != v; }
};

// long segments (step >= 512): one wave per segment, four 16-byte loads per lane in flight
__global__ __launch_bounds__(256) void minmax_long_kernel(const float *__restrict__ x, long long pitch, long long start,
                                                          long long stop, long long step, long long nseg,
                                                          float *__restrict__ out, long long out_pitch)
{
    const long long c = blockIdx.y;
    const float *row = x + c * pitch;
    const int lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nseg) return;
    const long long lo = start + i * step;
    const long long n = (lo + step < stop ? lo + step : stop) - lo;
    const float *seg = row + lo;
    MinMax m = {INFINITY, -INFINITY, false};
    long long base = 0;
    for (; base + 1024 <= n; base += 1024) {
        mm_f4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = *reinterpret_cast<const mm_f4 *>(seg + base + 256 * k + 4 * lane);
#pragma unroll
        for (int k = 0; k < 4; k++) { m.take(v[k].x); m.take(v[k].y); m.take(v[k].z); m.take(v[k].w); }
    }
    for (long long j = base + 4 * lane; j < n; j += 256) {
        if (j + 4 <= n) {
            const mm_f4 v = *reinterpret_cast<const mm_f4 *>(seg + j);
            m.take(v.x); m.take(v.y); m.take(v.z); m.take(v.w);
        } else {
            for (long long q = j; q < n; q++) m.take(seg[q]);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        m.mn = fminf(m.mn, __shfl_xor(m.mn, d, 64));
        m.mx = fmaxf(m.mx, __shfl_xor(m.mx, d, 64));
    }
    const bool bad = __builtin_amdgcn_ballot_w64(m.bad) != 0;
    if (lane == 0) {
        mm_f2 r;
        r.x = bad ? __builtin_nanf("") : m.mn;
        r.y = bad ? __builtin_nanf("") : m.mx;
        *reinterpret_cast<mm_f2 *>(out + c * out_pitch + 2 * i) = r;
    }
}

// shorter segments (step < 512): a wave stages the up to 2048 samples of a run of whole segments in LDS (coalesced
// 16-byte loads; sample j at dword j + j / 32, so that lanes which walk segments side by side -- strides that are
// powers of two -- land on different banks), then GL lanes take a segment each: GL = 1 up to step 32 (a lane walks its
// own segment), 16 up to 128, 64 beyond (lanes stride over the segment and combine by shuffles).
constexpr int MM_TILE = 2048;
template <int GL>
__global__ __launch_bounds__(256) void minmax_short_kernel(const float *__restrict__ x, long long pitch, long long start,
                                                           long long stop, int step, long long nseg, int segs_per_tile,
                                                           float *__restrict__ out, long long out_pitch)
{
    __shared__ float tiles[4][MM_TILE + MM_TILE / 32];
    const long long c = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    float *tile = tiles[wave];
    const long long i0 = ((long long)blockIdx.x * 4 + wave) * segs_per_tile;       // first segment of this wave's tile
    if (i0 >= nseg) return;                                                      // (no workgroup barrier below)
    const int nst = nseg - i0 < segs_per_tile ? (int)(nseg - i0) : segs_per_tile;
    const long long p0 = start + i0 * (long long)step;
    const float *src = x + c * pitch + p0;
    const int ns = (int)((p0 + (long long)nst * step < stop ? p0 + (long long)nst * step : stop) - p0);   // samples of the tile
    for (int j = 4 * lane; j < ns; j += 256) {
        float e[4];
        if (j + 4 <= ns) {
            const mm_f4 v = *reinterpret_cast<const mm_f4 *>(src + j);
            e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) e[q] = j + q < ns ? src[j + q] : 0.f;
        }
        const int d = j + (j >> 5);                      // (four consecutive samples share j / 32)
#pragma unroll
        for (int q = 0; q < 4; q++) tile[d + q] = e[q];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    constexpr int SPP = 64 / GL;                         // segments per pass
    const int g = lane / GL, l = lane % GL;
    float *orow = out + c * out_pitch + 2 * i0;
    for (int s0 = 0; s0 < nst; s0 += SPP) {
        const int sg = s0 + g;
        const bool active = sg < nst;
        const int lo = (active ? sg : 0) * step;
        int hi = lo + step;
        if (hi > ns) hi = ns;
        MinMax m = {INFINITY, -INFINITY, false};
        if (active)
            for (int j = lo + l; j < hi; j += GL) m.take(tile[j + (j >> 5)]);
#pragma unroll
        for (int d = GL / 2; d >= 1; d >>= 1) {
            m.mn = fminf(m.mn, __shfl_xor(m.mn, d, 64));
            m.mx = fmaxf(m.mx, __shfl_xor(m.mx, d, 64));
            const int ob = __shfl_xor((int)m.bad, d, 64);      // (every lane takes part in the shuffle: not behind the ||)
            m.bad = m.bad || ob != 0;
        }
        if (active && l == 0) {
            mm_f2 r;
            r.x = m.bad ? __builtin_nanf("") : m.mn;
            r.y = m.bad ? __builtin_nanf("") : m.mx;
            *reinterpret_cast<mm_f2 *>(orow + 2 * sg) = r;
        }
    }
}

// Power spectrum of the visible window (SpectrogramPlot.update_plot, spectrogramplot.py:158-160):
// mean over frames of one channel's (frames, F) slab, then decibel, then the -200 dB floor.
// Stage 1: partial[s][f] = sum over the s-th slice of frames (float64); stage 2 finishes.
__global__ __launch_bounds__(256) void mean_spectrum_partial(const float *__restrict__ spec, long long F,
                                                             long long i0, long long i1, int nsplit,
                                                             double *__restrict__ partial)
{
    const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int sidx = blockIdx.y;
    const long long n = i1 - i0;
    const long long a = i0 + n * sidx / nsplit, b = i0 + n * (sidx + 1) / nsplit;
    if (f >= F) return;
    double acc = 0.0;
    for (long long t = a; t < b; t++) acc += (double)spec[t * F + f];
    partial[(long long)sidx * F + f] = acc;
}

__global__ __launch_bounds__(256) void mean_spectrum_finish(const double *__restrict__ partial, long long F,
                                                            int nsplit, double inv_n, float inv_ref,
                                                            float min_power, float floor_db,
                                                            float *__restrict__ out)
{
    const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    double acc = 0.0;
    for (int s = 0; s < nsplit; s++) acc += partial[(long long)s * F + f];
    const float p = (float)(acc * inv_n);
    float d = (p <= min_power) ? -INFINITY : 10.0f * log10f(p * inv_ref);
    out[f] = d < floor_db ? floor_db : d;
}

__device__ __forceinline__ unsigned int mix64to32(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ULL;                      // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return (unsigned int)(z >> 32);
}

__global__ void synth_kernel(float *__restrict__ x, long long pitch, long long frames, double rate,
                             unsigned long long seed, long long c0, long long c_total)
{
    const long long c = blockIdx.y;
    const double cyc_per_sample = 1000.0 * (1.0 + (double)(c0 + c) / (double)c_total) / rate;
    float *row = x + c * pitch;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < frames;
         t += (long long)gridDim.x * blockDim.x) {
        unsigned int h = mix64to32(seed ^ ((unsigned long long)(c0 + c) << 40) ^ (unsigned long long)t);
        float u = (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;          // [-1, 1)
        double ph = cyc_per_sample * (double)t;
        ph -= floor(ph);
        row[t] = 0.5f * u + 0.5f * sinpif(2.0f * (float)ph);
    }
}

inline unsigned grid1d(long long n, int per_block, unsigned cap)
{
    long long b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

// ---- unwrap of wrapped-around (clipped) PCM: audioio's unwrap(), which the reference applies to every raw
// buffer it loads (Data.open -> set_unwrap, src/audian/data.py:180; CLI -u / -U, audian.py:1485-1512) ----------
// A recording whose true signal left [-1, 1) wraps around in the file: a step between successive samples
// larger than `thresh` is such a wrap, and from there on the samples are 2 (`step`) too high or too low.
// y[i] = x[i] + step * k[i],  k[i] = k[i-1] + (x[i] - x[i-1] < -thresh) - (x[i] - x[i-1] > thresh),  k[0] = 0,
// then clipped to +-1 (`clips`) or halved (`down_scale`).  A prefix sum of +-1 events along time per channel:
// chunks of UW_CHUNK samples count their events, one workgroup per channel scans the counts, and the chunks
// redo their events with the carried-in count (12 B/sample: the trace is read twice and written once).
constexpr int UW_ROW = 1024;                 // samples a 256-thread workgroup covers per step (float4 each)
constexpr int UW_ROWS = 16;
constexpr int UW_CHUNK = UW_ROW * UW_ROWS;

__device__ __forceinline__ int uw_event(float cur, float prev, float thresh)
{
    const float d = cur - prev;
    return (d < -thresh ? 1 : 0) - (d > thresh ? 1 : 0);
}

// events of samples [p, p + 4) of a row of n samples; sample 0 of the channel has no predecessor
__device__ __forceinline__ void uw_load(const float *x, long long p, long long n, float thresh, float v[4], int e[4])
{
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = p + k < n ? x[p + k] : 0.f;
    float prev = p > 0 && p - 1 < n ? x[p - 1] : v[0];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        e[k] = (p + k < n && p + k > 0) ? uw_event(v[k], prev, thresh) : 0;
        prev = v[k];
    }
}

__global__ __launch_bounds__(256) void unwrap_count_kernel(const float *__restrict__ x, long long pitch, long long n,
                                                           float thresh, int *__restrict__ counts, long long n_chunks)
{
    __shared__ int red[4];
    const long long chunk = blockIdx.x, ch = blockIdx.y;
    const float *xc = x + ch * pitch;
    int c = 0;
    for (int r = 0; r < UW_ROWS; r++) {
        const long long p = chunk * UW_CHUNK + (long long)r * UW_ROW + 4 * threadIdx.x;
        float v[4]; int e[4];
        uw_load(xc, p, n, thresh, v, e);
        c += e[0] + e[1] + e[2] + e[3];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[ch * n_chunks + chunk] = red[0] + red[1] + red[2] + red[3];
}

// counts[ch][chunk] -> events before the chunk (exclusive scan along a channel), in place
__global__ __launch_bounds__(256) void unwrap_scan_kernel(int *__restrict__ counts, long long n_chunks)
{
    __shared__ int part[256];
    int *c = counts + (long long)blockIdx.x * n_chunks;
    const long long per = (n_chunks + 255) / 256;
    const long long a = per * threadIdx.x, b = a + per < n_chunks ? a + per : n_chunks;
    int s = 0;
    for (long long i = a; i < b; i++) s += c[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int t = 0; t < 256; t++) { const int v = part[t]; part[t] = run; run += v; }
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (long long i = a; i < b; i++) { const int v = c[i]; c[i] = run; run += v; }
}

__global__ __launch_bounds__(256) void unwrap_apply_kernel(const float *__restrict__ x, long long x_pitch, long long n,
                                                           float thresh, float step, int clips, float scale,
                                                           const int *__restrict__ counts, long long n_chunks,
                                                           float *__restrict__ y, long long y_pitch)
{
    __shared__ int wsum[4];
    const long long chunk = blockIdx.x, ch = blockIdx.y;
    const float *xc = x + ch * x_pitch;
    float *yc = y + ch * y_pitch;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = counts[ch * n_chunks + chunk];
    for (int r = 0; r < UW_ROWS; r++) {
        const long long p = chunk * UW_CHUNK + (long long)r * UW_ROW + 4 * threadIdx.x;
        float v[4]; int e[4];
        uw_load(xc, p, n, thresh, v, e);
        // inclusive prefix over the row: inside the thread, over the wave, over the four waves
        e[1] += e[0]; e[2] += e[1]; e[3] += e[2];
        int incl = e[3];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = carry + incl - e[3];
        for (int w = 0; w < wave; w++) before += wsum[w];
        const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
        carry += total;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (p + k < n) {
                float o = v[k] + step * (float)(before + e[k]);
                if (clips) o = o < -0.5f * step ? -0.5f * step : (o > 0.5f * step ? 0.5f * step : o);   // (np.clip: NaN stays NaN)
                yc[p + k] = o * scale;
            }
        }
    }
}

// Two adjacent order statistics (ranks k and k + 1, zero based, ascending) of the non-negative floats
// x[i * row_stride + j], i < rows, j < cols -- the band BufferedSpectrogram.estimate_noiselevels takes its
// 95th percentile of (bufferedspectrogram.py:115-117).  For floats >= 0 the IEEE bit pattern orders like the
// value, so the rank is found by a radix select over the bits: three histogram passes (12 + 12 + 8 bits)
// per rank, each over the whole band, with the histogram in LDS.  One workgroup: the band is small
// (frames x F/16 values) and this runs once per trace.  NaNs sort above everything (as np.percentile puts
// them last).
__global__ __launch_bounds__(1024) void band_select_kernel(const float *__restrict__ x, long long rows, long long cols,
                                                          long long row_stride, long long k0, float *__restrict__ out)
{
    __shared__ unsigned int hist[4096];
    __shared__ unsigned int sel_prefix, sel_rank;
    const int tid = threadIdx.x;
    const long long n = rows * cols;
    for (int which = 0; which < 2; which++) {
        unsigned int prefix = 0;                       // the bits decided so far, right aligned
        long long rank = k0 + which;
        if (rank >= n) rank = n - 1;
        int decided = 0;
        for (int pass = 0; pass < 3; pass++) {
            const int bits = pass < 2 ? 12 : 8;
            const int shift = 32 - decided - bits;
            for (int i = tid; i < 4096; i += 1024) hist[i] = 0;
            __syncthreads();
            for (long long i = tid; i < n; i += 1024) {
                const float v = x[(i / cols) * row_stride + (i % cols)];
                unsigned int key = __float_as_uint(v);
                if (v != v) key = 0xffffffffu;
                if (decided == 0 || (key >> (32 - decided)) == prefix)
                    atomicAdd(&hist[(key >> shift) & ((1u << bits) - 1)], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                unsigned long long cum = 0;
                unsigned int b = 0;
                for (; b < (1u << bits); b++) {
                    if (cum + hist[b] > (unsigned long long)rank) break;
                    cum += hist[b];
                }
                if (b >= (1u << bits)) b = (1u << bits) - 1;
                sel_prefix = (prefix << bits) | b;
                sel_rank = (unsigned int)(rank - (long long)cum);
            }
            __syncthreads();
            prefix = sel_prefix;
            rank = sel_rank;
            decided += bits;
            __syncthreads();
        }
        if (tid == 0) out[which] = __uint_as_float(prefix);
    }
}

}  // namespace

// The streaming ceiling of this box as MI355X_MICROARCH.md measures it: ONE float4 per thread, 256-thread blocks, no
// loop -- reads and writes interleave at the finest grain (6.2-6.3 TB/s; grid-stride copies and hipMemcpy D2D reach
// 4.7-5.1, tools/copy_sweep.hip).  bench.py reports it next to the 8 TB/s spec peak.
__global__ __launch_bounds__(256) void copy_probe_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, long long n4)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) dst[i] = src[i];
}

extern "C" {

int hipdsp_decibel(hipdsp_ctx *ctx, const float *p, float *out, int64_t n, double ref_power,
                   double min_power)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_REQUIRE(n >= 0, "negative size");
    HD_REQUIRE(ref_power > 0, "ref_power must be positive");
    if (n == 0) return HIPDSP_OK;
    HD_REQUIRE(p != nullptr && out != nullptr, "NULL data pointer");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    const long long blocks = (n / 4 + 1 + 255) / 256;            // one thread per four values, one more for the tail
    HD_REQUIRE(blocks <= 0x7fffffffLL, "grid too large");
    hipLaunchKernelGGL(decibel_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, p, out,
                       (long long)n, (float)(1.0 / ref_power), (float)min_power);
    return hd_launch_status("decibel_kernel");
}

int hipdsp_decibel_image(hipdsp_ctx *ctx, const float *spec_tf, float *image_ft, int64_t frames,
                         int64_t nfreq, double ref_power, double min_power)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_REQUIRE(frames >= 0 && nfreq >= 0, "negative size");
    HD_REQUIRE(ref_power > 0, "ref_power must be positive");
    if (frames == 0 || nfreq == 0) return HIPDSP_OK;
    HD_REQUIRE(spec_tf != nullptr && image_ft != nullptr, "NULL data pointer");
    HD_REQUIRE((frames + 31) / 32 <= 65535, "too many frames for one image");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    dim3 grid((unsigned)((nfreq + 31) / 32), (unsigned)((frames + 31) / 32));
    hipLaunchKernelGGL(transpose_kernel<true>, grid, dim3(256), 0, ctx->stream, spec_tf, image_ft,
                       (long long)frames, (long long)nfreq, (float)(1.0 / ref_power), (float)min_power);
    return hd_launch_status("transpose_kernel");
}

int hipdsp_decibel_image_decimate(hipdsp_ctx *ctx, const float *spec_tf, float *image_fc, int64_t frames,
                                  int64_t nfreq, int64_t start, int64_t stop, int64_t step, double ref_power,
                                  double min_power)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_REQUIRE(frames >= 0 && nfreq >= 0, "negative size");
    HD_REQUIRE(step >= 1, "step must be >= 1");
    HD_REQUIRE(start >= 0 && start <= stop && stop <= frames, "frame range [%lld, %lld) not inside [0, %lld)",
               (long long)start, (long long)stop, (long long)frames);
    HD_REQUIRE(ref_power > 0, "ref_power must be positive");
    const long long ncols = (stop - start + step - 1) / step;
    if (ncols == 0 || nfreq == 0) return HIPDSP_OK;
    HD_REQUIRE(spec_tf != nullptr && image_fc != nullptr, "NULL data pointer");
    const int cpb = step >= 8 ? 8 : 32;
    HD_REQUIRE((ncols + cpb - 1) / cpb <= 65535, "too many columns for one image");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    dim3 grid((unsigned)((nfreq + 31) / 32), (unsigned)((ncols + cpb - 1) / cpb));
    if (cpb == 8)
        hipLaunchKernelGGL(db_image_decimate_kernel<8>, grid, dim3(256), 0, ctx->stream, spec_tf, image_fc,
                           (long long)start, (long long)stop, (long long)step, ncols, (long long)nfreq,
                           (float)(1.0 / ref_power), (float)min_power);
    else
        hipLaunchKernelGGL(db_image_decimate_kernel<32>, grid, dim3(256), 0, ctx->stream, spec_tf, image_fc,
                           (long long)start, (long long)stop, (long long)step, ncols, (long long)nfreq,
                           (float)(1.0 / ref_power), (float)min_power);
    return hd_launch_status("db_image_decimate_kernel");
}

static int pack_check(hipdsp_ctx *ctx, const void *a, const void *b, int64_t pitch, int64_t frames,
                      int64_t channels)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_REQUIRE(frames >= 0 && channels >= 0, "negative size");
    if (frames == 0 || channels == 0) return HIPDSP_OK;
    HD_REQUIRE(a != nullptr && b != nullptr, "NULL data pointer");
    HD_REQUIRE(pitch >= frames, "pitch smaller than frames");
    HD_REQUIRE((channels + 31) / 32 <= 65535, "too many channels");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    return HIPDSP_OK;
}

int hipdsp_pack_f64(hipdsp_ctx *ctx, const double *src_tc, float *dst, int64_t dst_pitch, int64_t frames,
                    int64_t channels)
{
    int rc = pack_check(ctx, src_tc, dst, dst_pitch, frames, channels);
    if (rc != HIPDSP_OK || frames == 0 || channels == 0) return rc;
    dim3 grid((unsigned)((frames + 31) / 32), (unsigned)((channels + 31) / 32));
    hipLaunchKernelGGL(pack_kernel<double>, grid, dim3(256), 0, ctx->stream, src_tc, dst,
                       (long long)dst_pitch, (long long)frames, (long long)channels);
    return hd_launch_status("pack_kernel<double>");
}

int hipdsp_pack_f32(hipdsp_ctx *ctx, const float *src_tc, float *dst, int64_t dst_pitch, int64_t frames,
                    int64_t channels)
{
    int rc = pack_check(ctx, src_tc, dst, dst_pitch, frames, channels);
    if (rc != HIPDSP_OK || frames == 0 || channels == 0) return rc;
    dim3 grid((unsigned)((frames + 31) / 32), (unsigned)((channels + 31) / 32));
    hipLaunchKernelGGL(pack_kernel<float>, grid, dim3(256), 0, ctx->stream, src_tc, dst,
                       (long long)dst_pitch, (long long)frames, (long long)channels);
    return hd_launch_status("pack_kernel<float>");
}

int hipdsp_unpack_f64(hipdsp_ctx *ctx, const float *src, int64_t src_pitch, double *dst_tc, int64_t frames,
                      int64_t channels)
{
    int rc = pack_check(ctx, src, dst_tc, src_pitch, frames, channels);
    if (rc != HIPDSP_OK || frames == 0 || channels == 0) return rc;
    dim3 grid((unsigned)((frames + 31) / 32), (unsigned)((channels + 31) / 32));
    hipLaunchKernelGGL(unpack_kernel, grid, dim3(256), 0, ctx->stream, src, (long long)src_pitch, dst_tc,
                       (long long)frames, (long long)channels);
    return hd_launch_status("unpack_kernel");
}

int hipdsp_unpack_spectrum_f64(hipdsp_ctx *ctx, const float *src, int64_t src_pitch, double *dst_tcf,
                               int64_t frames, int64_t channels, int64_t nfreq)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_REQUIRE(frames >= 0 && channels >= 0 && nfreq >= 0, "negative size");
    if (frames == 0 || channels == 0 || nfreq == 0) return HIPDSP_OK;
    HD_REQUIRE(src != nullptr && dst_tcf != nullptr, "NULL data pointer");
    HD_REQUIRE(channels <= 65535 && frames <= 0x7fffffffLL, "size out of range");
    if (src_pitch == 0) src_pitch = frames * nfreq;
    HD_REQUIRE(src_pitch >= frames * nfreq, "src_pitch smaller than one channel");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(unpack_spectrum_kernel, dim3((unsigned)frames, (unsigned)channels), dim3(256), 0,
                       ctx->stream, src, (long long)src_pitch, dst_tcf, (long long)frames,
                       (long long)channels, (long long)nfreq);
    return hd_launch_status("unpack_spectrum_kernel");
}

int hipdsp_channel_mean(hipdsp_ctx *ctx, const float *x, int64_t x_pitch, const int *host_channels, int count,
                        int64_t start, int64_t n, double heterodyne_cycles_per_sample, float *out)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_REQUIRE(count >= 1 && count <= 64, "1..64 channels per group, got %d", count);
    HD_REQUIRE(start >= 0 && n >= 0 && x_pitch >= start + n, "bad range");
    if (n == 0) return HIPDSP_OK;
    HD_REQUIRE(x != nullptr && out != nullptr && host_channels != nullptr, "NULL pointer");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    ChannelList cl;
    cl.count = count;
    for (int i = 0; i < 64; i++) cl.idx[i] = i < count ? host_channels[i] : 0;
    for (int i = 0; i < count; i++) HD_REQUIRE(cl.idx[i] >= 0, "negative channel index");
    hipLaunchKernelGGL(channel_mean_kernel, dim3(grid1d(n, 256, 4096)), dim3(256), 0, ctx->stream, x,
                       (long long)x_pitch, cl, (long long)start, (long long)n, heterodyne_cycles_per_sample, out);
    return hd_launch_status("channel_mean_kernel");
}

int hipdsp_stride_copy(hipdsp_ctx *ctx, const float *x, int64_t n, int64_t step, float *out)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_REQUIRE(n >= 0 && step >= 1, "bad argument");
    const long long m = (n + step - 1) / step;
    if (m == 0) return HIPDSP_OK;
    HD_REQUIRE(x != nullptr && out != nullptr, "NULL data pointer");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(stride_copy_kernel, dim3(grid1d(m, 256, 4096)), dim3(256), 0, ctx->stream, x,
                       (long long)step, m, out);
    return hd_launch_status("stride_copy_kernel");
}

int hipdsp_unwrap(hipdsp_ctx *ctx, const float *x, int64_t x_pitch, int64_t channels, int64_t frames, double thresh,
                  double ampl_max, int clips, int down_scale, float *y, int64_t y_pitch)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_REQUIRE(channels >= 0 && frames >= 0, "negative size");
    HD_REQUIRE(thresh > 0 && ampl_max > 0, "thresh and ampl_max must be positive");
    if (channels == 0 || frames == 0) return HIPDSP_OK;
    HD_REQUIRE(x != nullptr && y != nullptr, "NULL data pointer");
    HD_REQUIRE(x_pitch >= frames && y_pitch >= frames, "pitch smaller than row length");
    HD_REQUIRE(channels <= 65535, "more than 65535 channels");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    const long long n_chunks = (frames + UW_CHUNK - 1) / UW_CHUNK;
    HD_REQUIRE(n_chunks <= 0x7fffffffLL, "too many frames");
    void *work = nullptr;
    int rc = hipdsp_scratch(ctx, sizeof(int) * (size_t)n_chunks * (size_t)channels, &work);
    if (rc != HIPDSP_OK) return rc;
    int *counts = (int *)work;
    const dim3 grid((unsigned)n_chunks, (unsigned)channels);
    hipLaunchKernelGGL(unwrap_count_kernel, grid, dim3(256), 0, ctx->stream, x, (long long)x_pitch, (long long)frames,
                       (float)thresh, counts, n_chunks);
    hipLaunchKernelGGL(unwrap_scan_kernel, dim3((unsigned)channels), dim3(256), 0, ctx->stream, counts, n_chunks);
    hipLaunchKernelGGL(unwrap_apply_kernel, grid, dim3(256), 0, ctx->stream, x, (long long)x_pitch, (long long)frames,
                       (float)thresh, (float)(2.0 * ampl_max), clips, (clips || !down_scale) ? 1.0f : 0.5f, counts,
                       n_chunks, y, (long long)y_pitch);
    return hd_launch_status("unwrap kernels");
}

int hipdsp_band_order_stats(hipdsp_ctx *ctx, const float *x, int64_t rows, int64_t cols, int64_t row_stride,
                            int64_t rank, float *out2)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_REQUIRE(rows >= 1 && cols >= 1 && row_stride >= cols, "bad shape");
    HD_REQUIRE(rank >= 0 && rank < rows * cols, "rank %lld not in [0, %lld)", (long long)rank, (long long)(rows * cols));
    HD_REQUIRE(rows * cols < (1LL << 32), "band too large for the 32-bit rank counters");
    HD_REQUIRE(x != nullptr && out2 != nullptr, "NULL data pointer");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(band_select_kernel, dim3(1), dim3(1024), 0, ctx->stream, x, (long long)rows, (long long)cols,
                       (long long)row_stride, (long long)rank, out2);
    return hd_launch_status("band_select_kernel");
}

int hipdsp_max_nonneg(hipdsp_ctx *ctx, const float *x, int64_t n, float *out)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_REQUIRE(n >= 0 && out != nullptr, "bad argument");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    HD_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float), ctx->stream));
    if (n == 0) return HIPDSP_OK;
    HD_REQUIRE(x != nullptr, "NULL data pointer");
    hipLaunchKernelGGL(max_nonneg_kernel, dim3(grid1d(n, 4096, 1024)), dim3(256), 0, ctx->stream, x, (long long)n,
                       (unsigned int *)out);
    return hd_launch_status("max_nonneg_kernel");
}

int hipdsp_pcm_unpack(hipdsp_ctx *ctx, const void *pcm_tc, int sample_bytes, int64_t frames, int64_t channels,
                      double scale, float *dst, int64_t dst_pitch)
{
    int rc = pack_check(ctx, pcm_tc, dst, dst_pitch, frames, channels);
    if (rc != HIPDSP_OK || frames == 0 || channels == 0) return rc;
    const unsigned char *p = (const unsigned char *)pcm_tc;
    // 16-byte vectors on both sides when whole vectors of channels line up: 8 (int16) or 4 (int32) channels per vector,
    // frames that start on a 16-byte boundary
    if ((sample_bytes == 2 || sample_bytes == 4) && ((uintptr_t)pcm_tc & 15) == 0 && (channels * sample_bytes) % 16 == 0 &&
        (frames + 127) / 128 <= 0x7fffffffLL) {
        const int spv = 16 / sample_bytes;
        int tc = 64;
        while (tc > spv && channels % tc) tc >>= 1;
        if (channels % tc == 0 && channels / tc <= 65535) {
            const dim3 tgrid((unsigned)((frames + 127) / 128), (unsigned)(channels / tc));
#define HD_PCM_TILE(B, TCV)                                                                                          \
    hipLaunchKernelGGL((pcm_unpack_tile_kernel<B, TCV>), tgrid, dim3(256), 0, ctx->stream, p, dst, (long long)dst_pitch, \
                       (long long)frames, (long long)channels, (float)scale)
            if (sample_bytes == 2) {
                if (tc == 64) HD_PCM_TILE(2, 64); else if (tc == 32) HD_PCM_TILE(2, 32);
                else if (tc == 16) HD_PCM_TILE(2, 16); else HD_PCM_TILE(2, 8);
            } else {
                if (tc == 64) HD_PCM_TILE(4, 64); else if (tc == 32) HD_PCM_TILE(4, 32);
                else if (tc == 16) HD_PCM_TILE(4, 16); else if (tc == 8) HD_PCM_TILE(4, 8); else HD_PCM_TILE(4, 4);
            }
#undef HD_PCM_TILE
            return hd_launch_status("pcm_unpack_tile_kernel");
        }
    }
    dim3 grid((unsigned)((frames + 31) / 32), (unsigned)((channels + 31) / 32));
    switch (sample_bytes) {
    case 2: hipLaunchKernelGGL(pcm_unpack_kernel<2>, grid, dim3(256), 0, ctx->stream, p, dst, (long long)dst_pitch,
                               (long long)frames, (long long)channels, (float)scale); break;
    case 3: hipLaunchKernelGGL(pcm_unpack_kernel<3>, grid, dim3(256), 0, ctx->stream, p, dst, (long long)dst_pitch,
                               (long long)frames, (long long)channels, (float)scale); break;
    case 4: hipLaunchKernelGGL(pcm_unpack_kernel<4>, grid, dim3(256), 0, ctx->stream, p, dst, (long long)dst_pitch,
                               (long long)frames, (long long)channels, (float)scale); break;
    default:
        hipdsp_set_error("sample_bytes %d: signed PCM of 2, 3 or 4 bytes only", sample_bytes);
        return HIPDSP_ERR_UNSUPPORTED;
    }
    return hd_launch_status("pcm_unpack_kernel");
}

int hipdsp_minmax_decimate(hipdsp_ctx *ctx, const float *x, int64_t x_pitch, int64_t channels,
                           int64_t start, int64_t stop, int64_t step, float *out, int64_t out_pitch)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_REQUIRE(channels >= 0 && start >= 0 && stop >= start && step >= 1, "bad range");
    const long long nseg = (stop - start + step - 1) / step;
    if (channels == 0 || nseg == 0) return HIPDSP_OK;
    HD_REQUIRE(x != nullptr && out != nullptr, "NULL data pointer");
    HD_REQUIRE(x_pitch >= stop && out_pitch >= 2 * nseg, "pitch too small");
    HD_REQUIRE(channels <= 65535, "too many channels");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    if (step >= 512) {
        HD_REQUIRE((nseg + 3) / 4 <= 0x7fffffffLL, "grid too large");
        hipLaunchKernelGGL(minmax_long_kernel, dim3((unsigned)((nseg + 3) / 4), (unsigned)channels), dim3(256), 0, ctx->stream,
                           x, (long long)x_pitch, (long long)start, (long long)stop, (long long)step, nseg, out,
                           (long long)out_pitch);
        return hd_launch_status("minmax_long_kernel");
    }
    const int spt = MM_TILE / (int)step;                  // whole segments per 2048-sample tile (>= 4)
    const long long tiles = (nseg + spt - 1) / spt;
    HD_REQUIRE((tiles + 3) / 4 <= 0x7fffffffLL, "grid too large");
    dim3 grid((unsigned)((tiles + 3) / 4), (unsigned)channels), block(256);
#define HD_MINMAX(GLV)                                                                                            \
    hipLaunchKernelGGL(minmax_short_kernel<GLV>, grid, block, 0, ctx->stream, x, (long long)x_pitch, (long long)start, \
                       (long long)stop, (int)step, nseg, spt, out, (long long)out_pitch)
    if (step <= 32) HD_MINMAX(1);
    else if (step <= 128) HD_MINMAX(16);
    else HD_MINMAX(64);
#undef HD_MINMAX
    return hd_launch_status("minmax_short_kernel");
}

int hipdsp_mean_spectrum_db(hipdsp_ctx *ctx, const float *spec_tf, int64_t nfreq, int64_t i0, int64_t i1,
                            double ref_power, double min_power, double floor_db, float *out)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_REQUIRE(nfreq >= 1 && i0 >= 0 && i1 > i0, "bad range");
    HD_REQUIRE(ref_power > 0, "ref_power must be positive");
    HD_REQUIRE(spec_tf != nullptr && out != nullptr, "NULL data pointer");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    const long long n = i1 - i0;
    int nsplit = (int)(n < 64 ? n : 64);
    void *work = nullptr;
    int rc = hipdsp_scratch(ctx, sizeof(double) * (size_t)nsplit * (size_t)nfreq, &work);
    if (rc != HIPDSP_OK) return rc;
    unsigned gx = (unsigned)((nfreq + 255) / 256);
    hipLaunchKernelGGL(mean_spectrum_partial, dim3(gx, (unsigned)nsplit), dim3(256), 0, ctx->stream, spec_tf,
                       (long long)nfreq, (long long)i0, (long long)i1, nsplit, (double *)work);
    rc = hd_launch_status("mean_spectrum_partial");
    if (rc != HIPDSP_OK) return rc;
    hipLaunchKernelGGL(mean_spectrum_finish, dim3(gx), dim3(256), 0, ctx->stream, (const double *)work,
                       (long long)nfreq, nsplit, 1.0 / (double)n, (float)(1.0 / ref_power), (float)min_power,
                       (float)floor_db, out);
    return hd_launch_status("mean_spectrum_finish");
}

int hipdsp_copy_probe(hipdsp_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    HD_REQUIRE(ctx != nullptr && dst != nullptr && src != nullptr, "NULL argument");
    HD_REQUIRE(bytes % 16 == 0 && bytes / 16 / 256 < 0x7fffffffULL, "bytes must be a multiple of 16 (and below 8 TiB)");
    if (bytes == 0) return HIPDSP_OK;
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    const long long n4 = (long long)(bytes / 16);
    hipLaunchKernelGGL(copy_probe_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const float4 *)src, (float4 *)dst, n4);
    return hd_launch_status("copy_probe_kernel");
}

int hipdsp_synth(hipdsp_ctx *ctx, float *x, int64_t x_pitch, int64_t channels, int64_t frames, double rate,
                 uint64_t seed, int64_t c0, int64_t c_total)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_REQUIRE(frames >= 0 && channels >= 0 && c_total > 0 && rate > 0, "bad size");
    if (frames == 0 || channels == 0) return HIPDSP_OK;
    HD_REQUIRE(x != nullptr && x_pitch >= frames, "bad output");
    HD_REQUIRE(channels <= 65535, "too many channels");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(synth_kernel, dim3(grid1d(frames, 1024, 4096), (unsigned)channels), dim3(256), 0,
                       ctx->stream, x, (long long)x_pitch, (long long)frames, rate,
                       (unsigned long long)seed, (long long)c0, (long long)c_total);
    return hd_launch_status("synth_kernel");
}

}  // extern "C"
