// The spectrogram kernel's memory pattern without its arithmetic (nfft 2048, hop 1024, 64 ch x 600 s):
// per frame a wave reads 1024 new samples (8 x 8 B per lane) and writes 1025 bins either as the
// kernel does (17 x 4 B per lane, bins k and M-k) or as 16-byte stores of consecutive bins.
//   hipcc -O3 --offload-arch=gfx950 tools/spec_io_probe.hip -o /tmp/spec_io_probe && /tmp/spec_io_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

template <int MODE>
__global__ __launch_bounds__(256) void io(const float *__restrict__ x, long long x_pitch, float *__restrict__ out,
                                          long long out_pitch, long long frames, int fpw)
{
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const long long ch = blockIdx.y;
    const float *xc = x + ch * x_pitch;
    float *oc = out + ch * out_pitch;
    long long f0 = ((long long)blockIdx.x * 4 + wave) * fpw;
    for (int it = 0; it < fpw; it++) {
        const long long f = f0 + it;
        if (f >= frames) break;
        const float *seg = xc + f * 1024 + 1024;          // the new half of the frame
        float2 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = *reinterpret_cast<const float2 *>(seg + 2 * (l + 64 * k));
        float *o = oc + f * 1025;
        if (MODE == 0) {
#pragma unroll
            for (int m = 0; m < 8; m++) {
                o[l + 64 * m] = v[m].x;
                o[1024 - (l + 64 * m)] = v[m].y;
            }
            o[l == 0 ? 512 : l + 448] = v[0].x + v[1].y;
        } else {
            // 1025 floats as 4 x (64 lanes x 16 B) + one 4-byte tail
#pragma unroll
            for (int m = 0; m < 4; m++) {
                f4u t; t.x = v[2 * m].x; t.y = v[2 * m].y; t.z = v[2 * m + 1].x; t.w = v[2 * m + 1].y;
                *reinterpret_cast<f4u *>(o + 4 * l + 256 * m) = t;
            }
            if (l == 0) o[1024] = v[0].x;
        }
    }
}

int main()
{
    const long long C = 64, T = 57600000, frames = 56249, F = 1025;
    float *x, *out;
    CK(hipMalloc(&x, C * T * 4)); CK(hipMalloc(&out, C * (frames + 1) * F * 4));
    CK(hipMemset(x, 0, C * T * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int fpw : {8, 16, 32}) {
        dim3 grid((unsigned)((frames + 4 * fpw - 1) / (4 * fpw)), (unsigned)C);
        for (int mode = 0; mode < 2; mode++) {
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0));
                if (mode == 0) io<0><<<grid, 256>>>(x, T, out, (frames + 1) * F, frames, fpw);
                else io<1><<<grid, 256>>>(x, T, out, (frames + 1) * F, frames, fpw);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            double gb = (4.0 * C * T + 4.0 * C * frames * F) / 1e9;
            printf("frames/wave %2d  %s stores: %.3f ms  %.0f GB/s\n", fpw, mode == 0 ? "4-byte (kernel pattern)" : "16-byte consecutive   ", ms, gb / ms * 1e3);
        }
    }
    return 0;
}
