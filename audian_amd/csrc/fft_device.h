// fft_device.h -- device-side building blocks of the PSD kernels (csrc/spectrogram.hip) that the
// fused forward kernel of the chain (csrc/sos.hip) uses as well: complex helpers in packed fp32,
// in-register DFTs of radix 2...32, one Stockham stage over a padded LDS frame buffer, the DPP wave
// sum and the decibel conversion.  Everything is __forceinline__ and lives in an anonymous
// namespace of the including translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <type_traits>

namespace {

constexpr float DB_MIN_POWER = 1e-20f;      // thunderlab decibel default min_power

__device__ __forceinline__ float to_db(float p)
{
    return (p <= DB_MIN_POWER) ? -INFINITY : 10.0f * log10f(p);
}

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

// Sum over the 64 lanes of a wave, result in every lane: DPP adds inside the VALU (quad
// permutes, row mirrors, row broadcasts) instead of six dependent trips through the LDS
// crossbar (ds_bpermute), whose latency sat in front of every frame.
__device__ __forceinline__ float wave_sum(float v)
{
    auto step = [](float x, auto ctrl, auto row_mask) {
        const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value,
                                                  decltype(row_mask)::value, 0xf, false);
        return x + __int_as_float(t);
    };
    using I = std::integral_constant<int, 0>;
    (void)sizeof(I);
    v = step(v, std::integral_constant<int, 0xB1>(), std::integral_constant<int, 0xf>());    // quad_perm [1,0,3,2]
    v = step(v, std::integral_constant<int, 0x4E>(), std::integral_constant<int, 0xf>());    // quad_perm [2,3,0,1]
    v = step(v, std::integral_constant<int, 0x141>(), std::integral_constant<int, 0xf>());   // row_half_mirror
    v = step(v, std::integral_constant<int, 0x140>(), std::integral_constant<int, 0xf>());   // row_mirror
    v = step(v, std::integral_constant<int, 0x142>(), std::integral_constant<int, 0xa>());   // row_bcast:15 -> rows 1, 3
    v = step(v, std::integral_constant<int, 0x143>(), std::integral_constant<int, 0xc>());   // row_bcast:31 -> rows 2, 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

typedef float v2f __attribute__((ext_vector_type(2)));

// Packed-fp32 complex helpers (v_pk_*_f32 computes two lanes per instruction; op_sel picks the
// low/high dword of a source pair for the low lane, op_sel_hi for the high lane, neg_lo/neg_hi
// negate a source per lane).  hipcc's SLP vectoriser finds the packed form inside the small
// in-register DFTs but not across the twiddle tables and the split step of the big kernel.
__device__ __forceinline__ v2f as_v2f(float2 a) { v2f r = {a.x, a.y}; return r; }
__device__ __forceinline__ float2 as_f2(v2f a) { return make_float2(a.x, a.y); }
// a * b: same roundings as cmul() (product, then fma)
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f b)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(b));  // (-ay by, ay bx)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));   // (ax bx + t.x, ax by + t.y)
    return r;
}
// (-i a) * b = (ay bx + ax by, ay by - ax bx)
__device__ __forceinline__ v2f pk_cmul_negi(v2f a, v2f b)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(t) : "v"(a), "v"(b));               // (ay bx, ay by)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
__device__ __forceinline__ v2f pk_add_conj(v2f a, v2f b)      // a + conj(b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_sub_conj(v2f a, v2f b)      // a - conj(b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (a.x + b.x, a.x - b.x) and (a.y + b.y, a.y - b.y)
__device__ __forceinline__ v2f pk_sumdiff_x(v2f a, v2f b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_sumdiff_y(v2f a, v2f b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// 8-byte global load the compiler does not track (the caller counts vmcnt by hand)
// The destination is a tied operand ("+v"): the load lands in the very registers that held the
// previous frame's samples, so the value the loop carries never changes registers (a plain
// "=v" output lets hipcc pick fresh ones and copy them on the back edge -- while in flight).
__device__ __forceinline__ void asm_load8(v2f &r, const float *p, int imm)
{
    asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "+v"(r) : "v"(p), "n"(imm) : "memory");
}

// One Stockham stage on the PPL register values of this lane.
//   butterfly j = l + LPF*u reads in[j + t*M/R], twiddles by W^(k t), k = j % NS,
//   and writes out[(j/NS)*NS*R + k + t*NS].
// `tw` is this stage's own table, tw[(t-1)*NS + k] = exp(-2 pi i k t / (NS R)): lanes
// with consecutive k read consecutive entries (no LDS bank conflicts).
template <int R, int NS, int M, int LPF, bool LOAD, bool STORE, bool POWERS = false, bool COMPUTE = true>
__device__ __forceinline__ void stockham_stage(float2 *v, float2 *fb, const float2 *tw, int l)
{
    constexpr int PPL = M / LPF;
    constexpr int NB = PPL / R;          // butterflies per lane
    // Padded frame-buffer indices as (per-lane part) + (compile-time part): pad16(a + c) =
    // pad16(a) + c + c/16 whenever c is a multiple of 16, so the constant goes into the DS
    // instruction's offset field instead of a shift and two adds per access.
    constexpr bool SPLIT_LOAD = LPF % 16 == 0 && (M / R) % 16 == 0;
    constexpr bool STORE_R16 = NS == 1 && R == 16;                       // index = 17 j + t
    constexpr bool STORE_NS16 = NS % 16 == 0 && LPF % NS == 0;           // k, j/NS split by lane
    // all loads of the stage come before any store: the exchange is in place and one
    // butterfly's outputs land on another butterfly's inputs
    if (LOAD) {
        const int pl = pad16(l);
#pragma unroll
        for (int u = 0; u < NB; u++)
#pragma unroll
            for (int t = 0; t < R; t++) {
                const int c = LPF * u + t * (M / R);
                v[u * R + t] = fb[SPLIT_LOAD ? pl + c + c / 16 : pad16(l + c)];
            }
    }
#pragma unroll
    for (int u = 0; u < (COMPUTE ? NB : 0); u++) {
        const int j = l + LPF * u;
        float2 *b = v + u * R;
        if (NS > 1) {
            const int k = j % NS;
            if (POWERS) {            // table holds W^k only; W^(k t) by repeated multiplication
                const v2f w1 = as_v2f(tw[k]);
                v2f w = w1;
#pragma unroll
                for (int t = 1; t < R; t++) {
                    b[t] = as_f2(pk_cmul(as_v2f(b[t]), w));
                    if (t + 1 < R) w = pk_cmul(w, w1);
                }
            } else {
#pragma unroll
                for (int t = 1; t < R; t++) b[t] = as_f2(pk_cmul(as_v2f(b[t]), as_v2f(tw[(t - 1) * NS + k])));
            }
        }
        dft<R>(b);
    }
    if (STORE) {
        int sl = 0;                      // per-lane part of the store index
        if (STORE_R16) sl = 17 * l;
        else if (STORE_NS16) sl = (l / NS) * (NS * R + NS * R / 16) + pad16(l % NS);
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const int j = l + LPF * u;
            const int k = j % NS;
            const int base = (j / NS) * NS * R + k;
#pragma unroll
            for (int t = 0; t < R; t++) {
                int idx;
                if (STORE_R16) idx = sl + 17 * LPF * u + t;
                else if (STORE_NS16) idx = sl + u * (LPF / NS) * (NS * R + NS * R / 16) + t * (NS + NS / 16);
                else idx = pad16(base + t * NS);
                fb[idx] = v[u * R + t];
            }
        }
    }
}


template <> __device__ __forceinline__ void dft<32>(float2 *v)
{
    // radix-2 decimation in time over two 16-point transforms; W32^k = exp(-i pi k / 16)
    const float c[16] = {1.f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                         0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f,
                         0.19509032201612826785f, 0.f, -0.19509032201612826785f, -0.38268343236508977173f,
                         -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f,
                         -0.92387953251128675613f, -0.98078528040323044913f};
    const float sn[16] = {0.f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
                          0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f,
                          0.98078528040323044913f, 1.f, 0.98078528040323044913f, 0.92387953251128675613f,
                          0.83146961230254523708f, 0.70710678118654752440f, 0.55557023301960222474f,
                          0.38268343236508977173f, 0.19509032201612826785f};
    float2 e[16], o[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { e[k] = v[2 * k]; o[k] = v[2 * k + 1]; }
    dft<16>(e); dft<16>(o);
#pragma unroll
    for (int k = 1; k < 16; k++) {
        if (k == 8) o[k] = mul_negi(o[k]);
        else o[k] = cmul(o[k], make_float2(c[k], -sn[k]));
    }
#pragma unroll
    for (int k = 0; k < 16; k++) { v[k] = cadd(e[k], o[k]); v[k + 16] = csub(e[k], o[k]); }
}


}  // namespace
