// fft_device.h -- device-side building blocks of the PSD kernels (csrc/spectrogram.hip) that the
// fused forward kernel of the chain (csrc/sos.hip) uses as well: complex helpers in packed fp32,
// in-register DFTs of radix 2...32, one Stockham stage over a padded LDS frame buffer, the DPP wave
// sum and the decibel conversion.  Everything is __forceinline__ and lives in an anonymous
// namespace of the including translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <type_traits>

#pragma clang fp contract(fast)
namespace {

constexpr float DB_MIN_POWER = 1e-20f;      // thunderlab decibel default min_power

// 10 log10(p) as 10 log10(2) * v_log_f32(p): four VALU instructions per bin against the fourteen of log10f(), whose
// denormal scaling cannot trigger behind the floor (p > 1e-20) and whose two-part multiplication by log10(2) buys one unit
// in the last place -- 1.5e-5 dB at -200 dB, where the reference's float64 value is 0.01 dB wide at the parity tolerance.
// The fused dB epilogues are VALU-bound (770 against 480 instructions per 2048-frame with log10f, tools/spec_sq.sh); the
// stand-alone hipdsp_decibel (elementwise.hip) runs at the copy rate either way and keeps log10f.
__device__ __forceinline__ float to_db(float p)
{
    return (p <= DB_MIN_POWER) ? -INFINITY : 3.01029995663981195f * __builtin_amdgcn_logf(p);
}

typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 mul_negi(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f as_v2f(float2 a) { v2f r = {a.x, a.y}; return r; }
__device__ __forceinline__ float2 as_f2(v2f a) { return make_float2(a.x, a.y); }

// In-register DFTs in explicit packed fp32: a complex value is one aligned register pair, an
// addition one v_pk_add_f32, and the rotations by -i / +i ride on the op_sel / neg modifiers of the
// addition that consumes them (written with scalar float2 arithmetic, hipcc's SLP vectoriser found
// the packed adds too, but paid for them with ~25 register moves per 16-point transform).
__device__ __forceinline__ v2f pk_add_negi(v2f a, v2f b)      // a + (-i) b = (ax + by, ay - bx)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_sub_negi(v2f a, v2f b)      // a - (-i) b = (ax - by, ay + bx)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a * (c - i s) for a constant twiddle: c a + s (-i a) = (c ax + s ay, c ay - s ax)
__device__ __forceinline__ v2f pk_rot(v2f a, float c, float sn)
{
    const v2f cc = {c, c}, ss = {sn, sn};
    v2f t = a * cc, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(ss), "v"(t));
    return r;
}

__device__ __forceinline__ void dft4p(v2f &v0, v2f &v1, v2f &v2, v2f &v3)
{
    const v2f t0 = v0 + v2, t1 = v0 - v2, t2 = v1 + v3, d = v1 - v3;
    v0 = t0 + t2; v2 = t0 - t2;
    v1 = pk_add_negi(t1, d); v3 = pk_sub_negi(t1, d);
}

__device__ __forceinline__ void dft8p(v2f *v)                 // in place, natural order
{
    const float h = 0.70710678118654752440f;
    const v2f hh = {h, h};
    v2f e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4p(e0, e1, e2, e3);
    dft4p(o0, o1, o2, o3);
    const v2f p1 = pk_add_negi(o1, o1) * hh;                  // o1 * W8^1 = h (o1 + (-i) o1)
    const v2f p3 = pk_sub_negi(o3, o3) * hh;                  // o3 * W8^3 = -h (o3 - (-i) o3)
    v[0] = e0 + o0; v[4] = e0 - o0;
    v[1] = e1 + p1; v[5] = e1 - p1;
    v[2] = pk_add_negi(e2, o2); v[6] = pk_sub_negi(e2, o2);   // o2 * W8^2 = -i o2
    v[3] = e3 - p3; v[7] = e3 + p3;
}

__device__ __forceinline__ void dft16p(v2f *v)
{
    const float h = 0.70710678118654752440f;
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;   // cos, sin(pi/8)
    const v2f hh = {h, h};
    v2f e[8], o[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { e[k] = v[2 * k]; o[k] = v[2 * k + 1]; }
    dft8p(e); dft8p(o);
    const v2f q1 = pk_rot(o[1], c1, s1);                      // W16^1 = c1 - i s1
    const v2f q2 = pk_add_negi(o[2], o[2]) * hh;              // W16^2 = W8^1
    const v2f q3 = pk_rot(o[3], s1, c1);                      // W16^3 = s1 - i c1
    const v2f q5 = pk_rot(o[5], -s1, c1);                     // W16^5 = -s1 - i c1
    const v2f q6 = pk_sub_negi(o[6], o[6]) * hh;              // W16^6 = W8^3 = -h (1 + i): subtract below
    const v2f q7 = pk_rot(o[7], -c1, s1);                     // W16^7 = -c1 - i s1
    v[0] = e[0] + o[0]; v[8] = e[0] - o[0];
    v[1] = e[1] + q1;   v[9] = e[1] - q1;
    v[2] = e[2] + q2;   v[10] = e[2] - q2;
    v[3] = e[3] + q3;   v[11] = e[3] - q3;
    v[4] = pk_add_negi(e[4], o[4]); v[12] = pk_sub_negi(e[4], o[4]);   // W16^4 = -i
    v[5] = e[5] + q5;   v[13] = e[5] - q5;
    v[6] = e[6] - q6;   v[14] = e[6] + q6;
    v[7] = e[7] + q7;   v[15] = e[7] - q7;
}

template <int R> __device__ __forceinline__ void dft(float2 *v);

template <> __device__ __forceinline__ void dft<2>(float2 *v)
{
    float2 a = v[0], b = v[1];
    v[0] = cadd(a, b); v[1] = csub(a, b);
}

template <> __device__ __forceinline__ void dft<4>(float2 *v)
{
    v2f a = as_v2f(v[0]), b = as_v2f(v[1]), c = as_v2f(v[2]), d = as_v2f(v[3]);
    dft4p(a, b, c, d);
    v[0] = as_f2(a); v[1] = as_f2(b); v[2] = as_f2(c); v[3] = as_f2(d);
}

template <> __device__ __forceinline__ void dft<8>(float2 *v)
{
    v2f t[8];
#pragma unroll
    for (int k = 0; k < 8; k++) t[k] = as_v2f(v[k]);
    dft8p(t);
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = as_f2(t[k]);
}

template <> __device__ __forceinline__ void dft<16>(float2 *v)
{
    v2f t[16];
#pragma unroll
    for (int k = 0; k < 16; k++) t[k] = as_v2f(v[k]);
    dft16p(t);
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = as_f2(t[k]);
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

// Packed-fp32 complex helpers (v_pk_*_f32 computes two lanes per instruction; op_sel picks the
// low/high dword of a source pair for the low lane, op_sel_hi for the high lane, neg_lo/neg_hi
// negate a source per lane).  hipcc's SLP vectoriser finds the packed form inside the small
// in-register DFTs but not across the twiddle tables and the split step of the big kernel.
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
    // pad16 of a store index as (per-lane part) + (compile-time part) wherever the digits allow it, so that a stage's
    // stores are ONE address register and the offset fields of the DS instructions.  Left to hipcc, the 16-lane frames of
    // nfft 256 kept six address registers alive across the tile loop of the fused sweep -- and reloaded them from scratch
    // in every frame, with a full vmcnt(0) each (round 5: 14.3 instead of 11.1 ms).
    //   NS == 1, R == 8:  pad16(8 j + t) = 8 j + j / 2 + t                                       (t < 8)
    //   NS | 16 or 16 | NS, NS | LPF, 16 | NS R:  k = l % NS and j / NS = l / NS + u LPF / NS, and the pad of
    //   (k + t NS) is that of t NS alone (k < NS does not reach the next multiple of NS)
    // (STOCKHAM_SPLIT_MORE: the fused sweep's translation units only -- in the stand-alone PSD kernels the two new forms
    // bought 2.8 % without the dB image and cost 12 % with it at nfft 1024, profiles/r05w_entry_points_gate.log: their
    // register allocation is not this header's to disturb)
#ifdef STOCKHAM_SPLIT_MORE
    constexpr bool STORE_R8 = NS == 1 && R == 8 && LPF % 2 == 0;
    constexpr bool STORE_NS16 = NS > 1 && (NS % 16 == 0 || 16 % NS == 0) && LPF % NS == 0 && (NS * R) % 16 == 0;
#else
    constexpr bool STORE_R8 = false;
    constexpr bool STORE_NS16 = NS % 16 == 0 && LPF % NS == 0;           // k, j/NS split by lane
#endif
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
#ifdef STOCKHAM_LOADS_FIRST
        // (the fused sweep's FFT role: hipcc's scheduler, short of registers next to the IIR role, otherwise issues these
        // reads one at a time, each behind the multiplication of the value before it -- a full LDS round trip per value)
        __builtin_amdgcn_sched_barrier(0);
#endif
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
        else if (STORE_R8) sl = 8 * l + l / 2;
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
                else if (STORE_R8) idx = sl + u * (8 * LPF + LPF / 2) + t;
                else if (STORE_NS16) idx = sl + u * (LPF / NS) * (NS * R + NS * R / 16) + t * NS + (t * NS) / 16;
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
    v2f e[16], o[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { e[k] = as_v2f(v[2 * k]); o[k] = as_v2f(v[2 * k + 1]); }
    dft16p(e); dft16p(o);
    v[0] = as_f2(e[0] + o[0]); v[16] = as_f2(e[0] - o[0]);
#pragma unroll
    for (int k = 1; k < 16; k++) {
        if (k == 8) {                                              // W32^8 = -i
            v[k] = as_f2(pk_add_negi(e[k], o[k])); v[k + 16] = as_f2(pk_sub_negi(e[k], o[k]));
        } else {
            const v2f q = pk_rot(o[k], c[k], sn[k]);               // o * (c - i s)
            v[k] = as_f2(e[k] + q); v[k + 16] = as_f2(e[k] - q);
        }
    }
}


}  // namespace
#pragma clang fp contract(off)
