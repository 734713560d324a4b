// envsplit.hip -- the envelope's backward sweep with the roles split (experiment of round 4, context option "sos_split";
// BufferedEnvelope.process, src/audian/bufferedenvelope.py:34-41: the backward half of scipy's sosfiltfilt).
//
// env_bwd_kernel's waves do everything themselves: fetch a tile, recompute the forward cascade, run the backward one,
// store the tile.  Measured (profiles/r02_sweep_trace.log, r03e_bwd_ablate.log): 34 % of a wave's clocks pass while it
// tries to ISSUE its nine loads and eight stores, its arithmetic alone needs 4.2 ms of the sweep's 5.5-6.2 at BASELINE
// configs[2], its bytes 4.7 ms at the copy rate -- the waves that compute are the waves that stall on the memory pipeline.
// Here a workgroup (one per CU) is eight COMPUTE waves, two per SIMD, that never issue a vector-memory instruction for an
// interior tile, and four MOVER waves, one per SIMD, that do nothing else: a mover serves two compute waves, each of
// which owns two tile buffers in LDS (8 x 2 x 8 KB); it fetches tile i + 2 into the buffer tile i has just left,
// and stores tile i once the compute wave has written the envelope back into its buffer.  Hand-over by three monotonic
// counters per compute wave in LDS (filled, done; the mover keeps `drained` to itself), bounded polling with the
// context's fault word as in chain_fwd_kernel.  Tiles that are not interior (they touch T, `skip`, the envelope's first
// tile, the front of a shifted grid) keep env_bwd_kernel's own slow paths inside the compute wave; the mover only hands
// their buffers over.  Same arithmetic as env_bwd_kernel<SE, false, PIN, false, WPB, false> (the forward outputs pass
// through LDS as float32: with 12 waves per CU a wave has 168 registers, the register hand-over needs 190).
#include "sos_device.h"

namespace {

constexpr int NCW = 8, NMW = 4;

#define SPLIT_WAIT_FOR(word, want, what)                                                          \
    do {                                                                                          \
        if (!gave_up) {                                                                           \
            for (int spin_ = 0;; spin_++) {                                                       \
                if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&(word), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >= (want)) break; \
                if ((spin_ & 1023) == 1023) {                                                     \
                    const bool told = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&abort_wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0; \
                    if (told || spin_ >= (1 << 23) - 1) {                                         \
                        gave_up = true;                                                           \
                        if (!told && lane == 0) {                                                 \
                            __hip_atomic_store(&abort_wg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
                            a.fault[1] = (int)blockIdx.x; a.fault[2] = wave; a.fault[3] = (what); \
                            __hip_atomic_store(&a.fault[0], HD_FAULT_SPLIT_HANDOVER, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); \
                        }                                                                         \
                        break;                                                                    \
                    }                                                                             \
                }                                                                                 \
                __builtin_amdgcn_s_sleep(1);                                                      \
            }                                                                                     \
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");                      \
        }                                                                                         \
    } while (0)
#define SPLIT_POST(word, value)                                                                   \
    do {                                                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");                          \
        if (lane == 0) __hip_atomic_store(&(word), (value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
    } while (0)

struct SplitArgs {
    BwdArgs b;
    int *fault;
};

// the tiles a unit (channel, segment) walks, last sample first, as both roles see them
struct UnitWalk {
    long long ch, rt_lo, rt_start;
    int n;                       // tiles visited
    bool dead;                   // the channel's forward sweep ended non-finite: NaN everywhere, no sweep
};

template <int DE>
__device__ __forceinline__ UnitWalk unit_walk(const BwdArgs &a, const double *__restrict__ ckpt_all, long long unit)
{
    UnitWalk u;
    u.n = 0; u.dead = false; u.ch = 0; u.rt_lo = 0; u.rt_start = 0;
    if (unit >= a.units) return u;
    const int seg = (int)(unit % a.n_seg);
    u.ch = unit / a.n_seg;
    u.rt_lo = (long long)seg * a.seg_tiles;
    long long rt_hi = u.rt_lo + a.seg_tiles;
    if (rt_hi > a.n_tiles) rt_hi = a.n_tiles;
    u.rt_start = u.rt_lo - a.warm_tiles;
    if (u.rt_start < 0) u.rt_start = 0;
    const long long rt_stop = a.n_tiles - a.skip / TILE;            // tiles below `skip` are never visited
    long long end = rt_hi < rt_stop ? rt_hi : rt_stop;
    u.n = end > u.rt_start ? (int)(end - u.rt_start) : 0;
    const double *e = ckpt_all + u.ch * a.ckpt_pitch + a.n_tiles * DE;
    bool bad = false;
#pragma unroll
    for (int r = 0; r < DE; r++) bad = bad || !(fabs(e[r]) <= 1.7976931348623157e308);
    u.dead = bad;
    return u;
}

template <int SE>
__global__ __launch_bounds__(64 * (NCW + NMW)) void env_bwd_split_kernel(const SosPlanDev *__restrict__ P0,
                                                                         const double *__restrict__ ckpt_all, SplitArgs a)
{
    constexpr int DE = 2 * SE;
    __shared__ float4 tiles[NCW][2][64 * 8];
    __shared__ int filled[NCW], done[NCW];
    __shared__ int abort_wg;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < NCW) { filled[threadIdx.x] = 0; done[threadIdx.x] = 0; }
    if (threadIdx.x == 0) abort_wg = 0;
    __syncthreads();
    bool gave_up = false;
    const long long T = a.b.T;
    const int edge = a.b.edge;
    const long long top_full = T / TILE - 1;                     // host guarantees >= 0
    const long long env_tile0 = a.b.env0 > 0 ? a.b.env0 - a.b.env0 % TILE : -1;
    auto load_fast = [&](long long tidx) {
        return tidx >= 0 && tidx <= top_full && tidx * TILE >= a.b.lead && tidx * TILE != env_tile0;
    };

    if (wave < NCW) {
        // ================= compute role ==============================================================
        const int cw = wave;
        const UnitWalk u = unit_walk<DE>(a.b, ckpt_all, (long long)blockIdx.x * NCW + cw);
        const float *in = a.b.in + u.ch * a.b.in_pitch;
        float *out = a.b.out + u.ch * a.b.out_pitch;
        const double *ckpt = ckpt_all + u.ch * a.b.ckpt_pitch;
        if (u.dead) {
            // NaN everywhere in this channel (sos_device.h: FloodArgs): the unit fills its share itself, the mover skips it
            long long rt_hi = u.rt_lo + a.b.seg_tiles;
            if (rt_hi > a.b.n_tiles) rt_hi = a.b.n_tiles;
            long long p_lo = (a.b.n_tiles - rt_hi) * TILE, p_hi = (a.b.n_tiles - u.rt_lo) * TILE;
            if (p_lo < a.b.skip) p_lo = a.b.skip;
            if (p_hi > T) p_hi = T;
            for (long long p = p_lo + lane; p < p_hi; p += 64) out[p - a.b.skip] = __builtin_nanf("");
            return;
        }
        double cb_[DE];
#pragma unroll
        for (int r = 0; r < DE; r++) cb_[r] = 0.0;
        for (int i = 0; i < u.n; i++) {
            const long long rt = u.rt_start + i;
            const long long tidx = a.b.n_tiles - 1 - rt;
            const long long tile = tidx * TILE;
            float4 *lds = tiles[cw][i & 1];
            float *ldsf = reinterpret_cast<float *>(lds);
            SPLIT_WAIT_FOR(filled[cw], i + 1, i);
            if (!load_fast(tidx)) {
                // a tile that touches T, holds the `lead` samples in front of the trace or the envelope's first sample:
                // env_bwd_kernel's own path (the mover has only handed the buffer over)
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    float4 v = load_four_from(in, tile + 256 * k + 4 * lane, a.b.lead, T);
                    if (a.b.rectify) v = make_float4(fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w));
                    lds[lds_slot(8 * k + (lane >> 3), lane & 7)] = v;
                }
                WAVE_SYNC();
                if (tile + TILE > T) {
                    // right odd extension ext[T + i] = 2 r(T-1) - r(T-2-i), i < edge, straight from HBM
                    if (lane < edge) {
                        const long long pj = T + lane;
                        if (pj >= tile && pj < tile + TILE) {
                            float ra = in[T - 1], rb = in[T - 2 - lane];
                            if (a.b.rectify) { ra = fabsf(ra); rb = fabsf(rb); }
                            ldsf[lds_float_index((int)(pj - tile))] = 2.f * ra - rb;
                        }
                    }
                    WAVE_SYNC();
                }
                if (tile == env_tile0) (void)env_left_fill(ldsf, lane, (int)(a.b.env0 - tile), edge);
            }
            double cfw_[DE];
#pragma unroll
            for (int r = 0; r < DE; r++) cfw_[r] = ckpt[tidx * DE + r];
            // ---- forward cascade again, from the state that entered this tile (a tile the mover fetched is raw: the
            // rectification rides on the cascade's input; a tile of the slow path is rectified already and may hold
            // odd-extension values, which can be negative and must stay so)
            const bool rect_in = a.b.rectify && load_fast(tidx);
#define CASC_S SE
#define CASC_PLAN() PLAN_OF(P0)
#define CASC_IN(v) (rect_in ? fabsf(v) : (v))
#define CASC_PIN_GROUPS true
#define CASC_CARRY cfw_
#define CASC_GAIN a.b.gain
#include "sos_cascade.inc"
#undef CASC_GAIN
#undef CASC_CARRY
#undef CASC_IN
            WAVE_SYNC();
            if (rt == 0) {
                const int last = (int)(T + edge - 1 - tile);
                const float v0 = ldsf[lds_float_index(last)];
                WAVE_SYNC();
                for (int s2 = last + 1 + lane; s2 < TILE; s2 += 64) ldsf[lds_float_index(s2)] = v0;
                const SosPlanDev *P = PLAN_OF(P0);
#pragma unroll
                for (int r = 0; r < DE; r++) cb_[r] = P->zi[r] * (double)v0;
                WAVE_SYNC();
            }
            // ---- backward cascade over the forward outputs, last sample first; the clamp on the way into the tile
#define CASC_IN(v) (v)
#define CASC_CARRY cb_
#define CASC_REVERSED
#define CASC_TAP(j, e, y) do { if (a.b.clamp) e = max_zero(e); } while (0)
#include "sos_cascade.inc"
#undef CASC_TAP
#undef CASC_REVERSED
#undef CASC_CARRY
#undef CASC_IN
#undef CASC_PIN_GROUPS
#undef CASC_PLAN
#undef CASC_S
            WAVE_SYNC();
            if (rt >= u.rt_lo && !(tile >= a.b.skip && tile + TILE <= T)) {
                // a border tile of the output: stored here, element by element (the mover skips it)
#pragma unroll
                for (int k = 0; k < 8; k++)
                    store_four(out, tile + 256 * k + 4 * lane, lds[lds_slot(8 * k + (lane >> 3), lane & 7)], a.b.skip, T, a.b.skip);
                WAVE_SYNC();
            }
            SPLIT_POST(done[cw], i + 1);
        }
    } else {
        // ================= mover role ================================================================
        __builtin_amdgcn_s_setprio(3);
        const int mw = wave - NCW;
        UnitWalk u[2];
        const float *in[2];
        float *out[2];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            u[s] = unit_walk<DE>(a.b, ckpt_all, (long long)blockIdx.x * NCW + 2 * mw + s);
            if (u[s].dead) u[s].n = 0;
            in[s] = a.b.in + u[s].ch * a.b.in_pitch;
            out[s] = a.b.out + u[s].ch * a.b.out_pitch;
        }
        v4f r[2][8];
        bool pend[2] = {false, false};
        int i_fill[2] = {0, 0}, i_drain[2] = {0, 0};
        int idle = 0;
        while (i_drain[0] < u[0].n || i_drain[1] < u[1].n) {
            bool progress = false;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int cw = 2 * mw + s;
                if (pend[s]) {
                    // the fetch requested a round ago: into the buffer, then the compute wave may have it
                    float4 *lds = tiles[cw][(i_fill[s] - 1) & 1];
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        lds[lds_slot(8 * k + (lane >> 3), lane & 7)] = make_float4(r[s][k].x, r[s][k].y, r[s][k].z, r[s][k].w);
                    SPLIT_POST(filled[cw], i_fill[s]);
                    pend[s] = false;
                    progress = true;
                }
                if (i_drain[s] < u[s].n &&
                    __builtin_amdgcn_readfirstlane(__hip_atomic_load(&done[cw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) > i_drain[s]) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
                    const long long rt = u[s].rt_start + i_drain[s];
                    const long long tile = (a.b.n_tiles - 1 - rt) * TILE;
                    if (rt >= u[s].rt_lo && tile >= a.b.skip && tile + TILE <= T && !(a.b.debug & 1)) {   // ("sos_debug" 1: no stores)
                        const float4 *lds = tiles[cw][i_drain[s] & 1];
                        float4 v[8];
#pragma unroll
                        for (int k = 0; k < 8; k++) v[k] = lds[lds_slot(8 * k + (lane >> 3), lane & 7)];
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            f4u t; t.x = v[k].x; t.y = v[k].y; t.z = v[k].z; t.w = v[k].w;
                            *reinterpret_cast<f4u *>(out[s] + (tile + 256 * k + 4 * lane - a.b.skip)) = t;
                        }
                    }
                    i_drain[s]++;
                    progress = true;
                }
                if (i_fill[s] < u[s].n && i_fill[s] < i_drain[s] + 2) {
                    const long long tidx = a.b.n_tiles - 1 - (u[s].rt_start + i_fill[s]);
                    if (load_fast(tidx) && (a.b.debug & 2)) {        // ("sos_debug" 2: no fetches, whatever the buffer holds)
                        i_fill[s]++;
                        SPLIT_POST(filled[cw], i_fill[s]);
                    } else if (load_fast(tidx)) {
                        const float *src = in[s] + tidx * TILE;
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const f4u t = *reinterpret_cast<const f4u *>(src + 256 * k + 4 * lane);
                            r[s][k] = (v4f){t.x, t.y, t.z, t.w};
                        }
                        pend[s] = true;
                        i_fill[s]++;
                    } else {
                        i_fill[s]++;
                        SPLIT_POST(filled[cw], i_fill[s]);           // the compute wave fetches this one itself
                    }
                    progress = true;
                }
            }
            if (progress) {
                idle = 0;
            } else {
                __builtin_amdgcn_s_sleep(2);
                if ((++idle & 1023) == 0) {
                    const bool told = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&abort_wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0;
                    if (told || idle >= (1 << 23)) {
                        if (!told && lane == 0) {
                            __hip_atomic_store(&abort_wg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            a.fault[1] = (int)blockIdx.x; a.fault[2] = wave; a.fault[3] = i_drain[0];
                            __hip_atomic_store(&a.fault[0], HD_FAULT_SPLIT_HANDOVER, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                        break;
                    }
                }
            }
        }
    }
    (void)gave_up;
}

}  // namespace

int hd_launch_env_bwd_split(hipdsp_ctx *ctx, const SosPlanDev *edev, int SE, const BwdArgs &b)
{
    SplitArgs a;
    a.b = b;
    a.fault = ctx->fault_dev;
    const long long blocks = (b.units + NCW - 1) / NCW;
    HD_REQUIRE(blocks <= 0x7fffffffLL, "grid too large");
    const dim3 grid((unsigned)blocks), block(64 * (NCW + NMW));
    if (SE == 1) hipLaunchKernelGGL((env_bwd_split_kernel<1>), grid, block, 0, ctx->stream, edev, b.ckpt, a);
    else if (SE == 2) hipLaunchKernelGGL((env_bwd_split_kernel<2>), grid, block, 0, ctx->stream, edev, b.ckpt, a);
    else {
        hipdsp_set_error("the role-split backward sweep is built for one- and two-section plans");
        return HIPDSP_ERR_UNSUPPORTED;
    }
    return hd_launch_status("env_bwd_split_kernel");
}
