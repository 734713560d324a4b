// chain.hip -- the fused sweeps of the batch chain for gfx950: band-pass + envelope state sweep + spectrogram in one
// launch (hipdsp_chain_forward: IIR waves hand their tiles to FFT waves through LDS), and the frame-split variant of
// the backward sweep (hipdsp_chain_backward).  Reference: BufferedFilter / BufferedSpectrogram / BufferedEnvelope.process
// of bendalab/audian (src/audian/bufferedfilter.py:31-36, bufferedspectrogram.py:45-59, bufferedenvelope.py:34-41)
// recomputed depth-first by BufferedData.recompute_all (buffereddata.py:149-153).  The IIR role is sos.hip's
// sos_ckpt_kernel walk (shared cascade body: sos_cascade.inc), the FFT role spectrogram.hip's arithmetic (fft_device.h).
#include "chain_fwd.h"

namespace {

// ---- backward sweep of the batch chain with HALF of the spectrogram fused in ------------------------------
// The forward sweep is bound by VALU issue (its FFT waves are the critical path of every pair), the backward
// sweep by memory with its VALU half idle.  With "frame split" the forward sweep writes only the EVEN frames
// 2t (= tile t itself) and this kernel the ODD ones 2t+1 (second half of tile t + first half of tile t+1, which it
// walked one iteration earlier): both launches then carry one cascade pair and one FFT per tile and 10 bytes per
// sample (4 R + 4 W + 2 W of PSD).  A workgroup is NP IIR waves -- env_bwd_kernel<SE, true>'s walk, line for
// line, except that a tile goes into LDS RAW and is rectified on the way into the cascades -- and NP FFT waves
// with the same pairwise hand-over as in chain_fwd_kernel: H1 "the tile holds the filtered samples", H2 "copied"
// (the IIR wave may overwrite the tile with the forward cascade's outputs).  nfft 2048 / hop 1024 only.
// hipdsp_chain_backward: a channel whose forward sweep ended non-finite (slot n_tiles of its tile states, see
// sos_device.h: FloodArgs) has an envelope that is NaN everywhere
__global__ __launch_bounds__(256) void env_nan_fill_kernel(const double *__restrict__ ckpt, long long ckpt_pitch, long long n_tiles,
                                                           int DE, float *__restrict__ env, long long env_pitch, long long T)
{
    const long long ch = blockIdx.y;
    const double *e = ckpt + ch * ckpt_pitch + n_tiles * DE;
    bool bad = false;
    for (int r = 0; r < DE; r++) bad = bad || !(fabs(e[r]) <= 1.7976931348623157e308);
    if (!bad) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < T; i += (long long)gridDim.x * blockDim.x)
        env[ch * env_pitch + i] = __builtin_nanf("");
}

struct ChainBwdArgs {
    BwdArgs b;
    float *psd;
    long long psd_pitch;
    long long n_valid;
    const float *tables;
    float scale;
    long long units;
    int debug;
    int *fault;
};

template <int SE, int NP>
__global__ __launch_bounds__(128 * NP, NP / 2) void chain_bwd_kernel(const SosPlanDev *__restrict__ P0, ChainBwdArgs a)
{
    constexpr int DE = 2 * SE;
    constexpr int NFFT = 2048, M = NFFT / 2, F = M + 1, MP = M + M / 16;
    constexpr int R1 = 16, R2 = 16, R3 = 4;
    constexpr int TW2 = (R2 - 1) * R1, TW3 = R1 * R2, TWN = M / 2 + 1, NTAB = TW2 + TW3 + TWN + M;
    __shared__ float4 tiles[NP][64 * 8];
    __shared__ float2 fbs[NP][MP];
    __shared__ float2 tab[NTAB];
    __shared__ int ready[NP], taken[NP];
    __shared__ int abort_wg;
    if (threadIdx.x < NP) { ready[threadIdx.x] = 0; taken[threadIdx.x] = 0; }
    if (threadIdx.x == 0) abort_wg = 0;
    bool gave_up = false;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int pair = wave < NP ? wave : wave - NP;
    {
        const float2 *src = reinterpret_cast<const float2 *>(a.tables);
        for (int i = tid; i < NTAB; i += 128 * NP) tab[i] = src[i];
    }
    __syncthreads();

    const long long unit = (long long)blockIdx.x * NP + pair;
    const bool unit_ok = unit < a.units;
    const int seg = unit_ok ? (int)(unit % a.b.n_seg) : 0;
    const long long ch = unit_ok ? unit / a.b.n_seg : 0;
    const long long T = a.b.T;
    const int edge = a.b.edge;
    // reversed tile index rt = n_tiles-1 - (p / TILE): the unit owns rt in [rt_lo, rt_hi)
    const long long rt_lo = (long long)seg * a.b.seg_tiles;
    long long rt_hi = rt_lo + a.b.seg_tiles;
    if (rt_hi > a.b.n_tiles) rt_hi = a.b.n_tiles;
    long long rt_start = rt_lo - a.b.warm_tiles;
    if (rt_start < 0) rt_start = 0;
    if (!unit_ok) rt_hi = rt_start;

    if (wave < NP) {
        // ================= IIR role: env_bwd_kernel<SE, true> with the hand-overs added ==================
        float4 *lds = tiles[pair];
        float *ldsf = reinterpret_cast<float *>(lds);
        const float *in = a.b.in + ch * a.b.in_pitch;
        float *out = a.b.out + ch * a.b.out_pitch;
        const double *ckpt = a.b.ckpt + ch * a.b.ckpt_pitch;
        double cb_[DE];
#pragma unroll
        for (int r = 0; r < DE; r++) cb_[r] = 0.0;
        v4f nx[8], nck[SE];
        bool pre = false;
        auto fetch = [&](long long tidx) {
#pragma unroll
            for (int k = 0; k < 8; k++) nx[k] = asm_load16(in + tidx * TILE + 256 * k + 4 * lane);
#pragma unroll
            for (int i = 0; i < SE; i++) nck[i] = asm_load16(ckpt + tidx * DE + 2 * i);
        };
        const long long top_full = T / TILE - 1;                     // host guarantees >= 0
        auto prefetchable = [&](long long tidx) { return tidx >= 0 && tidx <= top_full; };
        {
            const long long t0 = a.b.n_tiles - 1 - rt_start;
            pre = rt_start < rt_hi && prefetchable(t0);
            fetch(pre ? t0 : top_full);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const double rgain = a.b.gain;
        int it = 0;
        for (long long rt = rt_start; rt < rt_hi; rt++) {
            it++;
            const long long tidx = a.b.n_tiles - 1 - rt;
            const long long tile = tidx * TILE;
            double cfw_[DE];
            // the FFT wave must have its copy of the previous tile before this one goes into LDS
            if (it > 1) CHAIN_WAIT_FOR(taken, it - 1, it);
            const bool fast = pre;
            if (fast) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    asm volatile("" : "+v"(nx[k]));
                    lds[lds_slot(8 * k + (lane >> 3), lane & 7)] = make_float4(nx[k].x, nx[k].y, nx[k].z, nx[k].w);
                }
#pragma unroll
                for (int i = 0; i < SE; i++) {
                    asm volatile("" : "+v"(nck[i]));
                    const v2d d = __builtin_bit_cast(v2d, nck[i]);
                    cfw_[2 * i] = d.x; cfw_[2 * i + 1] = d.y;
                }
            } else {
                // a tile that touches T: untracked loads from clamped addresses, zeros past T; RAW, like the fast path
#pragma unroll 1
                for (int k = 0; k < 8; k++) {
                    const long long p = tile + 256 * k + 4 * lane;
                    v4f t;
                    t.x = asm_load4(in + (p < T ? p : T - 1));
                    t.y = asm_load4(in + (p + 1 < T ? p + 1 : T - 1));
                    t.z = asm_load4(in + (p + 2 < T ? p + 2 : T - 1));
                    t.w = asm_load4(in + (p + 3 < T ? p + 3 : T - 1));
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("" : "+v"(t));
                    lds[lds_slot(8 * k + (lane >> 3), lane & 7)] =
                        make_float4(p < T ? t.x : 0.f, p + 1 < T ? t.y : 0.f, p + 2 < T ? t.z : 0.f, p + 3 < T ? t.w : 0.f);
                }
                v4f ck[SE];
#pragma unroll
                for (int i = 0; i < SE; i++) ck[i] = asm_load16(ckpt + tidx * DE + 2 * i);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < SE; i++) {
                    asm volatile("" : "+v"(ck[i]));
                    const v2d d = __builtin_bit_cast(v2d, ck[i]);
                    cfw_[2 * i] = d.x; cfw_[2 * i + 1] = d.y;
                }
            }
            WAVE_SYNC();
            CHAIN_POST(ready, it);                                  // H1: the tile holds the filtered samples
            {
                pre = rt + 1 < rt_hi && prefetchable(tidx - 1);
                fetch(pre ? tidx - 1 : top_full);
            }
            if (fast) {
                // ---- forward cascade again, from the state that entered this tile; rectification on the way in.
                // Phase 1 only reads the tile: H2 is needed before phase 3 overwrites it.
#define CASC_S SE
#define CASC_PLAN() PLAN_OF(P0)
#define CASC_CARRY cfw_
#define CASC_IN(v) (a.b.rectify ? fabsf(v) : (v))
#define CASC_GAIN rgain
#define CASC_ROLLED_GROUPS
#define CASC_STAMP(n) do { if ((n) == 1) CHAIN_WAIT_FOR(taken, it, it); } while (0)
#include "sos_cascade.inc"
#undef CASC_STAMP
#undef CASC_IN
#undef CASC_CARRY
            } else {
                CHAIN_WAIT_FOR(taken, it, it);                      // H2, then the tile is this wave's alone
                if (a.b.rectify) {
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        float4 v = lds[lds_slot(lane, q)];
                        v = make_float4(fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w));
                        lds[lds_slot(lane, q)] = v;
                    }
                }
                float ra = asm_load4(in + (T - 1));
                float rb = asm_load4(in + (lane < edge ? T - 2 - lane : 0));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(ra));
                asm volatile("" : "+v"(rb));
                WAVE_SYNC();
                // right odd extension ext[T + i] = 2 r(T-1) - r(T-2-i), i < edge
                if (lane < edge) {
                    const long long pj = T + lane;
                    if (pj >= tile && pj < tile + TILE) {
                        if (a.b.rectify) { ra = fabsf(ra); rb = fabsf(rb); }
                        ldsf[lds_float_index((int)(pj - tile))] = 2.f * ra - rb;
                    }
                }
                WAVE_SYNC();
#define CASC_CARRY cfw_
#define CASC_IN(v) (v)
#include "sos_cascade.inc"
#undef CASC_IN
#undef CASC_CARRY
            }
            WAVE_SYNC();
            if (rt == 0) {
                // scipy: backward pass starts from zi * y_fwd[-1]; pad the rest of the tile with it
                const int last = (int)(T + edge - 1 - tile);
                const float v0 = ldsf[lds_float_index(last)];
                WAVE_SYNC();
                for (int s2 = last + 1 + lane; s2 < TILE; s2 += 64) ldsf[lds_float_index(s2)] = v0;
                const SosPlanDev *P = PLAN_OF(P0);
#pragma unroll
                for (int r = 0; r < DE; r++) cb_[r] = P->zi[r] * (double)v0;
                WAVE_SYNC();
            }
            // ---- backward cascade over the forward outputs, last sample first (the gain belongs to the forward one)
#undef CASC_GAIN
#define CASC_CARRY cb_
#define CASC_IN(v) (v)
#define CASC_REVERSED
#include "sos_cascade.inc"
#undef CASC_REVERSED
#undef CASC_ROLLED_GROUPS
#undef CASC_S
#undef CASC_PLAN
#undef CASC_CARRY
#undef CASC_IN
            WAVE_SYNC();
            if (rt >= rt_lo) {
                if (tile + TILE <= T) {
                    // interior tile: exactly 8 vector stores, then the counted wait for the prefetch
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        float4 v = lds[lds_slot(8 * k + (lane >> 3), lane & 7)];
                        if (a.b.clamp) {
                            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        }
                        f4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
                        *reinterpret_cast<f4u *>(out + (tile + 256 * k + 4 * lane)) = t;
                    }
                    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                } else {
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        float4 v = lds[lds_slot(8 * k + (lane >> 3), lane & 7)];
                        if (a.b.clamp) {
                            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        }
                        store_four(out, tile + 256 * k + 4 * lane, v, 0, T, 0);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // warm-up tile: no stores to count
            }
            WAVE_SYNC();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // ================= FFT role: frame 2t+1 = second half of tile t + first half of tile t+1 ===========
        __builtin_amdgcn_s_setprio(3);
        float2 *fb = fbs[pair];
        const float *tlf = reinterpret_cast<const float *>(tiles[pair]);
        const float2 *tw2 = tab, *tw3 = tab + TW2, *twn = tab + TW2 + TW3, *win = tab + TW2 + TW3 + TWN;
        float *oc = a.psd + ch * a.psd_pitch;
        v2f cur_[16], nxt_[8];                     // this tile; the first half of the tile walked before it
#pragma unroll
        for (int j = 0; j < 16; j++) cur_[j] = (v2f){0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; j++) nxt_[j] = (v2f){0.f, 0.f};
        bool have_next = false;
        float piv = 0.f;
        bool have_piv = false;
        int it = 0;
        for (long long rt = rt_start; rt < rt_hi; rt++) {
            it++;
            const long long tidx = a.b.n_tiles - 1 - rt;
            CHAIN_WAIT_FOR(ready, it, it);
#pragma unroll
            for (int j = 0; j < 16; j++)
                cur_[j] = *reinterpret_cast<const v2f *>(tlf + lds_float_index(2 * lane + 128 * j));
            {
                const bool withhold = (a.debug & 8) && blockIdx.x == 0 && pair == 0 && rt == rt_start;
                if (!withhold) CHAIN_POST(taken, it);
            }
            const long long f = 2 * tidx + 1;
            if (rt >= rt_lo && have_next && f < a.n_valid) {
                v2f w[16];
#pragma unroll
                for (int i = 0; i < 8; i++) { w[i] = cur_[8 + i]; w[8 + i] = nxt_[i]; }
                psd_frame_pivot<NFFT, 64, R1, R2, R3, false>(w, fb, tw2, tw3, twn, win, lane, PsdScale{a.scale, 0.5f * a.scale, 2.f * a.scale}, true,
                                                             bin_sink<false>(oc + f * (long long)F, nullptr, 0), piv, have_piv);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) nxt_[j] = cur_[j];
            have_next = true;
        }
    }
}

#undef CHAIN_WAIT_FOR
#undef CHAIN_POST
#undef CHAIN_FAIR

}  // namespace

extern "C" {

int hipdsp_chain_forward(hipdsp_ctx *ctx, const hipdsp_sosplan *fplan, const hipdsp_sosplan *eplan,
                         const float *x, int64_t x_pitch, float *yf, int64_t yf_pitch, int64_t channels,
                         int64_t frames, int rectify, double gain, int nfft, int hop, double fs, float *psd,
                         float *db_out, int64_t frames_out, int64_t psd_pitch, int64_t spec_frames,
                         int64_t spec_first, int64_t env_first)
{
    HD_REQUIRE(ctx != nullptr && fplan != nullptr, "NULL argument");
    HD_REQUIRE(channels >= 0 && frames >= 0 && frames_out >= 0, "negative size");
    HD_REQUIRE(frames_out < (1LL << 31) - (1LL << 16), "frames_out %lld: the fused sweep counts frames in 32 bits", (long long)frames_out);
    HD_REQUIRE(spec_first >= 0 && spec_first <= frames, "spec_first %lld not in [0, frames=%lld]", (long long)spec_first,
               (long long)frames);
    HD_REQUIRE(spec_frames >= 0 && spec_first + spec_frames <= frames, "spec_first %lld + spec_frames %lld beyond frames=%lld",
               (long long)spec_first, (long long)spec_frames, (long long)frames);
    HD_REQUIRE(env_first >= 0 && env_first <= frames, "env_first %lld not in [0, frames=%lld]", (long long)env_first,
               (long long)frames);
    HD_REQUIRE(fs > 0, "fs must be positive");
    const int SF = fplan->host->n_sections, SE = eplan ? eplan->host->n_sections : 0;
    HD_REQUIRE(SF > 0 && (SE > 0 || eplan == nullptr), "plan has no coefficients");
    // shapes the kernel is built for: frames that are register windows of a 2048-sample tile
    const bool shape_ok = (nfft == 2048 && (hop == 1024 || hop == 512)) || (nfft == 1024 && (hop == 512 || hop == 256)) ||
                          (nfft == 512 && hop == 256) || (nfft == 256 && hop == 128);
    if (!shape_ok || SF > 4 || SE > 2 || frames < 4 * TILE ||
        fplan->host->warm >= (1LL << 40) || (eplan && eplan->host->warm >= (1LL << 40))) {
        hipdsp_set_error("the fused forward sweep covers nfft/hop 2048/1024, 2048/512, 1024/512, 1024/256, 512/256 and 256/128, a band-pass of "
                         "at most four and an envelope of at most two sections that decay, and traces of at least %d "
                         "frames: use hipdsp_sosfilt_envelope + hipdsp_spectrogram", 4 * TILE);
        return HIPDSP_ERR_UNSUPPORTED;
    }
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    {   // an earlier launch on this context may have reported a fault that nobody has looked at yet
        const int frc = hd_device_fault(ctx);
        if (frc != HIPDSP_OK) return frc;
    }
    const int edge = eplan ? eplan->host->edge : 0;
    if (eplan && frames - env_first <= edge) {
        hipdsp_set_error("The length of the input vector x must be greater than padlen, which is %d.", edge);
        return HIPDSP_ERR_TOO_SHORT;
    }
    if (channels == 0) return HIPDSP_OK;
    HD_REQUIRE(channels <= 65535, "more than 65535 channels");     // grid.y of the zero-tail launch
    HD_REQUIRE(x != nullptr && yf != nullptr && (psd != nullptr || frames_out == 0), "NULL data pointer");
    HD_REQUIRE(x_pitch >= frames && yf_pitch >= frames, "pitch smaller than row length");
    HD_NO_OVERLAP(x, x_pitch, frames, yf, yf_pitch, frames, channels, "x and yf");
    const long long F = nfft / 2 + 1;
    if (psd_pitch == 0) psd_pitch = frames_out * F;
    HD_REQUIRE(psd_pitch >= frames_out * F, "psd_pitch smaller than one channel");
    // frames inside the trace, as in hipdsp_spectrogram (bufferedspectrogram.py:46-49); the spectrogram may be
    // handed fewer samples than the filter produces (spec_frames: BufferedData.load_buffer's one frame "after")
    const long long sframes = spec_frames > 0 ? spec_frames : frames - spec_first;
    long long nsource = (frames_out - 1) * (long long)hop + nfft;
    if (nsource > sframes) nsource = sframes;
    long long n_valid = 0;
    if (frames_out > 0 && nsource >= nfft) n_valid = (nsource - (nfft - hop)) / hop;
    if (n_valid > frames_out) n_valid = frames_out;
    double wss = 0.0;
    for (int i = 0; i < nfft; i++) {
        const double w = 0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)nfft);
        wss += w * w;
    }
    ChainArgs a;
    memset(&a, 0, sizeof(a));
    int rc = hd_fft_tables(ctx, nfft, &a.tables);
    if (rc != HIPDSP_OK) return rc;
    // the sweep's tile grid: frame 0 of the spectrogram and the envelope's first sample wherever the caller has
    // scrolled to (sos_device.h: GridShift) -- the kernel walks T = frames + lead samples, x and yf shifted by -lead
    const GridShift gs = hd_grid_shift(spec_first, env_first, hop, edge, SE > 0);
    HD_REQUIRE(gs.lead < TILE, "grid shift %lld", gs.lead);
    const long long T = frames + gs.lead;
    const long long n_tiles = (T + edge + TILE - 1) / TILE;
    const long long ckpt_pitch = (n_tiles + 1) * 2 * SE;           // the layout the backward sweep expects
    void *work = nullptr;
    if (SE > 0) {
        rc = hipdsp_scratch(ctx, sizeof(double) * (size_t)ckpt_pitch * (size_t)channels, &work);
        if (rc != HIPDSP_OK) return rc;
    }
    hd_note_sweep(ctx, gs.lead, gs.env0, frames, channels, SE);    // what a backward sweep (phase 2) must agree with
    a.c.in = x - gs.lead; a.c.yf = yf - gs.lead; a.c.ckpt = (double *)work;
    a.c.in_pitch = x_pitch; a.c.yf_pitch = yf_pitch; a.c.ckpt_pitch = ckpt_pitch;
    a.c.T = T; a.c.edge = edge; a.c.rectify = rectify; a.c.gain = rectify ? gain : 1.0;
    a.c.lead = gs.lead; a.c.env0 = gs.env0; a.frame_off = gs.frame_off;
    a.psd = psd; a.db = db_out; a.psd_pitch = psd_pitch; a.n_valid = n_valid;
    a.scale = (float)(1.0 / (fs * wss));
    a.scale_half = 0.5f * a.scale; a.scale_twice = 2.f * a.scale;
    a.warm_total = fplan->host->warm;              // the envelope's states are handed over exactly (env_fix_kernel)
    a.debug = ctx->chain_debug;
    a.fault = ctx->fault_dev;
    a.split = (ctx->chain_split_frames && nfft == 2048 && hop == 1024 && !db_out && gs.lead == 0 && gs.env0 == 0) ? 1 : 0;
    constexpr int P = CHAIN_P;                                     // IIR waves (and FFT waves) per workgroup, one per CU
    // (hipdsp_chain_plan reports exactly this segmentation)
    plan_segments_chain(ctx, T, channels, a.warm_total, &a.c.seg_len, &a.c.n_seg);
    a.units = channels * a.c.n_seg;
    if (a.c.n_seg > 1) {                                           // non-finite samples: sos_device.h, FloodArgs
        rc = hd_seg_flags(ctx, (size_t)a.units, &a.c.flags);
        if (rc != HIPDSP_OK) return rc;
    }
    a.n_iter = (int)((a.warm_total + a.c.seg_len + edge + TILE - 1) / TILE) + 1;
    long long blocks = (a.units + P - 1) / P;
    {
        const long long cus = ctx->n_cus - ctx->chain_reserve_cus > 0 ? ctx->n_cus - ctx->chain_reserve_cus : 1;
        if (a.units < cus * P) {                                   // not every pair of the chip gets a unit: spread them
            blocks = a.units < cus ? a.units : cus;
            a.unit_stride = blocks;
        }
    }
    HD_REQUIRE(blocks <= 0x7fffffffLL, "grid too large");
    if (frames_out > n_valid && frames_out - n_valid <= 16 && !(ctx->chain_debug & 32)) {
        a.tail_end = frames_out;                                   // the few frames of the zero tail: inside the launch
    } else if (frames_out > n_valid) {                             // zero tail (bufferedspectrogram.py:59)
        const long long n = (frames_out - n_valid) * F;
        unsigned gx = (unsigned)((n + 1023) / 1024 > 4096 ? 4096 : (n + 1023) / 1024);
        hipLaunchKernelGGL(zero_rows_kernel, dim3(gx, (unsigned)channels), dim3(256), 0, ctx->stream,
                           psd + n_valid * F, (long long)psd_pitch, n, 0.f);
        if (db_out && !(ctx->chain_debug & 32))          // (bit 32: db_out is the stamp buffer of the diagnostic build)
            hipLaunchKernelGGL(zero_rows_kernel, dim3(gx, (unsigned)channels), dim3(256), 0, ctx->stream,
                               db_out + n_valid * F, (long long)psd_pitch, n, -INFINITY);
    }
    const SosPlanDev *edev_ = eplan ? eplan->dev : nullptr;
    const bool flags = (ctx->chain_debug & 4) == 0;       // bit 4: workgroup barriers instead of the pairwise flags
    // diagnostic build ("chain_debug" bit 32): db_out receives 16 clock sums per wave (needs >= blocks * 16 * 16 * 8 bytes)
    const bool stamp = (ctx->chain_debug & 32) && db_out && SF == 2 && SE == 1 && flags && nfft == 2048 && hop == 1024;
    // one translation unit per window shape holds the kernel's instantiations (chain_shape.inc): every shape with the
    // pairwise flags, PSD only or with the fused dB image; 2048 / 1024 with plans of one or two sections also with
    // workgroup barriers (the tests)
    int (*launch)(int, int, int, int, int, unsigned, hipStream_t, const SosPlanDev *, const SosPlanDev *, const void *, size_t) =
        (nfft == 2048 && hop == 1024) ? hd_chain_fwd_launch_2048_1024 : (nfft == 2048) ? hd_chain_fwd_launch_2048_512 :
        (nfft == 1024 && hop == 512) ? hd_chain_fwd_launch_1024_512 : (nfft == 1024) ? hd_chain_fwd_launch_1024_256 :
        (nfft == 512) ? hd_chain_fwd_launch_512_256 : hd_chain_fwd_launch_256_128;
    if (launch(SF, SE, flags ? 1 : 0, (db_out && !stamp) ? 1 : 0, stamp ? 1 : 0, (unsigned)blocks, ctx->stream, fplan->dev, edev_,
               &a, sizeof(a)) != 0) {
        hipdsp_set_error("the barrier variant of the fused sweep is built for nfft 2048 / hop 1024 and plans of one or two sections");
        return HIPDSP_ERR_UNSUPPORTED;
    }
    if (stamp) return hd_launch_status("chain_fwd_kernel");
    rc = hd_launch_status("chain_fwd_kernel");
    if (rc != HIPDSP_OK) return rc;
    FloodArgs fl;
    memset(&fl, 0, sizeof(fl));
    fl.flags = a.c.flags; fl.n_seg = a.c.n_seg; fl.seg_len = a.c.seg_len;
    fl.y = yf; fl.y_pitch = yf_pitch; fl.T = T; fl.skip = gs.lead;       // (sample p' of the sweep is yf[p' - lead])
    fl.psd = psd; fl.db = db_out; fl.psd_pitch = psd_pitch; fl.n_valid = n_valid;
    fl.F = (int)F; fl.nfft = nfft; fl.hop = hop; fl.frame_off = gs.frame_off;
    if (SE == 0) return launch_flood(ctx, fl, channels);
    return hd_launch_env_fix(ctx, eplan->dev, SE, (double *)work, ckpt_pitch, channels, a.c.n_seg, a.c.seg_len, n_tiles, &fl,
                             (int)((gs.env0 - gs.env0 % TILE) / a.c.seg_len));
}

int hipdsp_chain_backward(hipdsp_ctx *ctx, const hipdsp_sosplan *eplan, const float *yf, int64_t yf_pitch, float *env,
                          int64_t env_pitch, int64_t channels, int64_t frames, int rectify, double gain, int clamp,
                          int nfft, int hop, double fs, float *psd, int64_t frames_out, int64_t psd_pitch)
{
    HD_REQUIRE(ctx != nullptr && eplan != nullptr, "NULL argument");
    HD_REQUIRE(channels >= 0 && frames >= 0 && frames_out >= 0, "negative size");
    HD_REQUIRE(fs > 0, "fs must be positive");
    const int SE = eplan->host->n_sections;
    HD_REQUIRE(SE > 0, "plan has no coefficients");
    if (nfft != 2048 || hop != 1024 || SE > 2 || frames < 4 * TILE || eplan->host->warm >= (1LL << 40)) {
        hipdsp_set_error("the backward sweep with the odd frames fused in covers nfft 2048 / hop 1024, envelope plans of at "
                         "most two decaying sections and traces of at least %d frames", 4 * TILE);
        return HIPDSP_ERR_UNSUPPORTED;
    }
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    {
        const int frc = hd_device_fault(ctx);
        if (frc != HIPDSP_OK) return frc;
    }
    const int edge = eplan->host->edge;
    if (frames <= edge) {
        hipdsp_set_error("The length of the input vector x must be greater than padlen, which is %d.", edge);
        return HIPDSP_ERR_TOO_SHORT;
    }
    if (channels == 0) return HIPDSP_OK;
    HD_REQUIRE(channels <= 65535, "more than 65535 channels");
    HD_REQUIRE(yf != nullptr && env != nullptr && (psd != nullptr || frames_out == 0), "NULL data pointer");
    HD_REQUIRE(yf_pitch >= frames && env_pitch >= frames, "pitch smaller than row length");
    HD_NO_OVERLAP(yf, yf_pitch, frames, env, env_pitch, frames, channels, "yf and env");
    const long long F = nfft / 2 + 1;
    if (psd_pitch == 0) psd_pitch = frames_out * F;
    HD_REQUIRE(psd_pitch >= frames_out * F, "psd_pitch smaller than one channel");
    long long nsource = (frames_out - 1) * (long long)hop + nfft;
    if (nsource > frames) nsource = frames;
    long long n_valid = 0;
    if (frames_out > 0 && nsource >= nfft) n_valid = (nsource - (nfft - hop)) / hop;
    if (n_valid > frames_out) n_valid = frames_out;
    double wss = 0.0;
    for (int i = 0; i < nfft; i++) {
        const double w = 0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)nfft);
        wss += w * w;
    }
    ChainBwdArgs a;
    memset(&a, 0, sizeof(a));
    int rc = hd_fft_tables(ctx, nfft, &a.tables);
    if (rc != HIPDSP_OK) return rc;
    const long long n_tiles = (frames + edge + TILE - 1) / TILE;
    const long long ckpt_pitch = (n_tiles + 1) * 2 * SE;           // where the forward sweep parked the tile states
    void *work = nullptr;
    // the tile states in the scratch belong to the grid of the forward sweep that left them: this sweep walks the
    // unshifted grid only (after hipdsp_chain_forward with spec_first / env_first > 0 the states sit
    // on a grid of frames + lead samples and the odd frames belong elsewhere)
    if (ctx->sweep_frames != frames || ctx->sweep_channels != channels || ctx->sweep_sections != SE ||
        ctx->sweep_lead != 0 || ctx->sweep_env0 != 0) {
        hipdsp_set_error("hipdsp_chain_backward: the last forward sweep on this context left tile states for %lld channels x "
                         "%lld frames, %d sections, grid shift %lld, envelope from %lld -- not for %lld x %lld, %d on the "
                         "unshifted grid (hipdsp_chain_forward with spec_first = env_first = 0)",
                         ctx->sweep_channels, ctx->sweep_frames, ctx->sweep_sections, ctx->sweep_lead, ctx->sweep_env0,
                         (long long)channels, (long long)frames, SE);
        return HIPDSP_ERR_INVALID;
    }
    rc = hd_scratch_parked(ctx, sizeof(double) * (size_t)ckpt_pitch * (size_t)channels, &work);
    if (rc != HIPDSP_OK) return rc;
    a.b.in = yf; a.b.in_pitch = yf_pitch;
    a.b.out = env; a.b.out_pitch = env_pitch;
    a.b.ckpt = (const double *)work; a.b.ckpt_pitch = ckpt_pitch;
    a.b.T = frames; a.b.skip = 0; a.b.n_tiles = n_tiles; a.b.edge = edge;
    a.b.rectify = rectify; a.b.clamp = clamp; a.b.gain = rectify ? gain : 1.0;
    constexpr int P = 8;
    long long seg_len = 0;
    plan_segments_chain(ctx, n_tiles * TILE, channels, eplan->host->warm, &seg_len, &a.b.n_seg);
    a.b.seg_tiles = seg_len / TILE;
    a.b.warm_tiles = eplan->host->warm / TILE;
    a.psd = psd; a.psd_pitch = psd_pitch; a.n_valid = n_valid;
    a.scale = (float)(1.0 / (fs * wss));
    a.units = channels * a.b.n_seg;
    a.debug = ctx->chain_debug;
    a.fault = ctx->fault_dev;
    const long long blocks = (a.units + P - 1) / P;
    HD_REQUIRE(blocks <= 0x7fffffffLL, "grid too large");
    dim3 grid((unsigned)blocks), block(128 * P);
    if (SE == 1) hipLaunchKernelGGL((chain_bwd_kernel<1, P>), grid, block, 0, ctx->stream, eplan->dev, a);
    else hipLaunchKernelGGL((chain_bwd_kernel<2, P>), grid, block, 0, ctx->stream, eplan->dev, a);
    rc = hd_launch_status("chain_bwd_kernel");
    if (rc != HIPDSP_OK) return rc;
    // (its paired waves cannot leave early like env_bwd_kernel's: the NaN channels are overwritten afterwards)
    hipLaunchKernelGGL(env_nan_fill_kernel, dim3(64, (unsigned)channels), dim3(256), 0, ctx->stream, (const double *)work,
                       ckpt_pitch, n_tiles, 2 * SE, env, (long long)env_pitch, (long long)frames);
    return hd_launch_status("env_nan_fill_kernel");
}

int hipdsp_chain_backward_plan(hipdsp_ctx *ctx, const hipdsp_sosplan *eplan, int64_t channels, int64_t frames,
                               int64_t *first_border, int64_t *segment_frames, int *n_segments)
{
    HD_REQUIRE(ctx != nullptr && eplan != nullptr, "NULL argument");
    HD_REQUIRE(first_border != nullptr && segment_frames != nullptr && n_segments != nullptr, "NULL output");
    HD_REQUIRE(channels >= 1 && frames >= 1, "bad size");
    HD_REQUIRE(eplan->host->n_sections > 0, "plan has no coefficients");
    const long long n_tiles = (frames + eplan->host->edge + TILE - 1) / TILE;
    long long len = 0;
    int n = 0;
    plan_segments_chain(ctx, n_tiles * TILE, channels, eplan->host->warm, &len, &n);
    // segment s (walked from the END of the trace) covers frames [n_tiles*TILE - (s+1)*len, n_tiles*TILE - s*len)
    *first_border = n_tiles * TILE - len;
    *segment_frames = len;
    *n_segments = n;
    return HIPDSP_OK;
}

int hipdsp_chain_plan(hipdsp_ctx *ctx, const hipdsp_sosplan *fplan, const hipdsp_sosplan *eplan,
                      int64_t channels, int64_t frames, int64_t *segment_frames, int *n_segments)
{
    HD_REQUIRE(ctx != nullptr && fplan != nullptr, "NULL argument");      // (eplan may be NULL: no envelope)
    HD_REQUIRE(segment_frames != nullptr && n_segments != nullptr, "NULL output");
    HD_REQUIRE(channels >= 1 && frames >= 1, "bad size");
    HD_REQUIRE(fplan->host->n_sections > 0 && (eplan == nullptr || eplan->host->n_sections > 0), "plan has no coefficients");
    long long len = 0;
    int n = 0;
    plan_segments_chain(ctx, frames, channels, fplan->host->warm, &len, &n);
    *segment_frames = len;
    *n_segments = n;
    return HIPDSP_OK;
}

}  // extern "C"
