// sos_plan.hip -- host-side mathematics of the block-parallel biquad cascade (float64, no device code): the plan
// block of an SOS table (transition powers, phase-1 weights, sosfilt_zi, warm-up and pad lengths) and the segment
// planner.  Reference semantics: scipy.signal.sosfilt / sosfilt_zi / sosfiltfilt as audian calls them
// (src/audian/bufferedfilter.py:36, src/audian/bufferedenvelope.py:39).
#include "sos_plan.h"
#include <cmath>
#include <vector>

namespace {

// ---- host-side plan mathematics (float64) -----------------------------------

struct Mat {
    int d;
    double v[MAXD][MAXD];
};

Mat mat_identity(int d)
{
    Mat m; m.d = d;
    for (int i = 0; i < MAXD; i++) for (int j = 0; j < MAXD; j++) m.v[i][j] = (i == j && i < d) ? 1.0 : 0.0;
    return m;
}

Mat mat_mul(const Mat &a, const Mat &b)
{
    Mat m = mat_identity(a.d);
    for (int i = 0; i < a.d; i++)
        for (int j = 0; j < a.d; j++) {
            double s = 0.0;
            for (int k = 0; k < a.d; k++) s += a.v[i][k] * b.v[k][j];
            m.v[i][j] = s;
        }
    return m;
}

double mat_norm_inf(const Mat &a)
{
    double n = 0.0;
    for (int i = 0; i < a.d; i++) {
        double s = 0.0;
        for (int j = 0; j < a.d; j++) s += fabs(a.v[i][j]);
        if (!(s <= n)) n = s;      // NaN propagates as "large"
    }
    return n;
}

// One time step of the cascade (same arithmetic order as the kernel / scipy).
void cascade_step(const double coef[MAXS][5], int S, double *z, double x)
{
    double cur = x;
    for (int s = 0; s < S; s++) {
        double y = coef[s][0] * cur + z[2 * s];
        z[2 * s] = coef[s][1] * cur - coef[s][3] * y + z[2 * s + 1];
        z[2 * s + 1] = coef[s][2] * cur - coef[s][4] * y;
        cur = y;
    }
}

int fill_plan(SosPlanDev *p, const double *sos, int S)
{
    memset(p, 0, sizeof(*p));
    p->n_sections = S;
    const int D = 2 * S;
    for (int s = 0; s < S; s++) {
        const double *c = sos + 6 * s;
        if (c[3] != 1.0) {
            hipdsp_set_error("sos[%d][3] (a0) must be 1, got %g", s, c[3]);
            return HIPDSP_ERR_INVALID;
        }
        for (int k = 0; k < 6; k++)
            if (!std::isfinite(c[k])) {
                hipdsp_set_error("sos[%d][%d] is not finite", s, k);
                return HIPDSP_ERR_INVALID;
            }
        p->coef[s][0] = c[0]; p->coef[s][1] = c[1]; p->coef[s][2] = c[2];
        p->coef[s][3] = c[4]; p->coef[s][4] = c[5];
    }
    p->unit_tail = S > 1 ? 1 : 0;
    for (int s = 1; s < S; s++)
        if (!(p->coef[s][0] == 1.0 && p->coef[s][2] == 1.0 && fabs(p->coef[s][1]) == 2.0)) p->unit_tail = 0;
    // state-space (A, B): columns of A from unit states with zero input, B from unit input
    Mat A = mat_identity(D);
    double B[MAXD] = {0};
    for (int c = 0; c < D; c++) {
        double z[MAXD] = {0};
        z[c] = 1.0;
        cascade_step(p->coef, S, z, 0.0);
        for (int r = 0; r < D; r++) A.v[r][c] = z[r];
    }
    {
        double z[MAXD] = {0};
        cascade_step(p->coef, S, z, 1.0);
        for (int r = 0; r < D; r++) B[r] = z[r];
    }
    // G[j] = A^(L-1-j) B
    {
        double g[MAXD];
        for (int r = 0; r < D; r++) g[r] = B[r];
        for (int j = L - 1; j >= 0; j--) {
            for (int r = 0; r < D; r++) p->G[j * D + r] = g[r];
            double t[MAXD] = {0};
            for (int r = 0; r < D; r++)
                for (int c = 0; c < D; c++) t[r] += A.v[r][c] * g[c];
            for (int r = 0; r < D; r++) g[r] = t[r];
        }
    }
    // M[k] = A^(L*2^k); keep squaring up to A^TILE for the warm-up search
    Mat pw = A;                                   // A^1
    for (int k = 0; k < 5; k++) pw = mat_mul(pw, pw);   // A^32 = A^L
    static_assert(L == 32, "plan assumes L == 32");
    for (int k = 0; k < 6; k++) {
        for (int r = 0; r < D; r++)
            for (int c = 0; c < D; c++) p->M[k * D * D + r * D + c] = pw.v[r][c];
        pw = mat_mul(pw, pw);
    }
    for (int r = 0; r < D; r++)
        for (int c = 0; c < D; c++) p->AT[r * D + c] = pw.v[r][c];
    // pw == A^(L*64) == A^TILE.  warm = TILE * (smallest m with ||A^(TILE*m)|| < 2^-60)
    const double tol = ldexp(1.0, -60);
    const int MAXBITS = 40;
    std::vector<Mat> pows;
    pows.push_back(pw);
    long long m = 1;
    int top = 0;
    while (!(mat_norm_inf(pows[top]) < tol) && top < MAXBITS) {
        pows.push_back(mat_mul(pows[top], pows[top]));
        top++;
        m <<= 1;
    }
    if (!(mat_norm_inf(pows[top]) < tol)) {
        m = 1LL << 50;                            // does not decay: never segment
    } else if (top > 0) {
        // binary refinement: largest q with ||A^(TILE*q)|| >= tol, answer q + 1
        Mat acc = mat_identity(D);
        long long q = 0;
        for (int k = top - 1; k >= 0; k--) {
            Mat cand = mat_mul(acc, pows[k]);
            if (!(mat_norm_inf(cand) < tol)) { acc = cand; q += 1LL << k; }
        }
        m = q + 1;
    }
    p->warm = m * TILE;
    // scipy sosfilt_zi
    double scale = 1.0;
    for (int s = 0; s < S; s++) {
        const double *c = sos + 6 * s;
        double b0 = c[0], b1 = c[1], b2 = c[2], a1 = c[4], a2 = c[5];
        double B0 = b1 - a1 * b0, B1 = b2 - a2 * b0;
        double m00 = 1.0 + a1, m01 = -1.0, m10 = a2, m11 = 1.0;
        double det = m00 * m11 - m01 * m10;
        p->zi[2 * s] = scale * (B0 * m11 - m01 * B1) / det;
        p->zi[2 * s + 1] = scale * (m00 * B1 - m10 * B0) / det;
        scale *= (c[0] + c[1] + c[2]) / (c[3] + c[4] + c[5]);
    }
    // scipy sosfiltfilt: edge = 3*ntaps, ntaps = 2S+1 - min(#b2==0, #a2==0)
    int nb = 0, na = 0;
    for (int s = 0; s < S; s++) {
        if (sos[6 * s + 2] == 0.0) nb++;
        if (sos[6 * s + 5] == 0.0) na++;
    }
    p->edge = 3 * (2 * S + 1 - (nb < na ? nb : na));
    return HIPDSP_OK;
}

// Choose the number of time segments per channel.  One wave (the fused sweep: one pair of waves) per (channel,
// segment); a segment costs its own length plus the warm-up it re-reads, times what a tile step costs a wave that
// shares its CU with w - 1 others (tools/occupancy_sweep.py, profiles/r03_occupancy_sweep.log):
//   * single-wave sweeps (`per_simd` = 4): bound by the memory system from two waves per SIMD on -- 8, 12 or 16
//     waves per CU move the same bytes per second, so a tile step costs a wave s / 2 with s = ceil(w / 4) waves on
//     its SIMD -- and by a wave's own latency below (0.65 at one wave per SIMD): 8 per CU is never worse than 16
//     and re-reads half the warm-ups; a short job (BASELINE configs[1]) runs best at 4;
//   * fused sweeps (`per_simd` = 0): the pairs need each other's gaps -- 8 pairs per CU reach 0.66 tile steps per
//     microsecond, 4 pairs 0.57, 2 pairs 0.37 -- so the CU is filled whenever the job allows:
//     cost per tile step ~ w_max x (1 + 0.25 (1 - w / w_max)).
//   * the band-pass alone (sos_scan_kernel, `per_simd` = -1): no prefetch, every tile's load latency is hidden by the
//     other waves of the CU only -- 64 ch x 600 s take 12.97 / 7.49 / 5.98 / 5.56 ms with 4 / 8 / 12 / 16 waves per CU
//     (tools/sos_waves.py, profiles/r03_sos_waves.log), i.e. a tile step costs a wave 7 + 0.5625 w (16 at 16 waves);
//   * the band-pass + envelope-state sweep (sos_ckpt_kernel, prefetching; `per_simd` = -2): 7.46 / 6.31 / 6.01 / 5.93 ms
//     -- throughput is 80 % at one wave per SIMD already: 1.4 + 0.9125 w.
//   (Round 3 first planned both with the backward sweep's table: 8 waves per CU, 7.5 instead of 5.6 ms for a
//   BufferedFilter alone.)
// With more units than n_cus x w_max the waves run in rounds of that many.
double tile_step_cost(long long w, int w_max, int per_simd)
{
    if (per_simd > 0) {
        const long long sw = (w + per_simd - 1) / per_simd;
        return sw <= 1 ? 0.65 : 0.5 * (double)sw;
    }
    if (per_simd == -1) return 7.0 + 0.5625 * (double)w;
    if (per_simd == -2) return 1.4 + 0.9125 * (double)w;
    return (double)w_max * (1.0 + 0.25 * (1.0 - (double)w / (double)w_max));
}

void plan_segments_occ(long long n_cus, int w_max, int per_simd, int max_segments, long long N, long long channels,
                       long long warm, long long *seg_len, int *n_seg)
{
    if (n_cus < 1) n_cus = 1;
    if (w_max < 1) w_max = 1;
    long long max_seg = (N + TILE - 1) / TILE;            // at least one tile per segment
    if (warm >= (1LL << 40)) max_seg = 1;                 // non-decaying filter: never segment
    if (max_segments > 0 && max_seg > max_segments) max_seg = max_segments;
    if (max_seg > 65536) max_seg = 65536;
    if (max_seg < 1) max_seg = 1;
    const long long slots = n_cus * w_max;
    long long best_n = 1, best_len = (N + TILE - 1) / TILE * TILE;
    double best_cost = -1.0;
    auto consider = [&](long long n) {
        if (n < 1) n = 1;
        if (n > max_seg) n = max_seg;
        long long len = ((N + n - 1) / n + TILE - 1) / TILE * TILE;
        if (len < TILE) len = TILE;
        const long long cnt = (N + len - 1) / len;
        const long long units = channels * cnt;
        const double span = (double)(len + (cnt > 1 ? warm : 0));
        double cost;
        if (units <= slots) cost = span * tile_step_cost((units + n_cus - 1) / n_cus, w_max, per_simd);
        else cost = (double)((units + slots - 1) / slots) * span * tile_step_cost(w_max, w_max, per_simd);
        if (best_cost < 0 || cost < best_cost * (1.0 - 1e-9)) { best_cost = cost; best_n = cnt; best_len = len; }
    };
    // candidates: the segment counts that fill whole waves-per-CU levels (most waves first, so that ties keep the
    // chip full), whole rounds beyond that, and powers of two below one level
    for (int w = w_max; w >= 1; w--) consider(n_cus * w / channels);
    for (long long rounds = 2; rounds <= 64; rounds++) consider(rounds * slots / channels);
    for (int half = 1; half < 16; half++) consider((n_cus / channels) >> half);
    consider(1);
    *seg_len = best_len;
    *n_seg = (int)best_n;
}

}  // namespace

int hd_fill_plan(SosPlanDev *p, const double *sos, int S) { return fill_plan(p, sos, S); }

void hd_plan_segments_occ(long long n_cus, int w_max, int per_simd, int max_segments, long long N, long long channels,
                          long long warm, long long *seg_len, int *n_seg)
{
    plan_segments_occ(n_cus, w_max, per_simd, max_segments, N, channels, warm, seg_len, n_seg);
}

extern "C" {

int hipdsp_sos_plan_host(const double *host_sos, int n_sections, int64_t *warmup, int *edge, double *zi)
{
    HD_REQUIRE(host_sos != nullptr, "host_sos is NULL");
    if (n_sections < 1 || n_sections > MAXS) {
        hipdsp_set_error("n_sections %d not in 1..%d", n_sections, MAXS);
        return HIPDSP_ERR_UNSUPPORTED;
    }
    SosPlanDev tmp;
    int rc = fill_plan(&tmp, host_sos, n_sections);
    if (rc != HIPDSP_OK) return rc;
    if (warmup) *warmup = tmp.warm;
    if (edge) *edge = tmp.edge;
    if (zi)
        for (int k = 0; k < 2 * n_sections; k++) zi[k] = tmp.zi[k];
    return HIPDSP_OK;
}

int hipdsp_sos_segments_host(int64_t n_cus, int waves_max, int per_simd, int max_segments, int64_t frames,
                             int64_t channels, int64_t warmup, int64_t *segment_frames, int *n_segments)
{
    HD_REQUIRE(n_cus >= 1 && waves_max >= 1 && per_simd >= -2 && frames >= 1 && channels >= 1 && warmup >= 0 &&
               max_segments >= 0, "bad argument");
    HD_REQUIRE(segment_frames != nullptr && n_segments != nullptr, "NULL output");
    long long len = 0;
    int n = 0;
    plan_segments_occ(n_cus, waves_max, per_simd, max_segments, frames, channels, warmup, &len, &n);
    *segment_frames = len;
    *n_segments = n;
    return HIPDSP_OK;
}

}  // extern "C"
