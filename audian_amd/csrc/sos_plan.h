// sos_plan.h -- what the IIR kernels (sos.hip) and the host-side plan mathematics (sos_plan.hip) share: the tile
// geometry, the device plan block and the planner's entry points.  sos_plan.hip has no device code: it is also built
// with g++ under AddressSanitizer / UBSan by tests/test_shim_sanitizers.py.
#pragma once
#include "common.h"

constexpr int L = 32;               // samples per lane per tile
constexpr int TILE = 64 * L;        // samples per wave per tile
constexpr int MAXS = HIPDSP_MAX_SECTIONS;
constexpr int MAXD = 2 * MAXS;      // state dimension

struct SosPlanDev {
    double coef[MAXS][5];           // b0 b1 b2 a1 a2
    // the two tables are PACKED for the plan's own state dimension D = 2 * n_sections, so that a group of
    // consecutive entries is one run of memory (one batch of wide scalar loads, sos_cascade.inc)
    double G[L * MAXD];             // G[j * D + r] = (A^(L-1-j) B)[r]
    double M[6 * MAXD * MAXD];      // M[k * D * D + r * D + c] = (A^(L*2^k))[r][c]
    double zi[MAXD];                // scipy sosfilt_zi, flattened (z0,z1) per section
    double AT[MAXD * MAXD];         // AT[r * D + c] = (A^TILE)[r][c]: the state hand-over between time segments (env_fix_kernel)
    long long warm;                 // warm-up samples, multiple of TILE
    int n_sections;
    int edge;                       // sosfiltfilt pad length
    // scipy's Butterworth designs (butter(..., output='sos'): bufferedfilter.py:44-52, bufferedenvelope.py:47-52) put the
    // gain into the first section and leave every other numerator at exactly [1, +-2, 1]: for those sections phase 3 runs
    // y = x + z0; z0 = b1 x + z1 - a1 y; z1 = x - a2 y -- four operations per sample instead of five (the products with
    // b0 = b2 = 1 are the operand itself).  Non-zero: every section behind the first is of that form.
    int unit_tail;
};

// plan block from an SOS table (scipy layout, a0 == 1): HIPDSP_OK or an error with the message set
int hd_fill_plan(SosPlanDev *p, const double *sos, int S);
// the segment planner (see include/hip_dsp.h, hipdsp_sos_segments_host)
void hd_plan_segments_occ(long long n_cus, int w_max, int per_simd, int max_segments, long long N, long long channels,
                          long long warm, long long *seg_len, int *n_seg);
