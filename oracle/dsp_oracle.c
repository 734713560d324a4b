/*
 * dsp_oracle.c -- TEST INFRASTRUCTURE ONLY (parity oracle, never shipped, never
 * the thing measured except as bench.py's `cpu_baseline`).
 *
 * Plain-C float64 restatement of the arithmetic that audian's BufferedData hot
 * path delegates to scipy.signal (scipy 1.15.3; the reference pins no version,
 * pyproject.toml:9).  PARITY UNPINNED by the reference: it holds no golden vectors
 * for this path (it has no tests) and cannot be run here, so this oracle is pinned
 * against scipy-generated fixtures committed under tests/golden/ instead
 * (generator: tests/golden/make_golden.py; the reference's own call arguments).
 *
 * Reference call sites restated here (relative to /root/reference):
 *   src/audian/bufferedfilter.py:35-36     sosfilt(sos, source[:, c])[nbefore:]
 *   src/audian/bufferedenvelope.py:39-41   sosfiltfilt(sos, (pi/2)*|source|, axis=0)
 *   src/audian/bufferedspectrogram.py:51-59 thunderlab spectrogram -> scipy.signal.spectrogram
 *   src/audian/specitem.py:36              thunderlab decibel
 * scipy algorithms followed (scipy/signal/_signaltools.py, _spectral_py.py):
 *   _sosfilt DF-II-transposed loop, lfilter_zi/sosfilt_zi, odd_ext + sosfiltfilt,
 *   _spectral_helper(window='hann', detrend='constant', scaling='density',
 *   mode='psd', return_onesided=True).
 *
 * Build: make -C oracle   ->  oracle/liboracle.so  (gcc only, no dependencies)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* scipy _sosfilt: samples outer, sections inner, direct form II transposed.
 * sos: (S,6) rows [b0 b1 b2 a0 a1 a2] with a0 == 1; zi: (S,2) in/out state.
 * x, y are strided (stride in elements) so (T,C) column views work like the
 * reference's source[:, c]. */
void oracle_sosfilt(const double *sos, int n_sections, const double *x,
                    long x_stride, double *y, long y_stride, long n, double *zi)
{
    for (long i = 0; i < n; i++) {
        double cur = x[i * x_stride];
        for (int s = 0; s < n_sections; s++) {
            const double *c = sos + 6 * s;
            double *z = zi + 2 * s;
            double out = c[0] * cur + z[0];
            z[0] = c[1] * cur - c[4] * out + z[1];
            z[1] = c[2] * cur - c[5] * out;
            cur = out;
        }
        y[i * y_stride] = cur;
    }
}

/* scipy lfilter_zi for one biquad (a0 == 1): solve (I - companion(a).T) zi = B,
 * B = b[1:] - a[1:]*b[0];  I - A = [[1+a1, -1], [a2, 1]]. */
static void biquad_zi(const double *c, double *zi)
{
    double b0 = c[0], b1 = c[1], b2 = c[2], a1 = c[4], a2 = c[5];
    double B0 = b1 - a1 * b0, B1 = b2 - a2 * b0;
    double m00 = 1.0 + a1, m01 = -1.0, m10 = a2, m11 = 1.0;
    double det = m00 * m11 - m01 * m10;
    zi[0] = (B0 * m11 - m01 * B1) / det;
    zi[1] = (m00 * B1 - m10 * B0) / det;
}

/* scipy sosfilt_zi: per-section lfilter_zi scaled by the DC gain of the
 * preceding sections. */
void oracle_sosfilt_zi(const double *sos, int n_sections, double *zi)
{
    double scale = 1.0;
    for (int s = 0; s < n_sections; s++) {
        const double *c = sos + 6 * s;
        biquad_zi(c, zi + 2 * s);
        zi[2 * s] *= scale;
        zi[2 * s + 1] *= scale;
        scale *= (c[0] + c[1] + c[2]) / (c[3] + c[4] + c[5]);
    }
}

/* scipy sosfiltfilt default pad length: 3*ntaps, ntaps reduced by trailing
 * zero coefficients (first-order sections). */
int oracle_sosfiltfilt_edge(const double *sos, int n_sections)
{
    int nb = 0, na = 0;
    for (int s = 0; s < n_sections; s++) {
        if (sos[6 * s + 2] == 0.0) nb++;
        if (sos[6 * s + 5] == 0.0) na++;
    }
    int ntaps = 2 * n_sections + 1 - (nb < na ? nb : na);
    return 3 * ntaps;
}

/* scipy sosfiltfilt(padtype='odd', padlen=None) on one strided column.
 * Returns 0 on success, -1 if n <= edge (scipy raises ValueError), -2 on
 * allocation failure. */
int oracle_sosfiltfilt(const double *sos, int n_sections, const double *x,
                       long x_stride, double *y, long y_stride, long n)
{
    int edge = oracle_sosfiltfilt_edge(sos, n_sections);
    if (n <= edge) return -1;
    long m = n + 2 * (long)edge;
    double *ext = (double *)malloc(sizeof(double) * (size_t)m);
    double *zi0 = (double *)malloc(sizeof(double) * 2 * (size_t)n_sections);
    double *zi = (double *)malloc(sizeof(double) * 2 * (size_t)n_sections);
    if (!ext || !zi0 || !zi) { free(ext); free(zi0); free(zi); return -2; }
    /* odd extension: 2*x[0] - x[edge:0:-1], x, 2*x[-1] - x[-2:-(edge+2):-1] */
    double x0 = x[0], xl = x[(n - 1) * x_stride];
    for (int i = 0; i < edge; i++) {
        ext[i] = 2.0 * x0 - x[(long)(edge - i) * x_stride];
        ext[edge + n + i] = 2.0 * xl - x[(n - 2 - i) * x_stride];
    }
    for (long i = 0; i < n; i++) ext[edge + i] = x[i * x_stride];
    oracle_sosfilt_zi(sos, n_sections, zi0);
    /* forward, zi * ext[0] */
    for (int k = 0; k < 2 * n_sections; k++) zi[k] = zi0[k] * ext[0];
    oracle_sosfilt(sos, n_sections, ext, 1, ext, 1, m, zi);
    /* backward on the reversed forward output, zi * y[-1] */
    double ylast = ext[m - 1];
    for (int k = 0; k < 2 * n_sections; k++) zi[k] = zi0[k] * ylast;
    oracle_sosfilt(sos, n_sections, ext + (m - 1), -1, ext + (m - 1), -1, m, zi);
    for (long i = 0; i < n; i++) y[i * y_stride] = ext[edge + i];
    free(ext); free(zi0); free(zi);
    return 0;
}

/* ---- spectrogram ------------------------------------------------------- */

/* in-place iterative radix-2 complex FFT (n power of two), double. */
static void fft_pow2(double *re, double *im, long n)
{
    for (long i = 1, j = 0; i < n; i++) {
        long bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (long len = 2; len <= n; len <<= 1) {
        long half = len >> 1;
        for (long k = 0; k < half; k++) {
            double ang = -2.0 * M_PI * (double)k / (double)len;
            double wr = cos(ang), wi = sin(ang);
            for (long i = k; i < n; i += len) {
                long j = i + half;
                double tr = re[j] * wr - im[j] * wi;
                double ti = re[j] * wi + im[j] * wr;
                re[j] = re[i] - tr; im[j] = im[i] - ti;
                re[i] += tr; im[i] += ti;
            }
        }
    }
}

/* direct DFT for non power-of-two nfft (small sizes only; O(n^2)). */
static void dft_direct(const double *xr, double *re, double *im, long n)
{
    for (long k = 0; k <= n / 2; k++) {
        double sr = 0.0, si = 0.0;
        for (long i = 0; i < n; i++) {
            double ang = -2.0 * M_PI * (double)((k * i) % n) / (double)n;
            sr += xr[i] * cos(ang);
            si += xr[i] * sin(ang);
        }
        re[k] = sr; im[k] = si;
    }
}

/* scipy.signal.spectrogram(x, fs, window='hann', nperseg=nfft,
 * noverlap=nfft-hop, detrend='constant', scaling='density', mode='psd') on one
 * strided column, no boundary extension, no padding (_spectral_helper):
 *   n_seg = (n - noverlap) // hop; per segment subtract mean, multiply by the
 *   periodic Hann window, rfft, |X|^2 / (fs * sum(w^2)), double all bins except
 *   DC (and Nyquist when nfft is even).
 * out: (n_seg, F) row-major with row stride out_stride (elements), F = nfft/2+1.
 * Returns n_seg (>= 0) or -2 on allocation failure. */
long oracle_spectrogram(const double *x, long x_stride, long n, double fs,
                        long nfft, long hop, double *out, long out_stride)
{
    long F = nfft / 2 + 1;
    if (n < nfft) return 0;
    long nseg = (n - (nfft - hop)) / hop;
    double *w = (double *)malloc(sizeof(double) * (size_t)nfft);
    double *re = (double *)malloc(sizeof(double) * (size_t)nfft);
    double *im = (double *)malloc(sizeof(double) * (size_t)nfft);
    if (!w || !re || !im) { free(w); free(re); free(im); return -2; }
    double wss = 0.0;
    for (long i = 0; i < nfft; i++) {   /* get_window('hann', nfft) is periodic (fftbins=True) */
        w[i] = 0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)nfft);
        wss += w[i] * w[i];
    }
    double scale = 1.0 / (fs * wss);
    int pow2 = (nfft & (nfft - 1)) == 0;
    for (long k = 0; k < nseg; k++) {
        const double *seg = x + k * hop * x_stride;
        double mean = 0.0;
        for (long i = 0; i < nfft; i++) mean += seg[i * x_stride];
        mean /= (double)nfft;
        for (long i = 0; i < nfft; i++) {
            re[i] = (seg[i * x_stride] - mean) * w[i];
            im[i] = 0.0;
        }
        if (pow2) {
            fft_pow2(re, im, nfft);
        } else {
            double *tr = (double *)malloc(sizeof(double) * (size_t)nfft);
            if (!tr) { free(w); free(re); free(im); return -2; }
            memcpy(tr, re, sizeof(double) * (size_t)nfft);
            dft_direct(tr, re, im, nfft);
            free(tr);
        }
        double *o = out + k * out_stride;
        for (long f = 0; f < F; f++) {
            double p = (re[f] * re[f] + im[f] * im[f]) * scale;
            int edge_bin = (f == 0) || ((nfft % 2 == 0) && f == F - 1);
            o[f] = edge_bin ? p : 2.0 * p;
        }
    }
    free(w); free(re); free(im);
    return nseg;
}

/* thunderlab.powerspectrum.decibel(power, ref_power=1.0, min_power=1e-20):
 * 10*log10(power/ref), -inf where power <= min_power. */
void oracle_decibel(const double *p, double *out, long n, double ref_power,
                    double min_power)
{
    for (long i = 0; i < n; i++)
        out[i] = (p[i] <= min_power) ? -INFINITY : 10.0 * log10(p[i] / ref_power);
}
