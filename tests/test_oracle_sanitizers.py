"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only): a
driver program links dsp_oracle.c with -fsanitize and replays the golden inputs; any
out-of-bounds access or UB aborts it."""

import os
import subprocess

import numpy as np

from conftest import ROOT, load_golden

DRIVER = r'''
#include <stdio.h>
#include <stdlib.h>
void oracle_sosfilt(const double*, int, const double*, long, double*, long, long, double*);
int oracle_sosfiltfilt(const double*, int, const double*, long, double*, long, long);
long oracle_spectrogram(const double*, long, long, double, long, long, double*, long);
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb");
    long S, n, nfft, hop;
    if (fread(&S, sizeof(long), 1, f) != 1 || fread(&n, sizeof(long), 1, f) != 1 ||
        fread(&nfft, sizeof(long), 1, f) != 1 || fread(&hop, sizeof(long), 1, f) != 1) return 2;
    double *sos = malloc(sizeof(double)*6*S), *x = malloc(sizeof(double)*n), *y = malloc(sizeof(double)*n);
    double *zi = calloc(2*S, sizeof(double));
    if (fread(sos, sizeof(double), 6*S, f) != (size_t)(6*S) || fread(x, sizeof(double), n, f) != (size_t)n) return 2;
    fclose(f);
    oracle_sosfilt(sos, (int)S, x, 1, y, 1, n, zi);
    double acc = y[n-1];
    if (oracle_sosfiltfilt(sos, (int)S, x, 1, y, 1, n) != 0) return 3;
    acc += y[0];
    long F = nfft/2 + 1, nseg = (n - (nfft - hop))/hop;
    double *P = malloc(sizeof(double)*F*nseg);
    if (oracle_spectrogram(x, 1, n, 48000.0, nfft, hop, P, F) != nseg) return 4;
    acc += P[F*nseg - 1];
    printf("%.17g\n", acc);
    free(sos); free(x); free(y); free(zi); free(P);
    return 0;
}
'''


def test_oracle_c_is_clean_under_asan_ubsan(tmp_path):
    src = tmp_path/'driver.c'
    src.write_text(DRIVER)
    exe = tmp_path/'driver'
    subprocess.check_call(['gcc', '-O1', '-g', '-std=c99', '-fsanitize=address,undefined',
                           '-fno-sanitize-recover=all', '-ffp-contract=off', str(src),
                           os.path.join(ROOT, 'oracle', 'dsp_oracle.c'), '-lm', '-o', str(exe)])
    g = load_golden('sosfilt')
    for k, (nfft, hop) in [(1, (64, 13)), (5, (128, 128)), (3, (100, 30))]:
        sos = np.ascontiguousarray(g[f'sos_{k}'], dtype=np.float64)
        x = np.ascontiguousarray(g[f'x_{k}'][:, 0], dtype=np.float64)
        blob = tmp_path/f'in{k}.bin'
        with open(blob, 'wb') as f:
            np.array([len(sos), len(x), nfft, hop], dtype=np.int64).tofile(f)
            sos.tofile(f)
            x.tofile(f)
        env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1')
        r = subprocess.run([str(exe), str(blob)], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        assert np.isfinite(float(r.stdout))
