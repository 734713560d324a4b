"""Pin the CPU oracle against the scipy-generated golden fixtures (CPU only)."""

import numpy as np
import pytest

from conftest import load_golden, rel_err


def test_sosfilt_matches_scipy_golden(oracle):
    g = load_golden('sosfilt')
    for k in range(int(g['count'])):
        sos, x, y = g[f'sos_{k}'], g[f'x_{k}'], g[f'y_{k}']
        got = oracle.sosfilt(sos, x.astype(np.float64))
        assert got.shape == y.shape
        for c in range(y.shape[1]):
            assert rel_err(got[:, c], y[:, c]) < 1e-12, k


def test_sosfilt_with_zi(oracle):
    g = load_golden('sosfilt')
    zi = oracle.sosfilt_zi(g['zi_sos'])
    assert np.allclose(zi, g['zi_zi'], rtol=1e-12, atol=1e-15)
    x = g['zi_x'][:, 0].astype(np.float64)
    y, zf = oracle.sosfilt(g['zi_sos'], x, zi=zi*x[0])
    assert rel_err(y, g['zi_y']) < 1e-12
    assert np.allclose(zf, g['zi_zf'], rtol=1e-9, atol=1e-14)


def test_sosfiltfilt_envelope_matches_scipy_golden(oracle):
    g = load_golden('envelope')
    for k in range(int(g['count'])):
        sos, x, y = g[f'sos_{k}'], g[f'x_{k}'], g[f'y_{k}']
        assert oracle.sosfiltfilt_edge(sos) == int(g[f'edge_{k}'])
        got = oracle.sosfiltfilt(sos, (np.pi/2)*np.abs(x.astype(np.float64)))
        if float(g[f'hp_{k}']) == 0:
            got[got < 0] = 0
        for c in range(y.shape[1]):
            assert rel_err(got[:, c], y[:, c]) < 1e-10, k


def test_sosfiltfilt_too_short_raises(oracle):
    g = load_golden('envelope')
    sos = g['sos_0']
    edge = oracle.sosfiltfilt_edge(sos)
    with pytest.raises(ValueError):
        oracle.sosfiltfilt(sos, np.ones(edge))
    oracle.sosfiltfilt(sos, np.ones(edge + 1))


@pytest.mark.parametrize('impl', ['spectrogram', 'spectrogram_numpy'])
def test_spectrogram_matches_scipy_golden(oracle, impl):
    g = load_golden('spectrogram')
    fn = getattr(oracle, impl)
    for k in range(int(g['count'])):
        rate, nfft, hop = g[f'par_{k}']
        nfft, hop = int(nfft), int(hop)
        x, S = g[f'x_{k}'], g[f'S_{k}']
        f, t, got = fn(x.astype(np.float64), rate, nfft, nfft - hop)
        assert got.shape == S.shape, k
        assert np.allclose(f, g[f'f_{k}'])
        for c in range(S.shape[2]):
            for j in range(S.shape[1]):
                assert rel_err(got[:, j, c], S[:, j, c]) < 1e-11, (k, j, c)


def test_spectrogram_short_source(oracle):
    f, t, S = oracle.spectrogram(np.ones((100, 2)), 48000.0, 256, 128)
    assert S.shape == (129, 0, 2)


def test_decibel_matches_golden(oracle):
    g = load_golden('decibel')
    got = oracle.decibel(g['p'])
    inf = np.isinf(g['db'])
    assert np.array_equal(np.isinf(got), inf)
    assert np.all(got[inf] < 0)
    assert np.allclose(got[~inf], g['db'][~inf], rtol=1e-13, atol=1e-12)


def test_chain_process_bodies(oracle):
    """The reference's three process() bodies, chained as its trace graph does."""
    g = load_golden('chain')
    x = g['x'].astype(np.float64)
    rate = float(g['rate'])
    filt = np.zeros_like(x)
    oracle.filter_process(g['sos'], x, filt, 0)
    assert rel_err(filt, g['filt']) < 1e-12
    nd = g['spec'].shape[0] + 1          # one zero tail frame (nafter quirk)
    spec = np.full((nd, 2, 129), np.nan)
    oracle.spectrogram_process(filt, spec, rate, 256, 128)
    assert np.all(spec[-1] == 0)
    for j in range(nd - 1):
        assert rel_err(spec[j], g['spec'][j]) < 1e-10
    env = np.zeros_like(x)
    oracle.envelope_process(g['esos'], filt, env, 0)
    assert rel_err(env, g['env']) < 1e-10
    # nbefore > 0 and pass-through branches
    d2 = np.zeros((len(x) - 5, 2))
    oracle.filter_process(g['sos'], x, d2, 5)
    assert np.array_equal(d2, filt[5:])
    oracle.filter_process(None, x, d2, 5)
    assert np.array_equal(d2, x[5:])
    oracle.envelope_process(None, x, env, 0)
    assert np.all(env == 0)
