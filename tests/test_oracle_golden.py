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


def _scipy_signal():
    import pytest
    return pytest.importorskip('scipy.signal')


def test_oracle_against_live_scipy_random_cases(oracle):
    """Beyond the committed vectors: where scipy is installed (this image has it; the GPU box too),
    the oracle is re-checked against scipy itself on random designs, lengths and window
    parameters -- the reference's own calls (bufferedfilter.py:36, bufferedenvelope.py:39,
    bufferedspectrogram.py:51-56 through thunderlab)."""
    import numpy as np
    sig = _scipy_signal()
    rng = np.random.default_rng(2024)
    for _ in range(25):
        rate = float(rng.choice([8000.0, 44100.0, 96000.0, 192000.0]))
        order = int(rng.integers(1, 5))
        kind = rng.integers(0, 3)
        if kind == 0:
            sos = sig.butter(order, float(rng.uniform(5, 0.4*rate)), 'lowpass', fs=rate, output='sos')
        elif kind == 1:
            sos = sig.butter(order, float(rng.uniform(20, 0.3*rate)), 'highpass', fs=rate, output='sos')
        else:
            lo = float(rng.uniform(20, 0.1*rate))
            sos = sig.butter(min(order, 2), (lo, float(rng.uniform(2*lo, 0.45*rate))), 'bandpass', fs=rate, output='sos')
        n = int(rng.integers(40, 6000))
        x = rng.standard_normal((n, 2))
        want = np.column_stack([sig.sosfilt(sos, x[:, c]) for c in range(2)])
        assert np.allclose(oracle.sosfilt(sos, x), want, rtol=1e-12, atol=1e-14)
        assert np.allclose(oracle.sosfilt_zi(sos), sig.sosfilt_zi(sos), rtol=1e-12, atol=1e-15)
        if n > oracle.sosfiltfilt_edge(sos):
            r = (np.pi/2)*np.abs(x)
            want = sig.sosfiltfilt(sos, r, axis=0)
            got = oracle.sosfiltfilt(sos, r)
            assert np.max(np.abs(got - want)) <= 1e-9*max(1.0, np.max(np.abs(want)))
    for _ in range(15):
        rate = float(rng.choice([22050.0, 96000.0]))
        nfft = int(rng.choice([8, 64, 256, 300, 1024, 2048]))
        hop = int(rng.integers(1, nfft + 1))
        n = nfft + int(rng.integers(0, 9*nfft))
        x = rng.standard_normal((n, 2)) + 0.2
        f, t, S = sig.spectrogram(x, fs=rate, window='hann', nperseg=nfft, noverlap=nfft - hop, detrend='constant',
                                  scaling='density', mode='psd', axis=0)
        want = np.transpose(S, (0, 2, 1))                    # thunderlab's (F, T, C)
        fo, to, So = oracle.spectrogram(x, rate, nfft, nfft - hop)
        assert So.shape == want.shape and np.allclose(fo, f)
        assert np.max(np.abs(So - want)) <= 1e-10*np.max(np.abs(want))


@pytest.mark.parametrize('value', [np.nan, np.inf, -np.inf])
def test_non_finite_samples_behave_like_scipy(oracle, value):
    """The reference has no handling of its own for NaN or infinite samples: it gets what scipy does -- sosfilt's state
    stays NaN to the end of the slab, sosfiltfilt's output is NaN everywhere in that channel, a spectrogram frame that
    contains such a sample is NaN in every bin, decibel() passes NaN on.  The restatement must do the same, because the
    GPU path is checked against it (tests/test_gpu_parity.py::test_non_finite_*)."""
    ss = pytest.importorskip('scipy.signal')
    from audian_amd.design import butter_sos
    rate, T, nfft, hop = 48000.0, 9000, 256, 128
    x = np.random.default_rng(5).standard_normal((T, 3))
    x[4000, 1] = value
    for sos in (butter_sos(2, (300.0, 3000.0), 'bandpass', rate), butter_sos(2, 500.0, 'lowpass', rate),
                butter_sos(4, (300.0, 3000.0), 'bandpass', rate)):
        got = oracle.sosfilt(sos, x)
        want = np.stack([ss.sosfilt(sos, x[:, c]) for c in range(3)], axis=1)
        assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.isfinite(got), np.isfinite(want))
        assert not np.isfinite(got[4001:, 1]).any() and np.isfinite(got[:4000, 1]).all() and np.isfinite(got[:, [0, 2]]).all()
        env = np.zeros((T, 3))
        oracle.envelope_process(sos, x, env, 0)
        want_e = ss.sosfiltfilt(sos, (np.pi/2)*np.abs(x), axis=0)
        want_e[want_e < 0] = 0
        assert np.array_equal(np.isnan(env), np.isnan(want_e))
        assert np.isnan(env[:, 1]).all() and np.isfinite(env[:, [0, 2]]).all()
    nd = (T + hop - 1)//hop
    spec = np.zeros((nd, 3, nfft//2 + 1))
    oracle.spectrogram_process(x, spec, rate, nfft, hop)
    with np.errstate(invalid='ignore'):
        _, _, S = ss.spectrogram(x, fs=rate, window='hann', nperseg=nfft, noverlap=nfft - hop, detrend='constant',
                                 scaling='density', mode='psd', axis=0)
    S = np.transpose(S, (2, 1, 0))                     # scipy: (F, C, frames) -> (frames, C, F)
    assert np.array_equal(np.isnan(spec[:S.shape[0]]), np.isnan(S))
    hit = [k for k in range(S.shape[0]) if k*hop <= 4000 < k*hop + nfft]
    assert hit and all(np.isnan(spec[k, 1]).all() for k in hit) and np.isfinite(spec[:, [0, 2]]).all()
    assert np.isfinite(np.delete(spec[:, 1], hit, axis=0)).all()
    db = oracle.decibel(spec[:, 1])
    assert np.array_equal(np.isnan(db), np.isnan(spec[:, 1]))
