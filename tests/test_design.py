"""Butterworth SOS design (host, float64) against the scipy golden tables."""

import numpy as np
import pytest

from conftest import load_golden
from audian_amd.design import butter_sos


def test_design_matches_scipy_golden():
    g = load_golden('design')
    for i in range(len(g['order'])):
        btype = str(g['btype'][i])
        wn = (g['w0'][i], g['w1'][i]) if btype == 'bandpass' else g['w0'][i]
        sos = butter_sos(int(g['order'][i]), wn, btype, float(g['rate'][i]))
        want = g[f'sos_{i}']
        assert sos.shape == want.shape, i
        assert np.allclose(sos, want, rtol=1e-9, atol=1e-13), (i, btype, np.abs(sos - want).max())


def test_design_matches_live_scipy_if_present():
    signal = pytest.importorskip('scipy.signal')
    rng = np.random.default_rng(5)
    for _ in range(200):
        rate = float(rng.choice([8000.0, 44100.0, 48000.0, 96000.0, 192000.0, 500000.0]))
        order = int(rng.integers(1, 5))
        btype = str(rng.choice(['lowpass', 'highpass', 'bandpass']))
        lo = float(10.0**rng.uniform(0.5, np.log10(rate/2) - 0.3))
        hi = float(lo*10.0**rng.uniform(0.05, 1.0))
        if btype == 'bandpass':
            if hi >= rate/2:
                continue
            wn = (lo, hi)
        else:
            wn = lo
        want = signal.butter(order, wn, btype, fs=rate, output='sos')
        got = butter_sos(order, wn, btype, rate)
        assert got.shape == want.shape
        assert np.allclose(got, want, rtol=1e-8, atol=1e-12), (order, wn, btype, rate)


def test_design_errors_like_scipy():
    with pytest.raises(ValueError):
        butter_sos(2, 30000.0, 'lowpass', 48000.0)       # above Nyquist
    with pytest.raises(ValueError):
        butter_sos(2, 0.0, 'highpass', 48000.0)
    with pytest.raises(ValueError):
        butter_sos(2, (3000.0, 300.0), 'bandpass', 48000.0)
    with pytest.raises(ValueError):
        butter_sos(2, (0.0, 300.0), 'bandpass', 48000.0)
