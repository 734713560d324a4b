"""BASELINE.json configs[0] and configs[1] at their full sizes: the whole chain on the GPU
against the CPU oracle on every sample (they are small enough for the oracle)."""

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def run_chain(oracle, C, seconds, rate, nfft, hop, order, env_cutoff, seed):
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    T = int(seconds*rate)
    ctx = hipdsp.Context(0)
    dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
    hipdsp.synth(ctx, dx, T, C, T, rate, seed)
    sos = butter_sos(order, (300.0, 3000.0), 'bandpass', rate)
    esos = butter_sos(2, env_cutoff, 'lowpass', rate)
    df = hipdsp.DeviceArray(ctx, (C, T), np.float32)
    de = hipdsp.DeviceArray(ctx, (C, T), np.float32)
    nd = (T + hop - 1)//hop
    F = nfft//2 + 1
    ds = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32)
    hipdsp.sosfilt(ctx, hipdsp.SosPlan(ctx, sos), dx, T, df, T, C, T, 0)
    hipdsp.spectrogram(ctx, df, T, C, T, nfft, hop, rate, ds, nd)
    hipdsp.envelope(ctx, hipdsp.SosPlan(ctx, esos), df, T, de, T, C, T, 0)
    x = dx.to_host().T.astype(np.float64)
    filt = np.zeros_like(x)
    oracle.filter_process(sos, x, filt, 0)
    got = df.to_host()
    worst = max(rel_err(got[c], filt[:, c]) for c in range(C))
    env = np.zeros_like(x)
    oracle.envelope_process(esos, filt, env, 0)
    got = de.to_host()
    worst = max([worst] + [rel_err(got[c], env[:, c]) for c in range(C)])
    spec = np.zeros((nd, C, F))
    oracle.spectrogram_process(filt, spec, rate, nfft, hop)
    got = ds.to_host()
    for c in range(C):
        peak = np.max(np.abs(spec[:, c, :]), axis=1)
        err = np.max(np.abs(got[c] - spec[:, c, :]), axis=1)
        ok = peak > 0
        worst = max(worst, float(np.max(err[ok]/peak[ok])))
        assert np.all(got[c][~ok] == 0)
    return worst


def test_config0_one_channel_44k(oracle):
    """configs[0]: 1 ch x 60 s x 44.1 kHz (synthetic stand-in for the absent WAV), nfft 256
    hop 128, band-pass 300-3000 Hz order 2."""
    assert run_chain(oracle, 1, 60.0, 44100.0, 256, 128, 2, 500.0, 1234) < 1e-4


def test_config1_four_channels_48k(oracle):
    """configs[1]: 4 ch x 60 s x 48 kHz, nfft 1024 hop 256, SOS order-4 band-pass."""
    assert run_chain(oracle, 4, 60.0, 48000.0, 1024, 256, 4, 500.0, 1235) < 1e-4
