"""Generate the golden fixtures that pin the parity oracle.

Run in the build container (scipy 1.15.3 installed):

    python tests/golden/make_golden.py

The reference (bendalab/audian) holds no tests or golden vectors for its DSP path
and cannot be imported here (PyQt5/pyqtgraph/thunderlab/audioio are absent), so
every fixture is produced by calling scipy with EXACTLY the arguments of the
reference's call sites:

  src/audian/bufferedfilter.py:44-52   butter(order, Wn, btype, fs=rate, output='sos')
  src/audian/bufferedfilter.py:36      sosfilt(sos, source[:, c])
  src/audian/bufferedenvelope.py:39    sosfiltfilt(sos, (np.pi/2)*np.abs(source), axis=0)
  src/audian/bufferedspectrogram.py:51 thunderlab spectrogram == scipy.signal.spectrogram(
        x, fs, window='hann', nperseg=nfft, noverlap=nfft-hop, detrend='constant',
        scaling='density', mode='psd', axis=0), then (F,C,T)->(F,T,C)
  src/audian/specitem.py:36            thunderlab decibel (10*log10, -inf <= 1e-20)

This file is data generation only: no reference source text is stored.  Inputs are
kept as float32 (what the GPU path ingests); outputs as float64.
"""

import os

import numpy as np
from scipy import signal

HERE = os.path.dirname(os.path.abspath(__file__))


def synth(rng, n, channels, rate):
    """White noise in [-1, 1) plus a per-channel tone, as float32."""
    t = np.arange(n)/rate
    x = rng.uniform(-1.0, 1.0, size=(n, channels))
    for c in range(channels):
        x[:, c] = 0.5*x[:, c] + 0.5*np.sin(2*np.pi*1000.0*(1 + c/channels)*t)
    return x.astype(np.float32)


def design_cases():
    """All branches of BufferedFilter.update / BufferedEnvelope.update."""
    cases = []
    for rate in (44100.0, 48000.0, 96000.0, 192000.0):
        for order in (1, 2, 3, 4):
            cases.append(('lowpass', order, (0.1*rate,), rate))
            cases.append(('highpass', order, (300.0,), rate))
            cases.append(('bandpass', order, (300.0, 3000.0), rate))
        cases.append(('lowpass', 2, (20.0,), rate))        # config 3 envelope
        cases.append(('lowpass', 2, (500.0,), rate))       # envelope default
        cases.append(('bandpass', 2, (5.0, 3000.0), rate))  # low cut-off
        cases.append(('bandpass', 2, (10.0, 500.0), rate))  # bandpass envelope
        cases.append(('highpass', 2, (100.0,), rate))
        cases.append(('bandpass', 2, (2000.0, 4000.0), rate))
    return cases


def make_design():
    out = {}
    meta = []
    for i, (btype, order, wn, rate) in enumerate(design_cases()):
        w = wn[0] if len(wn) == 1 else wn
        sos = signal.butter(order, w, btype, fs=rate, output='sos')
        out[f'sos_{i}'] = sos
        meta.append((btype, order, wn[0], wn[1] if len(wn) > 1 else 0.0, rate))
    out['btype'] = np.array([m[0] for m in meta])
    out['order'] = np.array([m[1] for m in meta])
    out['w0'] = np.array([m[2] for m in meta])
    out['w1'] = np.array([m[3] for m in meta])
    out['rate'] = np.array([m[4] for m in meta])
    np.savez_compressed(os.path.join(HERE, 'design.npz'), **out)


def make_sosfilt():
    rng = np.random.default_rng(1001)
    out = {}
    k = 0
    specs = [
        # (rate, n, channels, btype, order, Wn)
        (44100.0, 3000, 1, 'bandpass', 2, (300.0, 3000.0)),   # config 1
        (48000.0, 3000, 2, 'bandpass', 4, (300.0, 3000.0)),   # config 2
        (96000.0, 3000, 3, 'bandpass', 2, (300.0, 3000.0)),   # config 3
        (96000.0, 4000, 1, 'lowpass', 2, (20.0,)),            # slow decay
        (96000.0, 4000, 1, 'bandpass', 2, (5.0, 3000.0)),     # fp32-state killer
        (192000.0, 2500, 2, 'highpass', 3, (100.0,)),         # odd order: 1st-order section
        (48000.0, 2500, 1, 'lowpass', 1, (4000.0,)),          # single first-order section
        (48000.0, 70, 1, 'lowpass', 2, (4000.0,)),            # shorter than one GPU tile
        (48000.0, 1, 2, 'bandpass', 2, (300.0, 3000.0)),      # single sample
    ]
    for rate, n, ch, btype, order, wn in specs:
        w = wn[0] if len(wn) == 1 else wn
        sos = signal.butter(order, w, btype, fs=rate, output='sos')
        x = synth(rng, n, ch, rate)
        y = np.zeros((n, ch))
        for c in range(ch):                      # the reference's per-channel loop
            y[:, c] = signal.sosfilt(sos, x[:, c].astype(np.float64))
        out[f'sos_{k}'] = sos
        out[f'x_{k}'] = x
        out[f'y_{k}'] = y
        out[f'rate_{k}'] = rate
        k += 1
    # with initial conditions (used by the envelope's forward pass)
    sos = signal.butter(2, 500.0, 'lowpass', fs=48000.0, output='sos')
    x = synth(rng, 2000, 1, 48000.0)
    zi = signal.sosfilt_zi(sos)
    y, zf = signal.sosfilt(sos, x[:, 0].astype(np.float64), zi=zi*float(x[0, 0]))
    out['zi_sos'] = sos
    out['zi_x'] = x
    out['zi_zi'] = zi
    out['zi_y'] = y
    out['zi_zf'] = zf
    out['count'] = k
    np.savez_compressed(os.path.join(HERE, 'sosfilt.npz'), **out)


def make_envelope():
    rng = np.random.default_rng(1002)
    out = {}
    k = 0
    specs = [
        # (rate, n, channels, env_cutoff, order, highpass_cutoff)
        (96000.0, 6000, 2, 20.0, 2, 0.0),      # config 3
        (48000.0, 3000, 2, 500.0, 2, 0.0),     # reference default
        (48000.0, 3000, 1, 500.0, 2, 10.0),    # bandpass envelope, no clamp
        (44100.0, 3000, 1, 200.0, 3, 0.0),     # odd order -> edge reduced
        (48000.0, 3000, 1, 1000.0, 1, 0.0),    # single first-order section: edge 6
        (48000.0, 16, 1, 500.0, 2, 0.0),       # just above padlen (9)
    ]
    for rate, n, ch, env, order, hp in specs:
        if hp > 0:
            sos = signal.butter(order, (hp, env), 'bandpass', fs=rate, output='sos')
        else:
            sos = signal.butter(order, env, 'lowpass', fs=rate, output='sos')
        x = synth(rng, n, ch, rate)
        y = signal.sosfiltfilt(sos, (np.pi/2)*np.abs(x.astype(np.float64)), axis=0)
        if hp == 0:
            y[y < 0] = 0
        out[f'sos_{k}'] = sos
        out[f'x_{k}'] = x
        out[f'y_{k}'] = y
        out[f'hp_{k}'] = hp
        out[f'edge_{k}'] = 3*(2*len(sos) + 1 - min((sos[:, 2] == 0).sum(),
                                                  (sos[:, 5] == 0).sum()))
        k += 1
    out['count'] = k
    np.savez_compressed(os.path.join(HERE, 'envelope.npz'), **out)


def ref_spectrogram(x, rate, nfft, hop):
    """thunderlab's wrapper around scipy: returns (freqs, times, Sxx[F, T', C])."""
    f, t, S = signal.spectrogram(x.astype(np.float64), fs=rate, window='hann',
                                 nperseg=nfft, noverlap=nfft - hop,
                                 detrend='constant', scaling='density',
                                 mode='psd', axis=0)
    return f, t, np.transpose(S, (0, 2, 1))


def make_spectrogram():
    rng = np.random.default_rng(1003)
    out = {}
    k = 0
    specs = [
        # (rate, n, channels, nfft, hop)
        (44100.0, 256*9 + 1, 1, 256, 128),     # config 1
        (48000.0, 1024*5 + 1, 2, 1024, 256),   # config 2
        (96000.0, 2048*4 + 1, 2, 2048, 1024),  # config 3
        (48000.0, 700, 1, 64, 13),             # odd hop
        (48000.0, 600, 2, 8, 4),               # minimum nfft of the reference
        (48000.0, 300, 1, 128, 128),           # no overlap
        (48000.0, 256, 1, 256, 128),           # exactly one frame
        (48000.0, 1000, 1, 100, 30),           # non power-of-two (oracle only)
        (192000.0, 4096*3, 1, 4096, 1024),     # larger nfft, 75 % overlap
    ]
    for rate, n, ch, nfft, hop in specs:
        x = synth(rng, n, ch, rate)
        # add a DC offset so that the constant detrend matters
        x = (x + np.float32(0.25)).astype(np.float32)
        f, t, S = ref_spectrogram(x, rate, nfft, hop)
        out[f'x_{k}'] = x
        out[f'S_{k}'] = S
        out[f'f_{k}'] = f
        out[f'par_{k}'] = np.array([rate, nfft, hop])
        k += 1
    out['count'] = k
    np.savez_compressed(os.path.join(HERE, 'spectrogram.npz'), **out)


def make_decibel():
    rng = np.random.default_rng(1004)
    p = np.concatenate([10.0**rng.uniform(-25, 3, size=500),
                        [0.0, 1e-20, 1.0000001e-20, 1e-21, 1.0, 123.0]])
    db = np.full(p.shape, -np.inf)
    m = p > 1e-20
    db[m] = 10.0*np.log10(p[m]/1.0)
    np.savez_compressed(os.path.join(HERE, 'decibel.npz'), p=p, db=db)


def make_chain():
    """A small end-to-end chain exactly as the reference's trace graph runs it:
    data -> BufferedFilter -> {BufferedSpectrogram, BufferedEnvelope}."""
    rng = np.random.default_rng(1005)
    rate = 48000.0
    n = 256*20 + 1
    x = synth(rng, n, 2, rate)
    sos = signal.butter(2, (300.0, 3000.0), 'bandpass', fs=rate, output='sos')
    filt = np.zeros((n, 2))
    for c in range(2):
        filt[:, c] = signal.sosfilt(sos, x[:, c].astype(np.float64))
    f, t, S = signal.spectrogram(filt, fs=rate, window='hann', nperseg=256,
                                 noverlap=128, detrend='constant',
                                 scaling='density', mode='psd', axis=0)
    S = np.transpose(S, (0, 2, 1)).transpose((1, 2, 0))     # (T', C, F)
    esos = signal.butter(2, 500.0, 'lowpass', fs=rate, output='sos')
    env = signal.sosfiltfilt(esos, (np.pi/2)*np.abs(filt), axis=0)
    env[env < 0] = 0
    np.savez_compressed(os.path.join(HERE, 'chain.npz'), x=x, sos=sos, filt=filt,
                        spec=S, esos=esos, env=env, rate=rate)


if __name__ == '__main__':
    make_design()
    make_sosfilt()
    make_envelope()
    make_spectrogram()
    make_decibel()
    make_chain()
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith('.npz'):
            print(fn, os.path.getsize(os.path.join(HERE, fn)))
