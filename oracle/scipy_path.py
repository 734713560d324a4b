"""The reference's CPU path, call for call -- TEST/BENCH INFRASTRUCTURE ONLY.

Restates the three ``process()`` bodies of the reference with the very scipy calls
they make (float64, time-major (T, C) buffers, single thread like the Qt GUI thread):

  src/audian/bufferedfilter.py:35-36      per-channel sosfilt loop on column views
  src/audian/bufferedspectrogram.py:51-59 thunderlab spectrogram (= scipy.signal.spectrogram,
                                          Hann / constant detrend / density / psd, axis=0)
                                          + the (F,C,T)->(F,T,C)->(T,C,F) transposes
  src/audian/bufferedenvelope.py:39-41    sosfiltfilt(sos, (pi/2)|x|, axis=0) + clamp

Used by bench.py's ``cpu_baseline`` leg and by tests that cross-check the oracle when
scipy is importable.  Nothing under audian_amd/ imports this.
"""

import numpy as np
from scipy import signal


def filter_process(sos, source, dest, nbefore=0):
    for c in range(source.shape[1]):
        dest[:, c] = signal.sosfilt(sos, source[:, c])[nbefore:]


def spectrogram_process(source, dest, rate, nfft, hop):
    nsource = (len(dest) - 1)*hop + nfft
    if nsource > len(source):
        nsource = len(source)
    if nsource >= nfft:
        with np.errstate(under='ignore'):
            freq, time, Sxx = signal.spectrogram(source[:nsource], fs=rate, window='hann',
                                                 nperseg=nfft, noverlap=nfft - hop,
                                                 detrend='constant', scaling='density',
                                                 mode='psd', axis=0)
            Sxx = np.transpose(Sxx, (0, 2, 1))           # thunderlab returns (F, T, C)
        n = Sxx.shape[1]
        dest[:n] = Sxx.transpose((1, 2, 0))
        dest[n:] = 0
    else:
        dest[:] = 0


def envelope_process(sos, source, dest, nbefore=0, highpass_cutoff=0):
    dest[:] = signal.sosfiltfilt(sos, (np.pi/2)*np.abs(source), axis=0)[nbefore:]
    if highpass_cutoff == 0:
        dest[dest < 0] = 0


def chain(x, rate, sos, esos, nfft, hop):
    """data -> filter -> {spectrogram, envelope} on a (T, C) float64 buffer."""
    T, C = x.shape
    filt = np.zeros((T, C))
    filter_process(sos, x, filt)
    nd = (T + hop - 1)//hop
    spec = np.zeros((nd, C, nfft//2 + 1))
    spectrogram_process(filt, spec, rate, nfft, hop)
    env = np.zeros((T, C))
    envelope_process(esos, filt, env)
    return filt, spec, env
