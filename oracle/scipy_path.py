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


def chain(x, rate, sos, esos, nfft, hop, stamps=None):
    """data -> filter -> {spectrogram, envelope} on a (T, C) float64 buffer.  `stamps` (a dict) receives the seconds
    each of the three process() bodies took (incl. the allocation of its destination buffer)."""
    import time
    T, C = x.shape
    t0 = time.perf_counter()
    filt = np.zeros((T, C))
    filter_process(sos, x, filt)
    t1 = time.perf_counter()
    nd = (T + hop - 1)//hop
    spec = np.zeros((nd, C, nfft//2 + 1))
    spectrogram_process(filt, spec, rate, nfft, hop)
    t2 = time.perf_counter()
    env = np.zeros((T, C))
    envelope_process(esos, filt, env)
    t3 = time.perf_counter()
    if stamps is not None:
        stamps.update(bandpass=t1 - t0, spectrogram=t2 - t1, envelope=t3 - t2)
    return filt, spec, env


def _worker(job):
    """One process = one block of channels through the reference's chain (own synthetic slab)."""
    import time
    c0, c1, C, T, rate, nfft, hop, hp, lp, order, env = job
    rng = np.random.default_rng(1234 + 2 + c0)
    t = np.arange(T)/rate
    x = rng.uniform(-1.0, 1.0, size=(T, c1 - c0))
    for c in range(c0, c1):
        x[:, c - c0] = 0.5*x[:, c - c0] + 0.5*np.sin(2*np.pi*1000.0*(1 + c/C)*t)
    sos = signal.butter(order, (hp, lp), 'bandpass', fs=rate, output='sos')
    esos = signal.butter(2, env, 'lowpass', fs=rate, output='sos')
    t0 = time.perf_counter()
    chain(x, rate, sos, esos, nfft, hop)
    return time.perf_counter() - t0


if __name__ == '__main__':
    # all-cores variant of the CPU baseline: channels split over processes.
    # Runs as its own process tree so that nothing here ever touches the GPU.
    import json
    import multiprocessing as mp
    import sys
    import time
    C, seconds, rate, nfft, hop, hp, lp, order, env, nproc = [float(v) for v in sys.argv[1:11]]
    C, nfft, hop, order, nproc = int(C), int(nfft), int(hop), int(order), int(nproc)
    T = int(seconds*rate)
    nproc = max(1, min(nproc, C))
    bounds = [round(i*C/nproc) for i in range(nproc + 1)]
    jobs = [(bounds[i], bounds[i + 1], C, T, rate, nfft, hop, hp, lp, order, env)
            for i in range(nproc) if bounds[i + 1] > bounds[i]]
    t0 = time.perf_counter()
    with mp.get_context('fork').Pool(len(jobs)) as pool:
        inner = pool.map(_worker, jobs)
    wall = time.perf_counter() - t0
    # throughput over the slowest worker's chain time (generation of the input excluded)
    print(json.dumps({'value': C*T/max(inner)/1e6, 'cores': len(jobs), 'wall_s': wall,
                      'chain_s_max': max(inner)}))
