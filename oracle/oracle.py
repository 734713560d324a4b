"""CPU parity oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Nothing under ``audian_amd/`` imports it; the product
path fails loudly when the HIP library is missing instead of falling back here.

It restates, in float64, the arithmetic that audian's ``BufferedData`` hot path
delegates to third-party code that is *not* part of ``/root/reference``:

* ``scipy.signal.sosfilt / sosfiltfilt / spectrogram`` (scipy is unpinned in the
  reference, ``pyproject.toml:9``; fixtures were generated with scipy 1.15.3),
* ``thunderlab.powerspectrum.spectrogram / decibel`` (``pyproject.toml:19``,
  thunderlab >= 1.6.0; source unavailable here, restated from the call sites
  ``src/audian/bufferedspectrogram.py:51-60`` and ``src/audian/specitem.py:36`` --
  "thunderlab-equivalence assumed").

PARITY UNPINNED by the reference: it has no tests and no golden vectors for this path and its
package cannot be imported here (PyQt), so nothing the reference itself holds or produces pins this
oracle.  What it is pinned by instead: scipy-generated fixtures under ``tests/golden/`` (generator script
``tests/golden/make_golden.py``, run in the build container where scipy 1.15.3 is
installed).  The recursive filters live in ``dsp_oracle.c`` (plain C, gcc); the
framed PSD exists both there and as a NumPy restatement below so the two check
each other.

Reference ``process()`` bodies restated at the bottom:
``src/audian/bufferedfilter.py:31-36``, ``src/audian/bufferedenvelope.py:34-41``,
``src/audian/bufferedspectrogram.py:45-59``.
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    """Compile ``dsp_oracle.c`` with gcc (building the checker is not using it)."""
    so = os.path.join(_HERE, 'liboracle.so')
    src = os.path.join(_HERE, 'dsp_oracle.c')
    if force or not os.path.exists(so) or \
       os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B', 'liboracle.so'])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build())
        lib.oracle_sosfilt.argtypes = [_dp, ctypes.c_int, _dp, ctypes.c_long,
                                       _dp, ctypes.c_long, ctypes.c_long, _dp]
        lib.oracle_sosfilt.restype = None
        lib.oracle_sosfilt_zi.argtypes = [_dp, ctypes.c_int, _dp]
        lib.oracle_sosfilt_zi.restype = None
        lib.oracle_sosfiltfilt_edge.argtypes = [_dp, ctypes.c_int]
        lib.oracle_sosfiltfilt_edge.restype = ctypes.c_int
        lib.oracle_sosfiltfilt.argtypes = [_dp, ctypes.c_int, _dp, ctypes.c_long,
                                           _dp, ctypes.c_long, ctypes.c_long]
        lib.oracle_sosfiltfilt.restype = ctypes.c_int
        lib.oracle_spectrogram.argtypes = [_dp, ctypes.c_long, ctypes.c_long,
                                           ctypes.c_double, ctypes.c_long,
                                           ctypes.c_long, _dp, ctypes.c_long]
        lib.oracle_spectrogram.restype = ctypes.c_long
        lib.oracle_decibel.argtypes = [_dp, _dp, ctypes.c_long, ctypes.c_double,
                                       ctypes.c_double]
        lib.oracle_decibel.restype = None
        _LIB = lib
    return _LIB


def _sos(sos):
    sos = np.ascontiguousarray(sos, dtype=np.float64)
    if sos.ndim != 2 or sos.shape[1] != 6:
        raise ValueError('sos must be shape (n_sections, 6)')
    if not np.all(sos[:, 3] == 1.0):
        raise ValueError('sos[:, 3] should be all ones')
    return sos


def _ptr(a, offset=0):
    return ctypes.cast(a.ctypes.data + 8*offset, _dp)


def sosfilt_zi(sos):
    """scipy.signal.sosfilt_zi (scipy/signal/_signaltools.py:4164-4176)."""
    sos = _sos(sos)
    zi = np.zeros((len(sos), 2))
    _lib().oracle_sosfilt_zi(_ptr(sos), len(sos), _ptr(zi))
    return zi


def sosfilt(sos, x, zi=None):
    """scipy.signal.sosfilt along axis 0 of a 1-D or (T, C) array, float64.

    ``zi``: optional (S, 2) initial state applied to every column; when given the
    final state(s) are returned too, like scipy.
    """
    sos = _sos(sos)
    x = np.asarray(x, dtype=np.float64)
    one_d = x.ndim == 1
    x2 = np.ascontiguousarray(x.reshape(len(x), -1))
    y = np.empty_like(x2)
    n, nc = x2.shape
    zf = np.zeros((nc, len(sos), 2))
    for c in range(nc):
        if zi is not None:
            zf[c] = zi
        _lib().oracle_sosfilt(_ptr(sos), len(sos), _ptr(x2, c), nc,
                              _ptr(y, c), nc, n, _ptr(zf[c]))
    y = y[:, 0] if one_d else y.reshape(x.shape)
    if zi is not None:
        return y, (zf[0] if one_d else zf)
    return y


def sosfiltfilt_edge(sos):
    sos = _sos(sos)
    return int(_lib().oracle_sosfiltfilt_edge(_ptr(sos), len(sos)))


def sosfiltfilt(sos, x):
    """scipy.signal.sosfiltfilt(sos, x, axis=0) with the default odd padding."""
    sos = _sos(sos)
    x = np.asarray(x, dtype=np.float64)
    one_d = x.ndim == 1
    x2 = np.ascontiguousarray(x.reshape(len(x), -1))
    y = np.empty_like(x2)
    n, nc = x2.shape
    for c in range(nc):
        rc = _lib().oracle_sosfiltfilt(_ptr(sos), len(sos), _ptr(x2, c), nc,
                                       _ptr(y, c), nc, n)
        if rc == -1:
            raise ValueError('The length of the input vector x must be greater '
                             'than padlen, which is %d.' % sosfiltfilt_edge(sos))
        if rc != 0:
            raise MemoryError('oracle_sosfiltfilt')
    return y[:, 0] if one_d else y.reshape(x.shape)


def spectrogram(data, rate, n_fft, n_overlap):
    """thunderlab.powerspectrum.spectrogram as audian calls it
    (src/audian/bufferedspectrogram.py:51-56): Hann window, constant detrend,
    density scaling, one-sided PSD.  ``data`` is (T,) or (T, C); returns
    ``freqs, times, Sxx`` with Sxx shaped (F, T') or (F, T', C).  C implementation.
    """
    data = np.asarray(data, dtype=np.float64)
    one_d = data.ndim == 1
    x2 = np.ascontiguousarray(data.reshape(len(data), -1))
    n, nc = x2.shape
    hop = n_fft - n_overlap
    F = n_fft//2 + 1
    nseg = (n - n_overlap)//hop if n >= n_fft else 0
    out = np.zeros((nc, max(nseg, 0), F))
    for c in range(nc):
        if nseg > 0:
            r = _lib().oracle_spectrogram(_ptr(x2, c), nc, n, float(rate),
                                          n_fft, hop, _ptr(out[c]), F)
            assert r == nseg
    freqs = np.arange(F)*rate/n_fft
    times = (np.arange(nseg)*hop + n_fft/2)/rate
    Sxx = out.transpose(2, 1, 0)      # (F, T', C)
    if one_d:
        Sxx = Sxx[:, :, 0]
    return freqs, times, Sxx


def spectrogram_numpy(data, rate, n_fft, n_overlap):
    """Same as `spectrogram` but NumPy only (np.fft.rfft); cross-checks the C FFT."""
    data = np.asarray(data, dtype=np.float64)
    one_d = data.ndim == 1
    x2 = data.reshape(len(data), -1)
    n, nc = x2.shape
    hop = n_fft - n_overlap
    F = n_fft//2 + 1
    nseg = (n - n_overlap)//hop if n >= n_fft else 0
    win = 0.5 - 0.5*np.cos(2*np.pi*np.arange(n_fft)/n_fft)
    scale = 1.0/(rate*np.sum(win**2))
    Sxx = np.zeros((F, nseg, nc))
    for k in range(nseg):
        seg = x2[k*hop:k*hop + n_fft, :]
        seg = (seg - seg.mean(axis=0, keepdims=True))*win[:, None]
        X = np.fft.rfft(seg, axis=0)
        P = (X.real**2 + X.imag**2)*scale
        if n_fft % 2:
            P[1:] *= 2
        else:
            P[1:-1] *= 2
        Sxx[:, k, :] = P
    freqs = np.arange(F)*rate/n_fft
    times = (np.arange(nseg)*hop + n_fft/2)/rate
    if one_d:
        Sxx = Sxx[:, :, 0]
    return freqs, times, Sxx


def decibel(power, ref_power=1.0, min_power=1e-20):
    """thunderlab.powerspectrum.decibel: 10*log10(power/ref), -inf at or below
    ``min_power`` (call sites src/audian/specitem.py:28,36,
    src/audian/spectrogramplot.py:159, src/audian/bufferedspectrogram.py:116-117)."""
    p = np.ascontiguousarray(power, dtype=np.float64)
    out = np.empty_like(p)
    _lib().oracle_decibel(_ptr(p.reshape(-1)), _ptr(out.reshape(-1)), p.size,
                          float(ref_power), float(min_power))
    return out


# ---- the reference's process() bodies, restated ------------------------------

def filter_process(sos, source, dest, nbefore):
    """src/audian/bufferedfilter.py:31-36."""
    if sos is None:
        dest[:, :] = source[nbefore:, :]
    else:
        for c in range(source.shape[1]):
            dest[:, c] = sosfilt(sos, source[:, c])[nbefore:]


def envelope_process(sos, source, dest, nbefore, highpass_cutoff=0):
    """src/audian/bufferedenvelope.py:34-41."""
    if sos is None:
        dest[:] = np.zeros_like(dest)
    else:
        dest[:] = sosfiltfilt(sos, (np.pi/2)*np.abs(source))[nbefore:]
        if highpass_cutoff == 0:
            dest[dest < 0] = 0


def spectrogram_process(source, dest, rate, nfft, hop):
    """src/audian/bufferedspectrogram.py:45-59 (dest is (T', C, F))."""
    nsource = (len(dest) - 1)*hop + nfft
    if nsource > len(source):
        nsource = len(source)
    if nsource >= nfft:
        freq, time, Sxx = spectrogram(source[:nsource], rate, nfft, nfft - hop)
        if Sxx.ndim == 2:
            Sxx = Sxx[:, :, None]
        n = Sxx.shape[1]
        dest[:n] = Sxx.transpose((1, 2, 0))
        dest[n:] = 0
        return freq
    dest[:] = 0
    return None


def minmax_decimate(data, start, stop, step):
    """TraceItem.update_plot's screen decimation (src/audian/traceitem.py:55-61; the same
    reduction in compresseddata.py:48-52): min and max of every `step` frames of
    data[start:stop] (1-D or (T, C)), interleaved min, max, ... along axis 0."""
    seg = np.arange(0, stop - start, step)
    block = np.asarray(data)[start:stop]
    out = np.zeros((2*len(seg),) + block.shape[1:])
    np.minimum.reduceat(block, seg, out=out[0::2])
    np.maximum.reduceat(block, seg, out=out[1::2])
    return out


def decimated_db_image(spec_tcf, start, stop, step, channel):
    """Screen-resolution spectrogram image of one channel: TraceItem.update_plot's reduction
    (np.maximum.reduceat over arange(0, stop - start, step), src/audian/traceitem.py:55-61) applied
    to the frames of buffer[start:stop, channel, :], then SpecItem's decibel(...).T
    (src/audian/specitem.py:36).  The reference itself only has this as a TODO (README.md:96)."""
    seg = np.arange(0, stop - start, step)
    block = np.asarray(spec_tcf)[start:stop, channel, :]
    return decibel(np.maximum.reduceat(block, seg, axis=0)).T


def mean_power_db(spec_tcf, i0, i1, channel, floor_db=-200.0):
    """SpectrogramPlot.update_plot's power spectrum (src/audian/spectrogramplot.py:158-160)."""
    power = np.mean(np.asarray(spec_tcf, dtype=np.float64)[i0:i1, channel, :], axis=0)
    power = decibel(power)
    power[power < floor_db] = floor_db
    return power


def play_data(data, rate, i0, i1, show_channels, heterodyne_freq=None, sos=None):
    """DataBrowser.play_region's arithmetic (src/audian/databrowser.py:1711-1727) on a (T, C)
    array: channel-group means, optional heterodyne * sin, sosfiltfilt with `sos`
    (= butter(2, 20000, 'low', fs=rate)), [::nstep]."""
    n2 = (len(show_channels) + 1)//2
    playdata = np.zeros((i1 - i0, min(2, len(show_channels))))
    playdata[:, 0] = np.mean(data[i0:i1, show_channels[:n2]], 1)
    if len(show_channels) > 1:
        playdata[:, 1] = np.mean(data[i0:i1, show_channels[n2:]], 1)
    if heterodyne_freq:
        heterodyne = np.sin(2*np.pi*heterodyne_freq*np.arange(len(playdata))/rate)
        playdata = (playdata.T*heterodyne).T
        fcutoff = 20000.0
        nstep = int(np.round(rate/(2*fcutoff)))
        if nstep < 1:
            nstep = 1
        playdata = sosfiltfilt(sos, playdata)[::nstep]
        rate /= nstep
    return playdata, rate


def unwrap(data, thresh, ampl_max=1.0, clips=False, down_scale=True):
    """audioio's unwrap() as the reference arms it on its raw loader (src/audian/data.py:180,
    src/audian/audian.py:1485-1512), restated from its documentation (source absent: parity unpinned):
    a step between successive samples beyond `thresh` is a wrap-around; from there on 2*ampl_max is
    subtracted (step up) or added (step down), cumulatively along axis 0.  float32 arithmetic like the
    device path (the raw buffer it is applied to holds the file's float32/PCM samples)."""
    x = np.asarray(data, dtype=np.float32)
    d = np.diff(x, axis=0)
    ev = (d < -np.float32(thresh)).astype(np.int64) - (d > np.float32(thresh)).astype(np.int64)
    k = np.zeros(x.shape, dtype=np.int64)
    k[1:] = np.cumsum(ev, axis=0)
    y = x + np.float32(2.0*ampl_max)*k.astype(np.float32)
    if clips:
        y = np.clip(y, -np.float32(ampl_max), np.float32(ampl_max))
    elif down_scale:
        y = y*np.float32(0.5)
    return y.astype(np.float32)
