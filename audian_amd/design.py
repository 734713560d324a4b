"""Host-side Butterworth design in float64 (NumPy only).

Replaces the ``scipy.signal.butter(order, Wn, btype, fs=rate, output='sos')`` calls of
the reference (``src/audian/bufferedfilter.py:44-52``,
``src/audian/bufferedenvelope.py:47-52``, /root/reference) so that the package has no
scipy dependency at run time.  The design stays on the host in float64 -- the
kernels take the float64 SOS table as is.

Algorithm (the published one scipy follows): analog Butterworth prototype ->
frequency transform in zero/pole/gain form -> bilinear transform with pre-warping ->
pole/zero pairing into second-order sections, worst (closest to the unit circle)
pole pair last, each paired with its nearest zeros.  Parity with scipy 1.15.3 is
pinned by ``tests/golden/design.npz``.
"""

import numpy as np


def _prototype_poles(order):
    m = np.arange(-order + 1, order, 2)
    return -np.exp(1j*np.pi*m/(2*order))


def _to_lowpass(p, k, wo):
    return np.zeros(0, complex), wo*p, k*wo**len(p)


def _to_highpass(p, k, wo):
    return np.zeros(len(p), complex), wo/p, k*np.real(1.0/np.prod(-p))


def _to_bandpass(p, k, wo, bw):
    n = len(p)
    c = p*bw/2
    root = np.sqrt(c**2 - wo**2)
    return np.zeros(n, complex), np.concatenate((c + root, c - root)), k*bw**n


def _bilinear(z, p, k):
    """Bilinear transform at sample rate 2 (normalised frequencies)."""
    fs2 = 4.0
    degree = len(p) - len(z)
    zd = (fs2 + z)/(fs2 - z)
    pd = (fs2 + p)/(fs2 - p)
    zd = np.concatenate((zd, -np.ones(degree)))
    kd = k*np.real(np.prod(fs2 - z)/np.prod(fs2 - p))
    return zd, pd, kd


def _split_conjugates(r):
    """Roots -> (one member with imag > 0 of each conjugate pair, real roots), both
    sorted by real part then |imag|; pair members are averaged."""
    r = np.atleast_1d(np.asarray(r, dtype=complex))
    if r.size == 0:
        return r, r.real
    tol = 100*np.finfo(float).eps
    r = r[np.lexsort((np.abs(r.imag), r.real))]
    is_real = np.abs(r.imag) <= tol*np.abs(r)
    reals = r[is_real].real
    c = r[~is_real]
    up, dn = c[c.imag > 0], c[c.imag < 0]
    if len(up) != len(dn):
        raise ValueError('complex root without a matching conjugate')
    # within runs of equal real part order by |imag| so that partners line up
    if len(up) > 1:
        same = np.diff(up.real) <= tol*np.abs(up[:-1])
        edges = np.diff(np.concatenate(([0], same.astype(int), [0])))
        for a, b in zip(np.nonzero(edges > 0)[0], np.nonzero(edges < 0)[0]):
            for half in (up, dn):
                seg = half[a:b + 1]
                seg[...] = seg[np.argsort(np.abs(seg.imag), kind='stable')]
    if np.any(np.abs(up - dn.conj()) > tol*np.abs(dn)):
        raise ValueError('complex root without a matching conjugate')
    return (up + dn.conj())/2, reals


def _quadratic(roots):
    """Monic real polynomial with the given (<= 2) roots, right-aligned in 3 slots."""
    out = np.zeros(3)
    if len(roots) == 0:
        out[2] = 1.0
    elif len(roots) == 1:
        out[1:] = [1.0, np.real(-roots[0])]
    else:
        out[:] = [1.0, np.real(-roots[0] - roots[1]), np.real(roots[0]*roots[1])]
    return out


def _nearest(cands, target, kind):
    order = np.argsort(np.abs(cands - target))
    if kind == 'any':
        return order[0]
    real = np.isreal(cands[order])
    return order[np.nonzero(real if kind == 'real' else ~real)[0][0]]


def zpk_to_sos(z, p, k):
    """Digital zeros/poles/gain -> second-order sections ('nearest' pairing)."""
    z = np.asarray(z, dtype=complex)
    p = np.asarray(p, dtype=complex)
    if len(z) == 0 and len(p) == 0:
        return np.array([[k, 0., 0., 1., 0., 0.]])
    p = np.concatenate((p, np.zeros(max(len(z) - len(p), 0))))
    z = np.concatenate((z, np.zeros(max(len(p) - len(z), 0))))
    n_sections = (len(p) + 1)//2
    if len(p) % 2 == 1:
        p = np.concatenate((p, [0.]))
        z = np.concatenate((z, [0.]))
    z = np.concatenate(_split_conjugates(z)).astype(complex)
    p = np.concatenate(_split_conjugates(p)).astype(complex)

    def worst(q):
        return np.argmin(np.abs(1 - np.abs(q)))

    def take(arr, i):
        return arr[i], np.delete(arr, i)

    sos = np.zeros((n_sections, 6))
    for si in range(n_sections - 1, -1, -1):
        p1, p = take(p, worst(p))
        if np.isreal(p1) and np.isreal(p).sum() == 0:
            # the last real pole: first-order section padded with a root at 0
            z1, z = take(z, _nearest(z, p1, 'real'))
            zs, ps = [z1, 0.0], [p1, 0.0]
        elif (len(p) + 1 == len(z) and not np.isreal(p1)
              and np.isreal(p).sum() == 1 and np.isreal(z).sum() == 1):
            # one real pole and one real zero remain: this pair needs a complex zero
            z1, z = take(z, _nearest(z, p1, 'complex'))
            zs, ps = [z1, z1.conj()], [p1, p1.conj()]
        else:
            if np.isreal(p1):
                ridx = np.flatnonzero(np.isreal(p))
                p2, p = take(p, ridx[worst(p[ridx])])
            else:
                p2 = p1.conj()
            ps = [p1, p2]
            if len(z) == 0:
                zs = []
            else:
                z1, z = take(z, _nearest(z, p1, 'any'))
                if not np.isreal(z1):
                    zs = [z1, z1.conj()]
                elif len(z) > 0:
                    z2, z = take(z, _nearest(z, p1, 'real'))
                    zs = [z1, z2]
                else:
                    zs = [z1]
        sos[si, :3] = _quadratic(zs)
        sos[si, 3:] = _quadratic(ps)
    assert len(p) == 0 and len(z) == 0
    sos[0, :3] *= k
    return sos


def butter_sos(order, Wn, btype, fs):
    """Digital Butterworth filter as a float64 SOS table.

    Same arguments and error behaviour as
    ``scipy.signal.butter(order, Wn, btype, fs=fs, output='sos')`` for the three
    types the reference uses ('lowpass', 'highpass', 'bandpass'): a critical
    frequency outside ``0 < Wn < fs/2`` raises ``ValueError`` (which
    ``BufferedEnvelope.update`` catches, src/audian/bufferedenvelope.py:53-54).
    """
    order = int(order)
    if order < 1:
        raise ValueError('Filter order must be a positive integer')
    fs = float(fs)
    wn = 2*np.atleast_1d(np.asarray(Wn, dtype=float))/fs
    if not np.all(wn > 0) or not np.all(wn < 1):
        raise ValueError('Digital filter critical frequencies must be 0 < Wn < fs/2 '
                         f'(fs={fs} -> fs/2={fs/2})')
    btype = {'low': 'lowpass', 'lp': 'lowpass', 'high': 'highpass', 'hp': 'highpass',
             'band': 'bandpass', 'bp': 'bandpass'}.get(btype, btype)
    warped = 4.0*np.tan(np.pi*wn/2.0)
    p = _prototype_poles(order)
    if btype in ('lowpass', 'highpass'):
        if len(warped) != 1:
            raise ValueError('Must specify a single critical frequency Wn for '
                             'lowpass or highpass filter')
        z, p, k = (_to_lowpass if btype == 'lowpass' else _to_highpass)(p, 1.0, warped[0])
    elif btype == 'bandpass':
        if len(warped) != 2:
            raise ValueError('Wn must specify start and stop frequencies for bandpass filter')
        if not wn[0] < wn[1]:
            raise ValueError('Wn[0] must be less than Wn[1]')
        z, p, k = _to_bandpass(p, 1.0, np.sqrt(warped[0]*warped[1]), warped[1] - warped[0])
    else:
        raise ValueError(f"'{btype}' is not a filter type used by audian")
    z, p, k = _bilinear(z, p, k)
    return zpk_to_sos(z, p, k)
