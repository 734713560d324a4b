"""Maximum sizes: ONE channel of more than 2**31 frames (6.2 hours at 96 kHz; 8.6 GB as float32) through every sweep --
where a 32-bit sample index, tile count or byte offset would wrap.  The oracle cannot filter two billion samples in
a test, so the check is a size-independent property: the input is PERIODIC (a block of P = 2**20 samples repeated),
so once the filters' transients are gone every output is periodic with P too -- and the first periods are checked
against the oracle.  A window behind frame 2**31 must therefore equal the same window a whole number of periods
earlier, which the oracle has vouched for."""

import numpy as np
import pytest

import gpu_helpers as gh
from conftest import rel_err

pytestmark = pytest.mark.gpu

P = 1 << 20
T = (1 << 31) + 50*2048 + 333         # not a multiple of the tile, the hop or the period


def test_one_channel_of_more_than_two_to_the_31_frames(oracle):
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rate, nfft, hop = 96000.0, 2048, 1024
    F, nd = nfft//2 + 1, (T + hop - 1)//hop
    rng = np.random.default_rng(2031)
    t = np.arange(P)/rate
    block = (0.5*rng.uniform(-1, 1, P) + 0.5*np.sin(2*np.pi*(rate*1000/P)*t)).astype(np.float32)   # 1000 whole cycles per block
    c = gh.ctx()
    dx = hipdsp.DeviceArray(c, (1, T), np.float32)
    hipdsp.lib.hipdsp_memcpy_h2d(c.handle, hipdsp._p(dx), block.ctypes.data, block.nbytes)
    have = P
    while have < T:                                        # doubling copies on the device
        n = min(have, T - have)
        hipdsp.memcpy2d(c, dx.view(have, (n,)), n*4, dx, n*4, n*4, 1)
        have += n
    c.synchronize()
    probe = dx.view(T - 5000, (5000,)).to_host()
    assert np.array_equal(probe, np.resize(block, T + P)[(T - 5000) % P:][:5000])
    sos, esos = butter_sos(2, (300.0, 3000.0), 'bandpass', rate), butter_sos(2, 20.0, 'lowpass', rate)
    fplan, eplan = hipdsp.SosPlan(c, sos), hipdsp.SosPlan(c, esos)
    yf, ye = hipdsp.DeviceArray(c, (1, T), np.float32), hipdsp.DeviceArray(c, (1, T), np.float32)
    ps, db = hipdsp.DeviceArray(c, (1, nd, F), np.float32), hipdsp.DeviceArray(c, (1, nd, F), np.float32)
    hipdsp.chain_forward(c, fplan, eplan, dx, T, yf, T, 1, T, nfft, hop, rate, ps, nd, db_out=db)
    hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, yf, T, ye, T, 1, T, phase=2)
    c.synchronize()

    # (1) the first periods against the oracle (band-pass, envelope of the first 3 P frames alone differs at its END only,
    #     where sosfiltfilt's backward pass starts: compare the second period)
    head = np.resize(block, 3*P).astype(np.float64)[:, None]
    want_f = oracle.sosfilt(sos, head)[:, 0]
    got_f = yf.view(0, (3*P,)).to_host()
    assert rel_err(got_f, want_f) < 1e-4
    want_e = np.zeros((3*P, 1))
    oracle.envelope_process(esos, got_f.astype(np.float64)[:, None], want_e, 0)
    got_e = ye.view(0, (3*P,)).to_host()
    assert rel_err(got_e[P:2*P], want_e[P:2*P, 0]) < 1e-4
    want_s = np.zeros((3*P//hop, 1, F))
    oracle.spectrogram_process(got_f.astype(np.float64)[:, None], want_s, rate, nfft, hop)
    got_s = ps.view(0, (3*P//hop, F)).to_host()
    for k in range(P//hop, 2*P//hop):
        assert rel_err(got_s[k], want_s[k, 0]) < 1e-4, k

    # (2) windows behind frame 2**31 (and in the middle) equal the vouched-for second period, whole periods earlier
    W = 6*2048
    for n0 in ((1 << 31) + 1111, (1 << 31) - 3000, 1500*P + 77, T - W - 40000):
        back = ((n0 - P)//P)*P                              # lands in [P, 2P)
        for arr, name in ((yf, 'filtered'), (ye, 'envelope')):
            a, b = arr.view(n0, (W,)).to_host(), arr.view(n0 - back, (W,)).to_host()
            assert np.isfinite(a).all() and np.abs(a).max() > 1e-3, (name, n0)
            assert np.abs(a - b).max()/np.abs(b).max() < 2e-6, (name, n0, np.abs(a - b).max())
        k0, kb = n0//hop, (n0 - back)//hop                   # back is a multiple of hop
        a, b = ps.view(k0*F, (4, F)).to_host(), ps.view(kb*F, (4, F)).to_host()
        assert np.abs(a - b).max()/np.abs(b).max() < 1e-5, ('PSD', n0)
        a, b = db.view(k0*F, (4, F)).to_host(), db.view(kb*F, (4, F)).to_host()
        # (bins more than six decades under their frame's peak are rounding noise of the float32 transform, down to the
        # floor of -200 dB = 1e-20, and the two windows' frames do not share their rounding: the pivot of a frame's mean is
        # the mean of the frame before it.  The PSD comparison above is the one with teeth; here: the dB image agrees
        # wherever it means something and stays down where it does not.)
        for j, (ra, rb) in enumerate(zip(a, b)):
            clear = np.isfinite(rb) & (rb > rb[np.isfinite(rb)].max() - 60.0)
            assert np.isfinite(ra[clear]).all() and np.abs(ra[clear] - rb[clear]).max() < 1e-2, ('dB', n0)
            assert np.all(ra[~clear] < rb[np.isfinite(rb)].max() - 55.0), ('dB under the noise', n0)
            # ... and the relaxation is pinned to a reason, not to a number: the bins it exempts are that far down in the
            # float64 oracle's spectrum of the same (vouched-for) frame too, where it has a spectrum there at all
            wdb = oracle.decibel(want_s[kb + j, 0])
            wfin = np.isfinite(wdb)
            assert np.all(wdb[~clear & wfin] < wdb[wfin].max() - 55.0), ('exempt bins are not small in the oracle', n0, j)
    # the last frames: valid ones finite and non-zero, the zero tail zero
    tail = ps.view((nd - 4)*F, (4, F)).to_host()
    n_valid = (min((nd - 1)*hop + nfft, T) - (nfft - hop))//hop
    for j in range(4):
        k = nd - 4 + j
        assert (np.all(tail[j] == 0) if k >= n_valid else (np.isfinite(tail[j]).all() and tail[j].max() > 0)), k

    # (3) the separate sweeps on the same trace: hipdsp_sosfilt (with nbefore), hipdsp_spectrogram, hipdsp_envelope
    skip = 12345
    y2 = hipdsp.DeviceArray(c, (1, T - skip), np.float32)
    hipdsp.sosfilt(c, fplan, dx, T, y2, T - skip, 1, T, skip)
    for n0 in ((1 << 31) + 1111, T - W):
        a, b = y2.view(n0 - skip, (W,)).to_host(), yf.view(n0, (W,)).to_host()
        assert np.abs(a - b).max()/np.abs(b).max() < 1e-6, ('sosfilt', n0)
    del y2
    s2 = hipdsp.DeviceArray(c, (1, nd, F), np.float32)
    hipdsp.spectrogram(c, yf, T, 1, T, nfft, hop, rate, s2, nd)
    for k0 in (((1 << 31) + 1111)//hop, nd - 8):
        a, b = s2.view(k0*F, (8, F)).to_host(), ps.view(k0*F, (8, F)).to_host()
        assert np.abs(a - b).max() <= 1e-5*np.abs(b).max(), ('spectrogram', k0)
    del s2
    e2 = hipdsp.DeviceArray(c, (1, T), np.float32)
    hipdsp.envelope(c, eplan, yf, T, e2, T, 1, T, 0)
    for n0 in ((1 << 31) + 1111, T - W, 0):
        a, b = e2.view(n0, (W,)).to_host(), ye.view(n0, (W,)).to_host()
        assert np.abs(a - b).max()/np.abs(b).max() < 1e-6, ('envelope', n0)
