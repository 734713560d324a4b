"""Seeded random sweeps over slab length, nbefore, pitch, segmentation and filter design for the
IIR entry points: lengths cluster around multiples of the 2048-sample tile and of the sosfiltfilt
pad length, where the tile-border, odd-extension and prefetch paths change."""

import numpy as np
import pytest

import gpu_helpers as gh
from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
TILE = 2048


def draw_length(rng, lo=1):
    kind = rng.integers(0, 4)
    if kind == 0:
        return int(rng.integers(lo, 200))
    if kind == 1:
        return max(lo, int(rng.integers(1, 12))*TILE + int(rng.integers(-40, 41)))
    if kind == 2:
        return int(rng.integers(lo, 40000))
    return max(lo, int(rng.integers(40, 160))*TILE + int(rng.integers(-TILE, TILE)))


def draw_design(rng, rate, envelope):
    from audian_amd.design import butter_sos
    order = int(rng.integers(1, 5 if not envelope else 4))
    kind = rng.integers(0, 3)
    if kind == 0:
        return butter_sos(order, float(rng.uniform(5.0, 0.4*rate)), 'lowpass', rate)
    if kind == 1 and not envelope:
        return butter_sos(order, float(rng.uniform(20.0, 0.3*rate)), 'highpass', rate)
    lo = float(rng.uniform(20.0, 0.1*rate))
    order = min(order, 2)
    return butter_sos(order, (lo, float(rng.uniform(2*lo, 0.45*rate))), 'bandpass', rate)


@pytest.mark.parametrize('seed', range(24))
def test_random_envelope_cases(oracle, seed):
    from audian_amd import hipdsp
    rng = np.random.default_rng(1000 + seed)
    rate = float(rng.choice([8000.0, 44100.0, 96000.0, 192000.0]))
    sos = draw_design(rng, rate, envelope=True)
    edge = oracle.sosfiltfilt_edge(sos)
    T = draw_length(rng, lo=edge + 1)
    C = int(rng.integers(1, 6))
    skip = int(rng.choice([0, 0, 1, rng.integers(0, T + 1), min(T, TILE), min(T, TILE + 1)]))
    pitch_in, pitch_out = T + int(rng.integers(0, 9)), max(T - skip, 1) + int(rng.integers(0, 9))
    clamp, rectify = bool(rng.integers(0, 2)), bool(rng.integers(0, 4))
    x = (rng.standard_normal((C, T))*rng.uniform(0.1, 3.0)).astype(np.float32)
    c = gh.ctx()
    c.set_max_segments(int(rng.choice([0, 0, 1, 3])))
    try:
        dx = hipdsp.DeviceArray(c, (C, pitch_in), np.float32)
        host = np.zeros((C, pitch_in), dtype=np.float32)
        host[:, :T] = x
        hipdsp.lib.hipdsp_memcpy_h2d(c.handle, hipdsp._p(dx), host.ctypes.data, host.nbytes)
        dy = hipdsp.DeviceArray(c, (C, pitch_out), np.float32)
        hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(dy), 0x7f, 4*C*pitch_out)
        hipdsp.envelope(c, hipdsp.SosPlan(c, sos), dx, pitch_in, dy, pitch_out, C, T, skip, rectify=rectify,
                        gain=np.pi/2 if rectify else 1.0, clamp=clamp)
        got = dy.to_host()
    finally:
        c.set_max_segments(0)
    src = x.T.astype(np.float64)
    want = oracle.sosfiltfilt(sos, (np.pi/2)*np.abs(src) if rectify else src)
    if clamp:
        want[want < 0] = 0
    for ch in range(C):
        if T - skip > 0:
            ref = want[skip:, ch]
            scale = max(np.max(np.abs(want[:, ch])), 1e-30)
            assert np.max(np.abs(got[ch, :T - skip] - ref))/scale < TOL, (seed, T, skip, ch)
        # nothing is written past the row
        tail = got[ch, max(T - skip, 0):]
        assert np.all(tail.view(np.uint32) == 0x7f7f7f7f), (seed, 'wrote past the row')


@pytest.mark.parametrize('seed', range(12))
def test_random_filter_envelope_chain_cases(oracle, seed):
    from audian_amd import hipdsp
    rng = np.random.default_rng(5000 + seed)
    rate = float(rng.choice([44100.0, 96000.0]))
    fsos = draw_design(rng, rate, envelope=False)
    esos = draw_design(rng, rate, envelope=True)
    edge = oracle.sosfiltfilt_edge(esos)
    T = draw_length(rng, lo=edge + 1)
    C = int(rng.integers(1, 5))
    x = rng.standard_normal((T, C)).astype(np.float32)
    c = gh.ctx()
    dx = gh.to_planar(c, x)
    yf = hipdsp.DeviceArray(c, (C, T), np.float32)
    ye = hipdsp.DeviceArray(c, (C, T), np.float32)
    hipdsp.sosfilt_envelope(c, hipdsp.SosPlan(c, fsos), hipdsp.SosPlan(c, esos), dx, T, yf, T, ye, T, C, T)
    gf, ge = yf.to_host(), ye.to_host()
    want_f = oracle.sosfilt(fsos, x.astype(np.float64))
    want_e = np.zeros_like(want_f)
    oracle.envelope_process(esos, gf.T.astype(np.float64), want_e, 0)
    for ch in range(C):
        assert rel_err(gf[ch], want_f[:, ch]) < TOL, (seed, T, ch)
        if np.max(np.abs(want_e[:, ch])) > 0:
            assert rel_err(ge[ch], want_e[:, ch]) < TOL, (seed, T, ch)


@pytest.mark.parametrize('seed', range(20))
def test_random_spectrogram_cases(oracle, seed):
    """Random window length (every kernel family: generic, two- and three-stage, workgroup, four-step,
    direct DFT), hop,
    slab length and destination length, incl. destinations longer than the source supports (zero
    tail) and shorter (fewer frames than fit)."""
    rng = np.random.default_rng(9000 + seed)
    rate = float(rng.choice([22050.0, 96000.0]))
    family = rng.integers(0, 4)
    if family == 0:
        nfft = int(2**rng.integers(3, 8))                 # generic radix-2 (8, 16), short two-stage kernels
    elif family == 1:
        nfft = int(2**rng.integers(8, 13))                # register/LDS kernels
    elif family == 2:
        nfft = int(rng.choice([8192, 16384, 32768, 65536, 131072]))   # workgroup FFT, frame-on-chip FFT
    else:
        nfft = int(rng.integers(9, 600))                  # whatever the clamp can produce
    hop = int(rng.choice([nfft//2, max(nfft//4, 1), nfft, int(rng.integers(1, nfft + 1))]))
    hop = max(hop, 1)
    nframes = int(rng.integers(1, 40))
    if nfft <= 256 and (nfft & (nfft - 1)) == 0 and rng.integers(0, 2):
        nframes = int(rng.integers(40, 6000))             # long runs: the short windows stream through an LDS ring (spec_pack.h)
    elif 4096 <= nfft <= 32768 and rng.integers(0, 2):
        nframes = int(rng.integers(40, 120))              # several frames per workgroup run (spec_wgs.h)
    T = (nframes - 1)*hop + nfft + int(rng.integers(-nfft//2, hop + 3))
    T = max(T, 0)
    C = int(rng.integers(1, 4))
    nd = max(1, (T + hop - 1)//hop + int(rng.integers(-2, 3)))
    x = (rng.standard_normal((T, C)) + 0.3).astype(np.float32)
    shape = int(rng.integers(0, 5))
    if shape == 1 and T > 0:                              # a large offset under a small signal (raw data of a DC-coupled sensor)
        x = (np.float32(10.0**rng.uniform(0, 4)*rng.choice([-1, 1])) + np.float32(10.0**rng.uniform(-3, 0))*x).astype(np.float32)
    elif shape == 2 and T > 0:                            # a pulse train on or next to the frame borders (pulse-type fish, clicks)
        x = (np.float32(1e-3)*x).astype(np.float32)
        x[int(rng.integers(0, min(hop, T)))::hop*int(rng.integers(1, 4))] += np.float32(10.0**rng.uniform(-2, 0.5)*rng.choice([-1, 1]))
    elif shape == 3 and T > 0:                            # steps in the level at a few frame borders, up to 300 times the noise
        x = (np.float32(1e-3)*x).astype(np.float32)
        for _ in range(int(rng.integers(1, 4))):
            x[int(rng.integers(0, max(T//hop, 1)))*hop:] += np.float32(rng.uniform(-0.3, 0.3))
    want_db = bool(rng.integers(0, 2))
    got = gh.gpu_spectrogram(x, rate, nfft, hop, nd, want_db=want_db)
    if want_db:
        got, gdb = got
        wdb = oracle.decibel(got)
        fin = np.isfinite(wdb)
        assert np.array_equal(np.isfinite(gdb), fin) and np.all(gdb[~fin] == -np.inf), (seed, nfft, hop)
        if fin.any():
            assert np.max(np.abs(gdb[fin] - wdb[fin])) < 1e-3, (seed, nfft, hop)
    want = np.full((nd, C, nfft//2 + 1), 7.0)
    if oracle.spectrogram_process(x.astype(np.float64), want, rate, nfft, hop) is None:
        want[:] = 0
    for k in range(nd):
        for ch in range(C):
            peak = np.max(np.abs(want[k, ch]))
            if peak == 0:
                assert np.all(got[k, ch] == 0), (seed, nfft, hop, k)
            else:
                assert np.max(np.abs(got[k, ch] - want[k, ch]))/peak < TOL, (seed, nfft, hop, T, k, ch)


@pytest.mark.parametrize('seed', range(12))
def test_random_sosfilt_cases(oracle, seed):
    rng = np.random.default_rng(7000 + seed)
    rate = float(rng.choice([8000.0, 48000.0, 192000.0]))
    sos = draw_design(rng, rate, envelope=False)
    T = draw_length(rng)
    C = int(rng.integers(1, 6))
    skip = int(rng.choice([0, 1, rng.integers(0, T + 1), min(T, TILE)]))
    x = rng.standard_normal((T, C)).astype(np.float32)
    got = gh.gpu_sosfilt(sos, x, skip=skip, max_segments=int(rng.choice([0, 0, 1, 5])))
    want = oracle.sosfilt(sos, x.astype(np.float64))[skip:]
    assert got.shape == want.shape
    for ch in range(C):
        if len(want):
            scale = max(np.max(np.abs(oracle.sosfilt(sos, x[:, ch].astype(np.float64)))), 1e-30)
            assert np.max(np.abs(got[:, ch] - want[:, ch]))/scale < TOL, (seed, T, skip, ch)


@pytest.mark.parametrize('seed', range(10))
def test_random_pitched_buffers(oracle, seed):
    """Row pitches larger than the rows (ring-buffer mirrors are laid out like that): nothing outside
    the valid part of a row may be read into the result or written."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rng = np.random.default_rng(3000 + seed)
    rate = 48000.0
    nfft = int(rng.choice([64, 256, 1024, 2048, 300]))
    hop = int(rng.choice([nfft//2, nfft//4, nfft]))
    C = int(rng.integers(1, 4))
    T = int(rng.integers(nfft, 12*nfft))
    nd = (T + hop - 1)//hop
    F = nfft//2 + 1
    xp, op = T + int(rng.integers(1, 50)), nd*F + int(rng.integers(1, 70))
    x = rng.standard_normal((C, T)).astype(np.float32)
    host = np.full((C, xp), np.float32(1e30))                  # poison behind every row
    host[:, :T] = x
    c = gh.ctx()
    dx = hipdsp.DeviceArray.from_host(c, host)
    out = hipdsp.DeviceArray(c, (C, op), np.float32)
    hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(out), 0x7f, 4*C*op)
    hipdsp.spectrogram(c, dx, xp, C, T, nfft, hop, rate, out, nd, out_pitch=op)
    got = out.to_host()
    want = np.zeros((nd, C, F))
    if oracle.spectrogram_process(x.T.astype(np.float64), want, rate, nfft, hop) is None:
        want[:] = 0
    for ch in range(C):
        g = got[ch, :nd*F].reshape(nd, F)
        for k in range(nd):
            peak = np.max(np.abs(want[k, ch]))
            if peak == 0:
                assert np.all(g[k] == 0)
            else:
                assert np.max(np.abs(g[k] - want[k, ch]))/peak < TOL, (seed, nfft, hop, k)
        assert np.all(got[ch, nd*F:].view(np.uint32) == 0x7f7f7f7f), (seed, 'wrote past the row')
    # the same for sosfilt
    sos = butter_sos(2, (300.0, 3000.0), 'bandpass', rate)
    skip = int(rng.integers(0, 5))
    yp = T - skip + int(rng.integers(1, 40))
    y = hipdsp.DeviceArray(c, (C, yp), np.float32)
    hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(y), 0x7f, 4*C*yp)
    hipdsp.sosfilt(c, hipdsp.SosPlan(c, sos), dx, xp, y, yp, C, T, skip)
    gy = y.to_host()
    wy = oracle.sosfilt(sos, x.T.astype(np.float64))[skip:]
    for ch in range(C):
        assert rel_err(gy[ch, :T - skip], wy[:, ch]) < TOL
        assert np.all(gy[ch, T - skip:].view(np.uint32) == 0x7f7f7f7f)


def test_bad_arguments_are_refused_not_launched():
    """Every entry point checks its arguments on the host: an error status, never a faulting kernel."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    c = gh.ctx()
    x = hipdsp.DeviceArray(c, (2, 5000), np.float32).zero_()
    y = hipdsp.DeviceArray(c, (2, 5000), np.float32)
    s = hipdsp.DeviceArray(c, (2, 40, 129), np.float32)
    plan = hipdsp.SosPlan(c, butter_sos(2, 1000.0, 'lowpass', 48000.0))
    for call in (
            lambda: hipdsp.sosfilt(c, plan, x, 4000, y, 5000, 2, 5000, 0),          # pitch < frames
            lambda: hipdsp.sosfilt(c, plan, x, 5000, y, 5000, 2, 5000, 5001),       # skip > frames
            lambda: hipdsp.sosfilt(c, plan, x, 5000, y, 100, 2, 5000, 0),           # output pitch too small
            lambda: hipdsp.sosfilt(c, plan, x, 5000, y, 5000, -1, 5000, 0),
            lambda: hipdsp.envelope(c, plan, x, 5000, y, 5000, 2, 5000, -1),
            lambda: hipdsp.envelope(c, plan, x, 5000, y, 5000, 2, 9, 0),            # not longer than padlen
            lambda: hipdsp.spectrogram(c, x, 5000, 2, 5000, 4, 2, 48000.0, s, 40),  # nfft < 8
            lambda: hipdsp.spectrogram(c, x, 5000, 2, 5000, 256, 0, 48000.0, s, 40),
            lambda: hipdsp.spectrogram(c, x, 5000, 2, 5000, 256, 300, 48000.0, s, 40),   # hop > nfft
            lambda: hipdsp.spectrogram(c, x, 5000, 2, 5000, 256, 128, -1.0, s, 40),
            lambda: hipdsp.spectrogram(c, x, 100, 2, 5000, 256, 128, 48000.0, s, 40),    # pitch < frames
            lambda: hipdsp.spectrogram(c, x, 5000, 2, 5000, 256, 128, 48000.0, s, 40, out_pitch=10),
            lambda: hipdsp.minmax_decimate(c, x, 5000, 2, 10, 5, 3, y, 5000),       # stop < start
            lambda: hipdsp.minmax_decimate(c, x, 5000, 2, 0, 5000, 0, y, 5000),     # step < 1
            lambda: hipdsp.decibel_image_decimate(c, s, y, 40, 129, 0, 41, 2),
            lambda: hipdsp.mean_spectrum_db(c, s, 129, 5, 5, y),                    # empty frame range
            lambda: hipdsp.decibel(c, x, y, 100, ref_power=0.0),
            lambda: hipdsp.sosfilt(c, plan, x, 5000, x, 5000, 2, 5000, 0),          # in place: segments race with each other's warm-up
            lambda: hipdsp.sosfilt(c, plan, x, 5000, x.view(4000, (1, 5000)), 5000, 1, 5000, 0),   # partial overlap
            lambda: hipdsp.sosfilt(c, None, x, 5000, x, 5000, 2, 5000, 0),          # the pass-through copy too
            lambda: hipdsp.envelope(c, plan, x, 5000, x, 5000, 2, 5000, 0),
            lambda: hipdsp.sosfilt_envelope(c, plan, plan, x, 5000, x, 5000, y, 5000, 2, 5000),
            lambda: hipdsp.sosfilt_envelope(c, plan, plan, x, 5000, y, 5000, y, 5000, 2, 5000),
    ):
        with pytest.raises((ValueError, IndexError)):
            call()
    with pytest.raises(NotImplementedError):
        hipdsp.spectrogram(c, x, 5000, 2, 5000, 1 << 20, 1 << 19, 48000.0, s, 1)
    with pytest.raises(NotImplementedError):
        hipdsp.SosPlan(c, np.tile(butter_sos(2, 1000.0, 'lowpass', 48000.0), (5, 1)))
    with pytest.raises(ValueError):
        hipdsp.SosPlan(c, np.array([[1.0, 0.0, 0.0, 2.0, 0.0, 0.0]]))
    c.synchronize()                                     # the context is still healthy
    hipdsp.sosfilt(c, plan, x, 5000, y, 5000, 2, 5000, 0)
    assert np.all(y.to_host() == 0)


@pytest.mark.parametrize('seed', range(16))
def test_random_chain_forward_cases(oracle, seed):
    """hipdsp_chain_forward + backward sweep under random trace lengths (around tile multiples),
    channel counts, pitches, segmentations, one- and two-section plans: everything against the
    oracle, the PSD also against the separate spectrogram call on the same filtered trace."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rng = np.random.default_rng(7000 + seed)
    rate = float(rng.choice([44100.0, 96000.0, 192000.0]))
    nfft, hop = 2048, 1024
    F = nfft//2 + 1
    T = int(rng.integers(4, 60))*TILE + int(rng.integers(-TILE + 1, TILE)) if rng.integers(0, 3) else \
        int(rng.integers(4*TILE, 6*TILE))
    T = max(T, 4*TILE)
    C = int(rng.integers(1, 7))
    lo = float(rng.uniform(50.0, 0.05*rate))
    sos = butter_sos(int(rng.integers(1, 3)), (lo, float(rng.uniform(2*lo, 0.4*rate))), 'bandpass', rate)
    esos = butter_sos(int(rng.integers(1, 5)), float(rng.uniform(5.0, 2000.0)), 'lowpass', rate)
    xp, fp = T + int(rng.integers(0, 9)), T + int(rng.integers(0, 9))
    nd = (T + hop - 1)//hop + int(rng.integers(-3, 4))
    pp = (nd + int(rng.integers(0, 3)))*F
    x = (rng.standard_normal((C, T))*rng.uniform(0.1, 3.0) + 0.1).astype(np.float32)
    c = gh.ctx()
    c.set_max_segments(int(rng.choice([0, 0, 1, 2, 5])))
    c.set_option('chain_debug', int(rng.choice([0, 4])))
    try:
        host = np.zeros((C, xp), dtype=np.float32)
        host[:, :T] = x
        dx = hipdsp.DeviceArray(c, (C, xp), np.float32)
        hipdsp.lib.hipdsp_memcpy_h2d(c.handle, hipdsp._p(dx), host.ctypes.data, host.nbytes)
        yf = hipdsp.DeviceArray(c, (C, fp), np.float32)
        ye = hipdsp.DeviceArray(c, (C, T), np.float32)
        ps = hipdsp.DeviceArray(c, (C*pp,), np.float32)
        hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(ps), 0x7f, 4*C*pp)
        fplan, eplan = hipdsp.SosPlan(c, sos), hipdsp.SosPlan(c, esos)
        hipdsp.chain_forward(c, fplan, eplan, dx, xp, yf, fp, C, T, nfft, hop, rate, ps, nd, psd_pitch=pp)
        hipdsp.sosfilt_envelope(c, fplan, eplan, dx, xp, yf, fp, ye, T, C, T, phase=2)
        s1 = hipdsp.DeviceArray(c, (C*pp,), np.float32)
        hipdsp.spectrogram(c, yf, fp, C, T, nfft, hop, rate, s1, nd, out_pitch=pp)
        gf = yf.to_host()[:, :T]
        ge = ye.to_host()
        gs = ps.to_host().reshape(C, pp)[:, :nd*F].reshape(C, nd, F)
        ss = s1.to_host().reshape(C, pp)[:, :nd*F].reshape(C, nd, F)
        guard = np.frombuffer(b'\x7f\x7f\x7f\x7f', dtype=np.float32)[0]
        assert np.all(ps.to_host().reshape(C, pp)[:, nd*F:] == guard)
        want_f = oracle.sosfilt(sos, x.T.astype(np.float64))
        want_e = np.zeros_like(want_f)
        oracle.envelope_process(esos, gf.T.astype(np.float64), want_e, 0)
        for ch in range(C):
            assert rel_err(gf[ch], want_f[:, ch]) < TOL, (seed, ch)
            assert rel_err(ge[ch], want_e[:, ch]) < TOL, (seed, ch)
            for j in range(nd):
                peak = np.max(np.abs(ss[ch, j]))
                if peak == 0:
                    assert np.all(gs[ch, j] == 0), (seed, j, ch)
                else:
                    assert np.max(np.abs(gs[ch, j] - ss[ch, j]))/peak < 1e-5, (seed, T, j, ch)
    finally:
        c.set_max_segments(0)
        c.set_option('chain_debug', 0)


@pytest.mark.parametrize('seed', range(24))
def test_random_scroll_positions_of_the_fused_launch(oracle, seed):
    """hipdsp_chain_forward's spec_first / env_first (the filtered buffer after a scroll: frame 0 of the spectrogram
    inside its first hop -- or anywhere --, the envelope from a later sample to the end) under random draws: every
    window shape, random offsets with a bias towards tile, lane-row and hop borders, spec_frames, pitches, channel
    counts, segmentations, barrier / flag hand-over, band-pass and low-pass envelopes.  Filtered trace against the
    oracle, PSD against the separate kernel on the sliced filtered trace, envelope against the oracle on the slice."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rng = np.random.default_rng(424200 + seed)
    rate = float(rng.choice([44100.0, 48000.0, 96000.0]))
    nfft, hop = [(2048, 1024), (2048, 512), (1024, 512), (1024, 256), (512, 256), (256, 128)][int(rng.integers(0, 6))]
    F = nfft//2 + 1
    T = max(4*TILE + 50, int(rng.integers(5, 40))*TILE + int(rng.integers(-TILE + 1, TILE)))
    C = int(rng.integers(1, 5))

    def near_border(limit):
        kind = int(rng.integers(0, 5))
        if kind == 0:
            return 0
        base = int(rng.integers(0, max(1, limit)))
        if kind == 1:
            base = (base//TILE)*TILE + int(rng.integers(-20, 21))
        elif kind == 2:
            base = (base//32)*32 + int(rng.integers(-1, 2))
        elif kind == 3:
            base = (base//hop)*hop + int(rng.integers(-2, 3))
        return int(min(max(base, 0), max(0, limit - 1)))
    spec_first = near_border(min(T - nfft - 1, 3*hop)) if rng.integers(0, 4) else near_border(T - nfft - 1)
    lo = float(rng.uniform(50.0, 0.05*rate))
    sos = butter_sos(int(rng.integers(1, 3)), (lo, float(rng.uniform(2*lo, 0.4*rate))), 'bandpass', rate)
    band_env = rng.integers(0, 4) == 0
    esos = butter_sos(1, (float(rng.uniform(2.0, 20.0)), float(rng.uniform(100.0, 1000.0))), 'bandpass', rate) if band_env \
        else butter_sos(int(rng.integers(1, 5)), float(rng.uniform(5.0, 2000.0)), 'lowpass', rate)
    edge = oracle.sosfiltfilt_edge(esos)
    env_first = near_border(T - edge - 2)
    nsrc = T - spec_first
    spec_frames = 0 if rng.integers(0, 2) else int(rng.integers(max(1, nsrc//2), nsrc + 1))
    xp, fp, ep = T + int(rng.integers(0, 9)), T + int(rng.integers(0, 9)), T - env_first + int(rng.integers(0, 9))
    nd = (nsrc + hop - 1)//hop + int(rng.integers(-2, 3))
    nd = max(nd, 1)
    x = (rng.standard_normal((C, T))*rng.uniform(0.1, 3.0) + 0.1).astype(np.float32)
    c = gh.ctx()
    c.set_max_segments(int(rng.choice([0, 0, 1, 3, 11])))
    c.set_option('chain_debug', int(rng.choice([0, 0, 4])) if (nfft, hop) == (2048, 1024) and len(sos) <= 2 else 0)
    try:
        host = np.zeros((C, xp), dtype=np.float32)
        host[:, :T] = x
        dx = hipdsp.DeviceArray(c, (C, xp), np.float32)
        hipdsp.lib.hipdsp_memcpy_h2d(c.handle, hipdsp._p(dx), host.ctypes.data, host.nbytes)
        yf = hipdsp.DeviceArray(c, (C, fp), np.float32)
        ye = hipdsp.DeviceArray(c, (C, ep), np.float32)
        ps = hipdsp.DeviceArray(c, (C, nd, F), np.float32)
        for arr, n in ((yf, C*fp), (ye, C*ep), (ps, C*nd*F)):
            hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(arr), 0x7f, 4*n)
        fplan, eplan = hipdsp.SosPlan(c, sos), hipdsp.SosPlan(c, esos)
        hipdsp.chain_forward(c, fplan, eplan, dx, xp, yf, fp, C, T, nfft, hop, rate, ps, nd, spec_frames=spec_frames,
                             spec_first=spec_first, env_first=env_first)
        hipdsp.sosfilt_envelope(c, fplan, eplan, dx, xp, yf, fp, ye, ep, C, T, clamp=not band_env, phase=2,
                                env_first=env_first)
        gf = yf.to_host()
        ge = ye.to_host()
        gs = ps.to_host()
        guard = np.frombuffer(b'\x7f\x7f\x7f\x7f', dtype=np.float32)[0]
        assert np.all(gf[:, T:] == guard) and np.all(ge[:, T - env_first:] == guard), seed
        gf, ge = gf[:, :T], ge[:, :T - env_first]
        s1 = hipdsp.DeviceArray(c, (C, nd, F), np.float32)
        hipdsp.spectrogram(c, yf.view(spec_first, (1,)), fp, C, spec_frames or nsrc, nfft, hop, rate, s1, nd)
        ss = s1.to_host()
        want_f = oracle.sosfilt(sos, x.T.astype(np.float64))
        src = gf.T[env_first:].astype(np.float64)
        want_e = oracle.sosfiltfilt(esos, (np.pi/2)*np.abs(src))
        if not band_env:
            want_e[want_e < 0] = 0
        what = (seed, T, nfft, hop, spec_first, env_first, spec_frames)
        for ch in range(C):
            assert rel_err(gf[ch], want_f[:, ch]) < TOL, what
            assert np.all(np.isfinite(ge[ch])), what
            assert rel_err(ge[ch], want_e[:, ch]) < TOL, what
            for j in range(nd):
                peak = np.max(np.abs(ss[ch, j]))
                if peak == 0:
                    assert np.all(gs[ch, j] == 0), what + (j,)
                else:
                    assert np.max(np.abs(gs[ch, j] - ss[ch, j]))/peak < 1e-5, what + (j,)
    finally:
        c.set_max_segments(0)
        c.set_option('chain_debug', 0)


@pytest.mark.parametrize('seed', range(20))
def test_random_chain_shapes_sections_and_modes(oracle, seed):
    """The round-2 generalisations of the fused sweep under random draws: every window shape the kernel is built
    for, band-passes of 1-4 sections, envelopes of 1-2 sections, dB epilogue (2048/1024), frame split with the
    role-split backward sweep (2048/1024), barrier / flag hand-over and every priority mode, random lengths around
    tile multiples, pitches, channel counts and segmentations.  Filtered trace and envelope against the oracle, the
    PSD against the oracle's spectrogram of the SAME filtered trace and against the separate kernel."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rng = np.random.default_rng(91000 + seed)
    rate = float(rng.choice([44100.0, 48000.0, 96000.0, 192000.0]))
    nfft, hop = [(2048, 1024), (2048, 512), (1024, 512), (1024, 256), (512, 256), (256, 128)][int(rng.integers(0, 6))]
    F = nfft//2 + 1
    T = int(rng.integers(4, 40))*TILE + int(rng.integers(-TILE + 1, TILE)) if rng.integers(0, 3) else \
        int(rng.integers(4*TILE, 6*TILE))
    T = max(T, 4*TILE)
    C = int(rng.integers(1, 6))
    lo = float(rng.uniform(50.0, 0.05*rate))
    sos = butter_sos(int(rng.integers(1, 5)), (lo, float(rng.uniform(2*lo, 0.4*rate))), 'bandpass', rate)
    esos = butter_sos(int(rng.integers(1, 5)), float(rng.uniform(5.0, 2000.0)), 'lowpass', rate)
    xp, fp = T + int(rng.integers(0, 9)), T + int(rng.integers(0, 9))
    nd = (T + hop - 1)//hop + int(rng.integers(-3, 4))
    pp = (nd + int(rng.integers(0, 3)))*F
    want_db = (nfft, hop) == (2048, 1024) and rng.integers(0, 3) == 0
    split = (nfft, hop) == (2048, 1024) and not want_db and rng.integers(0, 3) == 0
    x = (rng.standard_normal((C, T))*rng.uniform(0.1, 3.0) + rng.uniform(-0.5, 0.5)).astype(np.float32)
    c = gh.ctx()
    c.set_max_segments(int(rng.choice([0, 0, 1, 2, 5])))
    dbg = int(rng.choice([0, 0, 4, 64, 128]))
    if dbg == 4 and ((nfft, hop) != (2048, 1024) or len(sos) > 2):
        dbg = 0                      # (the barrier variant is built for the first shape and two sections only)
    c.set_option('chain_debug', dbg)
    c.set_option('chain_split_frames', int(split))
    try:
        host = np.zeros((C, xp), dtype=np.float32)
        host[:, :T] = x
        dx = hipdsp.DeviceArray(c, (C, xp), np.float32)
        hipdsp.lib.hipdsp_memcpy_h2d(c.handle, hipdsp._p(dx), host.ctypes.data, host.nbytes)
        yf = hipdsp.DeviceArray(c, (C, fp), np.float32)
        ye = hipdsp.DeviceArray(c, (C, T), np.float32)
        ps = hipdsp.DeviceArray(c, (C*pp,), np.float32)
        hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(ps), 0x7f, 4*C*pp)
        db = hipdsp.DeviceArray(c, (C*pp,), np.float32) if want_db else None
        fplan, eplan = hipdsp.SosPlan(c, sos), hipdsp.SosPlan(c, esos)
        hipdsp.chain_forward(c, fplan, eplan, dx, xp, yf, fp, C, T, nfft, hop, rate, ps, nd, psd_pitch=pp, db_out=db)
        if split:
            hipdsp.chain_backward(c, eplan, yf, fp, ye, T, C, T, nfft, hop, rate, ps, nd, psd_pitch=pp)
        else:
            hipdsp.sosfilt_envelope(c, fplan, eplan, dx, xp, yf, fp, ye, T, C, T, phase=2)
        c.set_option('chain_split_frames', 0)
        s1 = hipdsp.DeviceArray(c, (C*pp,), np.float32)
        hipdsp.spectrogram(c, yf, fp, C, T, nfft, hop, rate, s1, nd, out_pitch=pp)
        gf = yf.to_host()[:, :T]
        ge = ye.to_host()
        gs = ps.to_host().reshape(C, pp)[:, :nd*F].reshape(C, nd, F)
        ss = s1.to_host().reshape(C, pp)[:, :nd*F].reshape(C, nd, F)
        guard = np.frombuffer(b'\x7f\x7f\x7f\x7f', dtype=np.float32)[0]
        assert np.all(ps.to_host().reshape(C, pp)[:, nd*F:] == guard)
        want_f = oracle.sosfilt(sos, x.T.astype(np.float64))
        want_e = np.zeros_like(want_f)
        oracle.envelope_process(esos, gf.T.astype(np.float64), want_e, 0)
        want_s = np.zeros((nd, C, F))
        oracle.spectrogram_process(gf.T.astype(np.float64), want_s, rate, nfft, hop)
        what = (seed, nfft, hop, len(sos), len(esos), T, C, bool(want_db), bool(split))
        for ch in range(C):
            assert rel_err(gf[ch], want_f[:, ch]) < TOL, what + (ch,)
            assert rel_err(ge[ch], want_e[:, ch]) < TOL, what + (ch,)
            for j in range(nd):
                peak = np.max(np.abs(want_s[j, ch]))
                if peak == 0:
                    assert np.all(gs[ch, j] == 0), what + (j, ch)
                else:
                    assert np.max(np.abs(gs[ch, j] - want_s[j, ch]))/peak < TOL, what + (j, ch)
                    assert np.max(np.abs(gs[ch, j] - ss[ch, j]))/peak < 1e-5, what + (j, ch)
        if want_db:
            gdb = db.to_host().reshape(C, pp)[:, :nd*F].reshape(C, nd, F)
            wdb = oracle.decibel(gs)
            fin = np.isfinite(wdb)
            assert np.array_equal(np.isfinite(gdb), fin), what
            assert np.max(np.abs(gdb[fin] - wdb[fin])) < 1e-3, what
    finally:
        c.set_max_segments(0)
        c.set_option('chain_debug', 0)
        c.set_option('chain_split_frames', 0)


@pytest.mark.parametrize('seed', range(10))
def test_random_envelope_multi_cases(oracle, seed):
    """hipdsp_envelope_multi under random orders (1-12), cut-offs, splits of the cascade over plans, lengths around
    tile multiples and the pad length, nbefore, pitches and channel counts."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rng = np.random.default_rng(31000 + seed)
    rate = float(rng.choice([8000.0, 48000.0, 96000.0]))
    if rng.integers(0, 2):
        ehp = float(rng.uniform(1.0, 0.01*rate))
        sos = butter_sos(int(rng.integers(1, 7)), (ehp, float(rng.uniform(4*ehp, 0.4*rate))), 'bandpass', rate)
    else:
        ehp = 0.0
        sos = butter_sos(int(rng.integers(1, 13)), float(rng.uniform(0.002*rate, 0.4*rate)), 'lowpass', rate)
    split, left = [], len(sos)
    while left > 0:
        n = int(rng.integers(1, min(4, left) + 1))
        split.append(n); left -= n
    edge = oracle.sosfiltfilt_edge(sos)
    T = draw_length(rng, lo=1)
    C = int(rng.integers(1, 5))
    skip = int(rng.choice([0, 0, 1, rng.integers(0, T + 1)]))
    x = (rng.standard_normal((T, C))*rng.uniform(0.1, 3.0)).astype(np.float32)
    c = gh.ctx()
    dx = gh.to_planar(c, x)
    plans, i = [], 0
    for n in split:
        plans.append(hipdsp.SosPlan(c, sos[i:i + n]))
        i += n
    po = max(T - skip, 1) + int(rng.integers(0, 5))
    dy = hipdsp.DeviceArray(c, (C, po), np.float32)
    hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(dy), 0x7f, 4*C*po)
    what = (seed, T, len(sos), tuple(split), skip)
    if T <= edge:
        with pytest.raises(ValueError, match='padlen'):
            hipdsp.envelope_multi(c, plans, dx, T, dy, po, C, T, skip, clamp=ehp == 0)
        return
    hipdsp.envelope_multi(c, plans, dx, T, dy, po, C, T, skip, clamp=ehp == 0)
    got = dy.to_host()
    want = np.zeros((T - skip, C))
    oracle.envelope_process(sos, x.astype(np.float64), want, skip, highpass_cutoff=ehp)
    full = np.zeros((T, C))
    oracle.envelope_process(sos, x.astype(np.float64), full, 0, highpass_cutoff=ehp)
    for ch in range(C):
        if T - skip > 0:
            err = np.max(np.abs(got[ch, :T - skip] - want[:, ch]))
            assert err <= TOL*max(np.max(np.abs(full[:, ch])), 1e-30) + \
                4*np.finfo(np.float32).eps*(np.pi/2)*np.max(np.abs(x[:, ch])), what + (ch,)
        assert np.all(got[ch, max(T - skip, 0):].view(np.uint32) == 0x7f7f7f7f), what + ('wrote past the row',)


@pytest.mark.parametrize('seed', range(10))
def test_random_unwrap_cases(oracle, seed):
    """hipdsp_unwrap against the restated algorithm, bit for bit: random lengths around the 16384-sample chunks,
    wrap counts from none to many per chunk, thresholds, clipping and down-scaling, pitches."""
    from audian_amd import hipdsp
    rng = np.random.default_rng(32000 + seed)
    T = int(rng.choice([1, 2, int(rng.integers(3, 400)), 16384 + int(rng.integers(-3, 4)),
                        int(rng.integers(1, 9))*16384 + int(rng.integers(-50, 51)), int(rng.integers(1000, 300000))]))
    C = int(rng.integers(1, 5))
    t = np.arange(T)
    amp = float(rng.uniform(0.5, 6.0))
    true = np.stack([amp*np.sin(2*np.pi*float(rng.uniform(1e-5, 2e-3))*t + float(rng.uniform(0, 6.28))*(ch > 0)) +
                     float(rng.uniform(0, 0.2))*rng.standard_normal(T) for ch in range(C)], axis=1)
    wrapped = ((true + 1.0) % 2.0 - 1.0).astype(np.float32)
    px, py = T + int(rng.integers(0, 5)), T + int(rng.integers(0, 5))
    host = np.zeros((C, px), dtype=np.float32)
    host[:, :T] = wrapped.T
    c = gh.ctx()
    dx = hipdsp.DeviceArray(c, (C, px), np.float32)
    hipdsp.lib.hipdsp_memcpy_h2d(c.handle, hipdsp._p(dx), host.ctypes.data, host.nbytes)
    thresh = float(rng.choice([0.5, 1.0, 1.5, 1.9]))
    clips, down = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    dy = hipdsp.DeviceArray(c, (C, py), np.float32)
    hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(dy), 0x7f, 4*C*py)
    hipdsp.unwrap(c, dx, px, C, T, thresh, dy, py, clips=clips, down_scale=down)
    got = dy.to_host()
    want = oracle.unwrap(wrapped, thresh, clips=clips, down_scale=down)
    assert np.array_equal(got[:, :T].T, want), (seed, T, thresh, clips, down)
    assert np.all(got[:, T:].view(np.uint32) == 0x7f7f7f7f), (seed, 'wrote past the row')


@pytest.mark.parametrize('seed', range(10))
def test_random_screen_reductions(oracle, seed):
    """The SURVEY 8f rows under random shapes: min/max decimation of traces (bit-exact), decimated dB image and
    mean spectrum of a spectrogram slab, band order statistics (bit-exact), PCM ingest (bit-exact)."""
    from audian_amd import hipdsp
    rng = np.random.default_rng(33000 + seed)
    c = gh.ctx()
    # min/max decimation
    T, C = int(rng.integers(1, 200000)), int(rng.integers(1, 5))
    x = rng.standard_normal((T, C)).astype(np.float32)
    dx = gh.to_planar(c, x)
    for _ in range(3):
        start = int(rng.integers(0, T))
        stop = int(rng.integers(start + 1, T + 1))
        step = int(rng.choice([1, 2, 3, int(rng.integers(1, 5000)), stop - start]))
        nseg = (stop - start + step - 1)//step
        out = hipdsp.DeviceArray(c, (C, 2*nseg + 3), np.float32)
        hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(out), 0x7f, 4*C*(2*nseg + 3))
        hipdsp.minmax_decimate(c, dx, T, C, start, stop, step, out, 2*nseg + 3)
        got = out.to_host()
        want = oracle.minmax_decimate(x.astype(np.float64), start, stop, step)
        assert np.array_equal(got[:, :2*nseg].astype(np.float64), want.T), (seed, T, start, stop, step)
        assert np.all(got[:, 2*nseg:].view(np.uint32) == 0x7f7f7f7f), (seed, 'wrote past the row')
    # spectrogram slab: decimated dB image, mean spectrum, band order statistics
    frames, F = int(rng.integers(1, 3000)), int(rng.choice([5, 33, 129, 513, 1025, 2049]))
    spec = (10.0**rng.uniform(-24, 2, size=(frames, F))).astype(np.float32)
    spec[rng.random((frames, F)) < 0.02] = 0.0
    ds = hipdsp.DeviceArray.from_host(c, spec)
    start = int(rng.integers(0, frames))
    stop = int(rng.integers(start + 1, frames + 1))
    step = int(rng.choice([1, 2, int(rng.integers(1, 200)), stop - start]))
    ncols = (stop - start + step - 1)//step
    img = hipdsp.DeviceArray(c, (F, ncols), np.float32)
    hipdsp.decibel_image_decimate(c, ds, img, frames, F, start, stop, step)
    got = img.to_host()
    want = oracle.decimated_db_image(spec[:, None, :], start, stop, step, 0)
    fin = np.isfinite(want)
    assert got.shape == want.shape and np.array_equal(np.isfinite(got), fin), (seed, frames, F, start, stop, step)
    if fin.any():
        assert np.max(np.abs(got[fin] - want[fin])) < 1e-4, (seed, frames, F)
    out = hipdsp.DeviceArray(c, (F,), np.float32)
    hipdsp.mean_spectrum_db(c, ds, F, start, stop, out)
    assert np.max(np.abs(out.to_host().astype(np.float64) - oracle.mean_power_db(spec[:, None, :], start, stop, 0))) < 1e-3
    cols = int(rng.integers(1, F + 1))
    band = spec[:, F - cols:]
    srt = np.sort(band.ravel())
    n = frames*cols
    o2 = hipdsp.DeviceArray(c, (2,), np.float32)
    for rank in sorted({0, int(np.floor(0.95*(n - 1))), int(rng.integers(0, n)), n - 1}):
        hipdsp.band_order_stats(c, ds.view(F - cols, (1,)), frames, cols, F, rank, o2)
        g2 = o2.to_host()
        assert g2[0] == srt[rank] and g2[1] == srt[min(rank + 1, n - 1)], (seed, frames, cols, rank)
    # PCM ingest
    nbytes = int(rng.choice([2, 3, 4]))
    Tp, Cp = int(rng.integers(1, 50000)), int(rng.choice([1, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32, 64]))   # (whole 16-byte vectors of channels: the tiled kernel)
    raw = rng.integers(0, 256, size=(Tp, Cp, nbytes), dtype=np.uint8)
    ints = np.zeros((Tp, Cp), dtype=np.int64)
    for b in range(nbytes):
        ints |= raw[:, :, b].astype(np.int64) << (8*b)
    ints = np.where(ints >= 1 << (8*nbytes - 1), ints - (1 << 8*nbytes), ints)
    scale = 1.0/float(1 << (8*nbytes - 1))
    up = hipdsp.DeviceArray.from_host(c, raw.reshape(-1))
    pitch = Tp + int(rng.integers(0, 4))
    dst = hipdsp.DeviceArray(c, (Cp, pitch), np.float32)
    hipdsp.pcm_unpack(c, up, nbytes, Tp, Cp, scale, dst, pitch)
    assert np.array_equal(dst.to_host()[:, :Tp], (ints.T.astype(np.float64)*scale).astype(np.float32)), (seed, nbytes)


def _same_non_finite(got, want, what, scale=None):
    """`scale`: the largest finite value of the whole trace where `want` is only a window of it (the parity metric is per
    channel, SURVEY 7-3: a window of ONE sample next to the end of a sosfiltfilt, five decades under the envelope's size,
    is not a channel -- tools/fuzz_stress.py seed 90035 drew nbefore = T - 1)."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    bad = ~np.isfinite(want)
    assert np.array_equal(~np.isfinite(got), bad), (what, int((~np.isfinite(got)).sum()), int(bad.sum()))
    if (~bad).any() and np.abs(want[~bad]).max() > 0:
        assert np.abs(got[~bad] - want[~bad]).max()/max(np.abs(want[~bad]).max(), scale or 0.0) < TOL, what


@pytest.mark.parametrize('seed', range(12))
def test_random_non_finite_cases(oracle, seed):
    """A few NaN / infinite samples at random places of random channels, random lengths, designs, pitches, nbefore,
    segmentations and spectrogram windows, through hipdsp_sosfilt, hipdsp_envelope and (long traces) the fused sweeps:
    non-finite exactly where the oracle is, the rest within tolerance."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rng = np.random.default_rng(9100 + seed)
    rate = float(rng.choice([44100.0, 96000.0, 192000.0]))
    long_trace = bool(rng.integers(0, 3))
    T = int(rng.integers(4, 120))*TILE + int(rng.integers(-TILE + 1, TILE)) if long_trace else draw_length(rng, lo=64)
    T = max(T, 4*TILE) if long_trace else T
    C = int(rng.integers(1, 6))
    x = (rng.standard_normal((T, C))*rng.uniform(0.1, 3.0)).astype(np.float32)
    for _ in range(int(rng.integers(1, 4))):
        x[int(rng.integers(0, T)), int(rng.integers(0, C))] = rng.choice([np.nan, np.nan, np.inf, -np.inf])
    x64 = x.astype(np.float64)
    sos = draw_design(rng, rate, False)
    esos = draw_design(rng, rate, True)
    c = gh.ctx()
    opts = [{}, {'max_segments': 1}, {'max_segments': 3}, {'n_cus': 1024, 'sos_waves_per_cu': 16, 'sos_waves_min': 16}][int(rng.integers(0, 4))]
    try:
        for k, v in opts.items():
            c.set_option(k, v)
        skip = int(rng.integers(0, T)) if rng.integers(0, 2) else 0
        dx = gh.to_planar(c, x)
        plan, eplan = hipdsp.SosPlan(c, sos), hipdsp.SosPlan(c, esos)
        dy = hipdsp.DeviceArray(c, (C, max(T - skip, 1)), np.float32)
        hipdsp.sosfilt(c, plan, dx, T, dy, max(T - skip, 1), C, T, skip)
        want_f = oracle.sosfilt(sos, x64)
        _same_non_finite(dy.to_host()[:, :T - skip].T, want_f[skip:], (seed, 'sosfilt', skip, opts))
        edge = oracle.sosfiltfilt_edge(esos)
        if T > edge:
            eskip = int(rng.integers(0, T)) if rng.integers(0, 2) else 0
            de = hipdsp.DeviceArray(c, (C, max(T - eskip, 1)), np.float32)
            hipdsp.envelope(c, eplan, dx, T, de, max(T - eskip, 1), C, T, eskip)
            want_full = np.zeros((T, C))
            oracle.envelope_process(esos, x64, want_full, 0)
            want_e = want_full[eskip:]
            fin = np.isfinite(want_full)
            _same_non_finite(de.to_host()[:, :T - eskip].T, want_e, (seed, 'envelope', eskip, opts),
                             scale=np.abs(want_full[fin]).max() if fin.any() else None)
        if long_trace and len(sos) <= 4 and len(esos) <= 2:
            nfft, hop = [(2048, 1024), (2048, 512), (1024, 512), (1024, 256), (512, 256), (256, 128)][int(rng.integers(0, 6))]
            F, nd = nfft//2 + 1, (T + hop - 1)//hop
            with_env = bool(rng.integers(0, 2))
            yf, ye = hipdsp.DeviceArray(c, (C, T), np.float32), hipdsp.DeviceArray(c, (C, T), np.float32)
            ps = hipdsp.DeviceArray(c, (C, nd, F), np.float32)
            hipdsp.chain_forward(c, plan, eplan if with_env else None, dx, T, yf, T, C, T, nfft, hop, rate, ps, nd)
            gf = yf.to_host()
            _same_non_finite(gf.T, want_f, (seed, 'chain: filtered', opts))
            want_s = np.zeros((nd, C, F))
            oracle.spectrogram_process(gf.T.astype(np.float64), want_s, rate, nfft, hop)
            gs = ps.to_host()
            for ch in range(C):
                for k in range(nd):
                    _same_non_finite(gs[ch, k], want_s[k, ch], (seed, 'chain: PSD', ch, k, nfft, hop, opts))
            if with_env and T > edge:
                hipdsp.sosfilt_envelope(c, plan, eplan, dx, T, yf, T, ye, T, C, T, phase=2)
                want_e = np.zeros((T, C))
                oracle.envelope_process(esos, gf.T.astype(np.float64), want_e, 0)
                _same_non_finite(ye.to_host().T, want_e, (seed, 'chain: envelope', opts))
    finally:
        for k, v in (('max_segments', 0), ('n_cus', 256), ('sos_waves_per_cu', 0), ('sos_waves_min', 0)):
            c.set_option(k, v)
