"""HIP path vs the CPU oracle and the scipy golden fixtures, through the C ABI.

Tolerance (BASELINE.json north_star): 1e-4 relative float32, measured as
max|a-b| / max|b| per channel (per frame for PSDs) -- see conftest.rel_err.
"""

import numpy as np
import pytest

from conftest import load_golden, rel_err
import gpu_helpers as gh

pytestmark = pytest.mark.gpu

TOL = 1e-4


def synth(rng, n, channels, rate):
    t = np.arange(n)/rate
    x = rng.uniform(-1.0, 1.0, size=(n, channels))
    for c in range(channels):
        x[:, c] = 0.5*x[:, c] + 0.5*np.sin(2*np.pi*1000.0*(1 + c/channels)*t)
    return x.astype(np.float32)


def test_library_and_device():
    from audian_amd import _lib, hipdsp
    assert _lib.lib.hipdsp_version() == 102
    c = gh.ctx()
    a = hipdsp.DeviceArray.from_host(c, np.arange(10, dtype=np.float32))
    assert np.array_equal(a.to_host(), np.arange(10, dtype=np.float32))


def test_pack_unpack_roundtrip():
    rng = np.random.default_rng(1)
    for T, C in [(1, 1), (33, 3), (1000, 64), (4097, 5)]:
        x = rng.standard_normal((T, C))
        c = gh.ctx()
        y = gh.from_planar(c, gh.to_planar(c, x), T, C)
        assert np.array_equal(y, x.astype(np.float32).astype(np.float64))
        x32 = x.astype(np.float32)
        y = gh.from_planar(c, gh.to_planar(c, x32), T, C)
        assert np.array_equal(y, x32.astype(np.float64))


def test_sosfilt_golden():
    g = load_golden('sosfilt')
    for k in range(int(g['count'])):
        sos, x, y = g[f'sos_{k}'], g[f'x_{k}'], g[f'y_{k}']
        got = gh.gpu_sosfilt(sos, x)
        assert got.shape == y.shape
        for c in range(y.shape[1]):
            assert rel_err(got[:, c], y[:, c]) < TOL, (k, c)


@pytest.mark.parametrize('btype,order,wn,rate,T', [
    ('bandpass', 2, (300.0, 3000.0), 96000.0, 300001),
    ('bandpass', 4, (300.0, 3000.0), 48000.0, 200000),
    ('lowpass', 2, (20.0,), 96000.0, 1500000),
    ('bandpass', 2, (5.0, 3000.0), 96000.0, 700003),
    ('highpass', 3, (100.0,), 192000.0, 150000),
    ('lowpass', 1, (4000.0,), 48000.0, 70000),
])
def test_sosfilt_long_vs_oracle(oracle, btype, order, wn, rate, T):
    """Many tiles and several time segments (warm-up path) against the C oracle."""
    from audian_amd.design import butter_sos
    rng = np.random.default_rng(7)
    sos = butter_sos(order, wn if len(wn) > 1 else wn[0], btype, rate)
    x = synth(rng, T, 3, rate)
    want = oracle.sosfilt(sos, x.astype(np.float64))
    got = gh.gpu_sosfilt(sos, x)
    for c in range(3):
        assert rel_err(got[:, c], want[:, c]) < TOL
    # the result must not depend on the segmentation
    one = gh.gpu_sosfilt(sos, x, max_segments=1)
    for c in range(3):
        assert rel_err(one[:, c], want[:, c]) < TOL
        assert rel_err(got[:, c], one[:, c]) < 1e-6


def test_sosfilt_skip_and_passthrough(oracle):
    g = load_golden('sosfilt')
    sos, x, y = g['sos_2'], g['x_2'], g['y_2']
    for skip in (1, 5, 1024, len(x) - 1, len(x)):
        got = gh.gpu_sosfilt(sos, x, skip=skip)
        assert got.shape == (len(x) - skip, x.shape[1])
        if skip < len(x):
            assert rel_err(got, y[skip:]) < TOL
    got = gh.gpu_sosfilt(None, x, skip=3)           # sos is None: pass-through
    assert np.array_equal(got, x[3:].astype(np.float64))


def test_envelope_golden():
    g = load_golden('envelope')
    for k in range(int(g['count'])):
        sos, x, y = g[f'sos_{k}'], g[f'x_{k}'], g[f'y_{k}']
        got = gh.gpu_envelope(sos, x, clamp=float(g[f'hp_{k}']) == 0)
        for c in range(y.shape[1]):
            assert rel_err(got[:, c], y[:, c]) < TOL, (k, c)


def test_envelope_too_short_raises():
    g = load_golden('envelope')
    sos = g['sos_0']
    with pytest.raises(ValueError):
        gh.gpu_envelope(sos, np.ones((9, 1), dtype=np.float32))
    gh.gpu_envelope(sos, np.ones((10, 1), dtype=np.float32))
    got = gh.gpu_envelope(None, np.ones((100, 2), dtype=np.float32))    # sos None -> zeros
    assert np.all(got == 0)


@pytest.mark.parametrize('env,order,hp,rate,T', [
    (20.0, 2, 0.0, 96000.0, 1500000),
    (500.0, 2, 0.0, 48000.0, 400000),
    (500.0, 2, 10.0, 48000.0, 400000),
    (200.0, 3, 0.0, 44100.0, 250000),
])
def test_envelope_long_vs_oracle(oracle, env, order, hp, rate, T):
    from audian_amd.design import butter_sos
    rng = np.random.default_rng(8)
    sos = butter_sos(order, (hp, env), 'bandpass', rate) if hp > 0 else \
        butter_sos(order, env, 'lowpass', rate)
    x = synth(rng, T, 2, rate)
    want = np.zeros((T, 2))
    oracle.envelope_process(sos, x.astype(np.float64), want, 0, highpass_cutoff=hp)
    got = gh.gpu_envelope(sos, x, clamp=hp == 0)
    for c in range(2):
        assert rel_err(got[:, c], want[:, c]) < TOL
    got = gh.gpu_envelope(sos, x, skip=7, clamp=hp == 0)
    assert rel_err(got, want[7:]) < TOL
    # the same sweeps without the register prefetch (the variant short slabs always take)
    gh.ctx().set_option('sos_prefetch', 0)
    try:
        plain = gh.gpu_envelope(sos, x, clamp=hp == 0)
    finally:
        gh.ctx().set_option('sos_prefetch', 1)
    assert np.array_equal(plain, gh.gpu_envelope(sos, x, clamp=hp == 0))


@pytest.mark.parametrize('env,rate,T,C', [(20.0, 96000.0, 3000000, 1), (20.0, 96000.0, 1200000, 5), (5.0, 48000.0, 2000000, 2),
                                          (500.0, 48000.0, 900000, 3)])
def test_envelope_state_handover_is_exact_for_every_segmentation(oracle, env, rate, T, C):
    """The envelope's forward sweep starts every time segment from ZERO state and env_fix_kernel hands the true
    states over (exact, no warm-up): one segment, three, the planner's choice and one-tile segments
    (hundreds of segments far shorter than the filter's memory: the 20 Hz low-pass at 96 kHz remembers 26 tiles, so
    a state is the sum of up to 27 hand-overs) must agree to < 1e-6 with each other and to 1e-4 with the oracle --
    through hipdsp_envelope, hipdsp_sosfilt_envelope and hipdsp_chain_forward + backward sweep."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rng = np.random.default_rng(int(T + env))
    x = synth(rng, T, C, rate)
    esos = butter_sos(2, env, 'lowpass', rate)
    sos = butter_sos(2, (300.0, 3000.0), 'bandpass', rate)
    want = np.zeros((T, C))
    oracle.envelope_process(esos, x.astype(np.float64), want, 0)
    c = gh.ctx()
    dx = gh.to_planar(c, x)
    fplan, eplan = hipdsp.SosPlan(c, sos), hipdsp.SosPlan(c, esos)
    warm, _ = eplan.info()
    nfft, hop = 2048, 1024
    nd = (T + hop - 1)//hop
    results, chains = {}, {}
    try:
        for name, opts in [('one', {'max_segments': 1}), ('three', {'max_segments': 3}), ('planner', {}),
                           ('one-tile segments', {'n_cus': 1024, 'sos_waves_per_cu': 16, 'sos_waves_min': 16})]:
            for k, v in opts.items():
                c.set_option(k, v)
            de = hipdsp.DeviceArray(c, (C, T), np.float32)
            hipdsp.envelope(c, eplan, dx, T, de, T, C, T, 0)
            results[name] = de.to_host()
            yf, ye = hipdsp.DeviceArray(c, (C, T), np.float32), hipdsp.DeviceArray(c, (C, T), np.float32)
            ps = hipdsp.DeviceArray(c, (C, nd, nfft//2 + 1), np.float32)
            hipdsp.chain_forward(c, fplan, eplan, dx, T, yf, T, C, T, nfft, hop, rate, ps, nd)
            hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, yf, T, ye, T, C, T, phase=2)
            seg, n = hipdsp.chain_plan(c, fplan, eplan, C, T)
            chains[name] = (ye.to_host(), yf.to_host(), n, seg)
            if name == 'one-tile segments' and warm >= 8*2048:
                assert seg < warm//4 and n > 50          # many hand-overs inside the filter's memory
            c.set_option('max_segments', 0)
            c.set_option('n_cus', 256)
            c.set_option('sos_waves_per_cu', 0)
            c.set_option('sos_waves_min', 0)
    finally:
        for k, v in (('max_segments', 0), ('n_cus', 256), ('sos_waves_per_cu', 0), ('sos_waves_min', 0)):
            c.set_option(k, v)
    assert chains['one'][2] == 1 and chains['three'][2] <= 3
    ref = results['one']
    for name, got in results.items():
        for ch in range(C):
            assert rel_err(got[ch], want[:, ch]) < TOL, (name, ch)
            assert rel_err(got[ch], ref[ch]) < 1e-6, (name, ch)
    want_f = oracle.sosfilt(sos, x.astype(np.float64))
    ref_e = chains['one'][0]
    for name, (ge, gf, n, seg) in chains.items():
        want_e = np.zeros((T, C))
        oracle.envelope_process(esos, gf.T.astype(np.float64), want_e, 0)
        for ch in range(C):
            assert rel_err(gf[ch], want_f[:, ch]) < TOL, (name, ch)
            assert rel_err(ge[ch], want_e[:, ch]) < TOL, (name, ch)
            assert rel_err(ge[ch], ref_e[ch]) < 1e-6, (name, ch, n)


@pytest.mark.parametrize('order,env,hp', [(6, 40.0, 0.0), (8, 60.0, 0.0), (3, 300.0, 20.0), (4, 800.0, 50.0)])
def test_envelope_state_handover_with_three_and_four_sections(oracle, order, env, hp):
    """The same exactness for envelope plans of three and four sections (hipdsp_envelope alone: env_fix_kernel<3>,
    <4>; low-pass of order 6 / 8, band-pass envelope of order 3 / 4): one segment against one-tile segments."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rate, T, C = 48000.0, 700000, 2
    rng = np.random.default_rng(order)
    x = synth(rng, T, C, rate)
    sos = butter_sos(order, (hp, env), 'bandpass', rate) if hp > 0 else butter_sos(order, env, 'lowpass', rate)
    assert len(sos) in (3, 4)
    want = np.zeros((T, C))
    oracle.envelope_process(sos, x.astype(np.float64), want, 0, highpass_cutoff=hp)
    c = gh.ctx()
    try:
        one = gh.gpu_envelope(sos, x, clamp=hp == 0, max_segments=1)
        for k, v in (('n_cus', 1024), ('sos_waves_per_cu', 16), ('sos_waves_min', 16)):
            c.set_option(k, v)
        many = gh.gpu_envelope(sos, x, clamp=hp == 0)
    finally:
        for k, v in (('n_cus', 256), ('sos_waves_per_cu', 0), ('sos_waves_min', 0)):
            c.set_option(k, v)
    for ch in range(C):
        assert rel_err(one[:, ch], want[:, ch]) < TOL and rel_err(many[:, ch], want[:, ch]) < TOL, ch
        assert rel_err(many[:, ch], one[:, ch]) < 1e-6, ch


@pytest.mark.parametrize('T', [2050, 70000, 300000])
def test_envelope_skip_values_vs_oracle(oracle, T):
    """nbefore (skip) below, at and above tile borders: the backward sweep stops at the tile that
    holds `skip`, the forward state sweep still covers the whole slab."""
    from audian_amd.design import butter_sos
    rate = 48000.0
    rng = np.random.default_rng(T)
    sos = butter_sos(2, 500.0, 'lowpass', rate)
    x = synth(rng, T, 2, rate)
    want = np.zeros((T, 2))
    oracle.envelope_process(sos, x.astype(np.float64), want, 0)
    for skip in (0, 1, 2047, 2048, 2049, 5000, 65536, T - 1, T):
        if skip > T:
            continue
        got = gh.gpu_envelope(sos, x, skip=skip)
        assert got.shape[0] == T - skip
        if skip < T:
            assert rel_err(got, want[skip:]) < TOL, skip


def test_spectrogram_golden():
    g = load_golden('spectrogram')
    for k in range(int(g['count'])):
        rate, nfft, hop = g[f'par_{k}']
        nfft, hop = int(nfft), int(hop)
        x, S = g[f'x_{k}'], g[f'S_{k}']              # S: (F, T', C)
        nd = S.shape[1] + 2                          # two zero tail frames
        got = gh.gpu_spectrogram(x, rate, nfft, hop, nd)
        assert got.shape == (nd, x.shape[1], nfft//2 + 1)
        assert np.all(got[S.shape[1]:] == 0)
        for c in range(S.shape[2]):
            for j in range(S.shape[1]):
                assert rel_err(got[j, c, :], S[:, j, c]) < TOL, (k, j, c)


@pytest.mark.parametrize('nfft,hop', [(256, 128), (256, 37), (512, 256), (512, 128), (1024, 256),
                                      (2048, 1024), (2048, 512), (4096, 2048), (128, 64), (8192, 4096), (128, 17), (64, 32),
                                      (64, 64), (32, 16), (32, 5), (16, 8), (8, 4), (8192, 1000), (16384, 8192),
                                      (16384, 3000), (32768, 16384)])
def test_spectrogram_fast_and_generic_vs_oracle(oracle, nfft, hop):
    """Every supported size through its own kernel and through the generic radix-2 kernel."""
    rng = np.random.default_rng(nfft + hop)
    rate = 96000.0
    nframes = 70 if nfft < 8192 else 11
    T = (nframes - 1)*hop + nfft + 5
    x = (synth(rng, T, 3, rate) + np.float32(0.1)).astype(np.float32)
    nd = (T + hop - 1)//hop
    want = np.zeros((nd, 3, nfft//2 + 1))
    oracle.spectrogram_process(x.astype(np.float64), want, rate, nfft, hop)
    results = [gh.gpu_spectrogram(x, rate, nfft, hop, nd)]
    c = gh.ctx()
    try:
        for kern in (2, 3, 4):            # two-stage, three-stage register/LDS kernels, nfft 4096 as a streamed workgroup
            c.set_option('spec_kernel', kern)
            results.append(gh.gpu_spectrogram(x, rate, nfft, hop, nd))
        c.set_option('spec_kernel', 0)
        c.set_option('force_generic_fft', 1)
        results.append(gh.gpu_spectrogram(x, rate, nfft, hop, nd))
    finally:
        c.set_option('force_generic_fft', 0)
        c.set_option('spec_kernel', 0)
    for res in results:
        for ch in range(3):
            for j in range(nd):
                if np.max(np.abs(want[j, ch])) == 0:
                    assert np.all(res[j, ch] == 0)
                else:
                    assert rel_err(res[j, ch], want[j, ch]) < TOL, (nfft, hop, j, ch)


@pytest.mark.parametrize('nfft', [8, 16, 32, 64, 128, 256, 512, 1024])
def test_short_windows_stream_runs_of_frames_through_lds(oracle, nfft):
    """nfft 8 ... 1024 (the reference's default is 256 / 128; its selector starts at 8, databrowser.py:516; 512 and 1024 take
    this path for part of their hops only: spectrogram.hip's dispatch): a wave
    streams a run of consecutive frames through an LDS ring and stores whole batches of frames (spec_pack.h).  Every
    overlap the spin box can produce (databrowser.py:522-529: hop 1 ... nfft), runs long enough that every wave walks
    many batches and the ring wraps many times, odd channel pitches (4-byte aligned rows only), more output frames than
    the trace holds (zero tail), the dB image next to the PSD, against the oracle and the kernels replaced."""
    from audian_amd import hipdsp
    rate, C = 48000.0, 3
    c = gh.ctx()
    rng = np.random.default_rng(nfft)
    hops = sorted({1, 2, 3, nfft//4, nfft//2, nfft//2 + 1, nfft - 1, nfft, max(1, int(nfft*0.37))})
    for hop in hops:
        nframes = 5000 if hop > 2 else 1500
        T = (nframes - 1)*hop + nfft + int(rng.integers(0, hop + 1))
        pitch = T + int(rng.integers(0, 7))
        x = (synth(rng, T, C, rate) + np.float32(0.1)).astype(np.float32)
        dx = hipdsp.DeviceArray(c, (C, pitch), np.float32)
        dx.copy_from_host(np.pad(x.T, ((0, 0), (0, pitch - T)), constant_values=np.float32(7e9)))
        nd = (T + hop - 1)//hop + 3
        F = nfft//2 + 1
        want = np.zeros((nd, C, F))
        oracle.spectrogram_process(x.astype(np.float64), want, rate, nfft, hop)
        out = hipdsp.DeviceArray(c, (C, nd, F), np.float32)
        db = hipdsp.DeviceArray(c, (C, nd, F), np.float32)
        for fpw in (0, 1, 3):
            c.set_option('spec_fpw', fpw)
            try:
                for arr in (out, db):
                    hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(arr), 0x7f, 4*C*nd*F)
                hipdsp.spectrogram(c, dx, pitch, C, T, nfft, hop, rate, out, nd, db_out=db if fpw != 1 else None)
            finally:
                c.set_option('spec_fpw', 0)
            got = out.to_host()
            for ch in range(C):
                zero = np.max(np.abs(want[:, ch]), axis=1) == 0
                assert np.all(got[ch][zero] == 0), (nfft, hop, fpw)
                num = np.max(np.abs(got[ch][~zero] - want[~zero, ch]), axis=1)
                den = np.max(np.abs(want[~zero, ch]), axis=1)
                assert np.max(num/den) < TOL, (nfft, hop, fpw, ch, int(np.argmax(num/den)))
            if fpw != 1:
                gdb, wdb = db.to_host(), oracle.decibel(got.astype(np.float64))
                fin = np.isfinite(wdb)
                assert np.array_equal(np.isfinite(gdb), fin) and np.all(gdb[~fin] == -np.inf), (nfft, hop)
                assert np.max(np.abs(gdb[fin] - wdb[fin])) < 1e-3, (nfft, hop)
        c.set_option('spec_kernel', 2)
        try:
            old = hipdsp.DeviceArray(c, (C, nd, F), np.float32)
            hipdsp.spectrogram(c, dx, pitch, C, T, nfft, hop, rate, old, nd)
        finally:
            c.set_option('spec_kernel', 0)
        o = old.to_host()
        scale = np.maximum(np.max(np.abs(o), axis=2, keepdims=True), 1e-30)
        assert np.max(np.abs(got - o)/scale) < 1e-5, (nfft, hop)


@pytest.mark.parametrize('nfft,hop', [(64, 32), (256, 128), (256, 100), (512, 256), (1024, 256), (2048, 1024), (4096, 2048),
                                      (8192, 4096), (16384, 8192), (65536, 32768), (131072, 65536), (262144, 131072), (524288, 262144)])
def test_spectrogram_of_an_offset_plus_something_small(oracle, nfft, hop):
    """A trace that is a large offset plus a small signal -- raw data of the reference's default session (no filter,
    bufferedfilter.py:40-42) from a sensor with a DC offset, or a filter's decaying transient: detrend='constant'
    removes the offset, and a float32 sum of the samples would carry 1e-7 of the OFFSET into bins 0 and 1 of every
    frame (1e-4 of the frame's peak already at offset / signal = 100).  The stand-alone kernels take the frame mean
    relative to a pivot sample instead (tools/fuzz_stress.py found the case, seed 10268); a NaN as the pivot must not
    spread to frames that do not hold it."""
    from audian_amd import hipdsp
    rate, C = 96000.0, 2
    rng = np.random.default_rng(nfft + hop)
    nframes = 40 if nfft <= 2048 else 12
    T = (nframes - 1)*hop + nfft + 3
    for offset, small in ((1000.0, 0.05), (-3.0, 1e-3), (0.5, 1e-5)):
        x = (offset + small*rng.standard_normal((T, C))).astype(np.float32)
        nd = (T + hop - 1)//hop
        want = np.zeros((nd, C, nfft//2 + 1))
        oracle.spectrogram_process(x.astype(np.float64), want, rate, nfft, hop)
        got = gh.gpu_spectrogram(x, rate, nfft, hop, nd)
        for ch in range(C):
            for j in range(nd):
                if np.max(np.abs(want[j, ch])) == 0:
                    assert np.all(got[j, ch] == 0)
                else:
                    assert rel_err(got[j, ch], want[j, ch]) < TOL, (nfft, hop, offset, small, j, ch)
    x = rng.standard_normal((T, C)).astype(np.float32)
    x[0, 0] = np.nan                                   # the pivot of the first run of frames
    nd = (T + hop - 1)//hop
    want = np.zeros((nd, C, nfft//2 + 1))
    oracle.spectrogram_process(x.astype(np.float64), want, rate, nfft, hop)
    got = gh.gpu_spectrogram(x, rate, nfft, hop, nd)
    assert np.array_equal(np.isnan(got), np.isnan(want)), (nfft, hop)
    assert np.all(np.isnan(got[0, 0])) and np.all(np.isfinite(got[(nfft + hop - 1)//hop:, 0]))


@pytest.mark.parametrize('nfft,hop', [(64, 64), (256, 256), (256, 128), (512, 512), (1024, 1024), (2048, 2048), (2048, 1024),
                                      (4096, 4096), (8192, 8192), (16384, 16384), (65536, 65536), (131072, 131072), (131072, 65536),
                                      (262144, 262144), (524288, 262144)])
def test_spectrogram_of_pulses_at_the_frame_borders(oracle, nfft, hop):
    """A pulse train -- a pulse-type electric fish, clicks -- over a quiet baseline, with the pulses on the frame borders:
    the first sample of a frame is a thousand times the rest of it, and the Hann window gives that sample weight zero, so
    the frame's spectrum is the baseline's.  A frame mean taken relative to THAT sample (the pivot of round 4's first
    version) is the mean of differences of size A, wrong by 6e-8 A, and the Hann window puts 6e-8 A nfft / 2 of it into
    bins 0 and 1: 1e-3 of the frame's peak.  The pivot is the previous frame's mean (the first frame of a run takes two
    steps), which is as good as the signal is stationary over two frames and never worse than the plain sum."""
    rate, C = 96000.0, 2
    rng = np.random.default_rng(nfft)
    nframes = 24 if nfft <= 4096 else 6
    T = (nframes - 1)*hop + nfft + 5
    for amp, sigma, offset in ((1.0, 1e-3, 0.0), (-3.0, 1e-3, 0.2), (50.0, 2e-2, -1.0)):
        x = (offset + sigma*rng.standard_normal((T, C))).astype(np.float32)
        x[::hop, 0] += np.float32(amp)                 # channel 0: on every frame's first sample
        x[hop - 1::hop, 1] += np.float32(amp)          # channel 1: on every frame's last sample (weight 6e-10 at nfft 65536 ...)
        nd = (T + hop - 1)//hop
        want = np.zeros((nd, C, nfft//2 + 1))
        oracle.spectrogram_process(x.astype(np.float64), want, rate, nfft, hop)
        got = gh.gpu_spectrogram(x, rate, nfft, hop, nd)
        worst = 0.0
        for ch in range(C):
            for j in range(nd):
                if np.max(np.abs(want[j, ch])) == 0:
                    assert np.all(got[j, ch] == 0)
                else:
                    worst = max(worst, rel_err(got[j, ch], want[j, ch]))
        assert worst < TOL, (nfft, hop, amp, sigma, offset, worst)


@pytest.mark.parametrize('nfft', [64, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288])
def test_spectrogram_after_a_step_in_the_level(oracle, nfft):
    """A trace whose level jumps by a thousand times its noise between two frames (stimulus artefacts, a DC-coupled
    amplifier): the frame after the jump is flat again, the differences to the mean of the frame before it are all the
    size of the jump, and their float32 mean is good to 6e-8 of THAT.  Frames without overlap (the reference's overlap
    spin box goes down to 0 %, databrowser.py:522-529) and with half of it; the kernels sum what the subtraction left and
    take it out of bins 0 and 1 (spec_wgs.h, spec_fast's variants without register reuse), or take the frame mean in two
    steps (spec_chip.h); with half of the frame before inside the frame the plain scheme stays under the tolerance."""
    rate, sigma = 96000.0, 1e-3
    for hop in (nfft, nfft//2):
        rng = np.random.default_rng(nfft + hop)
        nframes = 24 if nfft <= 4096 else 8
        T = (nframes - 1)*hop + nfft + 5
        x = (sigma*rng.standard_normal((T, 2))).astype(np.float32)
        level = np.zeros(T, dtype=np.float32)
        for j in range(3, nframes, 4):
            level[j*hop:] += np.float32(1000.0*sigma*(1 if (j//4) % 2 == 0 else -0.7))
        x[:, 0] += level
        x[:, 1] += np.float32(0.3)*level + np.float32(5.0)
        nd = (T + hop - 1)//hop
        want = np.zeros((nd, 2, nfft//2 + 1))
        oracle.spectrogram_process(x.astype(np.float64), want, rate, nfft, hop)
        got = gh.gpu_spectrogram(x, rate, nfft, hop, nd)
        worst = max(rel_err(got[j, ch], want[j, ch]) for j in range(nd) for ch in range(2) if np.max(np.abs(want[j, ch])) > 0)
        assert worst < TOL, (nfft, hop, worst)


@pytest.mark.parametrize('nfft', [16, 256, 4096, 8192, 32768])
def test_spectrogram_steps_inside_long_runs_of_frames(oracle, nfft):
    """Steps in the level while a wave or workgroup is in the middle of a RUN of frames ("spec_fpw" 16; large batches
    get there by themselves): the pivot of a frame's mean is carried from frame to frame there.  Two finds of
    tools/fuzz_stress.py with random options: the workgroup kernels keep the overlapped half as differences to the pivot
    it was fetched under, and a pivot that moves far is not an exact float32 step -- half an ulp of it between the two
    halves is a step in the middle of the frame (bins 1, 3, 5 ...: 1.6e-4 at nfft 32768 behind 275 sigma; the half is
    fetched again then); and at eight or sixteen samples per frame the mean of the frame before follows a single pulse
    (those windows take the two steps of the first frame in every batch)."""
    from audian_amd import hipdsp
    rate, hop = 96000.0, nfft//2
    rng = np.random.default_rng(nfft)
    nframes = 56 if nfft >= 4096 else 3000
    T = (nframes - 1)*hop + nfft + 11
    x = (1e-3*(rng.standard_normal((T, 2)) + 0.3)).astype(np.float32)
    for j, d in ((30, 0.275), (33, 0.242), (50, 0.173)):
        x[(j if nfft >= 4096 else 40*j)*hop:, 0] += np.float32(d)
    x[1::24, 1] += np.float32(1.69)                      # a pulse in every third frame or so (of 8 or 16 samples)
    nd = (T + hop - 1)//hop
    want = np.zeros((nd, 2, nfft//2 + 1))
    oracle.spectrogram_process(x.astype(np.float64), want, rate, nfft, hop)
    c = gh.ctx()
    try:
        for fpw in (16, 3, 0):
            c.set_option('spec_fpw', fpw)
            got = gh.gpu_spectrogram(x, rate, nfft, hop, nd)
            worst = max(rel_err(got[j, ch], want[j, ch]) for j in range(nd) for ch in range(2) if np.max(np.abs(want[j, ch])) > 0)
            assert worst < TOL, (nfft, fpw, worst)
    finally:
        c.set_option('spec_fpw', 0)


@pytest.mark.parametrize('nfft,hop', [(2048, 1024), (1024, 256), (512, 256), (256, 128)])
def test_fused_sweep_of_pulses_at_the_frame_borders(oracle, nfft, hop):
    """The same pulse train through hipdsp_chain_forward: behind a wide first-order low-pass a pulse stays a few samples
    long, and every nfft-th sample carries one -- the frames that START there see it under window weights of 1e-5 and less,
    the frames that have it in their middle are all pulse.  The fused sweep's FFT waves take the pivot of a frame's mean
    from the frame before it (psd_frame, chain.hip), never from a sample."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rate, C = 96000.0, 2
    rng = np.random.default_rng(nfft + hop)
    T = 30*2048 + 777
    x = (1e-3*rng.standard_normal((T, C))).astype(np.float32)
    x[::nfft, 0] += np.float32(2.0)
    x[5::nfft, 1] -= np.float32(40.0)
    c = gh.ctx()
    dx = gh.to_planar(c, x)
    nd = (T + hop - 1)//hop
    F = nfft//2 + 1
    sos = butter_sos(1, 0.4*rate, 'lowpass', rate)
    esos = butter_sos(1, 500.0, 'lowpass', rate)
    fplan, eplan = hipdsp.SosPlan(c, sos), hipdsp.SosPlan(c, esos)
    for with_env in (True, False):
        yf = hipdsp.DeviceArray(c, (C, T), np.float32)
        ps = hipdsp.DeviceArray(c, (C, nd, F), np.float32)
        hipdsp.chain_forward(c, fplan, eplan if with_env else None, dx, T, yf, T, C, T, nfft, hop, rate, ps, nd)
        gf, gs = yf.to_host(), ps.to_host()
        want_s = np.zeros((nd, C, F))
        oracle.spectrogram_process(gf.T.astype(np.float64), want_s, rate, nfft, hop)
        worst = 0.0
        for ch in range(C):
            for j in range(nd):
                if np.max(np.abs(want_s[j, ch])) == 0:
                    assert np.all(gs[ch, j] == 0)
                else:
                    worst = max(worst, rel_err(gs[ch, j], want_s[j, ch]))
        assert worst < TOL, (nfft, hop, with_env, worst)


def test_spectrogram_short_source_and_db(oracle):
    x = np.ones((100, 2), dtype=np.float32)
    got = gh.gpu_spectrogram(x, 48000.0, 256, 128, 3)
    assert np.all(got == 0)
    g = load_golden('spectrogram')
    x, S = g['x_1'], g['S_1']
    got, db = gh.gpu_spectrogram(x, 48000.0, 1024, 256, S.shape[1] + 1, want_db=True)
    want = oracle.decibel(got)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(db), fin)
    assert np.all(db[~fin] == -np.inf)
    assert np.max(np.abs(db[fin] - want[fin])) < 1e-3


def test_decibel_golden():
    from audian_amd import hipdsp
    g = load_golden('decibel')
    c = gh.ctx()
    p = g['p'].astype(np.float32)
    want = np.full(p.shape, -np.inf)
    m = p > np.float32(1e-20)
    want[m] = 10*np.log10(p[m].astype(np.float64))
    dp = hipdsp.DeviceArray.from_host(c, p)
    out = hipdsp.DeviceArray(c, p.shape, np.float32)
    hipdsp.decibel(c, dp, out, p.size)
    got = out.to_host()
    assert np.array_equal(np.isinf(got), ~m)
    assert np.max(np.abs(got[m] - want[m])) < 1e-4
    # SpecItem image: decibel(buffer[:, ch, :].T)
    rng = np.random.default_rng(3)
    spec = (10.0**rng.uniform(-12, 2, size=(70, 129))).astype(np.float32)
    ds = hipdsp.DeviceArray.from_host(c, spec)
    img = hipdsp.DeviceArray(c, (129, 70), np.float32)
    hipdsp.decibel_image(c, ds, img, 70, 129)
    assert np.max(np.abs(img.to_host() - 10*np.log10(spec.astype(np.float64)).T)) < 1e-4


def test_chain_golden():
    """data -> filter -> {spectrogram, envelope}, stages resident on the device."""
    from audian_amd import hipdsp
    g = load_golden('chain')
    c = gh.ctx()
    x = g['x']
    T, C = x.shape
    dx = gh.to_planar(c, x)
    df = hipdsp.DeviceArray(c, (C, T), np.float32)
    hipdsp.sosfilt(c, hipdsp.SosPlan(c, g['sos']), dx, T, df, T, C, T, 0)
    nd = g['spec'].shape[0] + 1
    ds = hipdsp.DeviceArray(c, (C, nd, 129), np.float32)
    hipdsp.spectrogram(c, df, T, C, T, 256, 128, float(g['rate']), ds, nd)
    de = hipdsp.DeviceArray(c, (C, T), np.float32)
    hipdsp.envelope(c, hipdsp.SosPlan(c, g['esos']), df, T, de, T, C, T, 0)
    filt = gh.from_planar(c, df, T, C)
    env = gh.from_planar(c, de, T, C)
    spec = ds.to_host().transpose(1, 0, 2)
    for ch in range(C):
        assert rel_err(filt[:, ch], g['filt'][:, ch]) < TOL
        assert rel_err(env[:, ch], g['env'][:, ch]) < TOL
        for j in range(nd - 1):
            assert rel_err(spec[j, ch], g['spec'][j, ch]) < TOL
    assert np.all(spec[-1] == 0)


def test_synth_is_deterministic_and_bounded():
    from audian_amd import hipdsp
    c = gh.ctx()
    a = hipdsp.DeviceArray(c, (4, 96000), np.float32)
    b = hipdsp.DeviceArray(c, (2, 96000), np.float32)
    hipdsp.synth(c, a, 96000, 4, 96000, 96000.0, 1234)
    hipdsp.synth(c, b, 96000, 2, 96000, 96000.0, 1234, c0=2, c_total=4)
    ha, hb = a.to_host(), b.to_host()
    assert np.array_equal(ha[2:], hb)                 # channel shards agree with the whole
    assert np.all(np.abs(ha) <= 1.0)
    spec = np.abs(np.fft.rfft(ha[1]))                 # tone at 1000*(1 + 1/4) Hz
    assert np.argmax(spec[1:]) + 1 == 1250


def test_rccl_allgather_entry_point_single_rank():
    """hipdsp_comm_* / hipdsp_allgather_f32 with one rank (the box has one GPU): the
    communicator comes up and the gather reproduces the tile."""
    from audian_amd import hipdsp
    c = gh.ctx()
    comm = hipdsp.Comm(c, hipdsp.Comm.unique_id(), 0, 1)
    tile = np.random.default_rng(2).standard_normal((3, 50, 129)).astype(np.float32)
    send = hipdsp.DeviceArray.from_host(c, tile)
    recv = hipdsp.DeviceArray(c, tile.shape, np.float32).zero_()
    comm.allgather(send, recv, tile.size)
    c.synchronize()
    assert np.array_equal(recv.to_host(), tile)
    comm.close()


@pytest.mark.parametrize('step', [2, 3, 17, 63, 64, 100, 1000, 4097])
def test_minmax_decimation_bit_exact(oracle, step):
    """np.minimum/maximum.reduceat screen decimation, bit-exact (selection only)."""
    from audian_amd import hipdsp
    rng = np.random.default_rng(step)
    T, C = 50000 + step, 3
    x = rng.standard_normal((T, C)).astype(np.float32)
    c = gh.ctx()
    dx = gh.to_planar(c, x)
    for start, stop in [(0, T), (7, T - 5), (step*3, step*3 + 1), (100, 100 + 10*step)]:
        nseg = (stop - start + step - 1)//step
        out = hipdsp.DeviceArray(c, (C, 2*nseg), np.float32)
        hipdsp.minmax_decimate(c, dx, T, C, start, stop, step, out, 2*nseg)
        want = oracle.minmax_decimate(x.astype(np.float64), start, stop, step)     # (2n, C)
        assert np.array_equal(out.to_host().astype(np.float64), want.T), (step, start, stop)
    with pytest.raises(ValueError):
        hipdsp.minmax_decimate(c, dx, T, C, 10, 5, step, dx, 2)


@pytest.mark.parametrize('step', [1, 2, 4, 5, 31, 32, 33, 100, 128, 129, 480, 511, 512, 513, 2048, 28800])
def test_minmax_decimation_streams_any_step(oracle, step):
    """The same decimation as a read stream (16-byte loads whatever the step: segments of a tile staged in LDS below
    512 samples per segment, a wave per segment from there on): every regime and its borders, windows that start and
    stop anywhere, an output pitch with slack, NaN and infinities in the data (np.minimum / np.maximum: a NaN in a
    segment makes both of its values NaN)."""
    from audian_amd import hipdsp
    rng = np.random.default_rng(1000 + step)
    T, C = 100000 + 7*step, 2
    x = rng.standard_normal((T, C)).astype(np.float32)
    x[rng.integers(0, T, size=40), 0] = np.nan
    x[rng.integers(0, T, size=40), 1] = np.inf
    x[rng.integers(0, T, size=40), 1] = -np.inf
    c = gh.ctx()
    dx = gh.to_planar(c, x)
    for start, stop in [(0, T), (3, T - 1), (step + 1, min(T, step + 1 + 37*step + step//2)), (T - 1, T), (5, 5)]:
        nseg = (stop - start + step - 1)//step
        if nseg == 0:
            hipdsp.minmax_decimate(c, dx, T, C, start, stop, step, dx, 2)
            continue
        pitch = 2*nseg + 3
        out = hipdsp.DeviceArray.from_host(c, np.full((C, pitch), 5.0, dtype=np.float32))
        hipdsp.minmax_decimate(c, dx, T, C, start, stop, step, out, pitch)
        got = out.to_host()
        want = oracle.minmax_decimate(x.astype(np.float64), start, stop, step).T          # (C, 2n)
        assert np.array_equal(got[:, :2*nseg].astype(np.float64), want, equal_nan=True), (step, start, stop)
        assert np.all(got[:, 2*nseg:] == 5.0)


def test_mean_spectrum_db(oracle):
    """decibel(mean over frames) with the -200 dB floor."""
    from audian_amd import hipdsp
    rng = np.random.default_rng(4)
    c = gh.ctx()
    for frames, F in [(1, 129), (63, 513), (1000, 1025)]:
        spec = (10.0**rng.uniform(-24, 2, size=(frames, F))).astype(np.float32)
        spec[:, 3] = 0.0                              # -inf -> floored
        ds = hipdsp.DeviceArray.from_host(c, spec)
        out = hipdsp.DeviceArray(c, (F,), np.float32)
        for i0, i1 in [(0, frames), (frames//3, frames//3 + 1), (frames//2, frames)]:
            hipdsp.mean_spectrum_db(c, ds, F, i0, i1, out)
            want = oracle.mean_power_db(spec[:, None, :], i0, i1, 0)
            got = out.to_host().astype(np.float64)
            assert got[3] == -200.0
            assert np.max(np.abs(got - want)) < 1e-3, (frames, F, i0, i1)


@pytest.mark.parametrize('frames,F,start,stop,step', [(70, 129, 0, 70, 1), (1000, 1025, 3, 997, 28), (5000, 513, 100, 4999, 64),
                                                      (33, 33, 32, 33, 4), (400, 2049, 0, 400, 400), (10, 5, 4, 4, 3)])
def test_decimated_db_image(oracle, frames, F, start, stop, step):
    """SURVEY 8f-1 for the spectrogram: max over `step` frames (np.maximum.reduceat), then the dB image."""
    from audian_amd import hipdsp
    rng = np.random.default_rng(frames + step)
    c = gh.ctx()
    spec = (10.0**rng.uniform(-24, 2, size=(frames, F))).astype(np.float32)
    spec[frames//2, :] = 0.0
    if stop - start > 2*step:
        spec[start + step:start + 2*step, 1] = 0.0          # a whole segment at -inf
    ncols = (stop - start + step - 1)//step
    ds = hipdsp.DeviceArray.from_host(c, spec)
    img = hipdsp.DeviceArray(c, (F, max(ncols, 1)), np.float32)
    hipdsp.decibel_image_decimate(c, ds, img, frames, F, start, stop, step)
    if ncols == 0:
        return
    got = img.to_host()[:, :ncols] if ncols else None
    want = oracle.decimated_db_image(spec[:, None, :], start, stop, step, 0)
    fin = np.isfinite(want)
    assert got.shape == want.shape
    assert np.array_equal(np.isfinite(got), fin)
    assert np.max(np.abs(got[fin] - want[fin])) < 1e-4
    with pytest.raises(ValueError):
        hipdsp.decibel_image_decimate(c, ds, img, frames, F, 0, frames + 1, step)


@pytest.mark.parametrize('nfft,hop,nframes', [(8192, 2048, 6), (16384, 8192, 5), (32768, 4096, 4), (65536, 16384, 3),
                                              (131072, 65536, 4), (131072, 30001, 5), (262144, 131072, 3), (524288, 262144, 2)])
def test_spectrogram_large_nfft_four_step(oracle, nfft, hop, nframes):
    """The upper part of the reference's nfft selector (2^13 .. 2^19, databrowser.py:516):
    workgroup FFT (8192, 16384) and four-step FFT through the context scratch (larger), with a
    zero tail and the fused dB output."""
    rng = np.random.default_rng(nfft)
    rate = 96000.0
    T = (nframes - 1)*hop + nfft + 3
    x = (synth(rng, T, 2, rate) + np.float32(0.2)).astype(np.float32)
    nd = (T + hop - 1)//hop
    want = np.zeros((nd, 2, nfft//2 + 1))
    oracle.spectrogram_process(x.astype(np.float64), want, rate, nfft, hop)
    got, db = gh.gpu_spectrogram(x, rate, nfft, hop, nd, want_db=True)
    assert got.shape == want.shape
    for ch in range(2):
        for j in range(nd):
            if np.max(np.abs(want[j, ch])) == 0:
                assert np.all(got[j, ch] == 0) and np.all(db[j, ch] == -np.inf)
            else:
                assert rel_err(got[j, ch], want[j, ch]) < TOL, (nfft, j, ch)
    fin = got > 1e-20
    assert np.max(np.abs(db[fin] - 10*np.log10(got[fin]))) < 1e-3
    with pytest.raises(NotImplementedError):
        gh.gpu_spectrogram(x, rate, 1 << 20, 1 << 19, 1)


@pytest.mark.parametrize('nfft,hop,nframes', [(65536, 32768, 9), (65536, 8192, 21), (65536, 1001, 37), (65536, 65536, 5),
                                              (131072, 65536, 7), (131072, 16384, 11), (131072, 3001, 9), (131072, 131072, 3),
                                              (262144, 131072, 5), (262144, 50001, 4), (524288, 262144, 4), (524288, 524288, 2),
                                              (524288, 77777, 3)])
def test_spectrogram_65536_with_the_frame_on_chip(oracle, nfft, hop, nframes):
    """nfft 65536 (spec_chip.h: the frame in the registers of a 512-thread workgroup, two exchanges through LDS) and 131072
    (two such workgroups per frame, even and odd bins, behind a radix-2 step of decimation in frequency) against the
    oracle and against the four-step path through HBM they replace ("spec_kernel" 2): runs of frames that end inside a
    workgroup's run, hops that leave the frames at odd addresses, a tail of frames behind the last whole window, the dB
    image, and an offset of 300 times the signal's amplitude on one channel (the frame mean relative to a pivot)."""
    from audian_amd import hipdsp
    rate, C = 192000.0, 3
    rng = np.random.default_rng(hop)
    T = (nframes - 1)*hop + nfft + 17
    x = synth(rng, T, C, rate)
    x[:, 1] = np.float32(0.01)*x[:, 1] + np.float32(3.0)
    nd = nframes + 2
    want = np.zeros((nd, C, nfft//2 + 1))
    oracle.spectrogram_process(x.astype(np.float64), want, rate, nfft, hop)
    c = gh.ctx()
    try:
        for fpw in (0, 1, 4):
            c.set_option('spec_fpw', fpw)
            got, db = gh.gpu_spectrogram(x, rate, nfft, hop, nd, want_db=True)
            for ch in range(C):
                for j in range(nd):
                    if np.max(np.abs(want[j, ch])) == 0:
                        assert j >= nframes and np.all(got[j, ch] == 0) and np.all(db[j, ch] == -np.inf)
                    else:
                        assert rel_err(got[j, ch], want[j, ch]) < TOL, (hop, fpw, j, ch)
            fin = got > 1e-20
            assert np.max(np.abs(db[fin] - 10*np.log10(got[fin]))) < 1e-3
        c.set_option('spec_kernel', 2)
        old = gh.gpu_spectrogram(x, rate, nfft, hop, nd)
        for ch in range(C):
            assert rel_err(got[:, ch], old[:, ch]) < 2e-5
    finally:
        c.set_option('spec_kernel', 0)
        c.set_option('spec_fpw', 0)


@pytest.mark.parametrize('nfft,hop', [(100, 30), (6174, 3087), (1000, 1000), (4097, 2000), (9, 4)])
def test_spectrogram_arbitrary_nfft_direct_dft(oracle, nfft, hop):
    """nfft values the reference's clamp to len(source)//2 can produce (not powers of two)."""
    rng = np.random.default_rng(nfft)
    rate = 44100.0
    nframes = 4
    T = (nframes - 1)*hop + nfft + 2
    x = (synth(rng, T, 2, rate) + np.float32(0.3)).astype(np.float32)
    nd = (T + hop - 1)//hop
    want = np.zeros((nd, 2, nfft//2 + 1))
    oracle.spectrogram_process(x.astype(np.float64), want, rate, nfft, hop)
    got = gh.gpu_spectrogram(x, rate, nfft, hop, nd)
    for ch in range(2):
        for j in range(nd):
            if np.max(np.abs(want[j, ch])) == 0:
                assert np.all(got[j, ch] == 0)
            else:
                assert rel_err(got[j, ch], want[j, ch]) < TOL, (nfft, j, ch)


@pytest.mark.parametrize('T', [10, 33, 2047, 2048, 2049, 2040, 4095, 70000, 1500000])
def test_fused_filter_envelope_equals_separate_calls(oracle, T):
    """hipdsp_sosfilt_envelope (band-pass + envelope state sweep, then the backward sweep) against the two
    separate calls and against the oracle, incl. tile borders inside the odd extension."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rate, C = 96000.0, 3
    rng = np.random.default_rng(T)
    x = synth(rng, T, C, rate)
    c = gh.ctx()
    dx = gh.to_planar(c, x)
    cases = [((300.0, 3000.0), 2, 20.0, 0.0, 2), ((300.0, 3000.0), 1, 500.0, 0.0, 2), ((100.0, 20000.0), 2, 800.0, 50.0, 2)]
    if T > 4000:
        cases.append(((300.0, 3000.0), 4, 500.0, 0.0, 3))
    for band, order, env, ehp, eorder in cases:
        sos = butter_sos(order, band, 'bandpass', rate)
        esos = butter_sos(eorder, (ehp, env), 'bandpass', rate) if ehp > 0 else butter_sos(eorder, env, 'lowpass', rate)
        if T <= oracle.sosfiltfilt_edge(esos):
            continue
        fplan, eplan = hipdsp.SosPlan(c, sos), hipdsp.SosPlan(c, esos)
        yf = hipdsp.DeviceArray(c, (C, T), np.float32)
        ye = hipdsp.DeviceArray(c, (C, T), np.float32)
        hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, yf, T, ye, T, C, T, clamp=ehp == 0)
        f1 = hipdsp.DeviceArray(c, (C, T), np.float32)
        e1 = hipdsp.DeviceArray(c, (C, T), np.float32)
        hipdsp.sosfilt(c, fplan, dx, T, f1, T, C, T, 0)
        hipdsp.envelope(c, eplan, f1, T, e1, T, C, T, 0, clamp=ehp == 0)
        gf, ge, sf, se = yf.to_host(), ye.to_host(), f1.to_host(), e1.to_host()
        want_f = oracle.sosfilt(sos, x.astype(np.float64))
        want_e = np.zeros_like(want_f)
        oracle.envelope_process(esos, sf.T.astype(np.float64), want_e, 0, highpass_cutoff=ehp)
        for ch in range(C):
            assert rel_err(gf[ch], sf[ch]) < 1e-6, (T, order, ch)
            assert rel_err(ge[ch], se[ch]) < 2e-6, (T, env, ch)
            assert rel_err(gf[ch], want_f[:, ch]) < TOL
            assert rel_err(ge[ch], want_e[:, ch]) < TOL
    with pytest.raises(ValueError):
        plan = hipdsp.SosPlan(c, butter_sos(2, 20.0, 'lowpass', rate))
        hipdsp.sosfilt_envelope(c, plan, plan, dx, T, dx, T, dx, T, C, 9)


@pytest.mark.parametrize('handover', [0, 4])
@pytest.mark.parametrize('T,max_segments', [(8192, 0), (20480, 0), (70001, 0), (300000, 0), (300000, 3),
                                            (1500000, 0), (1500000, 1)])
def test_chain_forward_equals_separate_calls(oracle, T, max_segments, handover):
    """hipdsp_chain_forward (band-pass + envelope state sweep + spectrogram 2048/1024 in one pass) and
    the backward sweep after it, against the separate calls and the oracle: one and many segments
    (frames that straddle a segment border), traces that end inside a tile, zero tail."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rate, C, nfft, hop = 96000.0, 3, 2048, 1024
    rng = np.random.default_rng(T + max_segments)
    x = (synth(rng, T, C, rate) + np.float32(0.05)).astype(np.float32)
    c = gh.ctx()
    c.set_max_segments(max_segments)
    c.set_option('chain_debug', handover)      # 0: pairwise flags in LDS, 4: workgroup barriers
    try:
        dx = gh.to_planar(c, x)
        nd = (T + hop - 1)//hop
        F = nfft//2 + 1
        for band, order, env, eorder in (((300.0, 3000.0), 2, 20.0, 2), ((1000.0, 20000.0), 1, 500.0, 4)):
            sos = butter_sos(order, band, 'bandpass', rate)
            esos = butter_sos(eorder, env, 'lowpass', rate)
            fplan, eplan = hipdsp.SosPlan(c, sos), hipdsp.SosPlan(c, esos)
            yf = hipdsp.DeviceArray(c, (C, T), np.float32)
            ye = hipdsp.DeviceArray(c, (C, T), np.float32)
            ps = hipdsp.DeviceArray(c, (C, nd, F), np.float32)
            hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(ps), 0x7f, 4*C*nd*F)      # every bin must be written
            hipdsp.chain_forward(c, fplan, eplan, dx, T, yf, T, C, T, nfft, hop, rate, ps, nd)
            hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, yf, T, ye, T, C, T, phase=2)
            f1 = hipdsp.DeviceArray(c, (C, T), np.float32)
            e1 = hipdsp.DeviceArray(c, (C, T), np.float32)
            s1 = hipdsp.DeviceArray(c, (C, nd, F), np.float32)
            hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, f1, T, e1, T, C, T)
            hipdsp.spectrogram(c, f1, T, C, T, nfft, hop, rate, s1, nd)
            gf, ge, gs = yf.to_host(), ye.to_host(), ps.to_host()
            sf, se, ss = f1.to_host(), e1.to_host(), s1.to_host()
            want_f = oracle.sosfilt(sos, x.astype(np.float64))
            want_e = np.zeros_like(want_f)
            oracle.envelope_process(esos, sf.T.astype(np.float64), want_e, 0)
            want_s = np.zeros((nd, C, F))
            oracle.spectrogram_process(sf.T.astype(np.float64), want_s, rate, nfft, hop)
            for ch in range(C):
                assert rel_err(gf[ch], sf[ch]) < 1e-6, (T, order, ch)
                assert rel_err(ge[ch], se[ch]) < 2e-6, (T, env, ch)
                assert rel_err(gf[ch], want_f[:, ch]) < TOL
                assert rel_err(ge[ch], want_e[:, ch]) < TOL
                for j in range(nd):
                    if np.max(np.abs(want_s[j, ch])) == 0:
                        assert np.all(gs[ch, j] == 0) and np.all(ss[ch, j] == 0), (T, j, ch)
                    else:
                        assert rel_err(gs[ch, j], want_s[j, ch]) < TOL, (T, max_segments, j, ch)
                        assert rel_err(gs[ch, j], ss[ch, j]) < 1e-5, (T, max_segments, j, ch)
        # fewer destination frames than the trace supports, more (zero tail), and a channel pitch
        # with a gap: nothing outside [0, frames_out) of a channel may be touched
        for nf in (max(nd - 7, 1), nd + 5):
            pitch = (nf + 3)*F
            big = hipdsp.DeviceArray(c, (C, nf + 3, F), np.float32)
            hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(big), 0x7f, 4*C*pitch)
            bigdb = hipdsp.DeviceArray(c, (C, nf + 3, F), np.float32)
            hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(bigdb), 0x7f, 4*C*pitch)
            hipdsp.chain_forward(c, fplan, eplan, dx, T, yf, T, C, T, nfft, hop, rate, big, nf, psd_pitch=pitch,
                                 db_out=bigdb)
            got, gdb = big.to_host(), bigdb.to_host()
            guard = np.frombuffer(b'\x7f\x7f\x7f\x7f', dtype=np.float32)[0]
            assert np.all(got[:, nf:, :] == guard) and np.all(gdb[:, nf:, :] == guard)
            m = min(nf, nd)
            assert np.array_equal(got[:, :m, :], gs[:, :m, :])
            assert np.all(got[:, nd:nf, :] == 0) and np.all(gdb[:, nd:nf, :] == -np.inf)
            want_db = oracle.decibel(got[:, :nf, :])             # the fused dB epilogue (specitem.py:36)
            fin = np.isfinite(want_db)
            assert np.array_equal(np.isfinite(gdb[:, :nf, :]), fin)
            assert np.max(np.abs(gdb[:, :nf, :][fin] - want_db[fin])) < 1e-3
        with pytest.raises(NotImplementedError):
            hipdsp.chain_forward(c, fplan, eplan, dx, T, yf, T, C, T, 256, 64, rate, ps, nd)
    finally:
        c.set_max_segments(0)
        c.set_option('chain_debug', 0)


@pytest.mark.parametrize('rows,cols,stride', [(1, 1, 1), (7, 3, 5), (938, 64, 1025), (5000, 8, 129), (40000, 64, 1025)])
def test_band_order_stats_bit_exact(rows, cols, stride):
    """hipdsp_band_order_stats: the two order statistics np.percentile(.., 95) interpolates between, for the
    top-band slab of BufferedSpectrogram.estimate_noiselevels (bufferedspectrogram.py:115-117) -- exact,
    including ties, exact zeros (decibel -> -inf) and a rank that is the last element."""
    from audian_amd import hipdsp
    c = gh.ctx()
    rng = np.random.default_rng(rows*cols)
    full = (rng.standard_normal((rows, stride))**2).astype(np.float32)
    full[rng.random((rows, stride)) < 0.05] = 0.0                      # exact zeros
    full[:, :2] = full[0, 0]                                           # ties
    band = full[:, stride - cols:]
    d = hipdsp.DeviceArray.from_host(c, full)
    out = hipdsp.DeviceArray(c, (2,), np.float32)
    srt = np.sort(band.ravel())
    n = rows*cols
    for rank in sorted({0, int(np.floor(0.95*(n - 1))), n//2, n - 1}):
        hipdsp.band_order_stats(c, d.view(stride - cols, (1,)), rows, cols, stride, rank, out)
        got = out.to_host()
        assert got[0] == srt[rank] and got[1] == srt[min(rank + 1, n - 1)], (rank, got, srt[rank])
    # np.percentile itself from the two statistics
    pos = 0.95*(n - 1)
    k = int(np.floor(pos))
    hipdsp.band_order_stats(c, d.view(stride - cols, (1,)), rows, cols, stride, k, out)
    lo, hi = out.to_host().astype(np.float64)
    assert abs((lo + (pos - k)*(hi - lo)) - np.percentile(band.astype(np.float64), 95)) <= 1e-12*max(1.0, hi)
    with pytest.raises(ValueError):
        hipdsp.band_order_stats(c, d, rows, cols, stride, n, out)


@pytest.mark.parametrize('T', [1, 5, 1023, 1024, 16384, 16385, 100000, 1200000])
def test_unwrap_matches_the_restated_audioio_algorithm(oracle, T):
    """hipdsp_unwrap (src/audian/data.py:180 -> audioio's unwrap() on the raw loader's buffers): a signal
    that left [-1, 1) and wrapped around in the file is put back together.  audioio's source is not in
    the reference tree nor in this image, so the oracle is restated from its documentation -- "restated
    from documentation, unpinned" -- and the device path is held to that restatement bit for bit; in
    addition the true (unwrapped) signal must come back exactly wherever it can be represented."""
    from audian_amd import hipdsp
    c = gh.ctx()
    rate, C = 48000.0, 3
    t = np.arange(T)/rate
    rng = np.random.default_rng(T)
    # (every channel starts inside the range: a slab that begins wrapped cannot be told from one that does not)
    true = np.stack([2.6*np.sin(2*np.pi*(40.0 + 13*ch)*t) + 0.05*rng.standard_normal(T) for ch in range(C)], axis=1)
    wrapped = ((true + 1.0) % 2.0 - 1.0).astype(np.float32)          # what the 16-bit file would hold
    dx = gh.to_planar(c, wrapped)
    for thresh, clips, down in [(1.5, False, True), (1.5, True, False), (1.5, False, False), (1.0, False, True)]:
        dy = hipdsp.DeviceArray(c, (C, T), np.float32)
        hipdsp.unwrap(c, dx, T, C, T, thresh, dy, T, clips=clips, down_scale=down)
        got = dy.to_host().T
        want = oracle.unwrap(wrapped, thresh, clips=clips, down_scale=down)
        assert np.array_equal(got, want), (T, thresh, clips, down)
    if T > 1:
        # with neither clipping nor down-scaling the original comes back (steps of the true signal stay
        # far below the threshold at these frequencies)
        dy = hipdsp.DeviceArray(c, (C, T), np.float32)
        hipdsp.unwrap(c, dx, T, C, T, 1.5, dy, T, clips=False, down_scale=False)
        assert np.max(np.abs(dy.to_host().T - true)) < 1e-5
    with pytest.raises(ValueError):
        hipdsp.unwrap(c, dx, T, C, T, 0.0, dx, T)


@pytest.mark.parametrize('T', [60, 2500, 70000, 400000])
def test_envelope_cascades_longer_than_one_plan(oracle, T):
    """hipdsp_envelope_multi: BufferedEnvelope accepts any filter_order (bufferedenvelope.py:13-16,44-55);
    a band-pass envelope of order 5 has five sections, a low-pass of order 9 five, of order 12 six --
    more than one plan holds.  sosfiltfilt is run over chained plans (pad length and sosfilt_zi of the
    WHOLE cascade); against the oracle, against the single-plan kernel where the cascade fits one, and
    scipy's ValueError for slabs not longer than the pad length."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rate, C = 48000.0, 3
    rng = np.random.default_rng(T)
    x = synth(rng, T, C, rate)
    c = gh.ctx()
    dx = gh.to_planar(c, x)
    cases = [(butter_sos(5, (10.0, 500.0), 'bandpass', rate), 10.0, (4, 1)),
             (butter_sos(5, (10.0, 500.0), 'bandpass', rate), 10.0, (2, 2, 1)),
             (butter_sos(9, 800.0, 'lowpass', rate), 0.0, (4, 1)),
             (butter_sos(12, 2000.0, 'lowpass', rate), 0.0, (3, 3)),
             (butter_sos(8, (50.0, 4000.0), 'bandpass', rate), 50.0, (4, 4)),
             (butter_sos(4, (100.0, 900.0), 'bandpass', rate), 100.0, (2, 2))]
    for sos, ehp, split in cases:
        assert sum(split) == len(sos)
        plans, i = [], 0
        for n in split:
            plans.append(hipdsp.SosPlan(c, sos[i:i + n]))
            i += n
        edge = oracle.sosfiltfilt_edge(sos)
        for skip in (0, 7) if T > 100 else (0,):
            dy = hipdsp.DeviceArray(c, (C, max(T - skip, 1)), np.float32)
            if T <= edge:
                with pytest.raises(ValueError, match='padlen'):
                    hipdsp.envelope_multi(c, plans, dx, T, dy, max(T - skip, 1), C, T, skip, clamp=ehp == 0)
                continue
            hipdsp.envelope_multi(c, plans, dx, T, dy, T - skip, C, T, skip, clamp=ehp == 0)
            got = gh.from_planar(c, dy, T - skip, C, pitch=T - skip)
            want = np.zeros((T - skip, C))
            oracle.envelope_process(sos, x.astype(np.float64), want, skip, highpass_cutoff=ehp)
            for ch in range(C):
                assert rel_err(got[:, ch], want[:, ch]) < TOL, (T, len(sos), split, skip, ch)
            if len(sos) <= 4 and skip == 0:
                one = gh.gpu_envelope(sos, x, clamp=ehp == 0)
                for ch in range(C):
                    assert rel_err(got[:, ch], one[:, ch]) < 1e-5


@pytest.mark.parametrize('nfft,hop', [(2048, 512), (1024, 512), (1024, 256), (512, 256), (256, 128), (2048, 1024)])
@pytest.mark.parametrize('T,max_segments', [(8192, 0), (20481, 0), (70001, 0), (300000, 3), (1500000, 0)])
def test_chain_forward_other_windows_and_longer_bandpasses(oracle, T, max_segments, nfft, hop):
    """The fused forward sweep for every window the kernel is built for (frames are register windows of a
    2048-sample tile: 50 % and 75 % overlap at nfft 2048 and 1024; BASELINE configs[1] is 1024/256 with a
    four-section band-pass) and band-pass plans of three and four sections: against the separate calls and
    the oracle, one and many segments, traces that end inside a tile, every frame written, zero tail."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rate, C = 48000.0, 3
    rng = np.random.default_rng(T + max_segments + nfft + hop)
    x = (synth(rng, T, C, rate) + np.float32(0.05)).astype(np.float32)
    c = gh.ctx()
    c.set_max_segments(max_segments)
    try:
        dx = gh.to_planar(c, x)
        nd = (T + hop - 1)//hop
        F = nfft//2 + 1
        plans = [((300.0, 3000.0), 4, 20.0, 2), ((500.0, 9000.0), 3, 400.0, 4)]
        if (nfft, hop) != (2048, 1024):
            plans.append(((300.0, 3000.0), 2, 20.0, 2))
        for band, order, env, eorder in plans:
            sos = butter_sos(order, band, 'bandpass', rate)
            esos = butter_sos(eorder, env, 'lowpass', rate)
            fplan, eplan = hipdsp.SosPlan(c, sos), hipdsp.SosPlan(c, esos)
            yf = hipdsp.DeviceArray(c, (C, T), np.float32)
            ye = hipdsp.DeviceArray(c, (C, T), np.float32)
            ps = hipdsp.DeviceArray(c, (C, nd + 2, F), np.float32)
            hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(ps), 0x7f, 4*C*(nd + 2)*F)      # every bin must be written
            hipdsp.chain_forward(c, fplan, eplan, dx, T, yf, T, C, T, nfft, hop, rate, ps, nd + 2)
            hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, yf, T, ye, T, C, T, phase=2)
            f1 = hipdsp.DeviceArray(c, (C, T), np.float32)
            e1 = hipdsp.DeviceArray(c, (C, T), np.float32)
            s1 = hipdsp.DeviceArray(c, (C, nd + 2, F), np.float32)
            hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, f1, T, e1, T, C, T)
            hipdsp.spectrogram(c, f1, T, C, T, nfft, hop, rate, s1, nd + 2)
            gf, ge, gs = yf.to_host(), ye.to_host(), ps.to_host()
            sf, se, ss = f1.to_host(), e1.to_host(), s1.to_host()
            want_f = oracle.sosfilt(sos, x.astype(np.float64))
            want_e = np.zeros_like(want_f)
            oracle.envelope_process(esos, sf.T.astype(np.float64), want_e, 0)
            want_s = np.zeros((nd + 2, C, F))
            oracle.spectrogram_process(sf.T.astype(np.float64), want_s, rate, nfft, hop)
            for ch in range(C):
                assert rel_err(gf[ch], sf[ch]) < 1e-6, (T, order, ch)
                assert rel_err(ge[ch], se[ch]) < 2e-6, (T, env, ch)
                assert rel_err(gf[ch], want_f[:, ch]) < TOL
                assert rel_err(ge[ch], want_e[:, ch]) < TOL
                for j in range(nd + 2):
                    if np.max(np.abs(want_s[j, ch])) == 0:
                        assert np.all(gs[ch, j] == 0) and np.all(ss[ch, j] == 0), (T, j, ch)
                    else:
                        assert rel_err(gs[ch, j], want_s[j, ch]) < TOL, (T, max_segments, j, ch)
                        assert rel_err(gs[ch, j], ss[ch, j]) < 1e-5, (T, max_segments, j, ch)
            # the fused dB epilogue (SpecItem.update_plot's decibel(), specitem.py:36) for every window: same PSD
            # bit for bit, decibel(PSD) next to it, -inf at and below 1e-20 and in the zero tail
            db = hipdsp.DeviceArray(c, (C, nd + 2, F), np.float32)
            ps2 = hipdsp.DeviceArray(c, (C, nd + 2, F), np.float32)
            y2 = hipdsp.DeviceArray(c, (C, T), np.float32)
            for arr in (db, ps2):
                hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(arr), 0x7f, 4*C*(nd + 2)*F)
            hipdsp.chain_forward(c, fplan, eplan, dx, T, y2, T, C, T, nfft, hop, rate, ps2, nd + 2, db_out=db)
            assert np.array_equal(ps2.to_host(), gs) and np.array_equal(y2.to_host(), gf)
            gdb, wdb = db.to_host(), oracle.decibel(gs.astype(np.float64))
            fin = np.isfinite(wdb)
            assert np.array_equal(np.isfinite(gdb), fin) and np.all(gdb[~fin] == -np.inf)
            assert np.max(np.abs(gdb[fin] - wdb[fin])) < 1e-3, (T, nfft, hop)
            # ... and without an envelope behind the filter (eplan NULL: the reference's default trace set)
            hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(ps2), 0x7f, 4*C*(nd + 2)*F)
            hipdsp.chain_forward(c, fplan, None, dx, T, y2, T, C, T, nfft, hop, rate, ps2, nd + 2)
            assert np.array_equal(ps2.to_host(), gs) and np.array_equal(y2.to_host(), gf)
    finally:
        c.set_max_segments(0)


SCROLLED = [(0, 48000), (777, 48000), (1, 0), (255, 2047), (1000, 2040), (127, 100003), (5, 2049), (1023, 14), (300, 16),
            (64, 4095), (0, 1), (511, 6144)]


@pytest.mark.parametrize('nfft,hop', [(2048, 1024), (1024, 256), (256, 128), (512, 256), (2048, 512)])
@pytest.mark.parametrize('T,max_segments', [(8192, 0), (70001, 0), (300000, 0), (300000, 1), (300000, 37), (1500000, 0)])
def test_chain_forward_at_any_scroll_position(oracle, T, max_segments, nfft, hop):
    """After a scroll the filtered buffer starts at an arbitrary sample of the recording: the spectrogram's frame 0
    then starts spec_first = ceil(offset / hop) hop - offset samples into it and the envelope, whose second of pre-roll
    BufferedData.align_buffer trims (buffereddata.py:75-88), is sosfiltfilt of filtered[env_first:] only.  The fused
    launch takes both offsets (its tile grid shifts; the tile the envelope starts in holds scipy's odd extension and
    its steady state): against the oracle on the sliced filtered trace and against the separate calls on the same
    slices, for offsets on both sides of tile, lane-row and hop borders, one and many segments, with and without
    spec_frames, and through hipdsp_sosfilt_envelope's own forward sweep (sos_ckpt_kernel) as well."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rate, C = 48000.0, 2
    rng = np.random.default_rng(T + max_segments + nfft + hop)
    x = (synth(rng, T, C, rate) + np.float32(0.05)).astype(np.float32)
    c = gh.ctx()
    c.set_max_segments(max_segments)
    F = nfft//2 + 1
    sos = butter_sos(2, (300.0, 3000.0), 'bandpass', rate)
    fplan = hipdsp.SosPlan(c, sos)
    want_f = oracle.sosfilt(sos, x.astype(np.float64))
    try:
        dx = gh.to_planar(c, x)
        f1 = hipdsp.DeviceArray(c, (C, T), np.float32)
        hipdsp.sosfilt(c, fplan, dx, T, f1, T, C, T, 0)
        sf = f1.to_host()
        for i, (spec_first, env_first) in enumerate(SCROLLED):
            if env_first + 200 >= T or spec_first + nfft >= T:
                continue
            esos = butter_sos(*[(2, 20.0), (4, 300.0), (2, (5.0, 200.0))][i % 3],
                              'bandpass' if i % 3 == 2 else 'lowpass', rate)
            eplan = hipdsp.SosPlan(c, esos)
            clamp = i % 3 != 2
            nsrc = T - spec_first
            nd = (nsrc + hop - 1)//hop + 1
            spec_frames = 0 if i % 2 == 0 else max(nfft, nsrc - 3*hop - 7)
            yf = hipdsp.DeviceArray(c, (C, T), np.float32)
            ye = hipdsp.DeviceArray(c, (C, T - env_first), np.float32)
            ps = hipdsp.DeviceArray(c, (C, nd, F), np.float32)
            for arr, n in ((yf, C*T), (ye, C*(T - env_first)), (ps, C*nd*F)):
                hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(arr), 0x7f, 4*n)            # every value must be written
            hipdsp.chain_forward(c, fplan, eplan, dx, T, yf, T, C, T, nfft, hop, rate, ps, nd, spec_frames=spec_frames,
                                 spec_first=spec_first, env_first=env_first)
            hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, yf, T, ye, T - env_first, C, T, clamp=clamp, phase=2,
                                    env_first=env_first)
            gf, ge, gs = yf.to_host(), ye.to_host(), ps.to_host()
            want_e = np.zeros((T - env_first, C))
            oracle.envelope_process(esos, sf.T[env_first:].astype(np.float64), want_e, 0)
            if not clamp:
                want_e = oracle.sosfiltfilt(esos, (np.pi/2)*np.abs(sf.T[env_first:].astype(np.float64)))
            want_s = np.zeros((nd, C, F))
            src = sf.T[spec_first:spec_first + (spec_frames or nsrc)].astype(np.float64)
            oracle.spectrogram_process(src, want_s, rate, nfft, hop)
            # the separate calls on the same slices
            e1 = hipdsp.DeviceArray(c, (C, T - env_first), np.float32)
            hipdsp.envelope(c, eplan, f1.view(env_first, (1,)), T, e1, T - env_first, C, T - env_first, 0, clamp=clamp)
            se = e1.to_host()
            what = (T, max_segments, nfft, hop, spec_first, env_first)
            for ch in range(C):
                assert rel_err(gf[ch], sf[ch]) < 1e-6, what
                assert rel_err(gf[ch], want_f[:, ch]) < TOL, what
                assert np.all(np.isfinite(ge[ch])), what
                assert rel_err(ge[ch], se[ch]) < 5e-6, what
                assert rel_err(ge[ch], want_e[:, ch]) < TOL, what
                # ... also right at the envelope's first samples, where the extension and its steady state act
                scale = np.max(np.abs(want_e[:, ch]))
                assert np.max(np.abs(ge[ch][:4096] - want_e[:4096, ch])) < TOL*scale, what
                for j in range(nd):
                    if np.max(np.abs(want_s[j, ch])) == 0:
                        assert np.all(gs[ch, j] == 0), what + (j,)
                    else:
                        assert rel_err(gs[ch, j], want_s[j, ch]) < TOL, what + (j,)
            # the same envelope through hipdsp_sosfilt_envelope's own forward sweep (phase 0, and phases 1 + 2)
            for phases in ((0,), (1, 2)):
                y2 = hipdsp.DeviceArray(c, (C, T), np.float32)
                e2 = hipdsp.DeviceArray(c, (C, T - env_first), np.float32)
                hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(e2), 0x7f, 4*C*(T - env_first))
                for ph in phases:
                    hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, y2, T, e2, T - env_first, C, T, clamp=clamp, phase=ph,
                                            env_first=env_first)
                g2, gy = e2.to_host(), y2.to_host()
                for ch in range(C):
                    assert rel_err(gy[ch], sf[ch]) < 1e-6, what
                    assert rel_err(g2[ch], se[ch]) < 5e-6, what + (phases,)
                    assert rel_err(g2[ch], want_e[:, ch]) < TOL, what + (phases,)
            # a backward sweep must be given the envelope start of the forward sweep whose tile states it consumes
            with pytest.raises(ValueError):
                hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, yf, T, ye, T - env_first, C, T, phase=2,
                                        env_first=env_first + 1)
            # without an envelope behind the filter, and with the dB epilogue
            if i % 4 == 0:
                db = hipdsp.DeviceArray(c, (C, nd, F), np.float32)
                ps2 = hipdsp.DeviceArray(c, (C, nd, F), np.float32)
                y3 = hipdsp.DeviceArray(c, (C, T), np.float32)
                hipdsp.chain_forward(c, fplan, None, dx, T, y3, T, C, T, nfft, hop, rate, ps2, nd, spec_frames=spec_frames,
                                     spec_first=spec_first, db_out=db)
                g3 = ps2.to_host()
                assert np.array_equal(y3.to_host(), gf), what
                for ch in range(C):
                    for j in range(nd):
                        assert np.all(g3[ch, j] == 0) if np.max(np.abs(want_s[j, ch])) == 0 else \
                            rel_err(g3[ch, j], gs[ch, j]) < 1e-5, what + (j,)
                gdb, wdb = db.to_host(), oracle.decibel(g3.astype(np.float64))
                fin = np.isfinite(wdb)
                assert np.array_equal(np.isfinite(gdb), fin) and np.max(np.abs(gdb[fin] - wdb[fin])) < 1e-3, what
    finally:
        c.set_max_segments(0)


@pytest.mark.parametrize('T,C,env_first', [(300000, 3, 0), (1500000, 2, 0), (400000, 2, 48000), (70001, 1, 12345)])
def test_role_split_backward_sweep_is_the_same_envelope(oracle, T, C, env_first):
    """The experiment of round 4 ("sos_split": compute waves that issue no vector-memory instruction for interior tiles,
    mover waves that do nothing else, envsplit.hip; dropped as the default -- profiles/r04j_bwd_role_split_ab.log -- but
    kept reproducible): the same envelope as env_bwd_kernel up to the float32 hand-over between its two cascades, and
    the oracle's within the parity bar, for one and many segments and an envelope that starts inside the trace."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rate = 48000.0
    rng = np.random.default_rng(T + C)
    x = synth(rng, T, C, rate)
    c = gh.ctx()
    try:
        c.set_option('sos_split', 1)
        c.set_option('sos_split', 0)
    except NotImplementedError:
        pytest.skip('library built without envsplit.hip (make -C audian_amd/csrc SPLIT=1)')
    sos = butter_sos(2, (300.0, 3000.0), 'bandpass', rate)
    fplan = hipdsp.SosPlan(c, sos)
    dx = gh.to_planar(c, x)
    for esos in (butter_sos(2, 20.0, 'lowpass', rate), butter_sos(4, 300.0, 'lowpass', rate)):
        eplan = hipdsp.SosPlan(c, esos)
        yf = hipdsp.DeviceArray(c, (C, T), np.float32)
        got = []
        for mode in (0, 1):
            ye = hipdsp.DeviceArray(c, (C, T - env_first), np.float32)
            hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(ye), 0x7f, 4*C*(T - env_first))
            hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, yf, T, ye, T - env_first, C, T, phase=1, env_first=env_first)
            c.set_option('sos_split', mode)
            try:
                hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, yf, T, ye, T - env_first, C, T, phase=2, env_first=env_first)
            finally:
                c.set_option('sos_split', 0)
            got.append(ye.to_host())
        sf = yf.to_host()
        want = np.zeros((T - env_first, C))
        oracle.envelope_process(esos, sf.T[env_first:].astype(np.float64), want, 0)
        for ch in range(C):
            assert rel_err(got[1][ch], got[0][ch]) < 1e-4, (T, ch)
            assert rel_err(got[1][ch], want[:, ch]) < TOL, (T, ch)
            assert rel_err(got[0][ch], want[:, ch]) < TOL, (T, ch)


def test_envelope_start_inside_the_trace_too_short_raises():
    """frames - env_first <= padlen: scipy's ValueError, as for a slab of that length."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    c = gh.ctx()
    rate, C, T = 48000.0, 1, 20000
    fplan = hipdsp.SosPlan(c, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
    eplan = hipdsp.SosPlan(c, butter_sos(2, 20.0, 'lowpass', rate))
    edge = eplan.info()[1]
    dx = gh.to_planar(c, np.ones((T, C), dtype=np.float32))
    yf, ye = hipdsp.DeviceArray(c, (C, T), np.float32), hipdsp.DeviceArray(c, (C, T), np.float32)
    ps = hipdsp.DeviceArray(c, (C, 20, 1025), np.float32)
    with pytest.raises(ValueError, match='padlen'):
        hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, yf, T, ye, T, C, T, env_first=T - edge)
    with pytest.raises(ValueError, match='padlen'):
        hipdsp.chain_forward(c, fplan, eplan, dx, T, yf, T, C, T, 2048, 1024, rate, ps, 20, env_first=T - edge)
    hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, yf, T, ye, T, C, T, env_first=T - edge - 1)
    c.synchronize()


@pytest.mark.parametrize('T,max_segments', [(8192, 0), (20481, 0), (70001, 0), (300000, 0), (300000, 3), (1500000, 0),
                                            (1500000, 1)])
def test_chain_frame_split_forward_and_backward(oracle, T, max_segments):
    """Frame split of the batch chain ("chain_split_frames"): hipdsp_chain_forward writes the even frames,
    hipdsp_chain_backward the envelope AND the odd frames (second half of tile t + first half of tile t+1, walked
    backwards).  Together they must equal the unsplit pair of calls bit for bit in the filtered trace, to
    float32 rounding in envelope and PSD, and the oracle within 1e-4; every frame below n_valid written exactly
    once (the PSD is pre-filled with a guard pattern), zero tail, one and many segments, traces that end
    inside a tile."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rate, C, nfft, hop = 96000.0, 3, 2048, 1024
    rng = np.random.default_rng(T + max_segments)
    x = (synth(rng, T, C, rate) + np.float32(0.05)).astype(np.float32)
    c = gh.ctx()
    c.set_max_segments(max_segments)
    try:
        dx = gh.to_planar(c, x)
        nd = (T + hop - 1)//hop
        F = nfft//2 + 1
        for band, order, env, eorder in (((300.0, 3000.0), 2, 20.0, 2), ((1000.0, 20000.0), 1, 500.0, 4),
                                         ((300.0, 3000.0), 4, 100.0, 2)):
            sos = butter_sos(order, band, 'bandpass', rate)
            esos = butter_sos(eorder, env, 'lowpass', rate)
            fplan, eplan = hipdsp.SosPlan(c, sos), hipdsp.SosPlan(c, esos)
            yf = hipdsp.DeviceArray(c, (C, T), np.float32)
            ye = hipdsp.DeviceArray(c, (C, T), np.float32)
            ps = hipdsp.DeviceArray(c, (C, nd + 1, F), np.float32)
            hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(ps), 0x7f, 4*C*(nd + 1)*F)
            c.set_option('chain_split_frames', 1)
            hipdsp.chain_forward(c, fplan, eplan, dx, T, yf, T, C, T, nfft, hop, rate, ps, nd + 1)
            guard = np.frombuffer(b'\x7f\x7f\x7f\x7f', dtype=np.float32)[0]
            half = ps.to_host()
            n_valid = (min(nd*hop + nfft, T) - (nfft - hop))//hop if T >= nfft else 0
            for j in range(n_valid):
                assert np.all(half[:, j] == guard) == (j % 2 == 1), (T, j)      # odd frames still untouched
            hipdsp.chain_backward(c, eplan, yf, T, ye, T, C, T, nfft, hop, rate, ps, nd + 1)
            c.set_option('chain_split_frames', 0)
            f1 = hipdsp.DeviceArray(c, (C, T), np.float32)
            e1 = hipdsp.DeviceArray(c, (C, T), np.float32)
            s1 = hipdsp.DeviceArray(c, (C, nd + 1, F), np.float32)
            hipdsp.chain_forward(c, fplan, eplan, dx, T, f1, T, C, T, nfft, hop, rate, s1, nd + 1)
            hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, f1, T, e1, T, C, T, phase=2)
            gf, ge, gs = yf.to_host(), ye.to_host(), ps.to_host()
            sf, se, ss = f1.to_host(), e1.to_host(), s1.to_host()
            assert np.array_equal(gf, sf)
            want_e = np.zeros((T, C))
            oracle.envelope_process(esos, sf.T.astype(np.float64), want_e, 0)
            want_s = np.zeros((nd + 1, C, F))
            oracle.spectrogram_process(sf.T.astype(np.float64), want_s, rate, nfft, hop)
            for ch in range(C):
                assert rel_err(ge[ch], se[ch]) < 2e-6, (T, env, ch)
                assert rel_err(ge[ch], want_e[:, ch]) < TOL
                for j in range(nd + 1):
                    if np.max(np.abs(want_s[j, ch])) == 0:
                        assert np.all(gs[ch, j] == 0), (T, j, ch)
                    else:
                        assert rel_err(gs[ch, j], want_s[j, ch]) < TOL, (T, max_segments, j, ch)
                        assert rel_err(gs[ch, j], ss[ch, j]) < 1e-5, (T, max_segments, j, ch)
        with pytest.raises(NotImplementedError):
            hipdsp.chain_backward(c, eplan, yf, T, ye, T, C, T, 1024, 512, rate, ps, nd)
    finally:
        c.set_max_segments(0)
        c.set_option('chain_split_frames', 0)


@pytest.mark.parametrize('T,max_segments', [(70001, 0), (300000, 3)])
def test_cascades_whose_numerators_are_not_scipys_take_the_general_phase_3(oracle, T, max_segments):
    """Round 5: sections behind the first whose numerator is exactly [1, +-2, 1] -- every Butterworth design of scipy --
    run a four-operation phase 3 (SosPlanDev::unit_tail); any other table takes the general five-operation loop, which
    the Butterworth-only suite would no longer reach for cascades of two and more sections.  The same transfer function
    with its gain spread differently over the sections (not unit any more), and a table that is no Butterworth design at
    all: through hipdsp_sosfilt, hipdsp_envelope, hipdsp_sosfilt_envelope and hipdsp_chain_forward (which covers general
    numerators for one and two band-pass sections), against the oracle and against the unit-form run."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rate, C, nfft, hop = 48000.0, 2, 1024, 256
    rng = np.random.default_rng(T)
    x = synth(rng, T, C, rate)
    c = gh.ctx()
    nd, F = (T + hop - 1)//hop, nfft//2 + 1

    def spread(sos):                                  # the same filter, gain moved between the sections
        t = np.array(sos, dtype=np.float64)
        for i in range(1, len(t)):
            t[0, :3] *= 1.0/(1.5 + i)
            t[i, :3] *= 1.5 + i
        return t
    for order in (2, 3, 4):
        unit = butter_sos(order, (300.0, 3000.0), 'bandpass', rate)
        for sos in (spread(unit), unit*np.array([1, 1, 1, 1, 0.999, 0.998])):   # (the second: poles moved, no design of anybody's)
            want = oracle.sosfilt(sos, x.astype(np.float64))
            got = gh.gpu_sosfilt(sos, x, max_segments=max_segments)
            for ch in range(C):
                assert rel_err(got[:, ch], want[:, ch]) < TOL, (order, ch)
    # the envelope: a low-pass of order 4 (two sections) with the gain spread, both sweeps
    eunit = butter_sos(4, 300.0, 'lowpass', rate)
    esos = spread(eunit)
    sos = spread(butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
    filt = oracle.sosfilt(sos, x.astype(np.float64))
    yf32 = gh.gpu_sosfilt(sos, x).astype(np.float32)
    want_e = np.zeros((T, C))
    oracle.envelope_process(esos, yf32.astype(np.float64), want_e, 0)
    got_e = gh.gpu_envelope(esos, yf32, max_segments=max_segments)
    ref_e = gh.gpu_envelope(eunit, yf32, max_segments=max_segments)
    for ch in range(C):
        assert rel_err(got_e[:, ch], want_e[:, ch]) < TOL and rel_err(got_e[:, ch], ref_e[:, ch]) < 1e-5, ch
    # the fused launches with general numerators in both cascades (two band-pass sections: the general loop; the unit run next to it)
    c.set_max_segments(max_segments)
    try:
        dx = gh.to_planar(c, x)
        outs = []
        for fs_, es_ in ((sos, esos), (butter_sos(2, (300.0, 3000.0), 'bandpass', rate), eunit)):
            fplan, eplan = hipdsp.SosPlan(c, fs_), hipdsp.SosPlan(c, es_)
            yf, ye = (hipdsp.DeviceArray(c, (C, T), np.float32) for _ in range(2))
            ps = hipdsp.DeviceArray(c, (C, nd, F), np.float32)
            hipdsp.chain_forward(c, fplan, eplan, dx, T, yf, T, C, T, nfft, hop, rate, ps, nd)
            hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, yf, T, ye, T, C, T, phase=2)
            outs.append((yf.to_host(), ye.to_host(), ps.to_host()))
            y2, e2 = (hipdsp.DeviceArray(c, (C, T), np.float32) for _ in range(2))
            hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, y2, T, e2, T, C, T)          # sos_ckpt_kernel + env_bwd_kernel
            assert np.array_equal(y2.to_host(), outs[-1][0])
            for ch in range(C):
                assert rel_err(e2.to_host()[ch], outs[-1][1][ch]) < 2e-6, ch
        (gf, ge, gs), (uf, ue, us) = outs
        want_s = np.zeros((nd, C, F))
        oracle.spectrogram_process(gf.T.astype(np.float64), want_s, rate, nfft, hop)
        want_env = np.zeros((T, C))
        oracle.envelope_process(esos, gf.T.astype(np.float64), want_env, 0)
        for ch in range(C):
            assert rel_err(gf[ch], filt[:, ch]) < TOL and rel_err(gf[ch], uf[ch]) < 1e-5, ch
            assert rel_err(ge[ch], want_env[:, ch]) < TOL and rel_err(ge[ch], ue[ch]) < 1e-4, ch
            for j in range(nd):
                if np.max(np.abs(want_s[j, ch])) == 0:
                    assert np.all(gs[ch, j] == 0)
                else:
                    assert rel_err(gs[ch, j], want_s[j, ch]) < TOL, (j, ch)
    finally:
        c.set_max_segments(0)


def test_backward_sweeps_refuse_tile_states_that_are_not_theirs():
    """hipdsp_chain_backward walks the unshifted tile grid only -- behind a forward sweep whose grid was
    moved (spec_first / env_first > 0: the states sit on a grid of frames + lead samples) it must say so instead of
    reading them with the wrong pitch; and no backward sweep (phase 2 or hipdsp_chain_backward) may run on tile states
    another call has overwritten in the context's scratch since (here: the long-window spectrogram's work area)."""
    from audian_amd import hipdsp, _lib
    from audian_amd.design import butter_sos
    rate, C, T, nfft, hop = 96000.0, 2, 40000, 2048, 1024
    rng = np.random.default_rng(5)
    x = synth(rng, T, C, rate)
    c = gh.ctx()
    dx = gh.to_planar(c, x)
    nd = (T + hop - 1)//hop
    fplan = hipdsp.SosPlan(c, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
    eplan = hipdsp.SosPlan(c, butter_sos(2, 20.0, 'lowpass', rate))
    yf, ye = (hipdsp.DeviceArray(c, (C, T), np.float32) for _ in range(2))
    ps = hipdsp.DeviceArray(c, (C, nd, nfft//2 + 1), np.float32)
    c.set_option('chain_split_frames', 1)
    try:
        for kw in ({'spec_first': 300}, {'env_first': 5000}):
            hipdsp.chain_forward(c, fplan, eplan, dx, T, yf, T, C, T, nfft, hop, rate, ps, nd - 1, **kw)
            with pytest.raises(ValueError, match='unshifted grid'):       # (HIPDSP_ERR_INVALID)
                hipdsp.chain_backward(c, eplan, yf, T, ye, T, C, T, nfft, hop, rate, ps, nd)
        # the right forward sweep, but a call that uses the scratch in between
        for backward in (lambda: hipdsp.chain_backward(c, eplan, yf, T, ye, T, C, T, nfft, hop, rate, ps, nd),
                         lambda: hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, yf, T, ye, T, C, T, phase=2)):
            hipdsp.chain_forward(c, fplan, eplan, dx, T, yf, T, C, T, nfft, hop, rate, ps, nd)
            gh.gpu_spectrogram(x, rate, 32768, 16384, 2)       # (long windows work in the scratch of their context)
            c.reserve(1 << 20)                                  # hipdsp_ctx_reserve: asks for the scratch
            with pytest.raises(ValueError, match='tile states'):
                backward()
            hipdsp.chain_forward(c, fplan, eplan, dx, T, yf, T, C, T, nfft, hop, rate, ps, nd)
            _lib.check(hipdsp.lib.hipdsp_pool_trim(c.handle))   # gives the scratch back altogether
            with pytest.raises(ValueError, match='tile states'):
                backward()
        # and the pair that belongs together still works, twice over (a backward sweep does not consume the states)
        hipdsp.chain_forward(c, fplan, eplan, dx, T, yf, T, C, T, nfft, hop, rate, ps, nd)
        hipdsp.chain_backward(c, eplan, yf, T, ye, T, C, T, nfft, hop, rate, ps, nd)
        first = ye.to_host()
        hipdsp.chain_backward(c, eplan, yf, T, ye, T, C, T, nfft, hop, rate, ps, nd)
        assert np.array_equal(first, ye.to_host())
    finally:
        c.set_option('chain_split_frames', 0)


@pytest.mark.parametrize('nfft', [8, 64, 256, 512, 1024, 2048, 4096, 8192, 65536, 300])
def test_spectrogram_of_a_slab_shorter_than_one_window(nfft):
    """bufferedspectrogram.py:47-49: fewer source samples than one window -> the whole destination is zero (dB:
    -inf), for every kernel family, and NOTHING may read the slab (tools/fuzz_stress.py found the three-stage
    kernel requesting frame 0 before looking at the frame count: a memory fault when the slab ended at the end
    of a mapping).  The source pointer is the library's smallest allocation here."""
    from audian_amd import hipdsp
    c = gh.ctx()
    C, nd, F = 3, 5, nfft//2 + 1
    for T in (0, 1, nfft//2 + 1, nfft - 1):
        dx = hipdsp.DeviceArray(c, (C, max(T, 1)), np.float32)
        dx.zero_()
        out = hipdsp.DeviceArray(c, (C, nd, F), np.float32)
        db = hipdsp.DeviceArray(c, (C, nd, F), np.float32)
        hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(out), 0x7f, 4*C*nd*F)
        hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(db), 0x7f, 4*C*nd*F)
        hipdsp.spectrogram(c, dx, max(T, 1), C, T, nfft, max(nfft//4, 1), 48000.0, out, nd, db_out=db)
        assert np.all(out.to_host() == 0) and np.all(db.to_host() == -np.inf), (nfft, T)


def test_copy_probe_copies():
    """hipdsp_copy_probe (the measured device-copy ceiling bench.py reports): bit-exact copy, sizes that are not a
    multiple of the block, misuse rejected."""
    from audian_amd import hipdsp
    c = gh.ctx()
    rng = np.random.default_rng(3)
    for n in (4, 1024, 1024*257 + 12):
        x = rng.standard_normal(n).astype(np.float32)
        src = hipdsp.DeviceArray.from_host(c, x)
        dst = hipdsp.DeviceArray(c, (n,), np.float32).zero_()
        hipdsp.check(hipdsp.lib.hipdsp_copy_probe(c.handle, hipdsp._p(dst), hipdsp._p(src), 4*n))
        assert np.array_equal(dst.to_host(), x)
    with pytest.raises(ValueError):
        hipdsp.check(hipdsp.lib.hipdsp_copy_probe(c.handle, hipdsp._p(dst), hipdsp._p(src), 10))


def same_nan_and_close(got, want, what):
    """NaN exactly where the oracle has NaN; the rest within TOL of the oracle (relative to its largest finite value)."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, what
    bad = ~np.isfinite(want)
    assert np.array_equal(~np.isfinite(got), bad), (what, int((~np.isfinite(got)).sum()), int(bad.sum()))
    if (~bad).any() and np.abs(want[~bad]).max() > 0:
        assert np.abs(got[~bad] - want[~bad]).max()/np.abs(want[~bad]).max() < TOL, what


NAN_AT = {'first sample': 0, 'early': 70001, 'last sample of a tile': 40*2048 - 1, 'late': 555555, 'last sample': 599999}


@pytest.mark.parametrize('where', list(NAN_AT))
@pytest.mark.parametrize('value', [np.nan, np.inf])
def test_non_finite_samples_poison_what_the_reference_poisons(oracle, where, value):
    """scipy's sosfilt keeps a NaN state for ever: from a NaN (or infinite) sample on the filtered trace is NaN to the end
    of the slab, so is every spectrogram frame that reaches that far, and the envelope (sosfiltfilt: the backward pass
    starts from the forward pass's end) is NaN EVERYWHERE in that channel (bufferedfilter.py:36,
    bufferedspectrogram.py:51-58, bufferedenvelope.py:39-41).  A sweep cut into time segments would recover behind the
    next segment border; the segment flags, flood_channel() and the end-state slot (sos_device.h: FloodArgs) keep the
    reference's behaviour -- for one segment, the planner's choice and one-tile segments, through every entry point.
    The other channels must not notice."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rate, T, C, nfft, hop = 96000.0, 600000, 3, 2048, 1024
    F, nd = nfft//2 + 1, (T + hop - 1)//hop
    rng = np.random.default_rng(99)
    x = synth(rng, T, C, rate)
    x[NAN_AT[where], 1] = value
    sos, esos = butter_sos(2, (300.0, 3000.0), 'bandpass', rate), butter_sos(2, 20.0, 'lowpass', rate)
    x64 = x.astype(np.float64)
    want_f = oracle.sosfilt(sos, x64)
    assert not np.isfinite(want_f[NAN_AT[where] + 1:, 1]).any() and np.isfinite(want_f[:, [0, 2]]).all()
    want_e_alone = np.zeros((T, C))
    oracle.envelope_process(esos, x64, want_e_alone, 0)
    assert np.isnan(want_e_alone[:, 1]).all() and np.isfinite(want_e_alone[:, 0]).all()
    c = gh.ctx()
    dx = gh.to_planar(c, x)
    fplan, eplan = hipdsp.SosPlan(c, sos), hipdsp.SosPlan(c, esos)
    settings = [('one segment', {'max_segments': 1}), ('planner', {}),
                ('one-tile segments', {'n_cus': 1024, 'sos_waves_per_cu': 16, 'sos_waves_min': 16})]
    try:
        for name, opts in settings:
            for k, v in opts.items():
                c.set_option(k, v)
            # BufferedFilter alone, with and without frames dropped in front (nbefore)
            for skip in (0, 100000):
                dy = hipdsp.DeviceArray(c, (C, T - skip), np.float32)
                hipdsp.sosfilt(c, fplan, dx, T, dy, T - skip, C, T, skip)
                same_nan_and_close(dy.to_host().T, want_f[skip:], (name, 'sosfilt', skip))
            # BufferedEnvelope alone (the trace itself is rectified)
            de = hipdsp.DeviceArray(c, (C, T), np.float32)
            hipdsp.envelope(c, eplan, dx, T, de, T, C, T, 0)
            same_nan_and_close(de.to_host().T, want_e_alone, (name, 'envelope'))
            # filter + envelope in the two unfused sweeps
            yf, ye = hipdsp.DeviceArray(c, (C, T), np.float32), hipdsp.DeviceArray(c, (C, T), np.float32)
            hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, yf, T, ye, T, C, T)
            gf = yf.to_host()
            same_nan_and_close(gf.T, want_f, (name, 'sosfilt_envelope: filtered'))
            want_e = np.zeros((T, C))
            oracle.envelope_process(esos, gf.T.astype(np.float64), want_e, 0)
            same_nan_and_close(ye.to_host().T, want_e, (name, 'sosfilt_envelope: envelope'))
            # the fused forward sweep (PSD and dB) + backward sweep, with and without an envelope behind the filter
            for ep in (eplan, None):
                yf, ye = hipdsp.DeviceArray(c, (C, T), np.float32), hipdsp.DeviceArray(c, (C, T), np.float32)
                ps, db = (hipdsp.DeviceArray(c, (C, nd, F), np.float32) for _ in range(2))
                hipdsp.chain_forward(c, fplan, ep, dx, T, yf, T, C, T, nfft, hop, rate, ps, nd, db_out=db)
                gf = yf.to_host()
                same_nan_and_close(gf.T, want_f, (name, 'chain_forward: filtered', ep is None))
                want_s = np.zeros((nd, C, F))
                oracle.spectrogram_process(gf.T.astype(np.float64), want_s, rate, nfft, hop)
                got_s, got_db = ps.to_host(), db.to_host()
                first_bad = max(0, (NAN_AT[where] - nfft)//hop + 1)
                assert np.isnan(want_s[first_bad + 1:nd - 4, 1]).all() and np.isfinite(want_s[:, 0]).all()   # (zero tail behind)
                for ch in range(C):
                    for k in range(nd):
                        same_nan_and_close(got_s[ch, k], want_s[k, ch], (name, 'PSD', ch, k))
                    want_db = oracle.decibel(got_s[ch].astype(np.float64))       # the epilogue: decibel() of the PSD next to it
                    assert np.array_equal(np.isnan(got_db[ch]), np.isnan(want_db)), (name, 'dB', ch)
                    assert np.array_equal(np.isneginf(got_db[ch]), np.isneginf(want_db)), (name, 'dB floor and zero tail', ch)
                    ok = np.isfinite(want_db)
                    assert not ok.any() or np.abs(got_db[ch][ok] - want_db[ok]).max() < 1e-3, (name, 'dB', ch)
                if ep is not None:
                    hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, yf, T, ye, T, C, T, phase=2)
                    want_e = np.zeros((T, C))
                    oracle.envelope_process(esos, gf.T.astype(np.float64), want_e, 0)
                    same_nan_and_close(ye.to_host().T, want_e, (name, 'chain: envelope'))
            for k in opts:
                c.set_option(k, {'n_cus': 256}.get(k, 0))
    finally:
        for k, v in (('max_segments', 0), ('n_cus', 256), ('sos_waves_per_cu', 0), ('sos_waves_min', 0)):
            c.set_option(k, v)


def test_non_finite_samples_in_the_split_frames_sweeps(oracle):
    """The same through hipdsp_chain_forward (even frames) + hipdsp_chain_backward (odd frames + envelope)."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rate, T, C, nfft, hop = 96000.0, 600000, 3, 2048, 1024
    F, nd = nfft//2 + 1, (T + hop - 1)//hop
    x = synth(np.random.default_rng(98), T, C, rate)
    x[123456, 2] = np.nan
    sos, esos = butter_sos(2, (300.0, 3000.0), 'bandpass', rate), butter_sos(2, 20.0, 'lowpass', rate)
    want_f = oracle.sosfilt(sos, x.astype(np.float64))
    c = gh.ctx()
    dx = gh.to_planar(c, x)
    fplan, eplan = hipdsp.SosPlan(c, sos), hipdsp.SosPlan(c, esos)
    yf, ye = hipdsp.DeviceArray(c, (C, T), np.float32), hipdsp.DeviceArray(c, (C, T), np.float32)
    ps = hipdsp.DeviceArray(c, (C, nd, F), np.float32)
    c.set_option('chain_split_frames', 1)
    try:
        hipdsp.chain_forward(c, fplan, eplan, dx, T, yf, T, C, T, nfft, hop, rate, ps, nd)
        hipdsp.chain_backward(c, eplan, yf, T, ye, T, C, T, nfft, hop, rate, ps, nd)
    finally:
        c.set_option('chain_split_frames', 0)
    gf = yf.to_host()
    same_nan_and_close(gf.T, want_f, 'filtered')
    want_s, want_e = np.zeros((nd, C, F)), np.zeros((T, C))
    oracle.spectrogram_process(gf.T.astype(np.float64), want_s, rate, nfft, hop)
    oracle.envelope_process(esos, gf.T.astype(np.float64), want_e, 0)
    got_s = ps.to_host()
    for ch in range(C):
        for k in range(nd):
            same_nan_and_close(got_s[ch, k], want_s[k, ch], ('PSD', ch, k))
    same_nan_and_close(ye.to_host().T, want_e, 'envelope')
    assert np.isnan(want_e[:, 2]).all() and np.isfinite(want_e[:, :2]).all()


@pytest.mark.parametrize('nfft,hop', [(2048, 1024), (1024, 256), (256, 128), (4096, 1024), (100, 37)])
def test_non_finite_samples_in_the_spectrogram_alone(oracle, nfft, hop):
    """BufferedSpectrogram on a source with a NaN and an infinite sample: exactly the frames that contain one are NaN,
    in every bin (the frame's mean is NaN, bufferedspectrogram.py:51-56), their dB is NaN, the others do not notice --
    for every spectrogram kernel (fused-frame, two-stage, workgroup, generic)."""
    rate, T, C = 48000.0, 90000, 2
    x = synth(np.random.default_rng(nfft + hop), T, C, rate)
    x[33333, 0] = np.nan
    x[70001, 1] = -np.inf
    nd = (T + hop - 1)//hop
    want = np.zeros((nd, C, nfft//2 + 1))
    oracle.spectrogram_process(x.astype(np.float64), want, rate, nfft, hop)
    got, got_db = gh.gpu_spectrogram(x, rate, nfft, hop, nd, want_db=True)
    assert np.isnan(want).any() and np.isfinite(want).any()
    for ch in range(C):
        for k in range(nd):
            same_nan_and_close(got[k, ch], want[k, ch], (ch, k))
    want_db = oracle.decibel(got)
    assert np.array_equal(np.isnan(got_db), np.isnan(want_db))
    fin = np.isfinite(want_db)
    assert np.array_equal(np.isfinite(got_db), fin) and np.abs(got_db[fin] - want_db[fin]).max() < 1e-3


def test_repeated_runs_are_bit_identical():
    """No atomics, no data-dependent scheduling in the data path: the same call on the same input gives the same bits,
    whatever ran in between and whichever of two contexts (streams) runs it (tools/soak.py: 13 000 steps)."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rate, T, C, nfft, hop = 96000.0, 700000, 5, 2048, 1024
    F, nd = nfft//2 + 1, (T + hop - 1)//hop
    x = synth(np.random.default_rng(31), T, C, rate)
    sos, esos = butter_sos(2, (300.0, 3000.0), 'bandpass', rate), butter_sos(2, 20.0, 'lowpass', rate)
    runs = []
    for c in (gh.ctx(), hipdsp.Context(0), gh.ctx()):
        dx = gh.to_planar(c, x)
        fplan, eplan = hipdsp.SosPlan(c, sos), hipdsp.SosPlan(c, esos)
        yf, ye = hipdsp.DeviceArray(c, (C, T), np.float32), hipdsp.DeviceArray(c, (C, T), np.float32)
        ps, db = (hipdsp.DeviceArray(c, (C, nd, F), np.float32) for _ in range(2))
        for _ in range(2):
            hipdsp.chain_forward(c, fplan, eplan, dx, T, yf, T, C, T, nfft, hop, rate, ps, nd, db_out=db)
            hipdsp.sosfilt_envelope(c, fplan, eplan, dx, T, yf, T, ye, T, C, T, phase=2)
            runs.append([a.to_host() for a in (yf, ye, ps, db)])
            hipdsp.envelope(c, eplan, dx, T, ye, T, C, T, 0)          # something else in between
    for r in runs[1:]:
        for a, b in zip(r, runs[0]):
            assert np.array_equal(a, b)


@pytest.mark.parametrize('pos', [5, 250000, 399990])
def test_non_finite_samples_in_an_envelope_longer_than_one_plan(oracle, pos):
    """hipdsp_envelope_multi (cascades of more than four sections, chained plans): a NaN anywhere makes that channel's
    envelope NaN everywhere, the others do not notice -- with the sweeps cut into one-tile segments."""
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    rate, T, C = 48000.0, 400000, 3
    x = synth(np.random.default_rng(pos), T, C, rate)
    x[pos, 0] = np.nan
    sos = butter_sos(10, 800.0, 'lowpass', rate)                       # five sections: plans of 2 + 2 + 1
    want = np.zeros((T, C))
    oracle.envelope_process(sos, x.astype(np.float64), want, 0)
    assert np.isnan(want[:, 0]).all() and np.isfinite(want[:, 1:]).all()
    c = gh.ctx()
    dx = gh.to_planar(c, x)
    plans = [hipdsp.SosPlan(c, sos[0:2]), hipdsp.SosPlan(c, sos[2:4]), hipdsp.SosPlan(c, sos[4:5])]
    try:
        for opts in ({}, {'n_cus': 1024, 'sos_waves_per_cu': 16, 'sos_waves_min': 16}):
            for k, v in opts.items():
                c.set_option(k, v)
            dy = hipdsp.DeviceArray(c, (C, T), np.float32)
            hipdsp.envelope_multi(c, plans, dx, T, dy, T, C, T, 0)
            got = dy.to_host().T
            assert np.isnan(got[:, 0]).all(), opts
            for ch in (1, 2):
                assert rel_err(got[:, ch], want[:, ch]) < 2e-4, (opts, ch)     # (float32 hand-over between the plans)
    finally:
        for k, v in (('n_cus', 256), ('sos_waves_per_cu', 0), ('sos_waves_min', 0)):
            c.set_option(k, v)


def test_non_finite_samples_through_unwrap(oracle):
    """hipdsp_unwrap with clipping: np.clip leaves NaN alone (fminf(fmaxf()) would not); an infinite sample clips."""
    from audian_amd import hipdsp
    rng = np.random.default_rng(77)
    T, C = 30000, 2
    x = np.cumsum(rng.uniform(-0.2, 0.2, (T, C)), axis=0).astype(np.float32)
    x = ((x + 1.0) % 2.0 - 1.0).astype(np.float32)                    # wrapped into [-1, 1)
    x[1234, 0] = np.nan
    x[20000, 1] = np.inf
    c = gh.ctx()
    for clips in (False, True):
        with np.errstate(invalid='ignore'):
            want = oracle.unwrap(x, 1.5, 1.0, clips=clips, down_scale=True)
        dx = gh.to_planar(c, x)
        dy = hipdsp.DeviceArray(c, (C, T), np.float32)
        hipdsp.unwrap(c, dx, T, C, T, 1.5, dy, T, clips=clips, down_scale=True)
        got = dy.to_host().T
        assert np.array_equal(np.isnan(got), np.isnan(want)), clips
        ok = ~np.isnan(want)
        assert np.array_equal(got[ok], want[ok]), clips


def test_pass_through_copy_any_skip_pitch_and_tail():
    """BufferedFilter with sos None (the reference's DEFAULT session: cut-offs at 0 and Nyquist, bufferedfilter.py:32-33):
    dest = source[nbefore:], bit for bit, for every alignment of skip, pitches and length (the copy moves four floats
    per access with a scalar tail), nothing written past the row."""
    from audian_amd import hipdsp
    rng = np.random.default_rng(4)
    c = gh.ctx()
    for T, skip, xp_extra, yp_extra, C in [(1, 0, 0, 0, 1), (5, 2, 1, 3, 2), (4096, 1, 0, 0, 3), (4099, 3, 5, 7, 3), (100003, 50001, 2, 1, 4),
                                           (8, 8, 0, 0, 2), (1 << 20, 7, 0, 1, 2)]:
        xp, n = T + xp_extra, T - skip
        yp = max(n, 1) + yp_extra
        host = rng.standard_normal((C, xp)).astype(np.float32)
        dx = hipdsp.DeviceArray(c, (C, xp), np.float32)
        hipdsp.lib.hipdsp_memcpy_h2d(c.handle, hipdsp._p(dx), host.ctypes.data, host.nbytes)
        dy = hipdsp.DeviceArray(c, (C, yp), np.float32)
        hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(dy), 0x7f, 4*C*yp)
        hipdsp.sosfilt(c, None, dx, xp, dy, yp, C, T, skip)
        got = dy.to_host()
        assert np.array_equal(got[:, :n], host[:, skip:T]), (T, skip)
        assert np.all(got[:, n:].view(np.uint32) == 0x7f7f7f7f), (T, skip, 'wrote past the row')


def test_decibel_any_length_and_alignment(oracle):
    """hipdsp_decibel moves four values per access with a scalar tail: every length modulo 4, unaligned views, NaN and
    the 1e-20 floor, nothing written past the end."""
    from audian_amd import hipdsp
    rng = np.random.default_rng(8)
    c = gh.ctx()
    for n, off in [(1, 0), (3, 1), (4, 0), (5, 3), (1023, 2), (4096, 1), (100001, 3)]:
        p = (10.0**rng.uniform(-25, 3, size=n + off + 8)).astype(np.float32)
        p[rng.integers(0, len(p), 3)] = 0.0
        p[rng.integers(0, len(p))] = np.nan
        dp = hipdsp.DeviceArray.from_host(c, p)
        out = hipdsp.DeviceArray(c, (n + off + 8,), np.float32)
        hipdsp.lib.hipdsp_memset(c.handle, hipdsp._p(out), 0x7f, 4*(n + off + 8))
        hipdsp.decibel(c, dp.view(off, (n,)), out.view(off, (n,)), n)
        got = out.to_host()
        want = oracle.decibel(p[off:off + n].astype(np.float64))
        assert np.array_equal(np.isnan(got[off:off + n]), np.isnan(want)), (n, off)
        assert np.array_equal(np.isneginf(got[off:off + n]), np.isneginf(want)), (n, off)
        fin = np.isfinite(want)
        assert not fin.any() or np.max(np.abs(got[off:off + n][fin] - want[fin])) < 1e-3, (n, off)
        rest = np.concatenate([got[:off], got[off + n:]])
        assert np.all(rest.view(np.uint32) == 0x7f7f7f7f), (n, off)
