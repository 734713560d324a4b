"""BASELINE configs[2] at FULL size on the GPU (64 ch x 600 s x 96 kHz, 3.7 G samples,
offsets far beyond 2^31 bytes): windows deep inside the run against the oracle, plus
size-independent properties (Parseval, clamp, zero tail, linearity).

Every test on the `chain` fixture runs twice: on the three separate calls (sosfilt, spectrogram,
envelope) and on the path bench.py times (hipdsp_chain_forward + hipdsp_sosfilt_envelope phase 2:
64 channels x 32 segments, 8 units per workgroup); the fused path is additionally checked on both
sides of EVERY internal segment border of its plan (tests/test_gpu_fullsize.py::test_fused_segment_borders)."""

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

RATE, C, SECONDS, NFFT, HOP = 96000.0, 64, 600.0, 2048, 1024
T = int(RATE*SECONDS)
F = NFFT//2 + 1
ND = (T + HOP - 1)//HOP


@pytest.fixture(scope='module')
def base():
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos
    ctx = hipdsp.Context(0)
    sos = butter_sos(2, (300.0, 3000.0), 'bandpass', RATE)
    esos = butter_sos(2, 20.0, 'lowpass', RATE)
    dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
    hipdsp.synth(ctx, dx, T, C, T, RATE, 1236)
    plan, eplan = hipdsp.SosPlan(ctx, sos), hipdsp.SosPlan(ctx, esos)
    ctx.synchronize()
    yield dict(ctx=ctx, sos=sos, esos=esos, dx=dx, plan=plan, eplan=eplan)
    dx.free()


@pytest.fixture(scope='module', params=['separate', 'fused', 'fused-split'])
def chain(request, base):
    from audian_amd import hipdsp
    ctx, dx, plan, eplan = base['ctx'], base['dx'], base['plan'], base['eplan']
    df = hipdsp.DeviceArray(ctx, (C, T), np.float32)
    de = hipdsp.DeviceArray(ctx, (C, T), np.float32)
    ds = hipdsp.DeviceArray(ctx, (C, ND, F), np.float32)
    if request.param == 'separate':
        hipdsp.sosfilt(ctx, plan, dx, T, df, T, C, T, 0)
        hipdsp.spectrogram(ctx, df, T, C, T, NFFT, HOP, RATE, ds, ND)
        hipdsp.envelope(ctx, eplan, df, T, de, T, C, T, 0)
    elif request.param == 'fused':
        # exactly bench.py's step: fused forward sweep, then the envelope's backward sweep
        hipdsp.chain_forward(ctx, plan, eplan, dx, T, df, T, C, T, NFFT, HOP, RATE, ds, ND,
                             rectify=True, gain=np.pi/2)
        hipdsp.sosfilt_envelope(ctx, plan, eplan, dx, T, df, T, de, T, C, T, rectify=True,
                                gain=np.pi/2, clamp=True, phase=2)
    else:
        # the frame split: even frames from the forward, odd frames + envelope from the backward sweep
        hipdsp.lib.hipdsp_memset(ctx.handle, hipdsp._p(ds), 0x7f, 4*C*ND*F)      # every frame must be written
        ctx.set_option('chain_split_frames', 1)
        try:
            hipdsp.chain_forward(ctx, plan, eplan, dx, T, df, T, C, T, NFFT, HOP, RATE, ds, ND,
                                 rectify=True, gain=np.pi/2)
            hipdsp.chain_backward(ctx, eplan, df, T, de, T, C, T, NFFT, HOP, RATE, ds, ND,
                                  rectify=True, gain=np.pi/2, clamp=True)
        finally:
            ctx.set_option('chain_split_frames', 0)
    ctx.synchronize()
    yield dict(base, df=df, de=de, ds=ds, path=request.param)
    for a in (df, de, ds):
        a.free()


def window(arr, ch, off, n):
    return arr.view(ch*T + off, (n,)).to_host().astype(np.float64)


def test_filter_deep_windows_match_oracle(chain, oracle):
    """The oracle restarted W samples earlier from zero state reproduces the trace deep
    inside the recording (the band-pass forgets its past within ~2.3 k samples)."""
    rng = np.random.default_rng(0)
    lead, n = 20000, 8192
    for _ in range(12):
        ch = int(rng.integers(0, C))
        off = int(rng.integers(lead, T - n))
        x = window(chain['dx'], ch, off - lead, lead + n)
        want = oracle.sosfilt(chain['sos'], x)[lead:]
        got = window(chain['df'], ch, off, n)
        assert rel_err(got, want) < 1e-4, (ch, off)
    # both ends of the recording
    for ch in (0, C - 1):
        x = window(chain['dx'], ch, 0, n)
        assert rel_err(window(chain['df'], ch, 0, n), oracle.sosfilt(chain['sos'], x)) < 1e-4
        x = window(chain['dx'], ch, T - lead - n, lead + n)
        assert rel_err(window(chain['df'], ch, T - n, n), oracle.sosfilt(chain['sos'], x)[lead:]) < 1e-4


def test_envelope_deep_windows_match_oracle(chain, oracle):
    """20 Hz zero-phase envelope: needs ~0.6 s of context on both sides."""
    rng = np.random.default_rng(1)
    lead, n = 80000, 4096
    for _ in range(6):
        ch = int(rng.integers(0, C))
        off = int(rng.integers(lead, T - n - lead))
        f = window(chain['df'], ch, off - lead, n + 2*lead)[:, None]
        want = np.zeros_like(f)
        oracle.envelope_process(chain['esos'], f, want, 0)
        got = window(chain['de'], ch, off, n)
        assert rel_err(got, want[lead:lead + n, 0]) < 1e-4, (ch, off)
        assert np.all(got >= 0)
    for ch in (3, C - 2):                                   # true start and end (odd padding, zi)
        f = window(chain['df'], ch, 0, n + lead)[:, None]
        want = np.zeros_like(f)
        oracle.envelope_process(chain['esos'], f, want, 0)
        assert rel_err(window(chain['de'], ch, 0, n), want[:n, 0]) < 1e-4
        f = window(chain['df'], ch, T - n - lead, n + lead)[:, None]
        want = np.zeros_like(f)
        oracle.envelope_process(chain['esos'], f, want, 0)
        assert rel_err(window(chain['de'], ch, T - n, n), want[-n:, 0]) < 1e-4


def test_spectrogram_deep_frames_parseval_and_tail(chain, oracle):
    rng = np.random.default_rng(2)
    ds = chain['ds']
    win = 0.5 - 0.5*np.cos(2*np.pi*np.arange(NFFT)/NFFT)
    for _ in range(24):
        ch = int(rng.integers(0, C))
        k = int(rng.integers(0, ND - 2))
        seg = window(chain['df'], ch, k*HOP, NFFT)
        row = ds.view((ch*ND + k)*F, (F,)).to_host().astype(np.float64)
        want = np.zeros((1, 1, F))
        oracle.spectrogram_process(seg[:, None], want, RATE, NFFT, HOP)
        assert rel_err(row, want[0, 0]) < 1e-4, (ch, k)
        # Parseval for the one-sided density: sum(P) * fs/nfft == sum(((x - mean) w)^2) / sum(w^2)
        lhs = np.sum(row)*RATE/NFFT
        rhs = np.sum(((seg - seg.mean())*win)**2)/np.sum(win**2)
        assert abs(lhs - rhs) <= 1e-4*rhs
    # the nafter = 1 quirk of the facade gives len(source) = nd*hop + 1; here the source is
    # exactly T samples, so the last frame does not fit and must be zero
    n_valid = (T - (NFFT - HOP))//HOP
    assert n_valid == ND - 1
    for ch in (0, C - 1):
        assert np.all(ds.view((ch*ND + ND - 1)*F, (F,)).to_host() == 0)
        assert np.any(ds.view((ch*ND + ND - 2)*F, (F,)).to_host() != 0)


def test_filter_linearity_full_size(chain):
    """sosfilt(a*x + b*y) == a*sosfilt(x) + b*sosfilt(y) on the full batch, checked on
    random windows (y = a second synthetic batch)."""
    from audian_amd import hipdsp
    ctx = chain['ctx']
    dy = hipdsp.DeviceArray(ctx, (C, T), np.float32)
    fy = hipdsp.DeviceArray(ctx, (C, T), np.float32)
    hipdsp.synth(ctx, dy, T, C, T, RATE, 777)
    hipdsp.sosfilt(ctx, chain['plan'], dy, T, fy, T, C, T, 0)
    # z = x + y built by filtering the concatenated identity: use the envelope scratch-free
    # trick  sosfilt is linear, so filter y in place of x and compare windows of
    # f(x) + f(y) with the oracle applied to (x + y)
    rng = np.random.default_rng(3)
    lead, n = 20000, 4096
    from oracle import oracle
    for _ in range(6):
        ch = int(rng.integers(0, C))
        off = int(rng.integers(lead, T - n))
        xs = window(chain['dx'], ch, off - lead, lead + n) + window(dy, ch, off - lead, lead + n)
        want = oracle.sosfilt(chain['sos'], xs)[lead:]
        got = window(chain['df'], ch, off, n) + window(fy, ch, off, n)
        assert rel_err(got, want) < 1e-4
    dy.free()
    fy.free()


def test_fused_segment_borders(chain, oracle):
    """The fused forward sweep cuts every channel into segments (hipdsp_chain_plan: 32 at configs[2]);
    the unit that owns tile t writes frames 2t-1 and 2t, and frame 2t-1 of a segment's first tile takes
    its first half from that unit's own warm-up.  Check, at EVERY internal border b of the plan, on a
    different channel each time: the filtered trace and the envelope across b, and the three frames
    that end at, straddle and start at b."""
    if chain['path'] == 'separate':
        pytest.skip('borders of the fused plan')
    from audian_amd import hipdsp
    seg, nseg = hipdsp.chain_plan(chain['ctx'], chain['plan'], chain['eplan'], C, T)
    assert nseg >= 2 and seg % NFFT == 0 and (nseg - 1)*seg < T <= nseg*seg
    lead_f, lead_e, half = 20000, 80000, 4096
    for s_ in range(1, nseg):
        b = s_*seg
        ch = (7*s_ + 3) % C
        x = window(chain['dx'], ch, b - half - lead_f, lead_f + 2*half)
        want = oracle.sosfilt(chain['sos'], x)[lead_f:]
        assert rel_err(window(chain['df'], ch, b - half, 2*half), want) < 1e-4, (s_, ch)
        f = window(chain['df'], ch, b - half - lead_e, 2*half + 2*lead_e)[:, None]
        wenv = np.zeros_like(f)
        oracle.envelope_process(chain['esos'], f, wenv, 0)
        assert rel_err(window(chain['de'], ch, b - half, 2*half), wenv[lead_e:lead_e + 2*half, 0]) < 1e-4, (s_, ch)
        k0 = b//HOP
        for k in (k0 - 2, k0 - 1, k0):            # ends at b, straddles b, starts at b
            segx = window(chain['df'], ch, k*HOP, NFFT)
            row = chain['ds'].view((ch*ND + k)*F, (F,)).to_host().astype(np.float64)
            wpsd = np.zeros((1, 1, F))
            oracle.spectrogram_process(segx[:, None], wpsd, RATE, NFFT, HOP)
            assert rel_err(row, wpsd[0, 0]) < 1e-4, (s_, ch, k)
    if chain['path'] == 'fused-split':
        # the backward sweep of the frame split has borders of its own (counted from the end of the trace): the
        # envelope across each, and the odd frame whose second half comes from the unit's own warm-up tile
        fb, bseg, bn = hipdsp.chain_backward_plan(chain['ctx'], chain['eplan'], C, T)
        assert bn >= 2 and bseg % NFFT == 0
        for s_ in range(bn - 1):
            b = fb - s_*bseg
            if b + half + lead_e > T or b - half - lead_e < 0:
                continue
            ch = (11*s_ + 5) % C
            f = window(chain['df'], ch, b - half - lead_e, 2*half + 2*lead_e)[:, None]
            wenv = np.zeros_like(f)
            oracle.envelope_process(chain['esos'], f, wenv, 0)
            assert rel_err(window(chain['de'], ch, b - half, 2*half), wenv[lead_e:lead_e + 2*half, 0]) < 1e-4, (s_, ch)
            k0 = b//HOP
            for k in (k0 - 3, k0 - 2, k0 - 1, k0, k0 + 1):
                segx = window(chain['df'], ch, k*HOP, NFFT)
                row = chain['ds'].view((ch*ND + k)*F, (F,)).to_host().astype(np.float64)
                wpsd = np.zeros((1, 1, F))
                oracle.spectrogram_process(segx[:, None], wpsd, RATE, NFFT, HOP)
                assert rel_err(row, wpsd[0, 0]) < 1e-4, (s_, ch, k)
    # the last unit of the last channel (highest addresses of every array)
    ch = C - 1
    b = (nseg - 1)*seg
    x = window(chain['dx'], ch, b - lead_f, lead_f + (T - b))
    want = oracle.sosfilt(chain['sos'], x)[lead_f:]
    assert rel_err(window(chain['df'], ch, b, T - b)[-65536:], want[-65536:]) < 1e-4


def test_fused_fault_is_reported_not_swallowed(base):
    """A wave of the fused kernel that waits in vain for its partner must not end with HIPDSP_OK
    ("chain_debug" bit 8 makes one FFT wave withhold one hand-over): the next synchronisation
    raises, exactly once, and the context works again afterwards."""
    from audian_amd import hipdsp
    from audian_amd._lib import HipDspError
    ctx = base['ctx']
    Cs, Ts = 2, 64*2048
    nd = (Ts + HOP - 1)//HOP
    df = hipdsp.DeviceArray(ctx, (Cs, Ts), np.float32)
    ds = hipdsp.DeviceArray(ctx, (Cs, nd, F), np.float32)
    ctx.set_option('chain_debug', 8)
    try:
        hipdsp.chain_forward(ctx, base['plan'], base['eplan'], base['dx'], T, df, Ts, Cs, Ts, NFFT, HOP, RATE, ds, nd)
        with pytest.raises(HipDspError, match='gave up waiting'):
            ctx.synchronize()
    finally:
        ctx.set_option('chain_debug', 0)
    ctx.synchronize()                                  # reported once
    hipdsp.chain_forward(ctx, base['plan'], base['eplan'], base['dx'], T, df, Ts, Cs, Ts, NFFT, HOP, RATE, ds, nd)
    ctx.synchronize()
    # and the fault of a launch nobody synchronised on stops the next call of the chain
    ctx.set_option('chain_debug', 8)
    try:
        hipdsp.chain_forward(ctx, base['plan'], base['eplan'], base['dx'], T, df, Ts, Cs, Ts, NFFT, HOP, RATE, ds, nd)
        import time
        time.sleep(2.0)                                # the kernel's timeout is about a third of a second
        ctx.set_option('chain_debug', 0)
        with pytest.raises(HipDspError, match='gave up waiting'):
            hipdsp.sosfilt_envelope(ctx, base['plan'], base['eplan'], base['dx'], T, df, Ts, df, Ts, Cs, Ts, phase=2)
    finally:
        ctx.set_option('chain_debug', 0)
    ctx.synchronize()
    df.free()
    ds.free()
