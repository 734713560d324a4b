"""One line per compute entry point of the C ABI at BASELINE configs[2]'s shape (64 ch x 600 s x 96 kHz): time and
algorithmic GB/s -- so that a change to a shared piece (the segment planner, the cascade include, the block cache)
shows up wherever it lands, not only in the chain bench.py times.  (tools/next_rows_bench.py has the SURVEY 8f rows.)
    python tools/entry_points_bench.py [seconds=600]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos

ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()
C, rate = 64, 96000.0
T = int((float(sys.argv[1]) if len(sys.argv) > 1 else 600.0)*rate)
dx, df, de = (hipdsp.DeviceArray(ctx, (C, T), np.float32) for _ in range(3))
hipdsp.synth(ctx, dx, T, C, T, rate, 1236)
bp2 = hipdsp.SosPlan(ctx, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
bp4 = hipdsp.SosPlan(ctx, butter_sos(4, (300.0, 3000.0), 'bandpass', rate))
lp1 = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate))
lp2 = hipdsp.SosPlan(ctx, butter_sos(4, 20.0, 'lowpass', rate))
lp6 = [hipdsp.SosPlan(ctx, butter_sos(12, 500.0, 'lowpass', rate)[i:i + 3]) for i in (0, 3)]


def timed(f, n=5):
    f(); f()
    ctx.record(e0)
    for _ in range(n):
        f()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1)/n


def line(name, ms, nbytes):
    print(f'{name:78s} {ms:8.3f} ms {nbytes/ms/1e6:7.0f} GB/s', flush=True)


S = C*T
line('hipdsp_sosfilt, band-pass of 2 sections (BufferedFilter alone)', timed(lambda: hipdsp.sosfilt(ctx, bp2, dx, T, df, T, C, T, 0)), 8.0*S)
line('hipdsp_sosfilt, band-pass of 4 sections', timed(lambda: hipdsp.sosfilt(ctx, bp4, dx, T, df, T, C, T, 0)), 8.0*S)
line('hipdsp_sosfilt, no filter (copy)', timed(lambda: hipdsp.sosfilt(ctx, None, dx, T, df, T, C, T, 0)), 8.0*S)
hipdsp.sosfilt(ctx, bp2, dx, T, df, T, C, T, 0)
line('hipdsp_envelope, low-pass of 1 section (BufferedEnvelope alone)', timed(lambda: hipdsp.envelope(ctx, lp1, df, T, de, T, C, T, 0)), 12.0*S)
line('hipdsp_envelope, low-pass of 2 sections', timed(lambda: hipdsp.envelope(ctx, lp2, df, T, de, T, C, T, 0)), 12.0*S)
line('hipdsp_envelope_multi, low-pass of 6 sections as 3 + 3 (default options; temporaries in the context scratch)', timed(lambda: hipdsp.envelope_multi(ctx, lp6, df, T, de, T, C, T, 0), n=3), 12.0*S)
line('hipdsp_sosfilt_envelope, both sweeps (filter + envelope, unfused spectrogram)', timed(lambda: hipdsp.sosfilt_envelope(ctx, bp2, lp1, dx, T, df, T, de, T, C, T)), 16.0*S)
for nfft, hop in ((2048, 1024), (1024, 256), (256, 128), (8192, 4096), (65536, 32768)):
    F, nd = nfft//2 + 1, (T + hop - 1)//hop
    ds = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32)
    line(f'hipdsp_spectrogram {nfft}/{hop} (BufferedSpectrogram alone)', timed(lambda: hipdsp.spectrogram(ctx, df, T, C, T, nfft, hop, rate, ds, nd)), 4.0*S + 4.0*C*nd*F)
    if (nfft, hop) in ((2048, 1024), (1024, 256), (256, 128)):
        db = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32)
        line(f'hipdsp_spectrogram {nfft}/{hop} with the dB image', timed(lambda: hipdsp.spectrogram(ctx, df, T, C, T, nfft, hop, rate, ds, nd, db_out=db)), 4.0*S + 8.0*C*nd*F)
        line(f'hipdsp_chain_forward {nfft}/{hop}, 2 + 1 sections', timed(lambda: hipdsp.chain_forward(ctx, bp2, lp1, dx, T, df, T, C, T, nfft, hop, rate, ds, nd)), 8.0*S + 4.0*C*nd*F)
        line(f'hipdsp_chain_forward {nfft}/{hop}, 2 + 1 sections, with the dB image', timed(lambda: hipdsp.chain_forward(ctx, bp2, lp1, dx, T, df, T, C, T, nfft, hop, rate, ds, nd, db_out=db)), 8.0*S + 8.0*C*nd*F)
        line(f'hipdsp_chain_forward {nfft}/{hop}, 2 sections, no envelope', timed(lambda: hipdsp.chain_forward(ctx, bp2, None, dx, T, df, T, C, T, nfft, hop, rate, ds, nd)), 8.0*S + 4.0*C*nd*F)
        line(f'hipdsp_chain_forward {nfft}/{hop}, 4 + 2 sections', timed(lambda: hipdsp.chain_forward(ctx, bp4, lp2, dx, T, df, T, C, T, nfft, hop, rate, ds, nd)), 8.0*S + 4.0*C*nd*F)
        del db
    if (nfft, hop) == (2048, 1024):
        hipdsp.chain_forward(ctx, bp2, lp1, dx, T, df, T, C, T, nfft, hop, rate, ds, nd)
        line('hipdsp_sosfilt_envelope phase 2 (backward sweep behind the fused forward sweep)', timed(lambda: hipdsp.sosfilt_envelope(ctx, bp2, lp1, dx, T, df, T, de, T, C, T, phase=2)), 8.0*S)
        out = hipdsp.DeviceArray(ctx, (C*nd*F,), np.float32)
        line('hipdsp_decibel over the PSD', timed(lambda: hipdsp.decibel(ctx, ds, out, C*nd*F)), 8.0*C*nd*F)
        del out
    del ds

# ---- the reference's DEFAULT session through the plug-in surface: no filter set (bufferedfilter.py:40-42) + spectrogram
# 256 / 128 (plugins.py:11-13, bufferedspectrogram.py:14-16), whole recording resident: BufferedFilter.update() ->
# recompute_all().  The filtered trace's mirror is a view of the raw slab's device copy (no launch), the spectrogram is
# the only launch.
if os.environ.get('FACADE', '1') == '1':
    import time
    from audian_amd.bufferedfilter import BufferedFilter
    from audian_amd.bufferedspectrogram import BufferedSpectrogram
    from audian_amd.tracegraph import TraceGraph
    del df, de
    host = np.empty((T, C), dtype=np.float32)
    chunk = 1 << 20
    tmp = hipdsp.DeviceArray(ctx, (chunk, C), np.float64)
    for a in range(0, T, chunk):
        n = min(chunk, T - a)
        hipdsp.unpack(ctx, dx.view(a, (1,)), T, tmp, n, C)
        host[a:a + n] = tmp.to_host().reshape(-1)[:n*C].reshape(n, C)
    tmp.free()
    del dx
    hipdsp._default_ctx = ctx

    class Shown:
        def isVisible(self):
            return True

    g = TraceGraph(T/rate, 0.0)
    filt, spec = BufferedFilter(), BufferedSpectrogram()          # the defaults: 256, 50 % overlap
    g.add_trace(filt)
    g.add_trace(spec)
    g.setup_traces()
    g.open(host, rate, view=True)
    for t in g.traces:
        t.plot_items = [Shown() for _ in range(t.channels)]
    g.set_need_update()
    g.update_times(0.0, T/rate)
    filt.update()
    ctx.synchronize()
    before = dict(hipdsp.launches)
    t0 = time.perf_counter()
    for _ in range(5):
        filt.update()
    ctx.synchronize()
    ms = (time.perf_counter() - t0)/5*1e3
    per = {k: (v - before.get(k, 0))//5 for k, v in hipdsp.launches.items() if v != before.get(k, 0)}
    nd = len(spec._hostbuf)
    line(f'default session (no filter + 256/128) through BufferedFilter.update(), launches {per}', ms, 4.0*S + 4.0*C*nd*129)
