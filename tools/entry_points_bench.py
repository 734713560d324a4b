"""One line per compute entry point of the C ABI at BASELINE configs[2]'s shape (64 ch x 600 s x 96 kHz): time and
algorithmic GB/s -- so that a change to a shared piece (the segment planner, the cascade include, the block cache)
shows up wherever it lands, not only in the chain bench.py times.  (tools/next_rows_bench.py has the SURVEY 8f rows.)
    python tools/entry_points_bench.py [seconds=600]
    LIBS=tools/_ab/libr04.so,tree OUT_PREFIX=gpurun_out/r05_entry_points python tools/entry_points_bench.py
LIBS: several BUILDS of the library in this ONE process on the SAME device buffers, every entry point timed for each of
them in turn (ROUNDS times, the fastest counts), one log per build (<OUT_PREFIX>_<name>.log; "tree" = the tree's build,
also printed) -- what tools/entry_points_gate.py compares.  Separate processes place their buffers differently in HBM,
and the envelope's backward sweep and the PSD kernels with the dB image move by 5-12 % with that alone (identical machine
code: profiles/r05x_*, r05y_*); a box moves by 5 % against the next one."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


sys.path.insert(0, os.path.join(ROOT, 'tools'))
from _builds import load_build, build_name

libs = [s for s in os.environ.get('LIBS', 'tree').split(',') if s]
rounds = int(os.environ.get('ROUNDS', '2' if len(libs) > 1 else '1'))
C, rate = 64, 96000.0
T = int((float(sys.argv[1]) if len(sys.argv) > 1 else 600.0)*rate)
S = C*T
WINDOWS = ((2048, 1024), (1024, 256), (256, 128), (8192, 4096), (65536, 32768))
FUSED = ((2048, 1024), (1024, 256), (256, 128))

builds = []
for i, lib in enumerate(libs):
    name = build_name(lib)
    h, d = load_build(lib)
    ctx = h.Context(0)
    builds.append({'name': name, 'h': h, 'design': d, 'ctx': ctx, 'e0': ctx.event(), 'e1': ctx.event()})

# the buffers: allocated once (by the first build's context), every build sees the same pointers
b0 = builds[0]
shapes = {'dx': (C, T), 'df': (C, T), 'de': (C, T),
          'ds': (max(C*((T + hp - 1)//hp)*(n//2 + 1) for n, hp in WINDOWS),), 'db': (max(C*((T + hp - 1)//hp)*(n//2 + 1) for n, hp in FUSED),)}
owned = {k: b0['h'].DeviceArray(b0['ctx'], shp, np.float32) for k, shp in shapes.items()}
b0['h'].synth(b0['ctx'], owned['dx'], T, C, T, rate, 1236)
b0['ctx'].synchronize()
for b in builds:
    h, ctx, bs = b['h'], b['ctx'], b['design'].butter_sos
    b['buf'] = {k: (a if b is b0 else h.DeviceArray(ctx, a.shape, np.float32, ptr=a.ptr, owner=a)) for k, a in owned.items()}
    b['bp2'] = h.SosPlan(ctx, bs(2, (300.0, 3000.0), 'bandpass', rate))
    b['bp4'] = h.SosPlan(ctx, bs(4, (300.0, 3000.0), 'bandpass', rate))
    b['lp1'] = h.SosPlan(ctx, bs(2, 20.0, 'lowpass', rate))
    b['lp2'] = h.SosPlan(ctx, bs(4, 20.0, 'lowpass', rate))
    b['lp6'] = [h.SosPlan(ctx, bs(12, 500.0, 'lowpass', rate)[i:i + 3]) for i in (0, 3)]


def entries(b):
    """(name, callable, algorithmic bytes, timed calls, preparation) per entry point, for one build"""
    h, ctx = b['h'], b['ctx']
    dx, df, de, ds, db = (b['buf'][k] for k in ('dx', 'df', 'de', 'ds', 'db'))
    bp2, bp4, lp1, lp2, lp6 = b['bp2'], b['bp4'], b['lp1'], b['lp2'], b['lp6']
    filt = lambda: h.sosfilt(ctx, bp2, dx, T, df, T, C, T, 0)              # (what the envelope and the PSD lines read)
    out = [
        ('hipdsp_sosfilt, band-pass of 2 sections (BufferedFilter alone)', filt, 8.0*S, 5, None),
        ('hipdsp_sosfilt, band-pass of 4 sections', lambda: h.sosfilt(ctx, bp4, dx, T, df, T, C, T, 0), 8.0*S, 5, None),
        ('hipdsp_sosfilt, no filter (copy)', lambda: h.sosfilt(ctx, None, dx, T, df, T, C, T, 0), 8.0*S, 5, None),
        ('hipdsp_envelope, low-pass of 1 section (BufferedEnvelope alone)', lambda: h.envelope(ctx, lp1, df, T, de, T, C, T, 0), 12.0*S, 5, filt),
        ('hipdsp_envelope, low-pass of 2 sections', lambda: h.envelope(ctx, lp2, df, T, de, T, C, T, 0), 12.0*S, 5, filt),
        ('hipdsp_envelope_multi, low-pass of 6 sections as 3 + 3 (default options; temporaries in the context scratch)',
         lambda: h.envelope_multi(ctx, lp6, df, T, de, T, C, T, 0), 12.0*S, 3, filt),
        ('hipdsp_sosfilt_envelope, both sweeps (filter + envelope, unfused spectrogram)',
         lambda: h.sosfilt_envelope(ctx, bp2, lp1, dx, T, df, T, de, T, C, T), 16.0*S, 5, None),
    ]
    for nfft, hop in WINDOWS:
        F, nd = nfft//2 + 1, (T + hop - 1)//hop
        out.append((f'hipdsp_spectrogram {nfft}/{hop} (BufferedSpectrogram alone)',
                    lambda nfft=nfft, hop=hop, nd=nd: h.spectrogram(ctx, df, T, C, T, nfft, hop, rate, ds, nd), 4.0*S + 4.0*C*nd*F, 5, filt))
        if (nfft, hop) in FUSED:
            fwd = lambda nfft=nfft, hop=hop, nd=nd: h.chain_forward(ctx, bp2, lp1, dx, T, df, T, C, T, nfft, hop, rate, ds, nd)
            out += [
                (f'hipdsp_spectrogram {nfft}/{hop} with the dB image',
                 lambda nfft=nfft, hop=hop, nd=nd: h.spectrogram(ctx, df, T, C, T, nfft, hop, rate, ds, nd, db_out=db), 4.0*S + 8.0*C*nd*F, 5, filt),
                (f'hipdsp_chain_forward {nfft}/{hop}, 2 + 1 sections', fwd, 8.0*S + 4.0*C*nd*F, 5, None),
                (f'hipdsp_chain_forward {nfft}/{hop}, 2 + 1 sections, with the dB image',
                 lambda nfft=nfft, hop=hop, nd=nd: h.chain_forward(ctx, bp2, lp1, dx, T, df, T, C, T, nfft, hop, rate, ds, nd, db_out=db), 8.0*S + 8.0*C*nd*F, 5, None),
                (f'hipdsp_chain_forward {nfft}/{hop}, 2 sections, no envelope',
                 lambda nfft=nfft, hop=hop, nd=nd: h.chain_forward(ctx, bp2, None, dx, T, df, T, C, T, nfft, hop, rate, ds, nd), 8.0*S + 4.0*C*nd*F, 5, None),
                (f'hipdsp_chain_forward {nfft}/{hop}, 4 + 2 sections',
                 lambda nfft=nfft, hop=hop, nd=nd: h.chain_forward(ctx, bp4, lp2, dx, T, df, T, C, T, nfft, hop, rate, ds, nd), 8.0*S + 4.0*C*nd*F, 5, None),
            ]
        if (nfft, hop) == (2048, 1024):
            out += [
                ('hipdsp_sosfilt_envelope phase 2 (backward sweep behind the fused forward sweep)',
                 lambda: h.sosfilt_envelope(ctx, bp2, lp1, dx, T, df, T, de, T, C, T, phase=2), 8.0*S, 5, fwd),
                ('hipdsp_decibel over the PSD', lambda nd=nd, F=F: h.decibel(ctx, ds, db, C*nd*F), 8.0*C*nd*F, 5, None),
            ]
    return out


def timed(b, f, n):
    ctx = b['ctx']
    f(); f()
    ctx.record(b['e0'])
    for _ in range(n):
        f()
    ctx.record(b['e1'])
    return ctx.elapsed_ms(b['e0'], b['e1'])/n


tables = [entries(b) for b in builds]
logs = {b['name']: [] for b in builds}
for row in zip(*tables):
    best = {b['name']: 1e30 for b in builds}
    for _ in range(rounds):
        for b, (name, f, nbytes, n, prep) in zip(builds, row):
            if prep is not None:
                prep()                                   # (leaves what this entry point reads: the filtered trace, the tile states)
                b['ctx'].synchronize()
            best[b['name']] = min(best[b['name']], timed(b, f, n))
            b['ctx'].synchronize()
    name, nbytes = row[0][0], row[0][2]
    for b in builds:
        ms = best[b['name']]
        logs[b['name']].append(f'{name:78s} {ms:8.3f} ms {nbytes/ms/1e6:7.0f} GB/s')
    print(logs[builds[-1]['name']][-1] + ('' if len(builds) == 1 else '    | ' + '  '.join(f"{b['name']} {best[b['name']]:.3f}" for b in builds[:-1])), flush=True)
prefix = os.environ.get('OUT_PREFIX')
if prefix:
    for name, lines in logs.items():
        with open(f'{prefix}_{name}.log', 'w') as f:
            f.write('\n'.join(lines) + '\n')

# ---- the reference's DEFAULT session through the plug-in surface: no filter set (bufferedfilter.py:40-42) + spectrogram
# 256 / 128 (plugins.py:11-13, bufferedspectrogram.py:14-16), whole recording resident: BufferedFilter.update() ->
# recompute_all().  The filtered trace's mirror is a view of the raw slab's device copy (no launch), the spectrogram is
# the only launch.  (The tree's build alone: the facade is the package, not a library.)
if os.environ.get('FACADE', '1') == '1' and libs == ['tree']:
    import time
    from audian_amd import hipdsp
    from audian_amd.bufferedfilter import BufferedFilter
    from audian_amd.bufferedspectrogram import BufferedSpectrogram
    from audian_amd.tracegraph import TraceGraph
    h, ctx = b0['h'], b0['ctx']
    dx = owned['dx']
    for k in ('df', 'de', 'ds', 'db'):
        owned[k].free()
    host = np.empty((T, C), dtype=np.float32)
    chunk = 1 << 20
    tmp = h.DeviceArray(ctx, (chunk, C), np.float64)
    for a in range(0, T, chunk):
        n = min(chunk, T - a)
        h.unpack(ctx, dx.view(a, (1,)), T, tmp, n, C)
        host[a:a + n] = tmp.to_host().reshape(-1)[:n*C].reshape(n, C)
    tmp.free()
    dx.free()
    ctx2 = hipdsp.Context(0)
    hipdsp._default_ctx = ctx2

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
    ctx2.synchronize()
    before = dict(hipdsp.launches)
    t0 = time.perf_counter()
    for _ in range(5):
        filt.update()
    ctx2.synchronize()
    ms = (time.perf_counter() - t0)/5*1e3
    per = {k: (v - before.get(k, 0))//5 for k, v in hipdsp.launches.items() if v != before.get(k, 0)}
    nd = len(spec._hostbuf)
    print(f'{"default session (no filter + 256/128) through BufferedFilter.update(), launches " + str(per):78s} {ms:8.3f} ms {(4.0*S + 4.0*C*nd*129)/ms/1e6:7.0f} GB/s', flush=True)
