"""PSD kernel rate per window length (hop = nfft/HOPDIV, default 2), 64 ch x SECONDS_ (120) s x 96 kHz
resident in HBM: algorithmic bytes (4 per sample read + 4 per bin written) / event time.
Usage: [sizes ...]
    LIBS=tools/_ab/libr04.so,tree OUT_PREFIX=gpurun_out/r05_spec_sizes python tools/spec_sizes_bench.py
LIBS: several builds of the library in this one process on the same buffers, each window timed for each build in turn
(ROUNDS times, the fastest counts), one log per build -- see tools/entry_points_bench.py."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from _builds import load_build, build_name
sizes = [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536]
C, T, rate = 64, int(float(os.environ.get('SECONDS_', '120'))*96000), 96000.0
hopdiv = int(os.environ.get('HOPDIV', '2'))
libs = [s for s in os.environ.get('LIBS', 'tree').split(',') if s]
rounds = int(os.environ.get('ROUNDS', '3' if len(libs) > 1 else '1'))
builds = []
for lib in libs:
    h, _ = load_build(lib)
    ctx = h.Context(0)
    if os.environ.get('SPEC_KERNEL'):
        ctx.set_option('spec_kernel', int(os.environ['SPEC_KERNEL']))      # 2: the alternative kernel of a size
    if os.environ.get('FPW'):
        ctx.set_option('spec_fpw', int(os.environ['FPW']))                 # frames per wave / workgroup run instead of the heuristic
    builds.append({'name': build_name(lib), 'h': h, 'ctx': ctx, 'e0': ctx.event(), 'e1': ctx.event()})
b0 = builds[0]
# PAIRS="1024:100,256:37": arbitrary overlaps (the reference's overlap spin box, databrowser.py:522-529) instead of the size list
pairs = [(n, max(n//hopdiv, 1)) for n in sizes]
if os.environ.get('PAIRS'):
    pairs = [tuple(int(v) for v in p.split(':')) for p in os.environ['PAIRS'].split(',')]
tlen = lambda nfft: T if nfft >= 64 else T//8          # the tiny windows write 2x the input: keep the output small
nbins = max(C*((tlen(n) + hp - 1)//hp)*(n//2 + 1) for n, hp in pairs)
own = {'dx': b0['h'].DeviceArray(b0['ctx'], (C, T), np.float32), 'ds': b0['h'].DeviceArray(b0['ctx'], (nbins,), np.float32),
       'db': b0['h'].DeviceArray(b0['ctx'], (nbins,), np.float32)}
b0['h'].synth(b0['ctx'], own['dx'], T, C, T, rate, 7)
for b in builds:
    b['buf'] = {k: (a if b is b0 else b['h'].DeviceArray(b['ctx'], a.shape, np.float32, ptr=a.ptr, owner=a)) for k, a in own.items()}
ncalls = int(os.environ.get('TIMED_CALLS', '8'))       # (tools/spec_pmc.sh and spec_sq.sh count on 2 + 4 calls per variant)
# the clocks come up over a few hundred milliseconds of load: without this the first sizes of a short list read 10-15 % low
for _ in range(int(os.environ.get('WARM_CALLS', '60'))):
    b0['h'].spectrogram(b0['ctx'], own['dx'], T, C, T, 2048, 1024, rate, own['ds'], (T + 1023)//1024)
b0['ctx'].synchronize()
logs = {b['name']: [] for b in builds}
for nfft, hop in pairs:
    t = tlen(nfft)
    nd = (t + hop - 1)//hop
    for want_db in (False, True):
        best = {b['name']: 1e30 for b in builds}
        for _ in range(rounds):
            for b in builds:
                h, ctx, dx, ds = b['h'], b['ctx'], b['buf']['dx'], b['buf']['ds']
                db = b['buf']['db'] if want_db else None
                for _ in range(2):
                    h.spectrogram(ctx, dx, T, C, t, nfft, hop, rate, ds, nd, db_out=db)
                ctx.record(b['e0'])
                for _ in range(ncalls):
                    h.spectrogram(ctx, dx, T, C, t, nfft, hop, rate, ds, nd, db_out=db)
                ctx.record(b['e1'])
                best[b['name']] = min(best[b['name']], ctx.elapsed_ms(b['e0'], b['e1'])/ncalls)
                ctx.synchronize()
        gb = (4.0*C*t + (8.0 if want_db else 4.0)*C*nd*(nfft//2 + 1))/1e9
        for b in builds:
            ms = best[b['name']]
            logs[b['name']].append(f'nfft {nfft:6d} hop {hop:6d} {"PSD+dB" if want_db else "PSD   "}: {ms:8.3f} ms  {gb/ms*1e3:6.0f} GB/s')
        print(logs[builds[-1]['name']][-1] + ('' if len(builds) == 1 else '    | ' + '  '.join(f"{b['name']} {best[b['name']]:.3f}" for b in builds[:-1])), flush=True)
prefix = os.environ.get('OUT_PREFIX')
if prefix:
    for name, lines in logs.items():
        with open(f'{prefix}_{name}.log', 'w') as f:
            f.write('\n'.join(lines) + '\n')
