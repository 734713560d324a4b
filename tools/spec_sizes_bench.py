"""PSD kernel rate per window length (hop = nfft/HOPDIV, default 2), 64 ch x SECONDS_ (120) s x 96 kHz
resident in HBM: algorithmic bytes (4 per sample read + 4 per bin written) / event time.
Usage: [sizes ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
sizes = [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536]
C, T, rate = 64, int(float(os.environ.get('SECONDS_', '120'))*96000), 96000.0
hopdiv = int(os.environ.get('HOPDIV', '2'))
ctx = hipdsp.Context(0)
if os.environ.get('SPEC_KERNEL'):
    ctx.set_option('spec_kernel', int(os.environ['SPEC_KERNEL']))      # 2: the alternative kernel of a size
if os.environ.get('FPW'):
    ctx.set_option('spec_fpw', int(os.environ['FPW']))                 # frames per wave / workgroup run instead of the heuristic
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
e0, e1 = ctx.event(), ctx.event()
ncalls = int(os.environ.get('TIMED_CALLS', '8'))       # (tools/spec_pmc.sh and spec_sq.sh count on 2 + 4 calls per variant)
# the clocks come up over a few hundred milliseconds of load: without this the first sizes of a short list read 10-15 % low
_w = hipdsp.DeviceArray(ctx, (C, (T + 1023)//1024, 1025), np.float32)
for _ in range(int(os.environ.get('WARM_CALLS', '60'))):
    hipdsp.spectrogram(ctx, dx, T, C, T, 2048, 1024, rate, _w, (T + 1023)//1024)
ctx.synchronize()
_w.free()
# PAIRS="1024:100,256:37": arbitrary overlaps (the reference's overlap spin box, databrowser.py:522-529) instead of the size list
pairs = [(n, max(n//hopdiv, 1)) for n in sizes]
if os.environ.get('PAIRS'):
    pairs = [tuple(int(v) for v in p.split(':')) for p in os.environ['PAIRS'].split(',')]
for nfft, hop in pairs:
    t = T if nfft >= 64 else T//8          # the tiny windows write 2x the input: keep the output small
    nd = (t + hop - 1)//hop
    for want_db in (False, True):
        ds = hipdsp.DeviceArray(ctx, (C, nd, nfft//2 + 1), np.float32)
        db = hipdsp.DeviceArray(ctx, (C, nd, nfft//2 + 1), np.float32) if want_db else None
        for _ in range(2):
            hipdsp.spectrogram(ctx, dx, T, C, t, nfft, hop, rate, ds, nd, db_out=db)
        ctx.record(e0)
        for _ in range(ncalls):
            hipdsp.spectrogram(ctx, dx, T, C, t, nfft, hop, rate, ds, nd, db_out=db)
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1)/ncalls
        gb = (4.0*C*t + (8.0 if want_db else 4.0)*C*nd*(nfft//2 + 1))/1e9
        print(f'nfft {nfft:6d} hop {hop:6d} {"PSD+dB" if want_db else "PSD   "}: {ms:8.3f} ms  {gb/ms*1e3:6.0f} GB/s', flush=True)
        ds.free()
        if db is not None:
            db.free()
