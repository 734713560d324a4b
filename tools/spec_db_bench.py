import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
from audian_amd import hipdsp
C, T, rate = 64, int(120*96000), 96000.0
ctx = hipdsp.Context(0)
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
e0, e1 = ctx.event(), ctx.event()
for nfft, hop in ((2048, 1024), (2048, 512), (1024, 512), (4096, 2048)):
    nd = (T + hop - 1)//hop
    ds = hipdsp.DeviceArray(ctx, (C, nd, nfft//2 + 1), np.float32)
    db = hipdsp.DeviceArray(ctx, (C, nd, nfft//2 + 1), np.float32)
    for _ in range(3):
        hipdsp.spectrogram(ctx, dx, T, C, T, nfft, hop, rate, ds, nd, db_out=db)
    ctx.record(e0)
    for _ in range(5):
        hipdsp.spectrogram(ctx, dx, T, C, T, nfft, hop, rate, ds, nd, db_out=db)
    ctx.record(e1)
    ms = ctx.elapsed_ms(e0, e1)/5
    gb = (4.0*C*T + 8.0*C*nd*(nfft//2 + 1))/1e9
    print(f'nfft {nfft} hop {hop} with dB: {ms:.3f} ms  {gb/ms*1e3:.0f} GB/s', flush=True)
    ds.free(); db.free()
