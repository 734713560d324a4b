"""Experiment harness: time the spectrogram kernel under ctx options (not a test)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp

C, T, nfft, hop, rate = 64, int(120*96000), 2048, 1024, 96000.0
ctx = hipdsp.Context(0)
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
nd = (T + hop - 1)//hop
ds = hipdsp.DeviceArray(ctx, (C, nd, nfft//2 + 1), np.float32)
e0, e1 = ctx.event(), ctx.event()

def run(**opts):
    for k, v in opts.items():
        ctx.set_option(k, v)
    for _ in range(2):
        hipdsp.spectrogram(ctx, dx, T, C, T, nfft, hop, rate, ds, nd)
    ctx.record(e0)
    for _ in range(5):
        hipdsp.spectrogram(ctx, dx, T, C, T, nfft, hop, rate, ds, nd)
    ctx.record(e1)
    ms = ctx.elapsed_ms(e0, e1)/5
    gb = (4.0*C*T + 4.0*C*nd*(nfft//2 + 1))/1e9
    print(f'{opts}: {ms:.3f} ms  {gb/ms*1e3:.0f} GB/s', flush=True)

for kern, waves in ((3, 4), (3, 6), (3, 8), (0, 8)):
    for fpw in (8, 16, 64):
        run(spec_kernel=kern, spec_waves=waves, spec_fpw=fpw, spec_flags=0)
for flags in (1, 2, 3):
    run(spec_kernel=3, spec_waves=4, spec_fpw=16, spec_flags=flags)
