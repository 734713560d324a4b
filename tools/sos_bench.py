"""Experiment harness: time sosfilt / envelope under several shapes (not a test)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos

ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()

def run(C, seconds, rate, what, **kw):
    T = int(seconds*rate)
    dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
    dy = hipdsp.DeviceArray(ctx, (C, T), np.float32)
    hipdsp.synth(ctx, dx, T, C, T, rate, 7)
    if what == 'filt':
        plan = hipdsp.SosPlan(ctx, butter_sos(kw.get('order', 2), (300.0, 3000.0), 'bandpass', rate))
        f = lambda: hipdsp.sosfilt(ctx, plan, dx, T, dy, T, C, T, 0)
        nbytes = 8.0*C*T
    else:
        plan = hipdsp.SosPlan(ctx, butter_sos(2, kw.get('env', 20.0), 'lowpass', rate))
        f = lambda: hipdsp.envelope(ctx, plan, dx, T, dy, T, C, T, 0)
        nbytes = 12.0*C*T          # state sweep 4 B + backward sweep 8 B per sample
    for _ in range(2):
        f()
    ctx.record(e0)
    for _ in range(5):
        f()
    ctx.record(e1)
    ms = ctx.elapsed_ms(e0, e1)/5
    print(f'{what:5s} C={C:3d} {seconds:5.0f}s @{rate/1000:.0f}k {kw}: {ms:8.3f} ms {nbytes/ms/1e6:7.0f} GB/s warm={plan.info()[0]}', flush=True)
    dx.free(); dy.free()

run(64, 600, 96000.0, 'filt')
run(64, 600, 96000.0, 'env')
run(64, 120, 96000.0, 'filt')
run(64, 120, 96000.0, 'env')
run(16, 80, 192000.0, 'filt')
run(16, 80, 192000.0, 'env', env=500.0)
run(16, 80, 192000.0, 'env', env=20.0)
run(4, 60, 48000.0, 'filt', order=4)
run(4, 60, 48000.0, 'env', env=500.0)
run(1, 60, 44100.0, 'filt')
