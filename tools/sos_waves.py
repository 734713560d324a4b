import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos
ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()
C, rate = 64, 96000.0
T = int(300*rate)
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
dy = hipdsp.DeviceArray(ctx, (C, T), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
plan = hipdsp.SosPlan(ctx, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
eplan = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate))
for w in (8, 12, 16, 20, 24, 32, 40):
    ctx.set_option('sos_waves_per_cu', w)
    ctx.set_option('sos_waves_min', w)      # exactly that many (round 3: the planner may pick fewer otherwise)
    for name, f, nb in (('filt', lambda: hipdsp.sosfilt(ctx, plan, dx, T, dy, T, C, T, 0), 8.0*C*T),
                        ('env', lambda: hipdsp.envelope(ctx, eplan, dx, T, dy, T, C, T, 0), 12.0*C*T)):
        f(); f()
        ctx.record(e0)
        for _ in range(5): f()
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1)/5
        print(f'waves/CU {w:2d} {name}: {ms:.3f} ms {nb/ms/1e6:.0f} GB/s', flush=True)
