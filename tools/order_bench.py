"""Throughput of hipdsp_sosfilt / hipdsp_envelope against the Butterworth order (1-4 sections per plan)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos
ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()
C, rate = 64, 96000.0
T = int(120*rate)
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
dy = hipdsp.DeviceArray(ctx, (C, T), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
for order in (1, 2, 4, 6, 8):
    lp = hipdsp.SosPlan(ctx, butter_sos(order, 500.0, 'lowpass', rate))
    S = (order + 1)//2
    for name, f, nb in (('sosfilt', lambda: hipdsp.sosfilt(ctx, lp, dx, T, dy, T, C, T, 0), 8.0*C*T),
                        ('envelope', lambda: hipdsp.envelope(ctx, lp, dx, T, dy, T, C, T, 0), 12.0*C*T)):
        f(); f()
        ctx.record(e0)
        for _ in range(3): f()
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1)/3
        print(f'order {order} ({S} sections) {name:9s}: {ms:.3f} ms  {nb/ms/1e6:.0f} GB/s', flush=True)
