"""The backward sweep takes 5.7 or 6.1-6.4 ms from process to process on one box.  Is it how the three 14.7 GB
arrays were allocated (three hipMallocs against views into ONE allocation, which the driver may map with larger
page fragments)?   python tools/bwd_alloc_ab.py separate|one    (one line per process; run several)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos

mode = sys.argv[1] if len(sys.argv) > 1 else 'separate'
ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()
C, rate = 64, 96000.0
T = int(600*rate)
if mode == 'one':
    ctx.set_option('pool_limit_mb', 0)
    big = hipdsp.DeviceArray(ctx, (3*C*T,), np.float32)
    dx, dy, de = big.view(0, (C, T)), big.view(C*T, (C, T)), big.view(2*C*T, (C, T))
else:
    dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
    dy = hipdsp.DeviceArray(ctx, (C, T), np.float32)
    de = hipdsp.DeviceArray(ctx, (C, T), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
fplan = hipdsp.SosPlan(ctx, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
eplan = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate))
hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, dy, T, de, T, C, T, phase=1)


def timed(f, n=5):
    f()
    ctx.record(e0)
    for _ in range(n):
        f()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1)/n


v = sorted(timed(lambda: hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, dy, T, de, T, C, T, phase=2)) for _ in range(5))
print(f'{mode:8s}: backward sweep median {v[2]:.3f} ms  ({8*C*T/v[2]/1e6:.0f} GB/s)   bases {dx.ptr:#x} {dy.ptr:#x} {de.ptr:#x}')
