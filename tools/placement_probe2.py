"""Experiment: is it the SIZE / alignment of an allocation or its physical placement that moves the backward sweep?
Outputs of several sizes (each allocated three times), timed in turn on the same input.
    python tools/placement_probe2.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos

C, rate = 64, 96000.0
T = int(600*rate)
ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
df = hipdsp.DeviceArray(ctx, (C, T), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
fplan = hipdsp.SosPlan(ctx, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
eplan = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate))
first = hipdsp.DeviceArray(ctx, (C*T,), np.float32)
hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, first.view(0, (C, T)), T, C, T, phase=1)
ctx.synchronize()


def timed(out, n=5):
    f = lambda: hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, out, T, C, T, phase=2)
    f(); f()
    ctx.record(e0)
    for _ in range(n):
        f()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1)/n


exact = C*T
MiB2, GiB = (2 << 20)//4, (1 << 30)//4
sizes = {'exact': exact, 'up to 2 MiB': -(-exact//MiB2)*MiB2, 'up to 1 GiB': -(-exact//GiB)*GiB, '+ 1 GiB': exact + GiB,
         '+ 4 KiB': exact + 1024}
keep = []
for rnd in range(3):
    for name, n in sizes.items():
        a = hipdsp.DeviceArray(ctx, (n,), np.float32)
        keep.append(a)
        print(f'round {rnd}: {name:12s} {n*4/2**30:8.3f} GiB at {a.ptr:#x}: {timed(a.view(0, (C, T))):.3f} ms', flush=True)
print('again, in order of allocation:')
for a in keep:
    print(f'  {a.nbytes/2**30:8.3f} GiB at {a.ptr:#x}: {timed(a.view(0, (C, T))):.3f} ms', flush=True)
