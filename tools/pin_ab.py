"""A/B in one process: the envelope's backward sweep (configs[2] shape) with the plan tables fetched in
batches (CASC_PIN_GROUPS) against hipcc's just-in-time scalar loads ("sos_no_pin"), interleaved rounds."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos

ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()
C, rate = 64, 96000.0
T = int(float(os.environ.get('SECONDS_', '600'))*rate)
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


bwd = lambda: hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, dy, T, de, T, C, T, phase=2)
res = {0: [], 1: []}
for rnd in range(6):
    for nopin in (0, 1):
        ctx.set_option('sos_no_pin', nopin)
        res[nopin].append(timed(bwd))
ctx.set_option('sos_no_pin', 0)
for nopin, name in ((0, 'batched table loads'), (1, 'just-in-time loads ')):
    v = sorted(res[nopin])
    print(f'env_bwd<1>, {name}: median {v[len(v)//2]:.3f} ms  min {v[0]:.3f} ms   ({8*C*T/v[len(v)//2]/1e6:.0f} GB/s)')
