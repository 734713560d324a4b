"""A/B in one process: the envelope's backward sweep (configs[2] shape) planned for 8 .. 16 resident waves per CU
(context option "sos_waves_per_cu": fewer, longer segments) -- does the memory system prefer fewer streams?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos

ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()
C, rate = 64, 96000.0
T = int(600*rate)
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
dy = hipdsp.DeviceArray(ctx, (C, T), np.float32)
de = hipdsp.DeviceArray(ctx, (C, T), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
fplan = hipdsp.SosPlan(ctx, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
eplan = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate))


def timed(f, n=5):
    f()
    ctx.record(e0)
    for _ in range(n):
        f()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1)/n


res = {}
for rnd in range(3):
    for w in (8, 10, 12, 14, 16):
        ctx.set_option('sos_waves_per_cu', w)
        ctx.set_option('sos_waves_min', w)      # exactly that many (round 3: the planner may pick fewer otherwise)
        hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, dy, T, de, T, C, T, phase=1)   # checkpoints of this plan
        res.setdefault(w, []).append(timed(lambda: hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, dy, T, de, T, C, T, phase=2)))
for w, v in res.items():
    v = sorted(v)
    print(f'{w:2d} waves per CU: backward sweep median {v[len(v)//2]:.3f} ms  min {v[0]:.3f} ms  ({8*C*T/v[len(v)//2]/1e6:.0f} GB/s)')
