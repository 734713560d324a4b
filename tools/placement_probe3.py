"""Experiment (profiles/r03_placement_probe.log, fourth probe): does anything simpler than the backward sweep see the
fast and slow allocations?  Eight output allocations; per allocation the sweep writing into it, hipMemsetAsync over it,
the float4 copy probe writing into it and reading from it.
    python tools/placement_probe3.py"""
import os, sys, ctypes
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
outs = [hipdsp.DeviceArray(ctx, (C, T), np.float32) for _ in range(8)]
hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, outs[0], T, C, T, phase=1)
ctx.synchronize()
def timed(f, n=5):
    f(); f()
    ctx.record(e0)
    for _ in range(n): f()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1)/n
nbytes = C*T*4
for rep in range(2):
    for i, o in enumerate(outs):
        bwd = timed(lambda: hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, o, T, C, T, phase=2))
        fill = timed(lambda: hipdsp.check(hipdsp.lib.hipdsp_memset(ctx.handle, ctypes.c_void_p(o.ptr), 0, nbytes)))
        cpw = timed(lambda: hipdsp.check(hipdsp.lib.hipdsp_copy_probe(ctx.handle, ctypes.c_void_p(o.ptr), ctypes.c_void_p(df.ptr), nbytes)))
        cpr = timed(lambda: hipdsp.check(hipdsp.lib.hipdsp_copy_probe(ctx.handle, ctypes.c_void_p(dx.ptr), ctypes.c_void_p(o.ptr), nbytes)))
        print(f'pass {rep} alloc {i} at {o.ptr:#x}: env_bwd into it {bwd:.3f} ms | memset {fill:.3f} ms | copy INTO it {cpw:.3f} ms | copy FROM it {cpr:.3f} ms', flush=True)
        # restore dx (the copy FROM overwrote it) and the envelope state
    hipdsp.synth(ctx, dx, T, C, T, rate, 7)
    hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, outs[0], T, C, T, phase=1)
