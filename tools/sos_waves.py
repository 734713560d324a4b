"""The stand-alone sweeps (BufferedFilter alone: sos_scan_kernel; the unfused filter + envelope-state sweep:
sos_ckpt_kernel; BufferedEnvelope alone) for forced numbers of resident waves per CU and for the planner's choice.
    python tools/sos_waves.py [seconds=600]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos
ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()
C, rate = 64, 96000.0
T = int(float(sys.argv[1]) if len(sys.argv) > 1 else 600.0)*int(rate)
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
dy = hipdsp.DeviceArray(ctx, (C, T), np.float32)
de = hipdsp.DeviceArray(ctx, (C, T), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
plan = hipdsp.SosPlan(ctx, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
plan4 = hipdsp.SosPlan(ctx, butter_sos(4, (300.0, 3000.0), 'bandpass', rate))
eplan = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate))
cases = (('sosfilt, 2 sections', lambda: hipdsp.sosfilt(ctx, plan, dx, T, dy, T, C, T, 0), 8.0*C*T),
         ('sosfilt, 4 sections', lambda: hipdsp.sosfilt(ctx, plan4, dx, T, dy, T, C, T, 0), 8.0*C*T),
         ('filter + envelope states (sos_ckpt)', lambda: hipdsp.sosfilt_envelope(ctx, plan, eplan, dx, T, dy, T, de, T, C, T, phase=1), 8.0*C*T),
         ('envelope states alone (sos_ckpt<0,1>)', lambda: hipdsp.sosfilt_envelope(ctx, plan, eplan, dx, T, dy, T, de, T, C, T, phase=1) if False else hipdsp.envelope(ctx, eplan, dx, T, de, T, C, T, 0), 12.0*C*T))
for w in (0, 4, 8, 12, 16):
    ctx.set_option('sos_waves_per_cu', w)
    ctx.set_option('sos_waves_min', w)      # exactly that many (0: the planner's choice)
    for name, f, nb in cases:
        f(); f()
        ctx.record(e0)
        for _ in range(5): f()
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1)/5
        print(f"waves/CU {w if w else 'planner':>7} {name:40s}: {ms:.3f} ms {nb/ms/1e6:.0f} GB/s", flush=True)
