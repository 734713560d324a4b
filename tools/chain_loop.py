"""Loops the fused forward sweep (and the backward sweep) at configs[2] for a few seconds so that
tools/power_watch.sh can read board power and clocks under exactly this load."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos

C, rate, nfft, hop = 64, 96000.0, 2048, 1024
T = int(600*rate)
nd = (T + hop - 1)//hop
F = nfft//2 + 1
ctx = hipdsp.Context(0)
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
df = hipdsp.DeviceArray(ctx, (C, T), np.float32)
de = hipdsp.DeviceArray(ctx, (C, T), np.float32)
ds = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 1236)
fplan = hipdsp.SosPlan(ctx, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
eplan = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate))
n = int(os.environ.get('LOOP', '200'))
which = os.environ.get('WHICH', 'fwd')
ctx.synchronize()
t0 = time.perf_counter()
for i in range(n):
    if which in ('fwd', 'both'):
        hipdsp.chain_forward(ctx, fplan, eplan, dx, T, df, T, C, T, nfft, hop, rate, ds, nd)
    if which in ('bwd', 'both'):
        hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, de, T, C, T, phase=2)
    if which == 'spec':                 # the stand-alone spectrogram kernel on the filtered trace
        hipdsp.spectrogram(ctx, df, T, C, T, nfft, hop, rate, ds, nd)
    if which == 'sosfilt':              # the stand-alone band-pass
        hipdsp.sosfilt(ctx, fplan, dx, T, df, T, C, T, 0)
ctx.synchronize()
dt = time.perf_counter() - t0
print(f'{which}: {n} rounds in {dt:.2f} s = {dt/n*1e3:.3f} ms each')
