"""Experiment (profiles/r03_placement_probe.log, fifth probe): is it the ORDER of the allocations that decides an
allocation's class?  One process per call; the envelope buffer is the third or the fourth allocation.
    python tools/placement_probe4.py de_third | de_fourth      (alternate a few times in one gpurun call)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos
order = sys.argv[1]
C, rate, nfft, hop = 64, 96000.0, 2048, 1024
T = int(600*rate); F = nfft//2 + 1; nd = (T + hop - 1)//hop
ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
df = hipdsp.DeviceArray(ctx, (C, T), np.float32)
if order == 'de_third':
    de = hipdsp.DeviceArray(ctx, (C, T), np.float32); ds = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32)
else:
    ds = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32); de = hipdsp.DeviceArray(ctx, (C, T), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 1236)
fplan = hipdsp.SosPlan(ctx, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
eplan = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate))
def timed(f, n=10):
    f(); f()
    ctx.record(e0)
    for _ in range(n): f()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1)/n
fwd = timed(lambda: hipdsp.chain_forward(ctx, fplan, eplan, dx, T, df, T, C, T, nfft, hop, rate, ds, nd))
bwd = timed(lambda: hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, de, T, C, T, phase=2))
print(f'{order}: forward {fwd:.3f} ms  backward {bwd:.3f} ms  sum {fwd + bwd:.3f}')
