"""Do the waves of the envelope's backward sweep (env_bwd_kernel, one wave per segment, 16 per CU) progress
alike?  The context option "sos_trace" makes every wave report when it started and ended (100 MHz ticks) and
where it ran (HW_ID); the launch lasts as long as its slowest wave, and a tail with few waves left cannot
keep the HBM pipes full.
    python tools/sweep_trace.py        (on the GPU box)
"""
import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from audian_amd import hipdsp
from audian_amd.design import butter_sos
C, rate = 64, 96000.0
T = int(600*rate)
ctx = hipdsp.Context(0)
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32); df = hipdsp.DeviceArray(ctx, (C, T), np.float32)
de = hipdsp.DeviceArray(ctx, (C, T), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 1236)
fplan = hipdsp.SosPlan(ctx, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
eplan = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate))
W = 8192
tr = hipdsp.DeviceArray(ctx, (W, 9), np.int64)
tr.zero_()
hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, de, T, C, T)       # checkpoints + warm
ctx.set_option('sos_fair', int(os.environ.get('FAIR', '1')))
ctx.set_option('sos_trace_rows', W)
ctx.set_option('sos_trace', tr.ptr)
e0, e1 = ctx.event(), ctx.event()
ctx.record(e0)
hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, de, T, C, T, phase=2)
ctx.record(e1)
ms = ctx.elapsed_ms(e0, e1)
ctx.set_option('sos_trace', 0)
h = tr.to_host()
h = h[h[:, 1] > 0]
t0 = h[:, 0].min()
start = (h[:, 0] - t0)*1e-5
end = (h[:, 1] - t0)*1e-5
hw = h[:, 2]
slot = hw & 15
simd = (hw >> 4) & 3
print(f'backward sweep {ms:.3f} ms, {len(h)} waves; starts within {start.max():.3f} ms')
q = np.percentile(end, [0, 10, 50, 90, 100])
print('waves end at min %.3f  10%% %.3f  median %.3f  90%% %.3f  max %.3f ms' % tuple(q))
for s in np.unique(slot):
    m = slot == s
    print(f'  wave slot {s:2d} of its SIMD: {m.sum():5d} waves, mean end {end[m].mean():.3f} ms')
names = ['tile: prefetch registers -> LDS', 'prefetch of the next tile issued', 'forward cascade', 'backward cascade',
         'LDS -> 8 stores issued', 'wait for the prefetch (vmcnt 7)']
for s in np.unique(slot):
    m = slot == s
    acc = h[m, 3:9].mean(axis=0)
    print(f'  slot {s}: clocks per wave {acc.sum()/1e6:.2f} M: ' + ', '.join(f'{n} {100*v/acc.sum():.1f} %' for n, v in zip(names, acc)))
busy = np.array([(end > t).sum() for t in np.linspace(0, end.max(), 11)[:-1]])
print('waves still running at 0, 10, .. 90 % of the launch:', busy)
def timed(n=5):
    ctx.record(e0)
    for _ in range(n):
        hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, de, T, C, T, phase=2)
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1)/n
for rnd in range(3):
    for fair in (1, 0):
        ctx.set_option('sos_fair', fair)
        print(f'sos_fair {fair}: backward sweep {timed():.3f} ms')
