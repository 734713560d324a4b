"""Does a quick memset over an allocation predict how fast the envelope's backward sweep writes into it?  N envelope
buffers of configs[2]'s size, each timed with hipdsp_memset and as the output of the backward sweep behind ONE fused
forward sweep, round-robin (profiles/r03_placement_probe.log found allocations in three loose classes)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audian_amd import hipdsp
from audian_amd.design import butter_sos
C, rate = 64, 96000.0
T = int(float(os.environ.get('SECONDS_', '600'))*rate)
N = int(os.environ.get('BUFFERS', '6'))
nfft, hop = 2048, 1024
nd, F = (T + hop - 1)//hop, nfft//2 + 1
ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()
dx, df = (hipdsp.DeviceArray(ctx, (C, T), np.float32) for _ in range(2))
ds = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
bp = hipdsp.SosPlan(ctx, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
lp = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate))
hipdsp.chain_forward(ctx, bp, lp, dx, T, df, T, C, T, nfft, hop, rate, ds, nd)
outs = [hipdsp.DeviceArray(ctx, (C, T), np.float32) for _ in range(N)]
def timed(f, n=6):
    f(); f()
    ctx.record(e0)
    for _ in range(n):
        f()
    ctx.record(e1)
    ctx.synchronize()
    return ctx.elapsed_ms(e0, e1)/n
ms = {i: [1e9, 1e9] for i in range(N)}
for rnd in range(3):
    for i, de in enumerate(outs):
        t_set = timed(lambda: hipdsp.lib.hipdsp_memset(ctx.handle, hipdsp._p(de), 0, 4*C*T))
        t_bwd = timed(lambda: hipdsp.sosfilt_envelope(ctx, bp, lp, dx, T, df, T, de, T, C, T, phase=2))
        ms[i] = [min(ms[i][0], t_set), min(ms[i][1], t_bwd)]
for i, de in enumerate(outs):
    print(f'buffer {i} at {de.ptr:#x}: memset {ms[i][0]:.3f} ms, backward sweep {ms[i][1]:.3f} ms', flush=True)
a = np.array([ms[i] for i in range(N)])
print('correlation of the two times over the buffers: %.2f; backward sweep %.3f ... %.3f ms' % (np.corrcoef(a[:, 0], a[:, 1])[0, 1], a[:, 1].min(), a[:, 1].max()))
# the same question for the fused forward sweep (VALU-bound): the filtered trace goes into each of the buffers
# in turn, the PSD stays where it is
ms = {i: [1e9, 1e9] for i in range(N)}
for rnd in range(3):
    for i, de in enumerate(outs):
        t_set = timed(lambda: hipdsp.lib.hipdsp_memset(ctx.handle, hipdsp._p(de), 0, 4*C*T))
        t_fwd = timed(lambda: hipdsp.chain_forward(ctx, bp, lp, dx, T, de, T, C, T, nfft, hop, rate, ds, nd))
        ms[i] = [min(ms[i][0], t_set), min(ms[i][1], t_fwd)]
for i, de in enumerate(outs):
    print(f'buffer {i}: memset {ms[i][0]:.3f} ms, forward sweep with the filtered trace in it {ms[i][1]:.3f} ms', flush=True)
a = np.array([ms[i] for i in range(N)])
print('correlation: %.2f; forward sweep %.3f ... %.3f ms' % (np.corrcoef(a[:, 0], a[:, 1])[0, 1], a[:, 1].min(), a[:, 1].max()))
# and the allocator that chooses: three blocks by hipdsp_malloc_probed, each against the memset time of a plain one
for k in range(3):
    pr = hipdsp.DeviceArray(ctx, (C, T), np.float32, write_probe=4)
    t_set = timed(lambda: hipdsp.lib.hipdsp_memset(ctx.handle, hipdsp._p(pr), 0, 4*C*T))
    t_bwd = timed(lambda: hipdsp.sosfilt_envelope(ctx, bp, lp, dx, T, df, T, pr, T, C, T, phase=2)) if k == 0 else float('nan')
    print(f'hipdsp_malloc_probed(tries = 4) block {k}: memset {t_set:.3f} ms' + (f', backward sweep {t_bwd:.3f} ms' if k == 0 else ''), flush=True)
