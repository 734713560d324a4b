import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from audian_amd import hipdsp
from audian_amd.design import butter_sos
C, rate, nfft, hop = 64, 96000.0, 2048, 1024
T = int(600*rate); nd = (T + hop - 1)//hop; F = nfft//2 + 1
ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32); df = hipdsp.DeviceArray(ctx, (C, T), np.float32)
ds = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 1236)
fplan = hipdsp.SosPlan(ctx, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
eplan = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate))
def timed(f, n=5):
    f()
    ctx.record(e0)
    for _ in range(n): f()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1)/n
fused = lambda: hipdsp.chain_forward(ctx, fplan, eplan, dx, T, df, T, C, T, nfft, hop, rate, ds, nd)
res = {}
for rnd in range(5):
    for bits in [int(b) for b in os.environ.get('BITS', '0,64').split(',')]:
        ctx.set_option('chain_debug', bits)
        res.setdefault(bits, []).append(timed(fused))
for bits, v in res.items():
    v = sorted(v)
    print(f'chain_debug {bits:3d}: median {v[len(v)//2]:.3f} ms  min {v[0]:.3f} ms')
