"""Experiment: how much does the backward envelope sweep's time depend on WHERE its output (and input) buffers lie?
One process, one set of inputs; the output is a view at different offsets into one large allocation, then the same
with freshly allocated buffers.  (Round 3: two output buffers in one process differed by 9 %.)
    python tools/placement_probe.py"""
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
EXTRA = 1 << 28                                    # elements (1 GiB)
big = hipdsp.DeviceArray(ctx, (C*T + EXTRA,), np.float32)
hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, big.view(0, (C, T)), T, C, T, phase=1)
ctx.synchronize()


def timed(out, src=df, n=5):
    f = lambda: hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, src, T, out, T, C, T, phase=2)
    f()
    ctx.record(e0)
    for _ in range(n):
        f()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1)/n


print(f'base pointers: x {dx.ptr:#x} filtered {df.ptr:#x} big {big.ptr:#x}')
for rep in range(2):
    for off_bytes in (0, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 16 << 20, (16 << 20) + 4096, 256 << 20, 512 << 20, 1 << 30):
        out = big.view(off_bytes//4, (C, T))
        print(f'pass {rep}: output at +{off_bytes:>11d} B: {timed(out):.3f} ms')
fresh = []
for k in range(4):
    o = hipdsp.DeviceArray(ctx, (C, T), np.float32)
    fresh.append(o)
    print(f'fresh output buffer {k} at {o.ptr:#x} (distance to filtered {abs(o.ptr - df.ptr)/2**30:.2f} GiB): {timed(o):.3f} ms')
for k, o in enumerate(fresh):
    print(f'again, fresh {k}: {timed(o):.3f} ms')
# and the INPUT somewhere else (copy of the filtered trace into a fresh buffer)
for k, o in enumerate(fresh[:3]):
    hipdsp.memcpy2d(ctx, o, T*4, df, T*4, T*4, C)
    ctx.synchronize()
    print(f'input = fresh {k}, output at +0: {timed(big.view(0, (C, T)), src=o):.3f} ms')
