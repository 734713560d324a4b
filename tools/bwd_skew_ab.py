"""Does the RELATIVE placement of the trace the backward sweep reads and the envelope it writes matter?  Every wave
reads tile t of `filtered` and writes tile t of `envelope` at the same offset into two 14.7 GB arrays: if the two
bases are congruent modulo the memory channels' interleaving period, a wave's read and write streams always meet
on the same channel.  One process, the same physical buffers, the envelope written at different byte skews."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos

ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()
C, rate = 64, 96000.0
T = int(600*rate)
PAD = 1 << 21                                   # floats of slack behind the envelope
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
dy = hipdsp.DeviceArray(ctx, (C, T), np.float32)
de = hipdsp.DeviceArray(ctx, (C*T + PAD,), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
fplan = hipdsp.SosPlan(ctx, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
eplan = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate))
hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, dy, T, de.view(0, (C, T)), T, C, T, phase=1)
print('bases: x %#x  filtered %#x  envelope %#x' % (dx.ptr, dy.ptr, de.ptr))


def timed(f, n=5):
    f()
    ctx.record(e0)
    for _ in range(n):
        f()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1)/n


res = {}
skews = [0, 64, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, (1 << 20) + 4096 + 256]
for rnd in range(3):
    for skew in skews:                          # bytes
        out = de.view(skew//4, (C, T))
        res.setdefault(skew, []).append(timed(lambda: hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, dy, T, out, T, C, T, phase=2)))
for skew, v in res.items():
    v = sorted(v)
    print(f'envelope skewed by {skew:8d} B: backward sweep median {v[len(v)//2]:.3f} ms  ({8*C*T/v[len(v)//2]/1e6:.0f} GB/s)')
# and the pitch: channels 230.4 MB apart (T floats) against a pitch with a skew per channel
for pitch_extra in (0, 64, 1024, 16384):
    P = T + pitch_extra
    if C*P > C*T + PAD:
        break
    out = de.view(0, (C, P))
    v = sorted(timed(lambda: hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, dy, T, out, P, C, T, phase=2)) for _ in range(3))
    print(f'envelope pitch T + {pitch_extra:6d} floats: backward sweep median {v[1]:.3f} ms')
