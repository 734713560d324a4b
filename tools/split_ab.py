"""A/B in one process at configs[2]: the step as forward sweep (both frames of a tile) + backward sweep, against the
frame split (forward sweep: even frames; backward sweep: envelope + odd frames); interleaved rounds."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos

C, rate, nfft, hop = int(os.environ.get('CH', '64')), 96000.0, 2048, 1024
T = int(float(os.environ.get('SECONDS_', '600'))*rate)
nd = (T + hop - 1)//hop
F = nfft//2 + 1
ctx = hipdsp.Context(0)
e0, e1, e2 = ctx.event(), ctx.event(), ctx.event()
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
df = hipdsp.DeviceArray(ctx, (C, T), np.float32)
de = hipdsp.DeviceArray(ctx, (C, T), np.float32)
ds = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 1236)
fplan = hipdsp.SosPlan(ctx, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
eplan = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate))


def fwd():
    hipdsp.chain_forward(ctx, fplan, eplan, dx, T, df, T, C, T, nfft, hop, rate, ds, nd)


def bwd(split):
    if split:
        hipdsp.chain_backward(ctx, eplan, df, T, de, T, C, T, nfft, hop, rate, ds, nd)
    else:
        hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, de, T, C, T, phase=2)


res = {0: [], 1: []}
for rnd in range(5):
    for split in (0, 1):
        ctx.set_option('chain_split_frames', split)
        fwd(); bwd(split)
        tf = tb = 0.0
        n = 5
        for _ in range(n):
            ctx.record(e0); fwd(); ctx.record(e1); bwd(split); ctx.record(e2)
            tf += ctx.elapsed_ms(e0, e1)/n
            tb += ctx.elapsed_ms(e1, e2)/n
        res[split].append((tf + tb, tf, tb))
ctx.set_option('chain_split_frames', 0)
for split, name in ((0, 'both frames in the forward sweep'), (1, 'frame split                     ')):
    v = sorted(res[split])
    s, f, b = v[len(v)//2]
    gbf = (8.0*C*T + (2.0 if split else 4.0)*C*nd*F)/1e9
    gbb = (8.0*C*T + (2.0 if split else 0.0)*C*nd*F)/1e9
    print(f'{name}: step {s:.3f} ms = {C*T/s/1e3:.0f} Msamples/s | forward {f:.3f} ms ({gbf/f*1e3:.0f} GB/s of {gbf:.1f} GB) '
          f'| backward {b:.3f} ms ({gbb/b*1e3:.0f} GB/s of {gbb:.1f} GB)', flush=True)
