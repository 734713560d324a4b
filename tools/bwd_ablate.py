"""What does the envelope's backward sweep wait for?  Option "sos_debug" (results wrong): 1 = every tile is stored into the
channel's first tile (the writes never leave L2), 2 = every prefetch reads the channel's first tile (the reads hit L2):
the instruction streams stay the same, the HBM traffic goes away.
    python tools/bwd_ablate.py [channels=64] [seconds=600]
"""
import sys, os
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
import numpy as np
from audian_amd import hipdsp
from audian_amd.design import butter_sos
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
rate = 96000.0
T = int(secs*rate)
ctx = hipdsp.Context(0)
fplan = hipdsp.SosPlan(ctx, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
eplan = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate))
dx, df, de = (hipdsp.DeviceArray(ctx, (C, T), np.float32) for _ in range(3))
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, de, T, C, T, phase=1)
e0, e1 = ctx.event(), ctx.event()
bwd = lambda: hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, de, T, C, T, phase=2)
for w in (8, 16):
    ctx.set_option('sos_waves_per_cu', w); ctx.set_option('sos_waves_min', w)
    for dbg, what in [(0, 'as it is'), (1, 'writes stay in L2'), (2, 'reads hit L2'), (3, 'no HBM traffic at all')]:
        ctx.set_option('sos_debug', dbg)
        bwd(); ctx.synchronize()
        ts = []
        for _ in range(5):
            ctx.record(e0); bwd(); ctx.record(e1); ctx.synchronize(); ts.append(ctx.elapsed_ms(e0, e1))
        print(f'{w:2d} waves per CU, {what:24s}: {np.median(ts):7.3f} ms', flush=True)
ctx.set_option('sos_debug', 0)
