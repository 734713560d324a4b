"""How many waves (pairs) per CU should the sweeps use when the job does not fill the chip?  Forces the segment planner to
each level ("sos_waves_per_cu" = "sos_waves_min" = w; "chain_pairs" = p) and times the backward sweep
and the fused forward sweep, then the planner's own choice.
    python tools/occupancy_sweep.py CHANNELS SECONDS RATE [NFFT HOP ORDER]
"""
import sys, os
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
import numpy as np
from audian_amd import hipdsp
from audian_amd.design import butter_sos

C, secs, rate = int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3])
nfft, hop, order = (int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (2048, 1024, 2)
T = int(secs*rate)
nd = (T + hop - 1)//hop
F = nfft//2 + 1
ctx = hipdsp.Context(0)
fplan = hipdsp.SosPlan(ctx, butter_sos(order, (300.0, 3000.0), 'bandpass', rate))
eplan = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate))
dx, df, de = (hipdsp.DeviceArray(ctx, (C, T), np.float32) for _ in range(3))
ds = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
e0, e1 = ctx.event(), ctx.event()


def timed(fn, reps=7):
    fn(); ctx.synchronize()
    best = []
    for _ in range(reps):
        ctx.record(e0); fn(); ctx.record(e1); ctx.synchronize()
        best.append(ctx.elapsed_ms(e0, e1))
    return float(np.median(best))


fwd = lambda: hipdsp.chain_forward(ctx, fplan, eplan, dx, T, df, T, C, T, nfft, hop, rate, ds, nd)
bwd = lambda: hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, de, T, C, T, phase=2)
print(f'{C} ch x {secs:g} s x {rate/1000:g} kHz, {nfft}/{hop}, band-pass order {order}: {T//2048} tiles per channel')
fwd(); ctx.synchronize()
for w in (2, 4, 6, 8, 12, 16):
    ctx.set_option('sos_waves_per_cu', w); ctx.set_option('sos_waves_min', w)
    print(f'  backward sweep forced to {w:2d} waves per CU: {timed(bwd):8.4f} ms')
ctx.set_option('sos_waves_per_cu', 0); ctx.set_option('sos_waves_min', 0)
print(f'  backward sweep, planner\'s choice:        {timed(bwd):8.4f} ms')
for p in (1, 2, 3, 4, 6, 8):
    ctx.set_option('chain_pairs', p)
    seg, n = hipdsp.chain_plan(ctx, fplan, eplan, C, T)
    print(f'  fused forward sweep forced to {p} pairs per CU ({n} segments of {seg//2048} tiles): {timed(fwd):8.4f} ms')
ctx.set_option('chain_pairs', 0)
seg, n = hipdsp.chain_plan(ctx, fplan, eplan, C, T)
print(f'  fused forward sweep, planner\'s choice ({n} segments of {seg//2048} tiles):  {timed(fwd):8.4f} ms')
