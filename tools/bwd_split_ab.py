"""A/B in one process, on the same buffers: the envelope's backward sweep as it is (env_bwd_kernel) against the
role-split form (envsplit.hip, context option "sos_split": eight compute waves that issue no vector-memory instruction
for interior tiles + four mover waves per CU), BASELINE configs[2]'s shape by default.
    python tools/bwd_split_ab.py [channels] [seconds]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
rate, nfft, hop = 96000.0, 2048, 1024
T = int(secs*rate)
nd = (T + hop - 1)//hop
ctx = hipdsp.Context(0)
sos, esos = butter_sos(2, (300.0, 3000.0), 'bandpass', rate), butter_sos(2, 20.0, 'lowpass', rate)
fplan, eplan = hipdsp.SosPlan(ctx, sos), hipdsp.SosPlan(ctx, esos)
dx, df, de = (hipdsp.DeviceArray(ctx, (C, T), np.float32) for _ in range(3))
ds = hipdsp.DeviceArray(ctx, (C, nd, nfft//2 + 1), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
hipdsp.chain_forward(ctx, fplan, eplan, dx, T, df, T, C, T, nfft, hop, rate, ds, nd)
ctx.synchronize()
e0, e1 = ctx.event(), ctx.event()


def bwd():
    hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, de, T, C, T, phase=2)


def timed(n=3):
    bwd(); ctx.synchronize()
    ctx.record(e0)
    for _ in range(n):
        bwd()
    ctx.record(e1); ctx.synchronize()
    return ctx.elapsed_ms(e0, e1)/n


res = {0: [], 1: []}
heads = {}
for rnd in range(6):
    for mode in (0, 1):
        ctx.set_option('sos_split', mode)
        res[mode].append(timed())
        if rnd == 0:
            heads[mode] = [de.view(c*T + off, (min(T, 300000),)).to_host().copy()
                           for c in (0, C - 1) for off in (0, max(0, T//2 - 150000), max(0, T - 300000))]
if os.environ.get('ABLATE'):
    for mode in (0, 1):
        ctx.set_option('sos_split', mode)
        for bits, what in ((1, 'stores stay in L2 / are skipped'), (2, 'fetches hit L2 / are skipped'), (3, 'neither')):
            ctx.set_option('sos_debug', bits)
            print(f'sos_split {mode}, sos_debug {bits} ({what}): {timed():.3f} ms', flush=True)
        ctx.set_option('sos_debug', 0)
ctx.set_option('sos_split', 0)
same = all(np.array_equal(a, b) for a, b in zip(heads[0], heads[1]))
worst = max(float(np.abs(a - b).max()/max(np.abs(a).max(), 1e-30)) for a, b in zip(heads[0], heads[1]))
for mode, name in ((0, 'env_bwd_kernel      '), (1, 'env_bwd_split_kernel')):
    print(f'{name}: median {np.median(res[mode]):7.3f} ms  {[round(x, 3) for x in res[mode]]}')
print(f'{C} ch x {secs:g} s; windows at the start, the middle and the end of the first and the last channel identical: {same} '
      f'(largest difference relative to the largest value {worst:.3g})')
