"""The numbers behind BufferedFilter._plan_fusion's cost gate: for every window hipdsp_chain_forward covers and every
length of band-pass (1 .. 4 sections) and envelope plan (none, 1, 2 sections), the time of the fused launch and of the
launches it replaces -- hipdsp_sosfilt (no envelope) or the forward sweep of hipdsp_sosfilt_envelope (phase 1), plus
hipdsp_spectrogram -- at BASELINE configs[2]'s shape, as picoseconds per channel-sample.
    python tools/fusion_cost_bench.py [seconds=600] > audian_amd/fusion_costs.json
(a JSON object on stdout, progress on stderr).  tests/test_gpu_facade.py::test_fused_launch_never_loses re-measures a
subset against the committed table."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos
from audian_amd.bufferedspectrogram import FUSED_WINDOWS

ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()
C, rate = 64, 96000.0
T = int((float(sys.argv[1]) if len(sys.argv) > 1 else 600.0)*rate)
dx, df = (hipdsp.DeviceArray(ctx, (C, T), np.float32) for _ in range(2))
hipdsp.synth(ctx, dx, T, C, T, rate, 1236)
ds = hipdsp.DeviceArray(ctx, (max(C*((T + h - 1)//h)*(n//2 + 1) for n, h in FUSED_WINDOWS),), np.float32)
bp = {s: hipdsp.SosPlan(ctx, butter_sos(s, (300.0, 3000.0), 'bandpass', rate)) for s in (1, 2, 3, 4)}     # order N -> N sections
lp = {0: None, 1: hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate)), 2: hipdsp.SosPlan(ctx, butter_sos(4, 20.0, 'lowpass', rate))}
de = hipdsp.DeviceArray(ctx, (C, T), np.float32)


def timed(f, n=4):
    f(); f()
    ctx.record(e0)
    for _ in range(n):
        f()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1)/n*1e9/(C*T)          # ps per channel-sample


out = {'shape': f'{C} ch x {T/rate:.0f} s x {rate/1e3:.0f} kHz', 'unit': 'ps per channel-sample', 'spectrogram': {}, 'filter': {}, 'fused': {}}
for nfft, hop in sorted(FUSED_WINDOWS):
    nd = (T + hop - 1)//hop
    out['spectrogram'][f'{nfft}/{hop}'] = round(timed(lambda: hipdsp.spectrogram(ctx, df, T, C, T, nfft, hop, rate, ds, nd)), 3)
    print(f'spectrogram {nfft}/{hop}: {out["spectrogram"][f"{nfft}/{hop}"]}', file=sys.stderr, flush=True)
for sf in (1, 2, 3, 4):
    for se in (0, 1, 2):
        if se == 0:
            v = timed(lambda: hipdsp.sosfilt(ctx, bp[sf], dx, T, df, T, C, T, 0))
        else:
            v = timed(lambda: hipdsp.sosfilt_envelope(ctx, bp[sf], lp[se], dx, T, df, T, de, T, C, T, phase=1))
        out['filter'][f'{sf}+{se}'] = round(v, 3)
        print(f'filter {sf}+{se}: {v:.4f}', file=sys.stderr, flush=True)
for nfft, hop in sorted(FUSED_WINDOWS):
    nd = (T + hop - 1)//hop
    for sf in (1, 2, 3, 4):
        for se in (0, 1, 2):
            v = timed(lambda: hipdsp.chain_forward(ctx, bp[sf], lp[se], dx, T, df, T, C, T, nfft, hop, rate, ds, nd), n=3)
            out['fused'][f'{nfft}/{hop} {sf}+{se}'] = round(v, 3)
            sep = out['filter'][f'{sf}+{se}'] + out['spectrogram'][f'{nfft}/{hop}']
            print(f'fused {nfft}/{hop} {sf}+{se}: {v:.4f} against {sep:.4f} separate ({100*(v/sep - 1):+.1f} %)', file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
