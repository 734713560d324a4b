"""The fused forward sweep against the separate launches (band-pass + envelope state sweep, then the
spectrogram) for every window it is built for and for band-pass plans of 2 and 4 sections, at the size of
BASELINE configs[2] (64 ch x 600 s x 96 kHz); interleaved rounds in one process."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos

C, rate = 64, 96000.0
T = int(float(os.environ.get('SECONDS_', '600'))*rate)
ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
df = hipdsp.DeviceArray(ctx, (C, T), np.float32)
de = hipdsp.DeviceArray(ctx, (C, T), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 1236)
eplan = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate))


def timed(f, n=4):
    f()
    ctx.record(e0)
    for _ in range(n):
        f()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1)/n


for order in (2, 4):
    fplan = hipdsp.SosPlan(ctx, butter_sos(order, (300.0, 3000.0), 'bandpass', rate))
    for nfft, hop in [(2048, 1024), (1024, 512), (1024, 256), (2048, 512), (512, 256), (256, 128)]:
        nd = (T + hop - 1)//hop
        F = nfft//2 + 1
        ds = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32)
        fused = lambda: hipdsp.chain_forward(ctx, fplan, eplan, dx, T, df, T, C, T, nfft, hop, rate, ds, nd)

        def separate():
            hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, de, T, C, T, phase=1)
            hipdsp.spectrogram(ctx, df, T, C, T, nfft, hop, rate, ds, nd)
        noenv = lambda: hipdsp.chain_forward(ctx, fplan, None, dx, T, df, T, C, T, nfft, hop, rate, ds, nd)
        fu, se, ne = [], [], []
        for rnd in range(3):
            fu.append(timed(fused))
            se.append(timed(separate))
            ne.append(timed(noenv))
        fu, se, ne = sorted(fu)[1], sorted(se)[1], sorted(ne)[1]
        gb = (8.0*C*T + 4.0*C*nd*F)/1e9
        line = (f'band-pass {order} sections, nfft {nfft} hop {hop}: fused {fu:7.3f} ms ({gb/fu*1e3:5.0f} GB/s of its '
                f'{gb:.1f} GB) | separate {se:7.3f} ms | fused/separate {fu/se:.2f} | no envelope (eplan NULL) {ne:7.3f} ms '
                f'({gb/ne*1e3:5.0f} GB/s)')
        if order == 2 and 4*C*nd*F < 40e9:
            db = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32)
            wdb = lambda: hipdsp.chain_forward(ctx, fplan, eplan, dx, T, df, T, C, T, nfft, hop, rate, ds, nd, db_out=db)
            t = sorted(timed(wdb) for _ in range(3))[1]
            line += f' | with the dB output {t:7.3f} ms ({(gb + 4.0*C*nd*F/1e9)/t*1e3:5.0f} GB/s)'
            db.free()
        print(line, flush=True)
        ds.free()
