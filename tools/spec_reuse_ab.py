"""A/B in one process: the spectrogram kernel with and without register reuse of the overlapped part of
consecutive frames ("spec_no_half"), 64 ch x 120 s x 96 kHz, for the 50 % and 75 % overlaps."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp

C, T, rate = 64, int(120*96000), 96000.0
ctx = hipdsp.Context(0)
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
e0, e1 = ctx.event(), ctx.event()


def run(nfft, hop, db=False):
    nd = (T + hop - 1)//hop
    F = nfft//2 + 1
    ds = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32)
    dd = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32) if db else None
    res = {0: [], 1: []}
    for rnd in range(4):
        for no in (0, 1):
            ctx.set_option('spec_no_half', no)
            hipdsp.spectrogram(ctx, dx, T, C, T, nfft, hop, rate, ds, nd, db_out=dd)
            ctx.record(e0)
            for _ in range(5):
                hipdsp.spectrogram(ctx, dx, T, C, T, nfft, hop, rate, ds, nd, db_out=dd)
            ctx.record(e1)
            res[no].append(ctx.elapsed_ms(e0, e1)/5)
    ctx.set_option('spec_no_half', 0)
    gb = (4.0*C*T + (8.0 if db else 4.0)*C*nd*F)/1e9
    a, b = sorted(res[0])[len(res[0])//2], sorted(res[1])[len(res[1])//2]
    print(f'nfft {nfft:5d} hop {hop:5d}{" +dB" if db else "    "}: reuse {a:7.3f} ms {gb/a*1e3:5.0f} GB/s | every frame fetched whole {b:7.3f} ms {gb/b*1e3:5.0f} GB/s', flush=True)
    ds.free()
    if dd is not None:
        dd.free()


for nfft, hop in [(1024, 256), (2048, 512), (4096, 1024), (1024, 512), (2048, 1024), (4096, 2048)]:
    run(nfft, hop)
run(1024, 256, db=True)
