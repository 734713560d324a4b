"""Soak: the configs[2] step (fused forward sweep + backward sweep) over and over for a few minutes on one set of
buffers; every 250 steps three windows of every output are compared with what step 0 produced (the kernels have no
atomics in the data path: bit-identical or something is wrong) and the context's fault word is read.
    python tools/soak.py [minutes=3]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
C, rate, nfft, hop = 64, 96000.0, 2048, 1024
T = int(600*rate)
F, nd = nfft//2 + 1, (T + hop - 1)//hop
ctx = hipdsp.Context(0)
dx, df, de = (hipdsp.DeviceArray(ctx, (C, T), np.float32) for _ in range(3))
ds, db = (hipdsp.DeviceArray(ctx, (C, nd, F), np.float32) for _ in range(2))
hipdsp.synth(ctx, dx, T, C, T, rate, 1236)
fplan = hipdsp.SosPlan(ctx, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
eplan = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate))


def step():
    hipdsp.chain_forward(ctx, fplan, eplan, dx, T, df, T, C, T, nfft, hop, rate, ds, nd, db_out=db)
    hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, de, T, C, T, phase=2)


def sample():
    out = []
    for ch, n0 in ((0, 0), (C//2, T//2 - 777), (C - 1, T - 50000)):
        out.append(df.view(ch*T + n0, (50000,)).to_host())
        out.append(de.view(ch*T + n0, (50000,)).to_host())
        k0 = n0//hop
        out.append(ds.view((ch*nd + k0)*F, (40*F,)).to_host())
        out.append(db.view((ch*nd + k0)*F, (40*F,)).to_host())
    return out


step(); ctx.synchronize()
ref = sample()
t0, n, checks = time.time(), 0, 0
last = t0
while time.time() - t0 < 60*minutes:
    for _ in range(250):
        step()
    n += 250
    ctx.synchronize()                               # raises if a kernel left a fault word
    got = sample()
    checks += 1
    for i, (a, b) in enumerate(zip(got, ref)):
        if not np.array_equal(a, b, equal_nan=True):
            print(f'MISMATCH after {n} steps in sample {i}: {int((a != b).sum())} values differ', flush=True)
            sys.exit(1)
    if time.time() - last > 30:
        last = time.time()
        print(f'{n} steps, {checks} checks, {(time.time() - t0)/n*1e3:.2f} ms per step incl. checks', flush=True)
print(f'done: {n} steps in {time.time() - t0:.0f} s, {checks} checks of 12 windows each, all bit-identical to step 0, no fault')
