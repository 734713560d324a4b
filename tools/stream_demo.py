"""BASELINE configs[4]: streaming 16 ch x 192 kHz, resident window 80 s, live hp/lp sweep
over 300 hipGraph replays; prints ms per replay (target <= 33 ms = 30 FPS)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos

rate, C, seconds, nfft, hop = 192000.0, 16, 80.0, 2048, 1024
T = int(rate*seconds)
ctx = hipdsp.Context(0)
stream = ctx.create_stream()
ctx.set_stream(stream)
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 1238)
df = hipdsp.DeviceArray(ctx, (C, T), np.float32)
de = hipdsp.DeviceArray(ctx, (C, T), np.float32)
nd = (T + hop - 1)//hop
ds = hipdsp.DeviceArray(ctx, (C, nd, nfft//2 + 1), np.float32)
db = hipdsp.DeviceArray(ctx, (C, nd, nfft//2 + 1), np.float32)
plan = hipdsp.SosPlan(ctx, butter_sos(2, (100.0, 20000.0), 'bandpass', rate))
eplan = hipdsp.SosPlan(ctx, butter_sos(2, 500.0, 'lowpass', rate))

def chain():
    plan.upload()
    hipdsp.sosfilt(ctx, plan, dx, T, df, T, C, T, 0)
    hipdsp.spectrogram(ctx, df, T, C, T, nfft, hop, rate, ds, nd, db_out=db)
    hipdsp.envelope(ctx, eplan, df, T, de, T, C, T, 0)

chain(); ctx.synchronize()
ctx.graph_begin(); chain(); graph = ctx.graph_end()
n = 300
hps = np.linspace(100.0, 2000.0, n)
lps = np.linspace(20000.0, 4000.0, n)
ctx.synchronize()
t0 = time.perf_counter()
design = 0.0
for hp, lp in zip(hps, lps):
    d0 = time.perf_counter()
    plan.set_host(butter_sos(2, (hp, lp), 'bandpass', rate))
    design += time.perf_counter() - d0
    ctx.graph_launch(graph)
    ctx.synchronize()          # one frame per replay, as a display loop would
dt = time.perf_counter() - t0
print(f'{n} replays: {dt/n*1e3:.3f} ms per replay ({n/dt:.0f} FPS), of which host design+plan {design/n*1e3:.3f} ms; '
      f'{C*T/ (dt/n)/1e6:.0f} Msamples/s', flush=True)
