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

FUSED = os.environ.get('FUSED', '1') != '0'      # hipdsp_chain_forward + backward sweep (2 launches)


def chain():
    plan.upload()
    if FUSED:
        hipdsp.chain_forward(ctx, plan, eplan, dx, T, df, T, C, T, nfft, hop, rate, ds, nd, db_out=db)
        hipdsp.sosfilt_envelope(ctx, plan, eplan, dx, T, df, T, de, T, C, T, phase=2)
    else:
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

# ---- the same loop with LIVE data: every frame 1/30 s of new 16-bit PCM arrives, the resident window
# slides by that much (shift into the other of two windows + hipdsp_pcm_unpack of the new chunk, both
# inside the captured graph) and the whole chain is recomputed under the moving cut-offs.  Two graphs,
# one per direction of the ping-pong, so that every kernel keeps fixed addresses.
chunk = int(rate/30)
win = [dx, hipdsp.DeviceArray(ctx, (C, T), np.float32)]
staging = hipdsp.DeviceArray(ctx, (chunk, C), np.int16)
rng = np.random.default_rng(5)
pcm = [(rng.standard_normal((chunk, C))*3000).astype(np.int16) for _ in range(8)]


def live(k):
    src, dst = win[k], win[1 - k]
    hipdsp.memcpy2d(ctx, dst, 4*T, src.view(chunk, (1,)), 4*T, 4*(T - chunk), C)          # slide
    hipdsp.pcm_unpack(ctx, staging, 2, chunk, C, 1.0/32768, dst.view(T - chunk, (1,)), T)   # append
    plan.upload()
    if FUSED:
        hipdsp.chain_forward(ctx, plan, eplan, dst, T, df, T, C, T, nfft, hop, rate, ds, nd, db_out=db)
        hipdsp.sosfilt_envelope(ctx, plan, eplan, dst, T, df, T, de, T, C, T, phase=2)
    else:
        hipdsp.sosfilt(ctx, plan, dst, T, df, T, C, T, 0)
        hipdsp.spectrogram(ctx, df, T, C, T, nfft, hop, rate, ds, nd, db_out=db)
        hipdsp.envelope(ctx, eplan, df, T, de, T, C, T, 0)


graphs = []
for k in (0, 1):
    live(k); ctx.synchronize()
    ctx.graph_begin(); live(k); graphs.append(ctx.graph_end())
ctx.synchronize()
t0 = time.perf_counter()
for i, (hp, lp) in enumerate(zip(hps, lps)):
    hipdsp.lib.hipdsp_memcpy_h2d(ctx.handle, hipdsp._p(staging), pcm[i % len(pcm)].ctypes.data, 2*chunk*C)
    plan.set_host(butter_sos(2, (hp, lp), 'bandpass', rate))
    ctx.graph_launch(graphs[i % 2])
    ctx.synchronize()
dt = time.perf_counter() - t0
print(f'live: {n} frames of {chunk} new samples per channel: {dt/n*1e3:.3f} ms per frame ({n/dt:.0f} FPS; '
      f'30 FPS needs 33.3 ms)', flush=True)
