"""Measurement of the SURVEY 8f kernels (not a test): algorithmic GB/s on configs[2]-size data."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp

ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()
res = {}

def timeit(name, f, nbytes, reps=5):
    f(); f()
    ctx.record(e0)
    for _ in range(reps):
        f()
    ctx.record(e1)
    ms = ctx.elapsed_ms(e0, e1)/reps
    res[name] = {'ms': round(ms, 4), 'GBps': round(nbytes/ms/1e6, 1), 'bytes': nbytes}
    print(f'{name:46s} {ms:9.4f} ms  {nbytes/ms/1e6:8.1f} GB/s', flush=True)

C, rate = 64, 96000.0
T = int(600*rate)
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 5)
# overview of the whole recording at 2000 px, and a 10 s window at 2000 px
for label, start, stop in (('minmax whole 600 s -> 2000 px', 0, T), ('minmax 10 s window -> 2000 px', T//2, T//2 + int(10*rate))):
    step = max(1, (stop - start)//2000)
    nseg = (stop - start + step - 1)//step
    out = hipdsp.DeviceArray(ctx, (C, 2*nseg), np.float32)
    timeit(f'{label} (step {step}, 64 ch)', lambda: hipdsp.minmax_decimate(ctx, dx, T, C, start, stop, step, out, 2*nseg),
           4.0*C*(stop - start))
for step in (4, 32):
    stop = int(60*rate)
    nseg = (stop + step - 1)//step
    out = hipdsp.DeviceArray(ctx, (C, 2*nseg), np.float32)
    timeit(f'minmax 60 s step {step} (64 ch)', lambda: hipdsp.minmax_decimate(ctx, dx, T, C, 0, stop, step, out, 2*nseg),
           4.0*C*stop + 8.0*C*nseg)
# spectrogram reductions on an 80 s tile of one channel (7500 x 1025)
frames, F = 7500, 1025
spec = hipdsp.DeviceArray(ctx, (frames, F), np.float32)
hipdsp.synth(ctx, spec, frames*F, 1, frames*F, rate, 6)
o = hipdsp.DeviceArray(ctx, (F,), np.float32)
timeit('mean spectrum dB 7500 x 1025', lambda: hipdsp.mean_spectrum_db(ctx, spec, F, 0, frames, o), 4.0*frames*F)
top = hipdsp.DeviceArray(ctx, (1,), np.float32)
timeit('max reduction 7500 x 1025', lambda: hipdsp.max_nonneg(ctx, spec, frames*F, top), 4.0*frames*F)
img = hipdsp.DeviceArray(ctx, (F, frames), np.float32)
timeit('decibel image 7500 x 1025 -> (F, T)', lambda: hipdsp.decibel_image(ctx, spec, img, frames, F), 8.0*frames*F)
for step in (4, 28):
    ncols = (frames + step - 1)//step
    small = hipdsp.DeviceArray(ctx, (F, ncols), np.float32)
    timeit(f'decibel image 7500 x 1025 -> (F, {ncols}), max over {step} frames',
           lambda: hipdsp.decibel_image_decimate(ctx, spec, small, frames, F, 0, frames, step), 4.0*frames*F + 4.0*ncols*F)
# PCM ingest: 64 ch x 60 s int16
Tp = int(60*rate)
pcm = hipdsp.DeviceArray(ctx, (Tp, C), np.int16)
dst = hipdsp.DeviceArray(ctx, (C, Tp), np.float32)
timeit('pcm int16 -> planar f32, 64 ch x 60 s', lambda: hipdsp.pcm_unpack(ctx, pcm, 2, Tp, C, 1/32768.0, dst, Tp), 6.0*C*Tp)
f64 = hipdsp.DeviceArray(ctx, (Tp, C), np.float64)
timeit('pack f64 (T,C) -> planar f32, 64 ch x 60 s', lambda: hipdsp.pack(ctx, f64, dst, Tp, Tp, C), 12.0*C*Tp)
timeit('unpack planar f32 -> f64 (T,C), 64 ch x 60 s', lambda: hipdsp.unpack(ctx, dst, Tp, f64, Tp, C), 12.0*C*Tp)
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'next_rows_bench.json'), 'w'), indent=1)
