"""Spectrogram and envelope backward sweep on two streams (both read the filtered trace):
serial chain vs overlapped, configs[2] shape."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos

A = hipdsp.Context(0)
sA, sB = A.create_stream(), A.create_stream()
A.set_stream(sA)
B = hipdsp.Context(0, sB)
C, rate, nfft, hop = 64, 96000.0, 2048, 1024
T = int(float(os.environ.get('SECONDS_', '600'))*rate)
F, nd = nfft//2 + 1, (T - nfft)//hop + 1
dx = hipdsp.DeviceArray(A, (C, T), np.float32)
df = hipdsp.DeviceArray(A, (C, T), np.float32)
de = hipdsp.DeviceArray(A, (C, T), np.float32)
ds = hipdsp.DeviceArray(A, (C, nd, F), np.float32)
hipdsp.synth(A, dx, T, C, T, rate, 7)
fplanA = hipdsp.SosPlan(A, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
eplanA = hipdsp.SosPlan(A, butter_sos(2, 20.0, 'lowpass', rate))
eF, eS = A.event(), A.event()


def serial():
    hipdsp.sosfilt_envelope(A, fplanA, eplanA, dx, T, df, T, de, T, C, T, phase=1)
    hipdsp.spectrogram(A, df, T, C, T, nfft, hop, rate, ds, nd)
    hipdsp.sosfilt_envelope(A, fplanA, eplanA, dx, T, df, T, de, T, C, T, phase=2)


def overlapped():
    hipdsp.sosfilt_envelope(A, fplanA, eplanA, dx, T, df, T, de, T, C, T, phase=1)
    A.record(eF)
    B.wait_event(eF)
    hipdsp.spectrogram(B, df, T, C, T, nfft, hop, rate, ds, nd)
    B.record(eS)
    hipdsp.sosfilt_envelope(A, fplanA, eplanA, dx, T, df, T, de, T, C, T, phase=2)
    A.wait_event(eS)


for waves in [int(w) for w in os.environ.get('WAVES', '16,12,8').split(',')]:
    A.set_option('sos_waves_per_cu', waves)
    A.set_option('sos_waves_min', waves)      # exactly that many (round 3: the planner may pick fewer otherwise)
    for name, f in (('serial', serial), ('overlapped', overlapped), ('serial', serial), ('overlapped', overlapped)):
        for _ in range(3):
            f()
        A.synchronize(); B.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            f()
        A.synchronize(); B.synchronize()
        print(f'IIR waves/CU {waves:2d} {name:11s} {(time.perf_counter() - t0)*100:.3f} ms/step', flush=True)
sys.exit(0)

# ---- consecutive slabs pipelined: double-buffered filtered trace, three streams
sC = A.create_stream()
Cc = hipdsp.Context(0, sC)
df2 = [df, hipdsp.DeviceArray(A, (C, T), np.float32)]
eplanC = hipdsp.SosPlan(Cc, butter_sos(2, 20.0, 'lowpass', rate))
fplanC = hipdsp.SosPlan(Cc, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
evF = [A.event(), A.event()]
evS = [A.event(), A.event()]
evB = [A.event(), A.event()]
# the checkpoint scratch belongs to a context: the forward sweep must run on the context whose
# backward sweep consumes it, so both sweeps stay on A/Cc alternately
ctxs = [A, Cc]
plans = [(fplanA, eplanA), (fplanC, eplanC)]
count = [0]


def pipelined():
    i = count[0] % 2
    count[0] += 1
    X = ctxs[i]
    fp, ep = plans[i]
    # slab i: forward on its context (waits until the readers of this df buffer two slabs ago are done)
    if count[0] > 2:
        X.wait_event(evS[i])
    hipdsp.sosfilt_envelope(X, fp, ep, dx, T, df2[i], T, de, T, C, T, phase=1)
    X.record(evF[i])
    B.wait_event(evF[i])
    hipdsp.spectrogram(B, df2[i], T, C, T, nfft, hop, rate, ds, nd)
    B.record(evS[i])
    hipdsp.sosfilt_envelope(X, fp, ep, dx, T, df2[i], T, de, T, C, T, phase=2)


for name, f in (('pipelined', pipelined), ('pipelined', pipelined)):
    for _ in range(4):
        f()
    for c in (A, B, Cc):
        c.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        f()
    for c in (A, B, Cc):
        c.synchronize()
    print(f'{name:11s} {(time.perf_counter() - t0)*100:.3f} ms/step', flush=True)
