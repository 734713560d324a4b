"""Forward half of the batch chain at configs[2]: fused hipdsp_chain_forward against
hipdsp_sosfilt_envelope(phase 1) + hipdsp_spectrogram, and the whole step with the backward sweep."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos

C, rate, nfft, hop = 64, 96000.0, 2048, 1024
T = int(float(os.environ.get('SECONDS_', '600'))*rate)
nd = (T + hop - 1)//hop
F = nfft//2 + 1
ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
df = hipdsp.DeviceArray(ctx, (C, T), np.float32)
de = hipdsp.DeviceArray(ctx, (C, T), np.float32)
ds = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 1236)
fplan = hipdsp.SosPlan(ctx, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
eplan = hipdsp.SosPlan(ctx, butter_sos(int(os.environ.get('ENV_ORDER', '2')), 20.0, 'lowpass', rate))


def timed(f, n=5):
    f(); f()
    ctx.record(e0)
    for _ in range(n):
        f()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1)/n


def separate():
    hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, de, T, C, T, phase=1)
    hipdsp.spectrogram(ctx, df, T, C, T, nfft, hop, rate, ds, nd)


def fused():
    hipdsp.chain_forward(ctx, fplan, eplan, dx, T, df, T, C, T, nfft, hop, rate, ds, nd)


def bwd():
    hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, de, T, C, T, phase=2)


if os.environ.get('ABLATE'):
    for bits in [int(b) for b in os.environ['ABLATE'].split(',')]:
        ctx.set_option('chain_debug', bits)
        print(f'fused, debug bits {bits} (1: FFT waves only copy, 2: IIR waves without cascades): {timed(fused):.3f} ms', flush=True)
ctx.set_option('chain_debug', int(os.environ.get('CHAIN_DEBUG', '0')))
if os.environ.get('CLOCK'):
    ctx.set_option('chain_debug', 16)
    fused(); fused(); ctx.synchronize()
    raw = ds.view(0, (4,)).to_host().view(np.int64)
    print(f'engine clock during the fused kernel: {raw[0]/(raw[1]/100e6)/1e6:.0f} MHz '
          f'({raw[0]} shader clocks in {raw[1]/100e3:.3f} ms)', flush=True)
    ctx.set_option('chain_debug', 0)
s = timed(separate)
print(f'separate forward sweep + spectrogram: {s:.3f} ms', flush=True)
f = timed(fused)
print(f'fused forward sweep:                  {f:.3f} ms  ({12.0*C*T/f/1e6:.0f} GB/s algorithmic)', flush=True)
b = timed(bwd)
print(f'backward sweep:                       {b:.3f} ms', flush=True)
w = timed(lambda: (fused(), bwd()))
print(f'fused step (one stream):              {w:.3f} ms = {C*T/w/1e3:.0f} Msamples/s', flush=True)
w = timed(lambda: (separate(), bwd()))
print(f'separate step (one stream):           {w:.3f} ms = {C*T/w/1e3:.0f} Msamples/s', flush=True)
