"""Time the two envelope sweeps of the batch chain (configs[2] shape) for a range of resident
waves per CU."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos

ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()
C, rate = 64, 96000.0
T = int(float(os.environ.get('SECONDS_', '600'))*rate)
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
dy = hipdsp.DeviceArray(ctx, (C, T), np.float32)
de = hipdsp.DeviceArray(ctx, (C, T), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
fplan = hipdsp.SosPlan(ctx, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
eplan = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate))


def timed(f, n=5):
    f(); f()
    ctx.record(e0)
    for _ in range(n):
        f()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1)/n


waves = [int(w) for w in os.environ.get('WAVES', '12,16,20').split(',')]
for pf in [int(v) for v in os.environ.get('PREFETCH', '0').split(',')]:
  ctx.set_option('sos_prefetch', pf)
  print('prefetch', pf)
  for w in waves:
      ctx.set_option('sos_waves_per_cu', w)
      ctx.set_option('sos_waves_min', w)      # exactly that many (round 3: the planner may pick fewer otherwise)
      fw = timed(lambda: hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, dy, T, de, T, C, T, phase=1))
      bw = timed(lambda: hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, dy, T, de, T, C, T, phase=2))
      en = timed(lambda: hipdsp.envelope(ctx, eplan, dx, T, de, T, C, T, 0))
      print(f'waves/CU {w:2d}: band-pass + state sweep {fw:.3f} ms  backward sweep {bw:.3f} ms  '
            f'sum {fw + bw:.3f} ms | envelope alone {en:.3f} ms', flush=True)
