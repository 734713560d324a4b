"""hipdsp_spectrogram with the default kernel of a window and with the alternatives behind the option "spec_kernel", in ONE
process on the SAME buffers, round-robin (separate processes differ by 5-12 % with where their buffers land):
    PAIRS=512:256,512:100,1024:700 KERNELS=0,3 python tools/spec_kernel_ab.py
64 ch x SECONDS_ (120) s x 96 kHz; the fastest of ROUNDS (5) rounds of TIMED_CALLS (8) calls counts."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audian_amd import hipdsp
C, T, rate = 64, int(float(os.environ.get('SECONDS_', '120'))*96000), 96000.0
pairs = [tuple(int(v) for v in p.split(':')) for p in os.environ.get('PAIRS', '512:256').split(',')]
kernels = [int(k) for k in os.environ.get('KERNELS', '0,3').split(',')]
rounds, ncalls = int(os.environ.get('ROUNDS', '5')), int(os.environ.get('TIMED_CALLS', '8'))
ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()
nbins = max(C*((T + h - 1)//h)*(n//2 + 1) for n, h in pairs)
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
ds, db = (hipdsp.DeviceArray(ctx, (nbins,), np.float32) for _ in range(2))
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
for _ in range(60):
    hipdsp.spectrogram(ctx, dx, T, C, T, 2048, 1024, rate, ds, (T + 1023)//1024)
ctx.synchronize()
for nfft, hop in pairs:
    nd = (T + hop - 1)//hop
    for want_db in (False, True):
        best = {k: 1e30 for k in kernels}
        for _ in range(rounds):
            for k in kernels:
                ctx.set_option('spec_kernel', k)
                for _ in range(2):
                    hipdsp.spectrogram(ctx, dx, T, C, T, nfft, hop, rate, ds, nd, db_out=db if want_db else None)
                ctx.record(e0)
                for _ in range(ncalls):
                    hipdsp.spectrogram(ctx, dx, T, C, T, nfft, hop, rate, ds, nd, db_out=db if want_db else None)
                ctx.record(e1)
                ctx.synchronize()
                best[k] = min(best[k], ctx.elapsed_ms(e0, e1)/ncalls)
        ctx.set_option('spec_kernel', 0)
        print(f'nfft {nfft:6d} hop {hop:6d} {"PSD+dB" if want_db else "PSD   "}: ' +
              '   '.join(f'spec_kernel {k}: {best[k]:8.3f} ms' for k in kernels), flush=True)
