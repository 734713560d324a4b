"""hipdsp_chain_forward alone, CALLS times, for one window (NFFT / HOP) at BASELINE configs[2]'s size -- the profiling target of
tools/pmc_fwd_libs.sh (counters per build: AUDIAN_AMD_LIB selects the library).  ENV=0: no envelope behind the filter."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos

C, rate = 64, 96000.0
nfft, hop = int(os.environ.get('NFFT', '2048')), int(os.environ.get('HOP', '1024'))
T = int(float(os.environ.get('SECONDS_', '600'))*rate)
nd, F = (T + hop - 1)//hop, nfft//2 + 1
ctx = hipdsp.Context(0)
e0, e1 = ctx.event(), ctx.event()
dx, df = (hipdsp.DeviceArray(ctx, (C, T), np.float32) for _ in range(2))
ds = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 1236)
fplan = hipdsp.SosPlan(ctx, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
eplan = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate)) if os.environ.get('ENV', '1') == '1' else None
n = int(os.environ.get('CALLS', '6'))
fwd = lambda: hipdsp.chain_forward(ctx, fplan, eplan, dx, T, df, T, C, T, nfft, hop, rate, ds, nd)
fwd(); fwd()
ctx.record(e0)
for _ in range(n):
    fwd()
ctx.record(e1)
print(f'chain_forward {nfft}/{hop}: {ctx.elapsed_ms(e0, e1)/n:.3f} ms ({os.environ.get("AUDIAN_AMD_LIB", "tree build")})', flush=True)
