"""Where does spec_pack_kernel's time go?  nfft / hop at 64 ch x 120 s x 96 kHz: the kernel as it is, the kernel it
replaced ("spec_kernel" 3 / 2) and its ablations ("spec_debug": 1 no global stores beyond the first frames, 2 no fetches,
4 no transform), all in one process on the same buffers.  Usage: [nfft hop] ..."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
pairs = [(int(a), int(b)) for a, b in zip(sys.argv[1::2], sys.argv[2::2])] or [(256, 128), (128, 64), (64, 32)]
C, T, rate = 64, 120*96000, 96000.0
ctx = hipdsp.Context(0)
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
e0, e1 = ctx.event(), ctx.event()
for nfft, hop in pairs:
    t = T if nfft >= 64 else T//8
    nd = (t + hop - 1)//hop
    F = nfft//2 + 1
    ds = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32)
    gb = (4.0*C*t + 4.0*C*nd*F)/1e9
    def run(label):
        for _ in range(2):
            hipdsp.spectrogram(ctx, dx, T, C, t, nfft, hop, rate, ds, nd)
        ctx.record(e0)
        for _ in range(5):
            hipdsp.spectrogram(ctx, dx, T, C, t, nfft, hop, rate, ds, nd)
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1)/5
        print(f'nfft {nfft:4d} hop {hop:4d} {label:44s} {ms:8.3f} ms {gb/ms*1e3:6.0f} GB/s', flush=True)
    for rep in range(2):
        run('spec_pack_kernel')
        for kern in ((3, 2) if nfft == 256 else (2,)):
            ctx.set_option('spec_kernel', kern)
            run('the kernel it replaced (spec_kernel %d)' % kern)
            ctx.set_option('spec_kernel', 0)
    for bits, label in ((1, 'no global stores'), (2, 'no fetches'), (4, 'no transform'), (3, 'no stores, no fetches'),
                        (6, 'no fetches, no transform'), (5, 'no stores, no transform'), (7, 'LDS traffic and loop only')):
        ctx.set_option('spec_debug', bits)
        run('spec_debug %d: %s' % (bits, label))
    ctx.set_option('spec_debug', 0)
    ds.free()
