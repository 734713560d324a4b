"""A/B of two BUILDS of libhip_dsp in ONE process on the SAME device buffers (the backward sweep's time moves by 10 %
from process to process, more than most variants are worth): the default library through audian_amd.hipdsp, a second
build (tools/_ab/libnew.so, or argv[1]) through a bare ctypes handle with the same signature table.
    python tools/two_libs_ab.py [other.so] [channels] [seconds]"""
import ctypes, os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
import numpy as np
from audian_amd import hipdsp, _lib
from audian_amd.design import butter_sos

other = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'tools', '_ab', 'libnew.so')
C = int(sys.argv[2]) if len(sys.argv) > 2 else 64
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 600.0
rate = 96000.0
# SHAPES="2048:1024,1024:256,256:128": the forward sweep of every listed window, one table per shape (the backward sweep
# is timed with the first); default: the headline window only
shapes = [tuple(int(v) for v in p.split(':')) for p in os.environ.get('SHAPES', '2048:1024').split(',')]
nfft, hop = shapes[0]
T = int(secs*rate)
nd = max((T + h - 1)//h for _, h in shapes)
B = ctypes.CDLL(other)
for name, (args, res) in _lib._SIGNATURES.items():
    fn = getattr(B, name); fn.argtypes = args; fn.restype = res


def okB(rc):
    if rc:
        raise RuntimeError(B.hipdsp_last_error().decode())


ctx = hipdsp.Context(0)
sos, esos = butter_sos(2, (300.0, 3000.0), 'bandpass', rate), butter_sos(2, 20.0, 'lowpass', rate)
fplan, eplan = hipdsp.SosPlan(ctx, sos), hipdsp.SosPlan(ctx, esos)
dx, df, de, de2 = (hipdsp.DeviceArray(ctx, (C, T), np.float32) for _ in range(4))
ds = hipdsp.DeviceArray(ctx, (max(C*((T + h - 1)//h)*(n//2 + 1) for n, h in shapes),), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
ctx.synchronize()
vp = ctypes.c_void_p
cb = vp(); okB(B.hipdsp_ctx_create(0, None, ctypes.byref(cb)))
pf, pe = vp(), vp()
for h, tab in ((pf, sos), (pe, esos)):
    okB(B.hipdsp_sosplan_create(cb, ctypes.byref(h)))
    tab = np.ascontiguousarray(tab, dtype=np.float64)
    okB(B.hipdsp_sosplan_set(cb, h, vp(tab.ctypes.data), len(tab)))
P = lambda a: vp(a.ptr)
nds = lambda h: (T + h - 1)//h
fwdA = lambda n=nfft, h=hop: hipdsp.chain_forward(ctx, fplan, eplan, dx, T, df, T, C, T, n, h, rate, ds, nds(h))
bwdA = lambda: hipdsp.sosfilt_envelope(ctx, fplan, eplan, dx, T, df, T, de, T, C, T, phase=2)
fwdB = lambda n=nfft, h=hop: okB(B.hipdsp_chain_forward(cb, pf, pe, P(dx), T, P(df), T, C, T, 1, np.pi/2, n, h, rate, P(ds), None, nds(h), 0, 0, 0, 0))     # (a build of ABI 101 ignores the two trailing arguments)
# SAME_OUT=1: both builds write the same envelope buffer (where a buffer lies in HBM moves the sweep by several per cent,
# so two output buffers confound the comparison); the identity check then compares a copy taken after A's last run
same_out = os.environ.get('SAME_OUT', '0') == '1'
bwdB = lambda: okB(B.hipdsp_sosfilt_envelope(cb, pf, pe, P(dx), T, P(df), T, P(de if same_out else de2), T, C, T, 1, np.pi/2, 1, 2, 0))
e0, e1 = ctx.event(), ctx.event()


def timed(fn, sync, n=3):
    fn(); sync()
    ctx.record(e0)
    for _ in range(n):
        fn()
    ctx.record(e1); sync(); ctx.synchronize()
    return ctx.elapsed_ms(e0, e1)/n


syncB = lambda: okB(B.hipdsp_ctx_synchronize(cb))
fwdA(); ctx.synchronize(); fwdB(); syncB()          # each context's scratch holds its own tile states
res = {k: [] for k in ('fwd A', 'fwd B', 'bwd A', 'bwd B')}
for rnd in range(6):
    res['fwd A'].append(timed(fwdA, ctx.synchronize)); res['bwd A'].append(timed(bwdA, ctx.synchronize))
    res['fwd B'].append(timed(fwdB, syncB)); res['bwd B'].append(timed(bwdB, syncB))
bwdA(); ctx.synchronize(); headA = de.view(0, (min(T, 2000000),)).to_host().copy()
bwdB(); syncB(); headB = (de if same_out else de2).view(0, (min(T, 2000000),)).to_host()
same = np.array_equal(headA, headB)
print('largest difference of the two envelopes, relative to the largest value:', float(np.abs(headA - headB).max()/np.abs(headA).max()))
for k, v in res.items():
    print(f'{k}: median {np.median(v):7.3f} ms  {[round(x, 3) for x in v]}')
print('A = audian_amd/libhip_dsp.so, B =', other, '| envelopes identical:', same)
for n, h in shapes[1:]:
    ra, rb = [], []
    fwdA(n, h); ctx.synchronize(); fwdB(n, h); syncB()
    for rnd in range(6):
        ra.append(timed(lambda: fwdA(n, h), ctx.synchronize)); rb.append(timed(lambda: fwdB(n, h), syncB))
    print(f'fwd {n}/{h} A: median {np.median(ra):7.3f} ms  {[round(x, 3) for x in ra]}')
    print(f'fwd {n}/{h} B: median {np.median(rb):7.3f} ms  {[round(x, 3) for x in rb]}')
