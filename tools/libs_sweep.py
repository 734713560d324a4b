"""hipdsp_chain_forward of several BUILDS of the library in ONE process on the SAME device buffers, every listed window,
round-robin (boxes differ by 5 %, a box drifts by 1-2 % within minutes: only such a table compares builds).
    SHAPES=2048:1024,1024:256 python tools/libs_sweep.py a.so b.so ...      (ENV_=0: no envelope; ROUNDS, default 5)
Prints the median per build and window and the ratio to the FIRST build."""
import ctypes, os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
import numpy as np
from audian_amd import hipdsp, _lib
from audian_amd.design import butter_sos

libs = sys.argv[1:]
C, secs, rate = 64, float(os.environ.get('SECONDS_', '600')), 96000.0
shapes = [tuple(int(v) for v in p.split(':')) for p in os.environ.get('SHAPES', '2048:1024').split(',')]
T = int(secs*rate)
with_env = os.environ.get('ENV_', '1') == '1'
ctx = hipdsp.Context(0)
dx, df = (hipdsp.DeviceArray(ctx, (C, T), np.float32) for _ in range(2))
ds = hipdsp.DeviceArray(ctx, (max(C*((T + h - 1)//h)*(n//2 + 1) for n, h in shapes),), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 7)
ctx.synchronize()
sos, esos = butter_sos(2, (300.0, 3000.0), 'bandpass', rate), butter_sos(2, 20.0, 'lowpass', rate)
vp = ctypes.c_void_p
P = lambda a: vp(a.ptr)
builds = []
for path in libs:
    B = ctypes.CDLL(os.path.join(ROOT, path) if not os.path.isabs(path) else path)
    for name, (args, res) in _lib._SIGNATURES.items():
        fn = getattr(B, name); fn.argtypes = args; fn.restype = res

    def ok(rc, B=B):
        if rc:
            raise RuntimeError(B.hipdsp_last_error().decode())
    cb = vp(); ok(B.hipdsp_ctx_create(0, None, ctypes.byref(cb)))
    pf, pe = vp(), vp()
    for h, tab in ((pf, sos), (pe, esos)):
        ok(B.hipdsp_sosplan_create(cb, ctypes.byref(h)))
        tab = np.ascontiguousarray(tab, dtype=np.float64)
        ok(B.hipdsp_sosplan_set(cb, h, vp(tab.ctypes.data), len(tab)))
    e0, e1 = vp(), vp()
    ok(B.hipdsp_event_create(cb, ctypes.byref(e0))); ok(B.hipdsp_event_create(cb, ctypes.byref(e1)))
    builds.append((path, B, ok, cb, pf, pe if with_env else None, e0, e1))


def timed(b, n, h, reps=3):
    path, B, ok, cb, pf, pe, e0, e1 = b
    nd = (T + h - 1)//h
    f = lambda: ok(B.hipdsp_chain_forward(cb, pf, pe, P(dx), T, P(df), T, C, T, 1, np.pi/2, n, h, rate, P(ds), None, nd, 0, 0, 0, 0))
    f(); ok(B.hipdsp_ctx_synchronize(cb))
    ok(B.hipdsp_event_record(cb, e0))
    for _ in range(reps):
        f()
    ok(B.hipdsp_event_record(cb, e1)); ok(B.hipdsp_ctx_synchronize(cb))
    ms = ctypes.c_float()
    ok(B.hipdsp_event_elapsed_ms(cb, e0, e1, ctypes.byref(ms)))
    return ms.value/reps


rounds = int(os.environ.get('ROUNDS', '5'))
for n, h in shapes:
    res = {b[0]: [] for b in builds}
    for b in builds:
        timed(b, n, h, 1)
    for _ in range(rounds):
        for b in builds:
            res[b[0]].append(timed(b, n, h))
    base = np.median(res[builds[0][0]])
    print(f'{n}/{h}: ' + '   '.join(f'{os.path.basename(k)} {np.median(v):7.3f} ({100*(np.median(v)/base - 1):+5.1f} %)' for k, v in res.items()), flush=True)
