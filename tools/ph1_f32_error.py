"""What would float32 dot products in PHASE 1 of the block-parallel cascade cost in accuracy?  (CPU, numpy)
Phase 1 (sos_cascade.inc) forms, per lane, the zero-state end state of its 32 samples, f = sum_j G[j] x[j];
in float64 it costs a conversion and 2 n_sections multiply-adds per sample, a third of the sweeps' VALU time.
With a float32 table and float32 accumulation the state that enters the next lane is off by
delta = fl32(sum) - sum; the recursion itself stays float64.  This script injects exactly those deltas at the
lane borders of an otherwise exact float64 filter and reports the output error relative to max |y| -- the
north_star allows 1e-4, the float64 path delivers 6e-8 (the rounding of the float32 output).
"""
import sys
import numpy as np
from scipy.signal import sosfilt, butter
L = 32
def tables(sos):
    S = len(sos); D = 2*S
    def step(z, x):
        y, zo = sosfilt(sos, np.array([x], float), zi=z.reshape(S, 2))
        return zo.ravel()
    B = step(np.zeros(D), 1.0)
    A = np.stack([step(np.eye(D)[c], 0.0) for c in range(D)], axis=1)
    G = np.zeros((L, D)); P = np.eye(D)
    for j in range(L - 1, -1, -1):
        G[j] = P @ B; P = P @ A
    return G
def inject(sos, x, split=False):
    """output error caused by float32 phase-1 sums (max over samples), x float32"""
    S = len(sos); D = 2*S
    G = tables(sos)
    n = len(x)//L*L
    X = x[:n].reshape(-1, L)
    exact = X.astype(np.float64) @ G                          # (blocks, D)
    G32 = G.astype(np.float32)
    acc = np.zeros((X.shape[0], D), np.float32)
    for j in range(L):                                         # fused multiply-add: one rounding per term
        acc = (acc.astype(np.float64) + G32[j].astype(np.float64)[None, :]*X[:, j].astype(np.float64)[:, None]).astype(np.float32)
    delta = acc.astype(np.float64) - exact
    e = np.zeros(n); s = np.zeros(D)
    zeros = np.zeros(L)
    for b in range(X.shape[0]):
        eb, so = sosfilt(sos, zeros, zi=s.reshape(S, 2))
        e[b*L:(b + 1)*L] = eb
        s = so.ravel() + delta[b]
    return e
rate = 96000.0
rng = np.random.default_rng(1)
N = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
t = np.arange(N)/rate
signals = {
    'white noise': rng.standard_normal(N),
    'tones 440 + 1200 + 7000 Hz + noise': np.sin(2*np.pi*440*t) + 0.5*np.sin(2*np.pi*1200*t) + 0.3*np.sin(2*np.pi*7000*t) + 0.1*rng.standard_normal(N),
    'DC 1.0 + 1e-3 tone at 1 kHz': 1.0 + 1e-3*np.sin(2*np.pi*1000*t),
    '50 Hz hum 1.0 + 1e-3 tone at 1 kHz': np.sin(2*np.pi*50*t) + 1e-3*np.sin(2*np.pi*1000*t),
    '20 kHz 1.0 + 1e-3 tone at 1 kHz': np.sin(2*np.pi*20000*t) + 1e-3*np.sin(2*np.pi*1000*t),
}
# a rectified trace whose level steps by 1000 sigma (the envelope's memory then holds the large level while the input is small)
step = 1e-3*rng.standard_normal(N)
step[N//2:N//2 + N//8] *= 1000.0
signals['noise with a 1000 sigma burst'] = step
bp = butter(2, (300.0, 3000.0), 'bandpass', fs=rate, output='sos')
lp = butter(2, 20.0, 'lowpass', fs=rate, output='sos')
for name, x in signals.items():
    x = x.astype(np.float32)
    y = sosfilt(bp, x.astype(np.float64))
    e = inject(bp, x)
    skip = N//4
    ymax = np.abs(y[skip:]).max()
    r = (np.pi/2*np.abs(y.astype(np.float32))).astype(np.float32)
    env = sosfilt(lp, r.astype(np.float64))
    ee = inject(lp, r)
    print(f'{name:38s} band-pass: max|y| {ymax:.3e}, error {np.abs(e[skip:len(e)]).max()/ymax:.2e} of it | '
          f'envelope low-pass: max {np.abs(env[skip:]).max():.3e}, error {np.abs(ee[skip:]).max()/np.abs(env[skip:]).max():.2e} of it')

# other envelope cut-offs (the reference's spin box: databrowser.py:545-566), same signals, raw trace rectified
for fc in (5.0, 500.0):
    lpc = butter(2, fc, 'lowpass', fs=rate, output='sos')
    for name, x in signals.items():
        r = (np.pi/2*np.abs(x.astype(np.float32))).astype(np.float32)
        env = sosfilt(lpc, r.astype(np.float64))
        ee = inject(lpc, r)
        skip = N//4
        print(f'{name:38s} envelope low-pass {fc:5.0f} Hz of |x|: max {np.abs(env[skip:]).max():.3e}, error {np.abs(ee[skip:]).max()/np.abs(env[skip:]).max():.2e} of it, '
              f'worst error relative to the LOCAL envelope {np.max(np.abs(ee[skip:])/np.maximum(np.abs(env[skip:len(ee)]), 1e-300)):.2e}')
