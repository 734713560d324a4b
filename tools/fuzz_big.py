"""One-off: 40 random multi-megasample slabs through the filter+envelope chain and the stand-alone
envelope (random designs, channel counts, skip, planned waves per CU) against the oracle; needs an MI355X."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.chdir(ROOT)
import numpy as np
from oracle import oracle
from audian_amd import hipdsp
from audian_amd.design import butter_sos
import gpu_helpers as gh
TILE = 2048
bad = 0
c = gh.ctx()
for seed in range(40):
    rng = np.random.default_rng(77000 + seed)
    rate = float(rng.choice([48000.0, 96000.0, 192000.0]))
    T = int(rng.integers(200, 2500))*TILE + int(rng.integers(-TILE, TILE))
    C = int(rng.choice([1, 2, 3, 7, 16, 33]))
    fs = butter_sos(int(rng.integers(1, 3)), (float(rng.uniform(50, 500)), float(rng.uniform(1000, 0.4*rate))), 'bandpass', rate)
    es = butter_sos(int(rng.integers(1, 5)), float(rng.uniform(5, 2000)), 'lowpass', rate)
    waves = int(rng.choice([8, 12, 16]))
    c.set_option('sos_waves_per_cu', waves)
    c.set_option('sos_waves_min', waves)      # exactly that many (round 3: the planner may pick fewer otherwise)
    x = rng.standard_normal((T, C)).astype(np.float32)
    dx = gh.to_planar(c, x)
    yf = hipdsp.DeviceArray(c, (C, T), np.float32)
    ye = hipdsp.DeviceArray(c, (C, T), np.float32)
    hipdsp.sosfilt_envelope(c, hipdsp.SosPlan(c, fs), hipdsp.SosPlan(c, es), dx, T, yf, T, ye, T, C, T)
    gf, ge = yf.to_host(), ye.to_host()
    skip = int(rng.integers(0, T//2))
    y2 = hipdsp.DeviceArray(c, (C, T - skip), np.float32)
    hipdsp.envelope(c, hipdsp.SosPlan(c, es), yf, T, y2, T - skip, C, T, skip)
    g2 = y2.to_host()
    chs = sorted(set([0, C//2, C - 1]))
    for ch in chs:
        wf = oracle.sosfilt(fs, x[:, ch].astype(np.float64))
        we = oracle.sosfiltfilt(es, (np.pi/2)*np.abs(gf[ch].astype(np.float64)))
        we[we < 0] = 0
        ef = np.max(np.abs(gf[ch] - wf))/np.max(np.abs(wf))
        ee = np.max(np.abs(ge[ch] - we))/max(np.max(np.abs(we)), 1e-30)
        e2 = np.max(np.abs(g2[ch] - we[skip:]))/max(np.max(np.abs(we)), 1e-30)
        if not (ef < 1e-4 and ee < 1e-4 and e2 < 1e-4):
            bad += 1
            print('FAIL', seed, T, C, ch, waves, ef, ee, e2, flush=True)
    print(seed, T, C, waves, 'ok', flush=True)
c.set_option('sos_waves_per_cu', 0); c.set_option('sos_waves_min', 0)
print('done, failures', bad)
