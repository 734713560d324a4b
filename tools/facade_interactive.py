"""BASELINE configs[4] through the drop-in facade (not the C ABI): time DataBrowser.update_filter's
work -- BufferedFilter.update() -> recompute_all() through spectrogram and envelope -- under a
cut-off sweep, reading nothing back (no display).  Not a test."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd.bufferedfilter import BufferedFilter
from audian_amd.bufferedenvelope import BufferedEnvelope
from audian_amd.bufferedspectrogram import BufferedSpectrogram
from audian_amd.tracegraph import TraceGraph


class Item:
    def isVisible(self):
        return True


rate, C, seconds = 192000.0, 16, 260.0
rng = np.random.default_rng(1)
x = rng.uniform(-1, 1, size=(int(rate*seconds), C)).astype(np.float32)
g = TraceGraph(60.0, 20.0)
for t in (BufferedFilter(), BufferedSpectrogram(nfft=2048), BufferedEnvelope(envelope_cutoff=500.0)):
    g.add_trace(t)
g.setup_traces()
g.open(x, rate, view=os.environ.get('VIEW', '1') == '1')
for t in g.traces:
    t.plot_items = [Item() for _ in range(t.channels)]
g.set_need_update()
g.update_times(0.0, 10.0)
f = g['filtered']
print('resident frames', len(g.data.buffer), len(f.buffer), len(g['spectrogram'].buffer), len(g['envelope'].buffer), flush=True)
n = 30
t0 = time.perf_counter()
for i in range(n):
    f.highpass_cutoff = 100.0 + 60*i
    f.lowpass_cutoff = 20000.0 - 500*i
    f.update()
f.ctx.synchronize()
dt = (time.perf_counter() - t0)/n
print(f'facade update_filter: {dt*1e3:.2f} ms per recompute ({1/dt:.1f} FPS), no read-back', flush=True)
t0 = time.perf_counter()
for i in range(5):
    f.highpass_cutoff = 200.0 + 60*i
    f.update()
    img = g['spectrogram'].decibel_image(0)
    mm = f.minmax_decimate(f.offset, f.offset + len(f.buffer), len(f.buffer)//2000, channel=0)
dt = (time.perf_counter() - t0)/5
print(f'  + dB image of one channel + 2000 px min/max trace: {dt*1e3:.2f} ms', flush=True)
s = g['spectrogram']
t0 = time.perf_counter()
for i in range(5):
    f.highpass_cutoff = 200.0 + 60*i
    f.update()
    img = s.decimated_image(s.offset, s.offset + len(s.buffer), max(1, len(s.buffer)//2000), 0)
    mm = f.minmax_decimate(f.offset, f.offset + len(f.buffer), len(f.buffer)//2000, channel=0)
dt = (time.perf_counter() - t0)/5
print(f'  + the same with the image at screen resolution {img.shape}: {dt*1e3:.2f} ms', flush=True)

# scrolling: the visible window moves by 5 s per step; every trace keeps the overlapping part of its
# buffer and computes only what is new (plus the pre-roll the filters need)
g.update_times(60.0, 70.0)
f.ctx.synchronize()
steps = 16
off0 = g.data.offset
t0 = time.perf_counter()
for i in range(steps):
    g.update_times(65.0 + 5.0*i, 75.0 + 5.0*i)
f.ctx.synchronize()
dt = (time.perf_counter() - t0)/steps
print(f'scroll by 5 s: {dt*1e3:.2f} ms per step (raw buffer moved {(g.data.offset - off0)/rate:.0f} s in total)', flush=True)

# the same cut-off sweep in the MIDDLE of the recording: the filtered buffer starts at an arbitrary sample, the
# spectrogram's frame grid inside its first hop, the envelope one second later (pre-roll trimmed) -- still the fused
# launch (hipdsp_chain_forward's spec_first / env_first)
from audian_amd import hipdsp
g.update_times(130.0137, 140.0137)
f.update()
f.ctx.synchronize()
sp, en = g['spectrogram'], g['envelope']
before = dict(hipdsp.launches)
t0 = time.perf_counter()
for i in range(n):
    f.highpass_cutoff = 100.0 + 60*i
    f.lowpass_cutoff = 20000.0 - 500*i
    f.update()
f.ctx.synchronize()
dt_mid = (time.perf_counter() - t0)/n
per = {k: (v - before.get(k, 0))/n for k, v in hipdsp.launches.items() if v != before.get(k, 0)}
print(f'facade update_filter mid-recording (filtered offset {f.offset}, spectrogram first frame at sample '
      f'{sp._load_geometry(sp.offset, len(sp._hostbuf))[0]}, envelope at sample {en.offset - f.offset}): '
      f'{dt_mid*1e3:.2f} ms per recompute, launches per update {per}', flush=True)

# what audian's own plot items do, unchanged (specitem.py:36, traceitem.py:55-61): they slice one
# channel out of the buffer -- which now crosses PCIe as that one channel only
from audian_amd.bufferedspectrogram import decibel
f.highpass_cutoff = 333.0
f.update()
t0 = time.perf_counter()
img = decibel(s.buffer[:, 0, :].T)
t1 = time.perf_counter()
start, stop, step = f.offset, f.offset + len(f.buffer), len(f.buffer)//2000
seg = np.arange(0, stop - start, step)
col = f[start:stop, 0]
lo, hi = np.minimum.reduceat(col, seg), np.maximum.reduceat(col, seg)
t2 = time.perf_counter()
print(f'reference plot code on one channel: spectrogram slab + decibel {1e3*(t1 - t0):.1f} ms, '
      f'trace slice + reduceat {1e3*(t2 - t1):.1f} ms', flush=True)
