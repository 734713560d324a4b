"""Host logic of the BufferedData facade (no GPU): bookkeeping, index transforms
incl. the reference's seconds/rate quirk, ring-buffer recycling, dirty propagation.
Expected values are derived by hand from src/audian/buffereddata.py:33-153 and
src/audian/data.py:150-231 of the reference."""

from math import ceil, floor

import numpy as np
import pytest

from audian_amd.bufferedarray import ArrayLoader
from audian_amd.buffereddata import BufferedData, _merge, _covers
from audian_amd.bufferedspectrogram import BufferedSpectrogram, decibel
from audian_amd.tracegraph import TraceGraph


class Item:
    def __init__(self, visible=True):
        self.visible = visible

    def isVisible(self):
        return self.visible

    def setVisible(self, show):
        self.visible = show


class Doubler(BufferedData):
    """Stateless stand-in for a derived trace: dest = 2*source[nbefore:]."""

    def __init__(self, name='double', source='data', tbefore=0, tafter=0, step=1):
        super().__init__(name, source, tbefore=tbefore, tafter=tafter)
        self.step = step
        self.calls = []

    def open(self, source):
        super().open(source, self.step)

    def process(self, source, dest, nbefore):
        self.calls.append((len(source), len(dest), nbefore, self._pending.soffset,
                           self._pending.doffset))
        n = len(dest)
        dest[:] = 2*source[nbefore:nbefore + n*self.step:self.step][:n]


def ramp(frames, channels=2):
    return np.arange(frames, dtype=np.float64)[:, None] + 0.25*np.arange(channels)[None, :]


def make_graph(trace, frames=4000, rate=100.0, buffer_time=10.0, back_time=2.0):
    g = TraceGraph(buffer_time, back_time)
    g.add_trace(trace)
    g.setup_traces()
    g.open(ramp(frames), rate)
    return g


def test_interval_helpers():
    assert _merge([[5, 7], [0, 2], [2, 4], [6, 9]]) == [[0, 4], [5, 9]]
    assert _covers([[0, 4], [5, 9]], 1, 3) and not _covers([[0, 4], [5, 9]], 3, 6)


def test_array_loader_and_getitem():
    a = ArrayLoader(ramp(1000), 100.0, buffer_time=2.0, back_time=0.5)
    assert len(a) == 1000 and a.bufferframes == 200 and a.backframes == 50
    assert a.offset == 0 and len(a.buffer) == 200
    assert np.array_equal(a[10:20, 1], ramp(1000)[10:20, 1])
    assert a.offset == 0
    assert np.array_equal(a[500:520, 0], ramp(1000)[500:520, 0])     # moves the buffer
    assert a.offset == 450 and len(a.buffer) == 200
    assert a[999, 0] == 999.0
    assert a.offset + len(a.buffer) == 1000
    with pytest.raises(IndexError):
        a[1000]


def test_open_and_update_step():
    t = Doubler(tbefore=10)
    g = make_graph(t)
    assert (t.rate, t.frames, t.shape, t.channels) == (100.0, 4000, (4000, 2), 2)
    assert t.source is g.data and g.data.dests == [t]
    assert t.unit == g.data.unit and t.ampl_max == g.data.ampl_max
    # data buffer: buffer_time + tbefore (10) -> 20 s, back_time 2 + 10
    assert (g.tbefore, g.tafter) == (10, 0)
    assert g.data.bufferframes == 2000 and g.data.backframes == 1200
    s = Doubler(step=8)
    g = make_graph(s)
    assert s.rate == 100.0/8 and s.frames == 500 and s.shape == (500, 2)
    assert s.bufferframes == int((0/100.0)*s.rate)           # bufferframes was 0 before open


def test_load_buffer_quirk_and_align():
    """tbefore/tafter are divided by the rate in load_buffer (buffereddata.py:96,99):
    nbefore = floor(10/100) = 0, nafter = ceil(10/100) = 1."""
    t = Doubler(tbefore=10, tafter=10)
    g = make_graph(t)
    t.plot_items = [Item(), None]
    g.set_need_update()
    assert t.need_update and g.data.need_update      # propagated up from the visible leaf
    g.update_times(0.0, 5.0)
    # data buffer starts at 0 -> no front trim; it ends before EOF -> back trim floor(10*100)
    assert g.data.offset == 0 and len(g.data.buffer) == 3000   # 10 + 10 + 10 s
    assert t.offset == 0 and len(t.buffer) == 3000 - 1000
    ns, nd, nbefore, soff, doff = t.calls[-1]
    assert (nd, nbefore, soff, doff) == (2000, 0, 0, 0)
    assert ns == 2000 + 1                                       # nafter = 1 sample
    assert np.array_equal(t.buffer, 2*ramp(4000)[:2000])
    # scroll forward: data offset > 0 -> front trim floor(10*100) as well
    g.update_times(25.0, 30.0)
    d0 = g.data.offset
    assert d0 == int(15.0*100) - g.data.backframes + 0 or d0 >= 0
    assert t.offset == d0 + 1000
    assert np.array_equal(t.buffer, 2*ramp(4000)[t.offset:t.offset + len(t.buffer)])
    assert np.all(t.buffer_changed)


def test_move_buffer_recycles_overlap():
    t = Doubler()
    g = make_graph(t, buffer_time=10.0, back_time=0.0)
    t.plot_items = [Item(), Item()]
    g.set_need_update()
    g.update_times(0.0, 5.0)
    assert len(t.calls) == 1 and t.calls[0][1] == 1000
    g.update_times(6.0, 12.0)            # data moves to [600, 1600): 400 frames overlap
    assert g.data.offset == 600
    assert len(t.calls) == 2
    ns, nd, nbefore, soff, doff = t.calls[1]
    assert (nd, doff) == (600, 400)      # only the missing tail is computed
    assert soff == 400
    assert np.array_equal(t.buffer, 2*ramp(4000)[600:1600])
    g.update_times(3.0, 9.0)             # backwards: head is missing
    assert np.array_equal(t.buffer, 2*ramp(4000)[t.offset:t.offset + len(t.buffer)])
    assert t.calls[-1][4] == 0


def test_strided_trace_index_transform():
    """A trace at rate/step (the spectrogram's case): offsets via ceil/floor as in
    align_buffer (buffereddata.py:85-86) and load_buffer (:94-95)."""
    s = Doubler(step=8, tafter=10)
    g = make_graph(s, frames=4001)
    s.plot_items = [Item(), None]
    g.set_need_update()
    g.update_times(0.0, 5.0)
    n_data = len(g.data.buffer)
    trimmed = n_data - floor(10*100.0)
    assert s.offset == 0 and len(s.buffer) == floor(trimmed*s.rate/100.0)
    ns, nd, nbefore, soff, doff = s.calls[-1]
    assert ns == min(ceil(nd*8) + 1, n_data) and nbefore == 0


def test_need_update_propagation_and_recompute_all():
    g = TraceGraph(10.0, 2.0)
    a, b, c = Doubler('a', 'data'), Doubler('b', 'a'), Doubler('c', 'a')
    for t in (c, b, a):
        g.add_trace(t)
    g.setup_traces()
    assert [t.name for t in g.traces] == ['a', 'b', 'c'] or [t.name for t in g.traces] == ['a', 'c', 'b']
    g.open(ramp(3000), 100.0)
    assert a.dests == [g['b'], g['c']] or a.dests == [g['c'], g['b']]
    b.plot_items = [Item(True), None]
    c.plot_items = [Item(False), None]
    a.plot_items = [Item(False), None]
    g.set_need_update()
    assert b.need_update and not c.need_update
    assert a.need_update and g.data.need_update       # propagated up from the visible leaf
    g.update_times(0.0, 5.0)
    n0 = len(a.calls)
    a.recompute_all()
    assert len(a.calls) == n0 + 1 and len(b.calls) >= 2 and len(c.calls) == 0
    assert np.array_equal(b.buffer, 4*ramp(3000)[b.offset:b.offset + len(b.buffer)])
    b.set_visible(False)
    g.set_need_update()
    assert not b.need_update


def test_missing_source_is_reported():
    g = TraceGraph()
    g.add_trace(Doubler('x', 'nowhere'))
    with pytest.raises(ValueError):
        g.setup_traces()


def test_expand_times_accumulation():
    from audian_amd.bufferedfilter import BufferedFilter
    from audian_amd.bufferedenvelope import BufferedEnvelope
    g = TraceGraph()
    f, s, e = BufferedFilter.__new__(BufferedFilter), None, None
    # constructors only (no device needed before open)
    f = BufferedFilter()
    s = BufferedSpectrogram()
    e = BufferedEnvelope()
    for t in (f, s, e):
        g.add_trace(t)
    g.setup_traces()
    tb = [0]*3
    ta = [0]*3
    tbefore = tafter = 0
    for k in reversed(range(3)):
        b, a = g.traces[k].expand_times(tb[k], ta[k])
        i = g.sources[k]
        if i < 0:
            tbefore, tafter = max(tbefore, b), max(tafter, a)
        else:
            tb[i], ta[i] = max(tb[i], b), max(ta[i], a)
    assert (tbefore, tafter) == (11, 10)            # data.py:154-168 with the default traces
    assert (f.tbefore, f.tafter) == (1, 10) and (f.source_tbefore, f.source_tafter) == (10, 0)


def test_spectrogram_parameters():
    s = BufferedSpectrogram()
    assert (s.nfft, s.hop, s.overlap_frac) == (256, 128, 0.5)
    src = ArrayLoader(ramp(100000), 48000.0, buffer_time=1.0, back_time=0.0)
    s.open(src)
    assert s.hop == 128 and s.rate == 48000.0/128 and s.frames == ceil(100000/128)
    assert s.shape == (s.frames, 2, 129) and s.unit.endswith('^2/Hz')
    assert (s.ampl_min, s.ampl_max) == (0, 24000.0)
    assert s.fresolution == 48000.0/256 and s.tresolution == 128/48000.0
    assert len(s.frequencies) == 129
    s.update(nfft=4)                        # clamped to 8
    assert s.nfft == 8 and s.hop == 4 and s.shape[2] == 5
    s.update(nfft=2**31)                    # clamped to len(source)//2
    assert s.nfft == 50000
    s.update(nfft=1024, overlap_frac=2.0)   # overlap clamped to 0.99999 -> hop >= 1
    assert s.nfft == 1024 and s.hop == 1 and abs(s.overlap_frac - (1 - 1/1024)) < 1e-12
    s.update(overlap_frac=-1.0)
    assert s.hop == 1024 and s.overlap_frac == 0.0
    s.update(overlap_frac=0.75)
    assert s.hop == 256 and s.rate == 48000.0/256
    assert s.estimate_noiselevels(0) == (None, None)          # empty buffer


def test_host_decibel():
    p = np.array([0.0, 1e-20, 1e-19, 1.0, 100.0])
    d = decibel(p)
    assert d[0] == -np.inf and d[1] == -np.inf
    assert np.allclose(d[2:], [-190.0, 0.0, 20.0])
    assert decibel(10.0) == 10.0


def test_minmax_decimate_host_path():
    """Without a device mirror the facade falls back to the reference's own reduceat on the
    host buffer (no GPU involved here: the Doubler stub computes on the host)."""
    t = Doubler()
    g = make_graph(t, buffer_time=10.0, back_time=0.0)
    t.plot_items = [Item(), Item()]
    g.set_need_update()
    g.update_times(0.0, 5.0)
    got = t.minmax_decimate(t.offset + 3, t.offset + 503, 50, channel=1)
    blk = t.buffer[3:503, 1]
    seg = np.arange(0, 500, 50)
    want = np.zeros(20)
    np.minimum.reduceat(blk, seg, out=want[0::2])
    np.maximum.reduceat(blk, seg, out=want[1::2])
    assert np.array_equal(got, want)
    assert t.minmax_decimate(t.offset, t.offset + 1000, 7).shape == (2, 2*143)
    with pytest.raises(IndexError):
        t.minmax_decimate(t.offset, t.offset + len(t.buffer) + 1, 4)


def test_decimated_image_host_path(oracle):
    """BufferedSpectrogram.decimated_image without a device mirror: the NumPy reduceat + decibel."""
    from audian_amd.bufferedspectrogram import BufferedSpectrogram
    s = BufferedSpectrogram.__new__(BufferedSpectrogram)
    rng = np.random.default_rng(11)
    s.nfft, s.offset = 16, 100
    s._dev, s._dev_valid = None, []
    s._hostbuf = 10.0**rng.uniform(-22, 1, size=(50, 2, 9))
    s._stale = []
    got = s.decimated_image(103, 147, 6, 1)
    want = oracle.decimated_db_image(s._hostbuf, 3, 47, 6, 1)
    assert got.shape == (9, 8) and np.allclose(got, want, rtol=1e-6)
    assert s.decimated_image(110, 110, 3, 0).shape == (9, 0)
    with pytest.raises(IndexError):
        s.decimated_image(99, 120, 2, 0)


def _write_wav(path, data_int, nbytes, rate):
    import wave
    w = wave.open(str(path), 'wb')
    w.setnchannels(data_int.shape[1])
    w.setsampwidth(nbytes)
    w.setframerate(int(rate))
    if nbytes == 2:
        raw = data_int.astype('<i2').tobytes()
    elif nbytes == 4:
        raw = data_int.astype('<i4').tobytes()
    else:
        u = (data_int.astype(np.int64) & 0xFFFFFF).reshape(-1)
        raw = np.stack([u & 255, (u >> 8) & 255, (u >> 16) & 255], axis=1).astype(np.uint8).tobytes()
    w.writeframes(raw)
    w.close()


@pytest.mark.parametrize('nbytes', [2, 3, 4])
def test_wav_loader_scales_like_audioio(tmp_path, nbytes):
    from audian_amd.bufferedarray import WavLoader
    rng = np.random.default_rng(nbytes)
    bits = 8*nbytes
    ints = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), size=(5000, 3))
    ints[0] = [-(1 << (bits - 1)), (1 << (bits - 1)) - 1, 0]
    path = tmp_path/'x.wav'
    _write_wav(path, ints, nbytes, 8000)
    w = WavLoader(str(path), buffer_time=0.25, back_time=0.05)
    assert (w.rate, w.channels, w.frames, w.shape) == (8000.0, 3, 5000, (5000, 3))
    want = ints/float(1 << (bits - 1))
    assert np.array_equal(w[0:100, :], want[0:100])
    assert np.array_equal(w[4000:4500, 1], want[4000:4500, 1])       # moves the buffer
    assert w.offset > 0 and w[4999, 2] == want[4999, 2]
    raw = w.pcm_slab(10, 4)
    assert raw.dtype == np.uint8 and len(raw) == 4*3*nbytes
    w.close()


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_random_scroll_sequences_keep_buffers_consistent(seed):
    """Any sequence of window moves (forwards, backwards, jumps, EOF) leaves every derived
    trace's buffer equal to a from-scratch evaluation of its range (stateless stub traces at
    rate 1 and rate 1/8, pre/post-roll trimming as in align_buffer)."""
    rng = np.random.default_rng(seed)
    frames, rate = 60000, 100.0
    g = TraceGraph(buffer_time=20.0, back_time=5.0)
    a = Doubler('a', 'data', tbefore=3)
    b = Doubler('b', 'a', tafter=2)
    s = Doubler('s', 'a', step=8, tafter=4)
    for t in (s, b, a):
        g.add_trace(t)
    g.setup_traces()
    g.open(ramp(frames, 3), rate)
    for t in g.traces:
        t.plot_items = [Item(), None, None]
    g.set_need_update()
    t0 = 0.0
    ref = ramp(frames, 3)
    for _ in range(40):
        kind = rng.integers(0, 4)
        if kind == 0:
            t0 = max(0.0, t0 + rng.uniform(0.5, 8.0))
        elif kind == 1:
            t0 = max(0.0, t0 - rng.uniform(0.5, 8.0))
        elif kind == 2:
            t0 = rng.uniform(0.0, frames/rate - 1.0)
        else:
            t0 = frames/rate - rng.uniform(0.5, 5.0)
        t1 = min(frames/rate, t0 + rng.uniform(0.5, 10.0))
        g.update_times(t0, t1)
        d = g.data
        assert np.array_equal(d.buffer, ref[d.offset:d.offset + len(d.buffer)])
        assert np.array_equal(a.buffer, 2*ref[a.offset:a.offset + len(a.buffer)])
        assert np.array_equal(b.buffer, 4*ref[b.offset:b.offset + len(b.buffer)])
        want = 4*ref[::8][s.offset:s.offset + len(s.buffer)]
        assert np.array_equal(s.buffer, want)


@pytest.mark.parametrize('seed', range(6))
def test_load_geometry_is_what_process_gets(seed):
    """BufferedData._load_geometry (what a source's fused launch plans the derived traces' slabs with) is the very
    index arithmetic of load_buffer, quirk included: for random scrolls of random chains, every process() call saw
    the slab (first, count, lead) that _load_geometry predicts for its (offset, nframes)."""
    rng = np.random.default_rng(seed)

    class Spy(Doubler):
        def load_buffer(self, offset, nframes, buffer):
            self.predicted = self._load_geometry(offset, nframes)
            super().load_buffer(offset, nframes, buffer)

        def process(self, source, dest, nbefore):
            first, count, lead = self.predicted
            assert (self._pending.soffset, len(source), nbefore) == (first, max(count, 0), lead)
            super().process(source, dest, nbefore)

    step = int(rng.choice([1, 1, 4, 7]))
    t = Spy(tbefore=float(rng.choice([0, 0.5, 2])), tafter=float(rng.choice([0, 1, 3])), step=step)
    g = make_graph(t, frames=6000, rate=float(rng.choice([5.0, 100.0, 1000.0])), buffer_time=float(rng.uniform(2, 10)),
                   back_time=float(rng.uniform(0, 3)))
    t.plot_items = [Item(), Item()]
    g.set_need_update()
    span = 6000/g.data.rate
    for _ in range(25):
        t0 = float(rng.uniform(0, span*0.9))
        g.update_times(t0, min(span, t0 + float(rng.uniform(0.01, 0.3))*span))
    t.recompute_all()
    assert len(t.calls) >= 2
