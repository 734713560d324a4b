"""Base class for computed traces: the ``BufferedData`` plugin surface of audian
(``src/audian/buffereddata.py:10-153`` in /root/reference), re-stated on top of a
device-resident mirror.

What the PyQt browser sees is unchanged: ``open(source, step, more_shape)``,
``process(source, dest, nbefore)`` (the subclass hook), ``load_buffer``,
``align_buffer``, ``recompute``, ``recompute_all``, ``set_need_update``,
``expand_times``, ``update_step`` and the attributes ``rate, channels, frames, shape,
offset, buffer, buffer_changed, ampl_min, ampl_max, unit, source, dests, need_update,
plot_items``.  The bookkeeping follows the reference line by line in meaning --
including its seconds-divided-by-rate quirk in ``load_buffer``
(buffereddata.py:96,99), because that defines which samples reach the kernels.

What is new: every trace keeps its current buffer in HBM as planar float32
(channels, frames[, F]).  ``process()`` implementations write there and only mark the
host copy stale; reading ``trace.buffer`` (or ``trace[i0:i1, ch]``) copies back what
is stale, lazily.  A derived trace whose source is another ``BufferedData`` reads the
source's device mirror, so filter -> spectrogram -> envelope never leaves HBM
(SURVEY 7-4).
"""

import os
from math import floor, ceil

import numpy as np

from .bufferedarray import BufferedArray

_TRACE = bool(os.environ.get('AUDIAN_AMD_TRACE'))
# A trace's device mirror is what the kernels write into: the best of this many allocations by the time of a memset over
# each (hipdsp_malloc_probed; mirrors under 64 MiB: a plain allocation).  AUDIAN_AMD_WRITE_PROBE=1 turns the search off.
WRITE_PROBE = int(os.environ.get('AUDIAN_AMD_WRITE_PROBE', '4'))


class _Call(object):
    """Where the slab handed to process() sits: set by load_buffer for one call."""

    def __init__(self, soffset, snframes, doffset, dnframes):
        self.soffset, self.snframes = soffset, snframes
        self.doffset, self.dnframes = doffset, dnframes


def _merge(ranges):
    out = []
    for a, b in sorted(r for r in ranges if r[1] > r[0]):
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def _covers(ranges, a, b):
    return any(r[0] <= a and b <= r[1] for r in ranges) or b <= a


def _subtract(ranges, a, b):
    out = []
    for r0, r1 in ranges:
        if r1 <= a or r0 >= b:
            out.append([r0, r1])
        else:
            if r0 < a:
                out.append([r0, a])
            if r1 > b:
                out.append([b, r1])
    return out


class _LazyBuffer(object):
    """What ``trace.buffer`` returns while parts of the host copy are stale: it looks like the
    (frames, channels[, F]) float64 array, but only copies back from the device mirror the
    frame range an access touches (``len()`` and ``shape`` touch nothing).  Anything it does
    not know (``.T``, ufuncs on the whole array, ...) synchronises everything and falls
    through to the real ndarray."""

    def __init__(self, trace):
        object.__setattr__(self, '_t', trace)

    def __len__(self):
        return len(self._t._hostbuf)

    @property
    def shape(self):
        return self._t._hostbuf.shape

    @property
    def ndim(self):
        return self._t._hostbuf.ndim

    @property
    def dtype(self):
        return self._t._hostbuf.dtype

    @property
    def size(self):
        return self._t._hostbuf.size

    def _frames_of(self, key):
        first = key[0] if isinstance(key, tuple) else key
        n = len(self._t._hostbuf)
        if isinstance(first, slice):
            a, b, step = first.indices(n)
            return (a, b) if step > 0 else (0, n)
        if isinstance(first, (int, np.integer)):
            i = int(first) + (n if first < 0 else 0)
            return i, i + 1
        return 0, n

    def __getitem__(self, key):
        a, b = self._frames_of(key)
        one = self._t._read_one_channel(key, a, b)
        if one is not None:
            return one
        self._t._flush_range(a, b)
        return self._t._hostbuf[key]

    def __setitem__(self, key, value):
        a, b = self._frames_of(key)
        self._t._flush_range(a, b)
        self._t._hostbuf[key] = value
        self._t._dev_valid = _subtract(self._t._dev_valid, a, b)

    def __array__(self, dtype=None, copy=None):
        self._t._flush()
        return np.asarray(self._t._hostbuf, dtype=dtype)

    def __getattr__(self, name):
        self._t._flush()
        return getattr(self._t._hostbuf, name)

    def __iter__(self):
        self._t._flush()
        return iter(self._t._hostbuf)


def _delegate(name):
    def method(self, *args):
        self._t._flush()
        return getattr(self._t._hostbuf, name)(*args)
    method.__name__ = name
    return method


class _LazySource(object):
    """The ``source`` argument of process() while the source trace's host copy of that span is stale
    (its results live in the device mirror only): looks like ``source.buffer[first:first + count]`` of
    the reference contract (buffereddata.py:91-109) and copies the span back from the mirror the
    moment a subclass actually reads it; ``len()``, ``shape`` and ``dtype`` touch nothing, so the
    built-in process() implementations, which take the mirror instead, never trigger the copy."""

    def __init__(self, trace, first, count):
        object.__setattr__(self, '_t', trace)
        object.__setattr__(self, '_a', int(first))
        object.__setattr__(self, '_b', int(first) + int(count))

    def _real(self):
        self._t._flush_range(self._a, self._b)
        return self._t._hostbuf[self._a:self._b]

    def __len__(self):
        return self._b - self._a

    @property
    def shape(self):
        return (self._b - self._a,) + tuple(self._t._hostbuf.shape[1:])

    @property
    def ndim(self):
        return self._t._hostbuf.ndim

    @property
    def dtype(self):
        return self._t._hostbuf.dtype

    def __getitem__(self, key):
        return self._real()[key]

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self._real(), dtype=dtype)

    def __getattr__(self, name):
        return getattr(self._real(), name)

    def __iter__(self):
        return iter(self._real())


def _delegate_source(name):
    def method(self, *args):
        return getattr(self._real(), name)(*args)
    method.__name__ = name
    return method


for _name in ('__eq__', '__ne__', '__lt__', '__le__', '__gt__', '__ge__', '__add__', '__radd__',
              '__sub__', '__rsub__', '__mul__', '__rmul__', '__truediv__', '__rtruediv__',
              '__pow__', '__neg__', '__abs__', '__matmul__'):
    setattr(_LazyBuffer, _name, _delegate(_name))
    setattr(_LazySource, _name, _delegate_source(_name))
_LazyBuffer.__hash__ = None
_LazySource.__hash__ = None


class BufferedData(BufferedArray):

    _hostbuf = None
    _raw_cache = None
    _dev = None
    _dev_valid = ()
    _stale = ()
    _pending = None
    _carry = None
    _ctx = None
    _fused_token = False       # the source's own launch has already filled this trace's whole buffer (mirror)
    _alias_pitch = None        # the mirror is a VIEW of another device array with this row pitch (_alias_mirror)

    def __init__(self, name, source_name, tbefore=0, tafter=0, panel='none', panel_type='trace',
                 color='#00ee00', lw_thin=1.1, lw_thick=2):
        """Arguments as in audian (buffereddata.py:14-30).  `tbefore` / `tafter` are the margins
        (seconds) this trace needs from its SOURCE; its own margins start at zero and only grow
        through expand_times() when traces derived from it ask for more."""
        BufferedArray.__init__(self, verbose=0)
        self.name, self.source_name = name, source_name
        self.source, self.dests = None, []
        self.source_tbefore, self.source_tafter = tbefore, tafter
        self.tbefore = self.tafter = 0
        self.need_update = False
        # presentation attributes the plot code reads
        self.panel, self.panel_type = panel, panel_type
        self.color, self.lw_thin, self.lw_thick = color, lw_thin, lw_thick
        self.plot_items = []

    # ---- host buffer with lazy read-back ---------------------------------------
    @property
    def buffer(self):
        # the plain ndarray when the host copy is current, else a proxy that reads back only
        # what is accessed (the plot code asks for len(buffer) and per-channel slices)
        if self._stale:
            return _LazyBuffer(self)
        return self._hostbuf

    @buffer.setter
    def buffer(self, value):
        # a new host buffer invalidates the mirror (allocate_buffer, open)
        self._hostbuf = value
        self._stale = []
        self._dev = None
        self._dev_valid = []
        self._alias_pitch = None

    def _buf(self):
        return self._hostbuf

    def _prepare_keep(self, a, b):
        """Frames [a, b) survive a buffer move.  When the device mirror holds them, nothing is
        read back: the mirror is recycled on the device (_adopt_buffer) and what was stale on the
        host simply stays stale at its new position -- a scroll then costs no PCIe traffic and no
        host copy for results nobody has looked at.  Otherwise the stale part of the range is
        read back now, before the old buffer goes away."""
        self._carry = None
        if not self._stale:
            return True
        inside = [[max(r0, a), min(r1, b)] for r0, r1 in self._stale if min(r1, b) > max(r0, a)]
        if self._dev is not None and _covers(self._dev_valid, a, b):
            self._carry = inside
            self._stale = []                      # the old host buffer is about to be dropped
            return not _covers(inside, a, b)      # all of it stale: no host copy at all
        self._stale = inside
        if inside:
            self._flush()
        return True

    @property
    def ctx(self):
        if self._ctx is None:
            from . import hipdsp
            self._ctx = hipdsp.default_context()
        return self._ctx

    def _inner(self):
        """Elements per frame and channel (1 for traces, F for spectrograms)."""
        n = 1
        for s in self.shape[2:]:
            n *= int(s)
        return n

    def _pitch(self):
        """Elements between the rows (channels) of the device mirror."""
        return self._alias_pitch if self._alias_pitch is not None else len(self._hostbuf)*self._inner()

    def _mirror(self):
        """The device mirror of the current host buffer, allocated on demand (a mirror that is only a view of
        another array, _alias_mirror, becomes a real one first: whoever asks is about to write into it)."""
        from . import hipdsp
        if self._dev is not None and self._alias_pitch is not None:
            view, pitch, valid = self._dev, self._alias_pitch, list(self._dev_valid)
            self._dev, self._alias_pitch = None, None
            n = max(1, len(self._hostbuf)*self._inner())
            self._dev = hipdsp.DeviceArray(self.ctx, (max(1, self.channels), n), np.float32, write_probe=WRITE_PROBE)
            for a, b in valid:
                hipdsp.memcpy2d(self.ctx, self._dev.view(a, (1,)), 4*n, view.view(a, (1,)), 4*pitch, 4*(b - a),
                                self.channels)
            self._dev_valid = valid
            self.ctx.synchronize()
        if self._dev is None:
            n = max(1, len(self._hostbuf)*self._inner())
            self._dev = hipdsp.DeviceArray(self.ctx, (max(1, self.channels), n), np.float32, write_probe=WRITE_PROBE)
            self._dev_valid = []
        return self._dev

    def _alias_mirror(self, array, pitch, call):
        """The whole buffer of this trace IS frames [0, dnframes) of the rows of `array` (a device array that is not
        written again while this trace points at it: a slab's device copy, replaced -- never rewritten -- when the slab
        changes): no copy, the mirror becomes a view that keeps `array` alive.  Any write into the mirror, a buffer
        move or a new host buffer turns it back into (or replaces it with) a real one."""
        self._dev, self._alias_pitch = array, int(pitch)
        self._dev_valid = [[0, call.dnframes]]
        self._stale = [[0, call.dnframes]]

    def _flush(self):
        """Copy the stale frame ranges from the mirror into the host buffer."""
        from . import hipdsp
        stale, self._stale = self._stale, []
        inner = self._inner()
        pitch = self._pitch()
        for a, b in stale:
            n = b - a
            tmp = hipdsp.DeviceArray(self.ctx, (n, self.channels, inner) if inner > 1
                                     else (n, self.channels), np.float64)
            src = self._dev.view(a*inner, (1,))
            if inner > 1:
                hipdsp.unpack_spectrum(self.ctx, src, tmp, n, self.channels, inner, src_pitch=pitch)
                self._hostbuf[a:b] = tmp.to_host().reshape((n, self.channels) + tuple(self.shape[2:]))
            else:
                hipdsp.unpack(self.ctx, src, pitch, tmp, n, self.channels)
                self._hostbuf[a:b] = tmp.to_host()
            tmp.free()

    def _read_one_channel(self, key, a, b):
        """`buffer[frames, channel, ...]` as the plot items ask for it (traceitem.py:58-61,
        specitem.py:36): in the planar device layout one channel's frames are contiguous, so only
        they cross PCIe (float32) instead of every channel of the frame range (float64).  Returns
        None when the access is of another kind or the host copy is current there anyway."""
        if not (isinstance(key, tuple) and len(key) >= 2 and isinstance(key[1], (int, np.integer))):
            return None
        first, rest = key[0], key[2:]
        if isinstance(first, slice):
            step = first.indices(len(self._hostbuf))[2]
            if step < 1:
                return None
        elif not isinstance(first, (int, np.integer)):
            return None
        if any(not isinstance(r, (int, np.integer, slice)) for r in rest) or len(rest) > self._hostbuf.ndim - 2:
            return None
        if b <= a or self._dev is None or not _covers(self._dev_valid, a, b):
            return None
        if not any(min(r1, b) > max(r0, a) for r0, r1 in self._stale):
            return None                                   # nothing stale in there
        ch = int(key[1])
        if ch < 0:
            ch += self.channels
        if not 0 <= ch < self.channels:
            return None                                   # let numpy raise its IndexError
        inner = self._inner()
        block = self._dev.view(ch*self._pitch() + a*inner, ((b - a)*inner,)).to_host().astype(np.float64)
        block = block.reshape((b - a,) + tuple(self._hostbuf.shape[2:]))
        if isinstance(first, slice):
            out = block[::step]
            return out[(slice(None),) + tuple(rest)] if rest else out
        out = block[0]
        return out[tuple(rest)] if rest else out

    def _flush_range(self, a, b):
        """Read back only the stale parts of frames [a, b)."""
        if not self._stale or b <= a:
            return
        need = []
        for r0, r1 in self._stale:
            lo, hi = max(r0, a), min(r1, b)
            if hi > lo:
                need.append([lo, hi])
        if not need:
            return
        rest = list(self._stale)
        for lo, hi in need:
            rest = _subtract(rest, lo, hi)
        self._stale = need
        self._flush()
        self._stale = rest

    def _adopt_buffer(self, new, offset, old_offset, old_nframes, keep0, keep1):
        """move_buffer recycled the host buffer: recycle the mirror the same way."""
        from . import hipdsp
        old_dev, old_valid, old_pitch = self._dev, list(self._dev_valid), self._pitch()
        self.buffer = new            # setter drops mirror and stale marks (host is current)
        self.offset = offset
        if old_dev is not None and keep1 > keep0 and \
           _covers(old_valid, keep0 - old_offset, keep1 - old_offset) and len(new) > 0:
            inner = self._inner()
            dev = self._mirror()
            hipdsp.memcpy2d(self.ctx, dev.view((keep0 - offset)*inner, (1,)), 4*len(new)*inner,
                            old_dev.view((keep0 - old_offset)*inner, (1,)), 4*old_pitch,
                            4*(keep1 - keep0)*inner, self.channels)
            self._dev_valid = [[keep0 - offset, keep1 - offset]]
            if self._carry:
                shift = old_offset - offset
                self._stale = _merge([[r0 + shift, r1 + shift] for r0, r1 in self._carry])
            self.ctx.synchronize()   # old mirror may be freed now
        self._carry = None
        if old_dev is not None:
            old_dev.free()

    # ---- helpers for process() implementations -----------------------------------
    def _take_call(self, source, dest):
        call, self._pending = self._pending, None
        if call is not None and (call.snframes != len(source) or call.dnframes != len(dest)):
            call = None
        return call

    def _take_fused(self, call):
        """True when this call of process() is the recompute of the whole buffer that the SOURCE's fused
        launch has already carried out (BufferedFilter.recompute_all: one launch fills the filtered trace
        and the traces derived from it): the mirror is marked valid, nothing is launched."""
        token, self._fused_token = self._fused_token, False
        if not token or call is None or call.doffset != 0 or call.dnframes != len(self._hostbuf) or \
           self._dev is None:
            return False
        self._dev_valid = [[0, call.dnframes]]
        self._stale = [[0, call.dnframes]]
        return True

    def _builtin(self, cls):
        """process() and recompute() are the built-in ones (a subclass that overrides either computes
        in its own way and takes no part in fused launches)."""
        return type(self).process is cls.process and type(self).recompute is BufferedData.recompute

    def _device_source(self, source, call):
        """(device pointer holder, pitch in elements) of the source slab, planar float32.
        Uses the source trace's mirror when it is valid there, else uploads `source`."""
        from . import hipdsp
        src = self.source
        if call is not None and isinstance(src, BufferedData) and src._dev is not None and \
           _covers(src._dev_valid, call.soffset, call.soffset + call.snframes):
            inner = src._inner()
            return src._dev.view(call.soffset*inner, (1,)), src._pitch(), None
        if call is not None and isinstance(src, BufferedData) and src._stale:
            source = src.buffer[call.soffset:call.soffset + call.snframes]     # flushes
        n = len(source)
        if call is not None and n > 0 and hasattr(src, 'pcm_slab') and getattr(src, 'unwrap_thresh', 0.0) <= 1e-3:
            # the loader can hand over the file's own integers: upload those (2-4 bytes per
            # sample instead of 8) and convert on the device
            # (an interactive cut-off sweep recomputes from the SAME raw slab: keep its device copy)
            pkey = ('pcm', id(src), src.offset + call.soffset, n)
            if self._raw_cache is not None and self._raw_cache[0] == pkey:
                return self._raw_cache[1], n, None
            raw = src.pcm_slab(src.offset + call.soffset, n)
            up = hipdsp.DeviceArray.from_host(self.ctx, raw)
            planar = hipdsp.DeviceArray(self.ctx, (max(1, src.channels), n), np.float32)
            hipdsp.pcm_unpack(self.ctx, up, src.sample_bytes, n, src.channels, src.scale, planar, n)
            self.ctx.synchronize()
            self._raw_cache = (pkey, planar)
            return planar, n, None
        key = None
        if call is not None and n > 0 and not isinstance(src, BufferedData):
            # same for float slabs, keyed by the slab's identity, position and a fingerprint of a
            # strided subsample (a loader may rewrite its buffer in place)
            sample = source[::max(1, n//4096)]
            key = (id(src), src.offset, self._source_len(), call.soffset, n,
                   hash(np.ascontiguousarray(sample).tobytes()))
            if self._raw_cache is not None and self._raw_cache[0] == key:
                return self._raw_cache[1], n, None
        dtype = np.float32 if source.dtype == np.float32 else np.float64
        host = np.ascontiguousarray(source.reshape(n, -1), dtype=dtype)
        up = hipdsp.DeviceArray.from_host(self.ctx, host)
        planar = hipdsp.DeviceArray(self.ctx, (max(1, host.shape[1]), max(1, n)), np.float32)
        if n > 0:
            hipdsp.pack(self.ctx, up, planar, n, n, host.shape[1], src_dtype=dtype)
        if key is not None:
            self.ctx.synchronize()
            self._raw_cache = (key, planar)
            return planar, n, None
        return planar, max(1, n), up

    def _device_dest(self, dest, call):
        """(device pointer holder, pitch in elements, is_mirror) for the output slab."""
        from . import hipdsp
        inner = self._inner()
        if call is not None:
            dev = self._mirror()
            return dev.view(call.doffset*inner, (1,)), len(self._hostbuf)*inner, True
        n = max(1, len(dest))
        return hipdsp.DeviceArray(self.ctx, (max(1, self.channels), n*inner), np.float32), n*inner, False

    def _finish_dest(self, dest, ddst, pitch, is_mirror, call):
        """Mark the mirror range written (host stale), or copy a temporary result to dest."""
        from . import hipdsp
        if is_mirror:
            a, b = call.doffset, call.doffset + call.dnframes
            self._dev_valid = _merge(list(self._dev_valid) + [[a, b]])
            self._stale = _merge(list(self._stale) + [[a, b]])
            return
        n, inner = len(dest), self._inner()
        if n == 0:
            return
        if inner > 1:
            tmp = hipdsp.DeviceArray(self.ctx, (n, self.channels, inner), np.float64)
            hipdsp.unpack_spectrum(self.ctx, ddst, tmp, n, self.channels, inner, src_pitch=pitch)
        else:
            tmp = hipdsp.DeviceArray(self.ctx, (n, self.channels), np.float64)
            hipdsp.unpack(self.ctx, ddst, pitch, tmp, n, self.channels)
        dest[...] = tmp.to_host().reshape(dest.shape)

    def minmax_decimate(self, start, stop, step, channel=None):
        """Min/max screen decimation of frames [start, stop) (absolute frame indices inside
        the current buffer), `step` frames per plot point: what TraceItem.update_plot computes
        with np.minimum/maximum.reduceat (src/audian/traceitem.py:55-61), interleaved as
        min, max, min, max, ...  Runs on the device mirror when it is valid there and returns
        float64 like the reference's plot_data: (2*n,) for one channel, (channels, 2*n) for all.
        """
        from . import hipdsp
        if self._inner() != 1:
            raise ValueError('minmax_decimate is for traces, not spectrograms')
        a, b = int(start) - self.offset, int(stop) - self.offset
        n = len(self._hostbuf)
        if a < 0 or b > n or b < a or step < 1:
            raise IndexError('range outside the loaded buffer')
        nseg = (b - a + step - 1)//step
        if nseg == 0:
            return np.zeros(0) if channel is not None else np.zeros((self.channels, 0))
        if self._dev is not None and _covers(self._dev_valid, a, b):
            out = hipdsp.DeviceArray(self.ctx, (self.channels, 2*nseg), np.float32)
            hipdsp.minmax_decimate(self.ctx, self._dev, self._pitch(), self.channels, a, b, step, out, 2*nseg)
            res = out.to_host().astype(np.float64)
        else:
            seg = np.arange(0, b - a, step)
            buf = self.buffer[a:b]
            res = np.empty((self.channels, 2*nseg))
            res[:, 0::2] = np.minimum.reduceat(buf, seg, axis=0).T
            res[:, 1::2] = np.maximum.reduceat(buf, seg, axis=0).T
        return res[channel] if channel is not None else res

    # ---- the reference's surface (src/audian/buffereddata.py), restated --------------
    def expand_times(self, tbefore, tafter):
        """Widen this trace's own margins by what a derived trace needs; returns what the source
        then has to provide (buffereddata.py:33-36)."""
        self.tbefore, self.tafter = self.tbefore + tbefore, self.tafter + tafter
        return tbefore + self.source_tbefore, tafter + self.source_tafter

    def update_step(self, step=1, more_shape=None):
        """Frame bookkeeping for one frame of this trace per `step` source frames
        (buffereddata.py:39-56): rate, frames, shape, offset round UP to whole frames; the buffer
        keeps its length in seconds unless the source holds the whole recording."""
        src = self.source
        seconds = self.bufferframes/self.rate            # at the rate valid so far
        step = max(step, 1)

        def frames_of(n):
            return -(-n//step)

        self.rate = src.rate/step
        self.frames = frames_of(src.frames)
        self.shape = (self.frames, self.channels) + tuple(more_shape or ())
        self.ndim, self.size = len(self.shape), self.frames*self.channels
        whole_recording = src.bufferframes == src.frames
        self.bufferframes = self.frames if whole_recording else int(seconds*self.rate)
        self.offset = frames_of(src.offset)
        self.follow = 0

    def open(self, source, step=1, more_shape=None):
        """Become a destination of `source`: inherit channels, rate, amplitude range and unit,
        start with an empty buffer (buffereddata.py:59-72)."""
        source.dests.append(self)
        self.source = source
        for attr in ('ampl_min', 'ampl_max', 'unit', 'channels', 'rate'):
            setattr(self, attr, getattr(source, attr))
        self.bufferframes = self.backframes = 0
        self.buffer_changed = np.zeros(self.channels, dtype=bool)
        self.buffer = np.zeros((0, self.channels))
        self.plot_items = [None]*self.channels
        self.update_step(step, more_shape)

    def _source_len(self):
        src = self.source
        return len(src._hostbuf) if isinstance(src, BufferedData) else len(src.buffer)

    def _source_buffer(self):
        src = self.source
        return src._hostbuf if isinstance(src, BufferedData) else src.buffer

    def align_buffer(self):
        """Put this buffer over what the source holds minus the margins the computation needs:
        `source_tbefore` at the front unless the source starts at the beginning of the recording,
        `source_tafter` at the back unless it reaches the end (buffereddata.py:75-88)."""
        src = self.source
        first, count = src.offset, self._source_len()
        reaches_end = first + count >= src.frames
        if first > 0:
            margin = floor(self.source_tbefore*src.rate)
            first, count = first + margin, count - margin
        if not reaches_end:
            count -= floor(self.source_tafter*src.rate)
        offset = ceil(first*self.rate/src.rate)
        self.move_buffer(offset, floor((first + count)*self.rate/src.rate) - offset)
        self.bufferframes = len(self._hostbuf)

    def _load_geometry(self, offset, nframes):
        """Which slab of the source's buffer load_buffer hands to process() for frames
        [offset, offset + nframes) of this trace: (first, count, lead) -- `count` source frames from
        `first` (relative to the source's buffer), the first `lead` of them ahead of the span
        (buffereddata.py:93-107)."""
        src = self.source
        # the same span in source frames
        first = floor(offset*src.rate/self.rate)
        count = ceil((offset + nframes)*src.rate/self.rate) - first
        # Margins.  The reference DIVIDES the margins in seconds by the rate where it means to
        # multiply (buffereddata.py:96,99), which for any real rate gives no frame before and one
        # frame after.  Kept as it is: it decides what process() gets to see.
        lead = floor(self.source_tbefore/src.rate)
        tail = ceil(self.source_tafter/src.rate)
        first -= lead + src.offset               # from here on relative to the source's buffer
        count += lead + tail
        if first < 0:                             # the source's buffer starts later than that
            lead, count, first = lead + first, count + first, 0
        count = min(count, self._source_len() - first)
        return first, count, lead

    def load_buffer(self, offset, nframes, buffer):
        """Fill `buffer` (frames [offset, offset + nframes) of this trace) from the source's
        buffer through process() (buffereddata.py:91-109)."""
        if _TRACE:
            print(f'load {self.name} {offset/self.rate:.3f} - {(offset + nframes)/self.rate:.3f}')
        src = self.source
        first, count, lead = self._load_geometry(offset, nframes)
        source = self._source_buffer()[first:first + count]
        if isinstance(src, BufferedData) and any(min(r1, first + count) > max(r0, first) for r0, r1 in src._stale):
            # the raw host slice would show stale frames to a subclass that reads `source` itself
            source = _LazySource(src, first, len(source))
        self._pending = _Call(first, len(source), offset - self.offset, len(buffer))
        try:
            self.process(source, buffer, lead)
        finally:
            self._pending = None

    def process(self, source, dest, nbefore):
        raise NotImplementedError

    def reload_buffer(self):
        # everything is recomputed: nothing stale needs to be read back first
        if len(self._hostbuf) > 0:
            self._stale = []
            self._dev_valid = []
            self.load_buffer(self.offset, len(self._hostbuf), self._hostbuf)
            self.buffer_changed[:] = True

    def recompute(self):
        """(Re)allocate for the current geometry and compute the whole buffer (buffereddata.py:112-115)."""
        if self._source_len() > 0:
            self.allocate_buffer()
        self.reload_buffer()

    def is_visible(self):
        return any(item is not None and item.isVisible() for item in self.plot_items)

    def set_visible(self, show):
        for item in self.plot_items:
            if item is not None:
                item.setVisible(show)

    def set_need_update(self):
        """A trace needs computing if one of its plot items is shown; from every leaf of the
        dependency tree the need then climbs up to all sources (buffereddata.py:131-146)."""
        self.need_update = self.is_visible()
        for dest in self.dests:
            dest.set_need_update()
        if self.dests:
            return
        node = self
        while hasattr(node, 'source'):
            node.source.need_update = node.need_update or node.source.need_update
            node = node.source

    def recompute_all(self):
        """Recompute this trace and, recursively, the traces derived from it -- those that are
        needed (buffereddata.py:149-153)."""
        if not self.need_update:
            return
        self.recompute()
        for dest in self.dests:
            dest.recompute_all()
