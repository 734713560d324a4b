"""Minimal ring-buffer base class standing in for ``audioio.BufferedArray``.

audioio is not part of the reference tree (``src/audian/buffereddata.py:7`` imports
it), so this module supplies the contract the ``BufferedData`` surface relies on,
derived from the reference's call sites:

  attributes  rate, channels, frames, shape, ndim, size, offset, buffer, bufferframes,
              backframes, follow, ampl_min, ampl_max, unit, buffer_changed, verbose
  methods     __len__ (frames), __getitem__ (moves the buffer on demand,
              data.py:112), update_time(t0, t1) (data.py:227), update_buffer,
              move_buffer(offset, nframes) (buffereddata.py:87: keeps the overlapping
              part, calls load_buffer for what is missing, sets buffer_changed),
              allocate_buffer() (buffereddata.py:114), reload_buffer() (:115),
              load_buffer(offset, nframes, view) -- the subclass hook.

It is written from that contract, not from audioio's source.
"""

import numpy as np


class BufferedArray(object):

    def __init__(self, verbose=0):
        self.rate = 0.0
        self.channels = 0
        self.frames = 0
        self.shape = (0, 0)
        self.ndim = 2
        self.size = 0
        self.offset = 0
        self.bufferframes = 0
        self.backframes = 0
        self.follow = 0
        self.ampl_min = -1.0
        self.ampl_max = 1.0
        self.unit = ''
        self.verbose = verbose
        self.buffer_changed = np.zeros(0, dtype=bool)
        self.buffer = np.zeros((0, 0))
        self.unwrap_thresh = 0.0
        self.unwrap_clips = False
        self.unwrap_down_scale = True
        self.unwrap_ampl = 1.0

    def __len__(self):
        return self.frames

    # -- unwrap of clipped recordings (audioio: BufferedArray.set_unwrap / unwrap()) -----------
    def set_unwrap(self, thresh, clips=False, down_scale=True, unit=''):
        """Arm audioio's unwrap() for every slab this loader reads from now on, as the reference does
        right after opening the recording (``self.data.set_unwrap(unwrap, unwrap_clip, False, unit)``,
        src/audian/data.py:180; CLI ``-u`` / ``-U``, src/audian/audian.py:1485-1512).  ``thresh`` <= 1e-3
        turns it off.  Without clipping and down-scaling the amplitude range doubles.  audioio's source
        is not available here: restated from its documentation -- UNVERIFIED against audioio (parity
        unpinned; `unit` is accepted and ignored).  Like audioio's per-buffer call, every slab a loader
        reads is unwrapped on its own, starting from zero offset at its first frame: a buffer move that
        keeps an overlap and loads the rest can therefore carry different offsets in the kept and the new
        part of a recording that is wrapped at the seam (tests/test_gpu_facade.py pins exactly this
        behaviour, not audioio's)."""
        self.unwrap_ampl = float(self.ampl_max if self.unwrap_thresh <= 1e-3 else self.unwrap_ampl)
        self.unwrap_thresh = float(thresh)
        self.unwrap_clips = bool(clips)
        self.unwrap_down_scale = bool(down_scale)
        if self.unwrap_thresh > 1e-3:
            grow = 1.0 if (self.unwrap_clips or self.unwrap_down_scale) else 2.0
            self.ampl_min, self.ampl_max = -grow*self.unwrap_ampl, grow*self.unwrap_ampl
            if getattr(self, 'view', False):
                self.view = False              # the buffer must be this loader's own copy now
                self.buffer = np.zeros((0, self.channels))
                self.move_buffer(self.offset, self.bufferframes)
                return
        else:
            self.ampl_min, self.ampl_max = -self.unwrap_ampl, self.unwrap_ampl
        if len(self._buf()) > 0:
            self.reload_buffer()

    def _apply_unwrap(self, buffer):
        """Unwrap a freshly loaded (frames, channels) slab in place (device kernels: hipdsp_unwrap);
        like audioio, every slab starts again from zero offset."""
        if self.unwrap_thresh <= 1e-3 or len(buffer) == 0:
            return
        from . import hipdsp
        ctx = hipdsp.default_context()
        n, nch = buffer.shape
        host = np.ascontiguousarray(buffer, dtype=np.float32)
        up = hipdsp.DeviceArray.from_host(ctx, host)
        planar = hipdsp.DeviceArray(ctx, (nch, n), np.float32)
        hipdsp.pack(ctx, up, planar, n, n, nch, src_dtype=np.float32)
        out = hipdsp.DeviceArray(ctx, (nch, n), np.float32)
        hipdsp.unwrap(ctx, planar, n, nch, n, self.unwrap_thresh, out, n, ampl_max=self.unwrap_ampl,
                      clips=self.unwrap_clips, down_scale=self.unwrap_down_scale)
        tmp = hipdsp.DeviceArray(ctx, (n, nch), np.float64)
        hipdsp.unpack(ctx, out, n, tmp, n, nch)
        buffer[:, :] = tmp.to_host()
        for d in (up, planar, out, tmp):
            d.free()

    # -- subclass hook ---------------------------------------------------------
    def load_buffer(self, offset, nframes, buffer):
        raise NotImplementedError

    # -- buffer management -----------------------------------------------------
    def _buf(self):
        """The buffer as stored (subclasses with a lazy host copy return it unsynced)."""
        return self.buffer

    def _prepare_keep(self, a, b):
        """Hook: frames [a, b) of the current buffer are about to be copied.  Returns False when
        the host copy of that range is not worth copying (a subclass keeps it elsewhere)."""
        return True

    def _blank(self, nframes):
        return np.zeros((int(nframes),) + tuple(self.shape[1:]))

    def allocate_buffer(self, nframes=None, force=False):
        """Size ``buffer`` to ``bufferframes`` frames (clipped to the data)."""
        if nframes is None:
            nframes = self.bufferframes
        if self.offset + nframes > self.frames:
            nframes = max(0, self.frames - self.offset)
        cur = self._buf()
        if force or nframes != len(cur) or tuple(cur.shape[1:]) != tuple(self.shape[1:]):
            self.buffer = self._blank(nframes)

    def reload_buffer(self):
        """Recompute the whole current buffer in place."""
        cur = self._buf()
        if len(cur) > 0:
            self.load_buffer(self.offset, len(cur), cur)
            self.buffer_changed[:] = True

    def move_buffer(self, offset, nframes):
        """Make the buffer cover frames [offset, offset + nframes): the part that
        overlaps the current buffer is kept, the rest comes from ``load_buffer``."""
        offset = int(max(0, offset))
        nframes = int(max(0, min(nframes, self.frames - offset)))
        old, old_off = self._buf(), self.offset
        if offset == old_off and nframes == len(old):
            return
        new = self._blank(nframes)
        keep0 = max(offset, old_off)
        keep1 = min(offset + nframes, old_off + len(old))
        if tuple(old.shape[1:]) != tuple(new.shape[1:]):
            keep0 = keep1 = 0
        todo = []
        if keep1 > keep0:
            if self._prepare_keep(keep0 - old_off, keep1 - old_off) is not False:
                new[keep0 - offset:keep1 - offset] = old[keep0 - old_off:keep1 - old_off]
            if keep0 > offset:
                todo.append((offset, keep0 - offset))
            if keep1 < offset + nframes:
                todo.append((keep1, offset + nframes - keep1))
        elif nframes > 0:
            todo.append((offset, nframes))
        self._adopt_buffer(new, offset, old_off, len(old), keep0, keep1)
        for r_offset, r_nframes in todo:
            self.load_buffer(r_offset, r_nframes,
                             self.buffer[r_offset - offset:r_offset - offset + r_nframes])
        self.buffer_changed[:] = True

    def _adopt_buffer(self, new, offset, old_offset, old_nframes, keep0, keep1):
        """Install the recycled buffer (hook for subclasses that mirror it elsewhere)."""
        self.buffer = new
        self.offset = offset

    def _buffer_position(self, start, stop):
        """Where to put the buffer so that frames [start, stop) are inside it."""
        nframes = max(self.bufferframes, stop - start)
        offset = start - self.backframes
        if offset + nframes > self.frames:
            offset = self.frames - nframes
        if offset < 0:
            offset = 0
        if offset + nframes > self.frames:
            nframes = self.frames - offset
        return offset, nframes

    def update_buffer(self, start, stop):
        start = int(max(0, start))
        stop = int(min(self.frames, stop))
        if stop <= start:
            return
        if start < self.offset or stop > self.offset + len(self.buffer):
            offset, nframes = self._buffer_position(start, stop)
            self.move_buffer(offset, nframes)

    def update_time(self, start, stop):
        self.update_buffer(int(start*self.rate), int(stop*self.rate) + 1)

    def __getitem__(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        first, rest = key[0], key[1:]
        if isinstance(first, slice):
            start, stop, step = first.indices(self.frames)
            if step < 1:
                raise IndexError('negative steps are not supported')
            if stop <= start:
                return self.buffer[(slice(0, 0),) + rest]
            self.update_buffer(start, stop)
            return self.buffer[(slice(start - self.offset, stop - self.offset, step),) + rest]
        index = int(first)
        if index < 0:
            index += self.frames
        if index < 0 or index >= self.frames:
            raise IndexError('frame index out of range')
        self.update_buffer(index, index + 1)
        return self.buffer[(index - self.offset,) + rest]


class ArrayLoader(BufferedArray):
    """An in-memory (frames, channels) recording behind the BufferedArray interface:
    stands in for ``thunderlab.dataloader.DataLoader`` (``src/audian/data.py:172``)
    with its ``buffer_time`` / ``back_time`` arguments."""

    def __init__(self, data, rate, buffer_time=60.0, back_time=20.0, unit='a.u.',
                 ampl_max=1.0, verbose=0, view=False):
        super().__init__(verbose)
        self.view = view          # buffer = a window onto `data` instead of a float64 copy of it
        data = np.asarray(data)
        if data.ndim == 1:
            data = data[:, None]
        self.data = data
        self.rate = float(rate)
        self.frames, self.channels = data.shape
        self.shape = (self.frames, self.channels)
        self.ndim = 2
        self.size = self.frames*self.channels
        self.unit = unit
        self.ampl_min = -ampl_max
        self.ampl_max = ampl_max
        self.bufferframes = min(self.frames, int(buffer_time*self.rate))
        self.backframes = int(back_time*self.rate)
        self.buffer_changed = np.zeros(self.channels, dtype=bool)
        self.buffer = np.zeros((0, self.channels))
        self.name = 'data'
        self.dests = []
        self.need_update = False
        self.plot_items = [None]*self.channels
        self.move_buffer(0, self.bufferframes)

    def load_buffer(self, offset, nframes, buffer):
        buffer[:, :] = self.data[offset:offset + nframes, :]
        self._apply_unwrap(buffer)

    def move_buffer(self, offset, nframes):
        if not self.view:
            return BufferedArray.move_buffer(self, offset, nframes)
        # the recording is in memory anyway: moving the buffer is re-slicing it
        offset = int(max(0, offset))
        nframes = int(max(0, min(nframes, self.frames - offset)))
        self.buffer = self.data[offset:offset + nframes]
        self.offset = offset
        self.buffer_changed[:] = True


class WavLoader(BufferedArray):
    """A PCM WAV file behind the BufferedArray interface (stdlib ``wave``): stands in for
    ``thunderlab.dataloader.DataLoader`` on plain WAV recordings such as the reference's
    ``data/Gryllus_campestris.wav``.  Samples become float64 in [-1, 1) exactly as audioio
    scales them (integer / 2**(bits-1)); ``pcm_slab`` additionally hands the file's own
    bytes to the device path (``hipdsp_pcm_unpack``)."""

    def __init__(self, path, buffer_time=60.0, back_time=20.0, unit='a.u.', verbose=0):
        import wave
        super().__init__(verbose)
        self._wav = wave.open(path, 'rb')
        if self._wav.getcomptype() != 'NONE' or self._wav.getsampwidth() not in (2, 3, 4):
            raise ValueError('only uncompressed 16/24/32-bit PCM WAV files are supported')
        self.filepath = path
        self.sample_bytes = self._wav.getsampwidth()
        self.scale = 1.0/float(1 << (8*self.sample_bytes - 1))
        self.rate = float(self._wav.getframerate())
        self.channels = self._wav.getnchannels()
        self.frames = self._wav.getnframes()
        self.shape = (self.frames, self.channels)
        self.ndim = 2
        self.size = self.frames*self.channels
        self.unit = unit
        self.ampl_min, self.ampl_max = -1.0, 1.0
        self.bufferframes = min(self.frames, int(buffer_time*self.rate))
        self.backframes = int(back_time*self.rate)
        self.buffer_changed = np.zeros(self.channels, dtype=bool)
        self.buffer = np.zeros((0, self.channels))
        self.name = 'data'
        self.dests = []
        self.need_update = False
        self.plot_items = [None]*self.channels
        self.move_buffer(0, self.bufferframes)

    def pcm_slab(self, offset, nframes):
        """Raw interleaved bytes of frames [offset, offset + nframes) as a uint8 array."""
        self._wav.setpos(int(offset))
        raw = self._wav.readframes(int(nframes))
        return np.frombuffer(raw, dtype=np.uint8)

    def load_buffer(self, offset, nframes, buffer):
        raw = self.pcm_slab(offset, nframes)
        nb = self.sample_bytes
        if nb == 2:
            ints = raw.view('<i2').astype(np.int64)
        elif nb == 4:
            ints = raw.view('<i4').astype(np.int64)
        else:
            b = raw.reshape(-1, 3).astype(np.int64)
            ints = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            ints = np.where(ints >= 1 << 23, ints - (1 << 24), ints)
        buffer[:, :] = ints.reshape(-1, self.channels)*self.scale
        self._apply_unwrap(buffer)

    def close(self):
        self._wav.close()
